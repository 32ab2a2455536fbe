"""CPU suite: the HIP kernel SOURCES of the U-Net forward, executed through the
tests/emu functional emulator at a tiny size, against the reference-generated
golden (tolerance 1e-3 on log-probs per BASELINE.json north_star)."""
import os

import numpy as np

from oracle import unet_oracle as uo

TOL = 1e-3


def test_forward_emulated_matches_reference_golden(emu_engine, golden_dir):
    g = np.load(os.path.join(golden_dir, "unet_c3.npz"))
    emu_engine.load_state_dict(0, uo.synthetic_state_dict(3))
    x = g["rand32_x"][:1]
    emu_engine.set_precision("f32")
    try:
        lab, logp = emu_engine.forward(0, x)
    finally:
        emu_engine.set_precision("split_f16")
    assert np.abs(logp - g["rand32_logp"][:1]).max() < TOL
    bad = lab != g["rand32_lab"][:1]
    assert not np.any(bad & (g["rand32_margin"][:1].astype(np.float32) > 2 * TOL))


def test_forward_emulated_split_f16(emu_engine, golden_dir):
    """The 3-product split-f16 convolution path (v_mfma_f32_32x32x16_f16 + LDS-DMA staging, emulated)."""
    g = np.load(os.path.join(golden_dir, "unet_c3.npz"))
    emu_engine.load_state_dict(0, uo.synthetic_state_dict(3))
    emu_engine.set_precision("split_f16")
    lab, logp = emu_engine.forward(0, g["rand32_x"][:1])
    assert np.abs(logp - g["rand32_logp"][:1]).max() < TOL
    bad = lab != g["rand32_lab"][:1]
    assert not np.any(bad & (g["rand32_margin"][:1].astype(np.float32) > 2 * TOL))


def test_fused_head_labels_and_log_probabilities(emu_engine, golden_dir):
    """The head runs inside the last conv's epilogue on its fp32 results, with or without log-probabilities: the labels of both
    forms are the argmax of the same numbers.  The stand-alone head kernel (lm_set_fusion without bit 3) reads the stored 22-bit
    tensor: log-probabilities within a few 1e-6, labels equal away from near-ties."""
    g = np.load(os.path.join(golden_dir, "unet_c6.npz"))
    try:
        for c in (3, 6):
            emu_engine.load_state_dict(0, uo.synthetic_state_dict(c))
            x = g["rand32_x"][:1]
            emu_engine.set_fusion(11)
            lab_fused = emu_engine.forward(0, x, want_logp=False)[0]
            lab_plain, logp = emu_engine.forward(0, x)
            assert np.array_equal(lab_fused, lab_plain) and np.array_equal(lab_plain, logp.argmax(1))
            emu_engine.set_fusion(3)
            lab_k, logp_k = emu_engine.forward(0, x)
            d = float(np.abs(logp - logp_k).max())
            srt = np.sort(logp, axis=1)
            assert d < 2e-5 and not np.any((lab_plain != lab_k) & (srt[:, -1] - srt[:, -2] > 4 * d + 1e-6))
    finally:
        emu_engine.set_fusion(11)


def test_first_conv_inside_the_loader_is_bit_identical(emu_engine, golden_dir):
    """lm_set_fusion bit 0: down_path.0's first conv (resunet.py:93-95) computed by the vector ALU inside the loader of its second conv
    keeps the stand-alone kernel's operation order -- same labels, same log-probability bytes (32-wide geometry of the persistent
    kernel, two work items per slice so that the item switch with its double-buffered input patch is exercised)."""
    g = np.load(os.path.join(golden_dir, "unet_c3.npz"))
    emu_engine.load_state_dict(0, uo.synthetic_state_dict(3))
    x = np.concatenate([g["rand32_x"][:1].reshape(1, 32, 32)] * 2, axis=2)  # one 32 x 64 slice
    try:
        emu_engine.set_fusion(10)  # the default without the first conv in the loader
        lab0, logp0 = emu_engine.forward(0, x)
        emu_engine.set_fusion(11)
        lab1, logp1 = emu_engine.forward(0, x)
    finally:
        emu_engine.set_fusion(11)
    assert np.array_equal(lab0, lab1) and np.array_equal(logp0, logp1)


def out_of_f16_range_state_dict(c=3, scale=1e5):
    """A model whose FIRST layer's activations are ~1e5 (beyond the f16 maximum of 65504) while everything behind it is
    ordinary: the first BatchNorm's affine is scaled up and the second conv's weights down by the same factor (the reference
    computes this in fp32 without noticing)."""
    import torch

    sd = uo.synthetic_state_dict(c)
    sd["down_path.0.block.2.weight"] = sd["down_path.0.block.2.weight"] * scale
    sd["down_path.0.block.2.bias"] = sd["down_path.0.block.2.bias"] * scale
    sd["down_path.0.block.3.weight"] = sd["down_path.0.block.3.weight"] / scale
    return sd


def test_f16_range_guard_falls_back_to_exact_fp32(emu_engine, golden_dir):
    """Activations beyond the f16 range: the split-f16 kernels flag them, the engine re-runs the model on the exact-fp32
    kernels in the same call and pins it there -- the result meets the 1e-3 bar instead of silently being wrong."""
    import torch

    g = np.load(os.path.join(golden_dir, "unet_c3.npz"))
    x = g["rand32_x"][:1]
    sd = out_of_f16_range_state_dict(3)
    with torch.inference_mode():
        ref = uo.forward(sd, torch.from_numpy(x if x.ndim == 4 else x[:, None])).numpy()
    emu_engine.set_precision("split_f16")
    emu_engine.load_state_dict(0, sd)
    assert emu_engine.model_precision(0) == "split_f16"
    lab, logp = emu_engine.forward(0, x)
    assert emu_engine.model_precision(0) == "f32"  # the guard tripped and pinned the model
    assert np.abs(logp - ref).max() < TOL
    srt = np.sort(ref, axis=1)
    assert not np.any((lab != ref.argmax(1)) & (srt[:, -1] - srt[:, -2] > 2 * TOL))
    lab2 = emu_engine.forward(0, x, want_logp=False)[0]  # later forwards of this model: fp32 straight away
    assert np.array_equal(lab2, lab)
    # an ordinary model in the same engine keeps the fast path
    emu_engine.load_state_dict(0, uo.synthetic_state_dict(3))
    emu_engine.forward(0, x)
    assert emu_engine.model_precision(0) == "split_f16"


def rescaled_batchnorm_state_dict(n_classes=3, seed=5):
    """The synthetic network with every BatchNorm output channel multiplied by a factor f of either sign and up to 1.5 decades either
    way (gamma and beta times f) and the consumers' input-channel weights divided by f: the SAME function, but BatchNorm scales like
    a trained network may have them -- negative, tiny, large, wildly different within a layer.  The engine folds the scale into the
    consumers' packed weights (LM_H3_FOLD_SCALE) with one power of two per layer: this is the case that would hurt it."""
    import torch

    sd = {k: v.clone() for k, v in uo.synthetic_state_dict(n_classes).items()}
    g = torch.Generator().manual_seed(seed)

    def factors(c):
        f = torch.pow(10.0, torch.rand(c, generator=g) * 3.0 - 1.5) * torch.where(torch.rand(c, generator=g) < 0.3, -1.0, 1.0)
        f[0], f[1] = -1.0, 1e-2  # (two fixed cases per layer)
        return f

    def scale_bn(prefix, f):
        sd[prefix + ".weight"] = sd[prefix + ".weight"] * f
        sd[prefix + ".bias"] = sd[prefix + ".bias"] * f

    def unscale_consumer(conv, f, c0=0):
        w = sd[conv + ".weight"]
        w[:, c0 : c0 + len(f)] = w[:, c0 : c0 + len(f)] / f.view(1, -1, 1, 1)

    for i in range(5):
        c = 64 << i
        p = f"down_path.{i}.block"
        f = factors(c)
        scale_bn(p + ".2", f)
        unscale_consumer(p + ".3", f)
        f = factors(c)
        scale_bn(p + ".5", f)
        if i < 4:
            unscale_consumer(f"down_path.{i + 1}.block.0", f)                       # (through the average pool)
            unscale_consumer(f"up_path.{3 - i}.conv_block.block.0", f, c0=c)       # the skip half of torch.cat([up, bridge], 1)
        else:
            unscale_consumer("up_path.0.up.1", f)                                   # (through the bilinear upsample)
    for j in range(4):
        c = 512 >> j
        p = f"up_path.{j}.conv_block.block"
        f = factors(c)
        scale_bn(p + ".2", f)
        unscale_consumer(p + ".3", f)
        f = factors(c)
        scale_bn(p + ".5", f)
        unscale_consumer(f"up_path.{j + 1}.up.1" if j < 3 else "last", f)
    return sd


def check_rescaled_batchnorm(engine, x):
    """engine(sd') against the oracle on sd' (and sd' computes what sd computes: the rescaling is function-preserving)."""
    import torch

    sd, sd2 = uo.synthetic_state_dict(3), rescaled_batchnorm_state_dict(3)
    with torch.inference_mode():
        ref = uo.forward(sd2, torch.from_numpy(x)).numpy()
        same = uo.forward(sd, torch.from_numpy(x)).numpy()
    assert np.abs(ref - same).max() < 2e-4  # (the construction is right: only fp32 rounding differs)
    engine.load_state_dict(0, sd2)
    lab, logp = engine.forward(0, x)
    assert engine.model_precision(0) == "split_f16"  # (no range-guard fall-back: the stored tensors keep BatchNorm's magnitude)
    err = np.abs(logp - ref).max()
    assert err < TOL, err
    srt = np.sort(ref, axis=1)
    margin = srt[:, -1] - srt[:, -2]
    assert not np.any((lab != ref.argmax(1)) & (margin > 2 * TOL))
    return err


def test_batchnorm_scales_of_any_sign_and_magnitude(emu_engine, golden_dir):
    g = np.load(os.path.join(golden_dir, "unet_c3.npz"))
    check_rescaled_batchnorm(emu_engine, g["rand32_x"][:1])


def test_accuracy_guard_pins_a_model_the_split_kernels_cannot_resolve(emu_engine, golden_dir, monkeypatch):
    """The load-time accuracy guard (nn_engine.hip: model_probe; on in the product build, opt-in on the emulator where a probe is two
    emulated forwards): the Appendix-D model passes and stays on the split-f16 kernels; the same network with a head calibrated to a
    logit standard deviation of 30 is further than 5e-4 from the exact-fp32 kernels on the probe slice, gets pinned to them at load
    -- with a notice, like the range guard's -- and its result meets the 1e-3 bar."""
    import torch

    g = np.load(os.path.join(golden_dir, "unet_c3.npz"))
    x = g["rand32_x"][:1]
    base = uo.synthetic_state_dict(3)
    monkeypatch.setenv("LM_ACC_GUARD", "5e-4")
    try:
        emu_engine.set_precision("split_f16")
        emu_engine.load_state_dict(0, base)
        err, pinned = emu_engine.model_probe(0)
        assert err is not None and 0 < err < 5e-4 and not pinned and emu_engine.model_precision(0) == "split_f16"
        sd = uo.calibrate_head(base, torch.from_numpy(x), 30.0)
        emu_engine.load_state_dict(0, sd)
        err, pinned = emu_engine.model_probe(0)
        assert err is not None and err > 5e-4 and pinned and emu_engine.model_precision(0) == "f32"
        with torch.inference_mode():
            ref = uo.forward(sd, torch.from_numpy(x)).numpy()
        lab, logp = emu_engine.forward(0, x)
        assert np.abs(logp - ref).max() < TOL
        # the guard is per loaded model; "0" switches it off
        monkeypatch.setenv("LM_ACC_GUARD", "0")
        emu_engine.load_state_dict(0, sd)
        assert emu_engine.model_probe(0) == (None, False) and emu_engine.model_precision(0) == "split_f16"
    finally:
        monkeypatch.delenv("LM_ACC_GUARD", raising=False)
        emu_engine.load_state_dict(0, base)


def wide_batchnorm_heavy_tail_state_dict(n_classes=3, seed=9, decades=3.0, outlier=60.0, frac=5e-4):
    """BatchNorm scales spread over `decades` decades within every layer (gamma and beta times f; NOT compensated in the consumers: a
    different network, with channels that really are 30x louder or quieter than their neighbours, and a few near-dead ones at 1e-4)
    AND conv weights with `outlier`-fold outliers -- the two things a trained checkpoint can have that the seeded stand-in does
    not.  Both sides (oracle and engine) get the same tensors."""
    import torch

    sd = {k: v.clone() for k, v in uo.synthetic_state_dict(n_classes).items()}
    g = torch.Generator().manual_seed(seed)
    for k in list(sd):
        v = sd[k]
        if k.endswith(".weight") and v.ndim == 1 and (".block.2." in k or ".block.5." in k):  # BatchNorm gamma (and its beta)
            c = v.shape[0]
            f = torch.pow(10.0, (torch.rand(c, generator=g) - 0.5) * decades) * torch.where(torch.rand(c, generator=g) < 0.3, -1.0, 1.0)
            f = f / f.abs().pow(2).mean().sqrt()  # (keep the layer's output power: the activations stay in the stand-in's range)
            f[0], f[1] = 1e-4, -1e-4              # near-dead channels
            sd[k] = v * f
            sd[k[:-6] + "bias"] = sd[k[:-6] + "bias"] * f
        elif k.endswith(".weight") and v.ndim == 4 and v.shape[-1] == 3 and v.shape[1] >= 64:
            mask = torch.rand(v.shape, generator=g) < frac
            sd[k] = torch.where(mask, v * outlier, v)
    return sd


def test_split_k_3x3_instantiation_emulated(golden_dir):
    """The accuracy guard's middle tier (nn_kernels_h3.hip: the KS instantiation of the persistent 3x3 kernel + splitk_reduce3_h3_kernel):
    LM_H3_KSPLIT_K=576 cuts every accumulator chain at 576 products, which at 32 x 32 puts the K = 1152 layers of the two persistent
    geometries (32-wide: up_path.3's first conv; 16-wide: down_path.1's second conv, with its pooled output) on the split-K path:
    the result meets the reference golden, and differs from the single-chain form in the last bits only.  Own process (the hook is
    read once)."""
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys, os, numpy as np; sys.path.insert(0, %r)\n"
            "from lungmask_amd import _native as nat\n"
            "from lungmask_amd.build import build_emu\n"
            "from oracle import unet_oracle as uo\n"
            "g = np.load(os.path.join(%r, 'unet_c3.npz'))\n"
            "eng = nat.Engine(0, nat.Library(build_emu(), allow_emulation=True))\n"
            "eng.load_state_dict(0, uo.synthetic_state_dict(3))\n"
            "x = np.concatenate([g['rand32_x'][:1].reshape(1, 32, 32)] * 2, axis=0)\n"
            "lab, logp = eng.forward(0, x)\n"
            "err = float(np.abs(logp[:1] - g['rand32_logp'][:1]).max()); print('ERR', err)\n"
            "assert err < 1e-3 and np.array_equal(logp[0], logp[1])\n"
            "np.save(sys.argv[1], logp)\n") % (os.path.dirname(here), golden_dir)
    import tempfile

    out = {}
    with tempfile.TemporaryDirectory() as d:
        for k in ("0", "576"):
            f = os.path.join(d, f"logp{k}.npy")
            r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, LM_H3_KSPLIT_K=k, OMP_NUM_THREADS="8"), capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, r.stdout + r.stderr
            out[k] = np.load(f)
    d = float(np.abs(out["0"] - out["576"]).max())
    assert 0 < d < 2e-4, d  # (another summation order: not the same bits, the same numbers)
