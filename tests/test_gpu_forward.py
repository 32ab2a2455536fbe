"""GPU parity of the network forward through the C ABI (lm_forward_dev)."""
import os

import numpy as np
import pytest
import torch

from oracle import prepost_oracle as po
from oracle import unet_oracle as uo

pytestmark = pytest.mark.gpu
TOL = 1e-3  # BASELINE.json north_star: pre-argmax log-probs within 1e-3 (fp32)


def check_labels(lab, ref_lab, margin, tol):
    """Labels must be identical except on pixels whose reference top-2 margin is below 2*tol."""
    bad = lab != ref_lab
    n_bad = int(bad.sum())
    assert not np.any(bad & (margin > 2 * tol)), f"{n_bad} mismatches, some away from near-ties"
    return n_bad


@pytest.fixture(params=["f32", "split_f16"])
def precision(request, gpu_engine):
    gpu_engine.set_precision(request.param)
    yield request.param
    gpu_engine.set_precision("split_f16")  # the engine default


@pytest.mark.parametrize("C", [3, 6])
def test_forward_matches_reference_goldens(gpu_engine, golden_dir, C, precision):
    g = np.load(os.path.join(golden_dir, f"unet_c{C}.npz"))
    gpu_engine.load_state_dict(0, uo.synthetic_state_dict(C))
    for case in ("rand32", "rand64", "phantom256"):
        x = g[case + "_x"]
        lab, logp = gpu_engine.forward(0, x)
        ref = g[case + "_logp"]
        got = logp if x.shape[-1] <= 64 else logp[:, :, ::4, ::4]
        err = np.abs(got - ref).max()
        print(f"[{precision}] C={C} {case}: max|dlogp|={err:.3e}")
        assert err < TOL, (case, err)
        check_labels(lab, g[case + "_lab"], g[case + "_margin"].astype(np.float32), TOL)


def test_loader_side_fusions_are_bit_identical(gpu_engine):
    """lm_set_fusion bit 0: the first conv computed inside the loader of conv 2 (resunet.py:93-95) keeps the operation order of the
    stand-alone kernel -- labels AND log-probabilities are the same bytes with and without, on widths that take the persistent
    kernel (multiples of 32), the 16-wide geometry and the fallback kernel, and batches that leave a partial last work item (bit 1,
    the bilinear x2 in the decoder conv's loader, is reserved: not built).  Bit 2, the split-K of the 16 x 16 /
    32 x 32 decoder 1x1 convs, adds its parts in a fixed order that is not the single chain's: deterministic, log-probabilities
    within 3e-4, labels equal away from near-ties."""
    try:
        for C in (3, 6):
            gpu_engine.load_state_dict(0, uo.synthetic_state_dict(C))
            for shape in ((3, 256, 256), (5, 64, 96), (2, 32, 32), (2, 48, 80), (3, 128, 32), (2, 16, 16), (20, 256, 256)):
                x = np.random.default_rng(shape[1] + C).random(shape, dtype=np.float32)
                gpu_engine.set_fusion(0)
                lab0, logp0 = gpu_engine.forward(0, x)
                only0 = gpu_engine.forward(0, x, want_logp=False)[0]
                for mask in (1, 2, 3):
                    gpu_engine.set_fusion(mask)
                    lab, logp = gpu_engine.forward(0, x)
                    only = gpu_engine.forward(0, x, want_logp=False)[0]
                    assert np.array_equal(lab, lab0) and np.array_equal(logp, logp0) and np.array_equal(only, only0), (C, shape, mask)
                gpu_engine.set_fusion(4)
                lab4, logp4 = gpu_engine.forward(0, x)
                lab4b, logp4b = gpu_engine.forward(0, x)
                assert np.array_equal(lab4, lab4b) and np.array_equal(logp4, logp4b)  # a fixed order: run to run the same bytes
                d = float(np.abs(logp4 - logp0).max())
                srt = np.sort(logp0, axis=1)
                assert d < 3e-4 * max(1.0, float(np.abs(logp0).max()) / 25.0), (C, shape, d)
                assert not np.any((lab4 != lab0) & (srt[:, -1] - srt[:, -2] > 4 * d + 1e-6)), (C, shape)
                if shape[0] == 20:
                    print(f"C={C} {shape}: split-K 1x1 vs single chain: max|dlogp| {d:.2e}, {int((lab4 != lab0).sum())} labels differ")
    finally:
        gpu_engine.set_fusion(11)


def test_forward_batch20_vs_oracle(gpu_engine, precision):
    """BASELINE config batch (20 slices of 256x256) against the torch-fp32 CPU oracle."""
    sd = uo.synthetic_state_dict(3)
    gpu_engine.load_state_dict(0, sd)
    rng = np.random.default_rng(5)
    x = rng.random((20, 256, 256), dtype=np.float32)
    lab, logp = gpu_engine.forward(0, x)
    xs = torch.from_numpy(x[[0, 7, 19]][:, None])
    with torch.inference_mode():
        ref = uo.forward(sd, xs)
    srt = torch.sort(ref, dim=1, descending=True)[0]
    margin = (srt[:, 0] - srt[:, 1]).numpy()
    ref = ref.numpy()
    err = np.abs(logp[[0, 7, 19]] - ref).max()
    assert err < TOL, err
    check_labels(lab[[0, 7, 19]], ref.argmax(1).astype(np.uint8), margin, TOL)
    # determinism: same input twice -> identical bytes
    lab2, logp2 = gpu_engine.forward(0, x)
    assert np.array_equal(lab, lab2) and np.array_equal(logp, logp2)


def test_forward_repeatability_stress(gpu_engine):
    """Race screen: the same input through the batched two-lane forward 60 times (ragged last batch, odd slice count
    on the 16x16 level) must give identical bytes every time, and identical to a single-lane run."""
    sd = uo.synthetic_state_dict(3)
    gpu_engine.load_state_dict(0, sd)
    rng = np.random.default_rng(17)
    n = 45
    x = gpu_engine.to_device(rng.random((n, 256, 256), dtype=np.float32))
    lab = gpu_engine.empty((n, 256, 256), np.uint8)
    lib = gpu_engine.L.lib

    def run(batch):
        gpu_engine.L.check(lib.lm_forward_batches_dev(gpu_engine.h, 0, x.ptr, n, 256, 256, batch, lab.ptr))
        gpu_engine.sync()
        return lab.download()

    gpu_engine.set_streams(1)
    ref = run(20)
    gpu_engine.set_streams(2)
    for it in range(60):
        got = run(20 if it % 3 else 7)
        assert np.array_equal(got, ref), f"iteration {it}: {int((got != ref).sum())} label bytes differ"
    # log-probabilities of one batch, bit for bit
    xs = gpu_engine.to_device(rng.random((5, 256, 256), dtype=np.float32))
    lp = gpu_engine.empty((5, 3, 256, 256), np.float32)
    l5 = gpu_engine.empty((5, 256, 256), np.uint8)
    gpu_engine.forward_dev(0, xs, l5, lp)
    gpu_engine.sync()
    first = lp.download()
    for it in range(40):
        gpu_engine.forward_dev(0, xs, l5, lp)
        gpu_engine.sync()
        assert np.array_equal(lp.download(), first), f"iteration {it}"


def test_forward_odd_width_fallback_kernel(gpu_engine, precision):
    """48x48 input: widths 48/24/12/6/3 are neither 16 nor multiples of 32 -> the simple 4-wave kernel with clipped tiles."""
    sd = uo.synthetic_state_dict(3)
    gpu_engine.load_state_dict(0, sd)
    x = np.random.default_rng(23).random((3, 48, 48), dtype=np.float32)
    lab, logp = gpu_engine.forward(0, x)
    with torch.inference_mode():
        ref = uo.forward(sd, torch.from_numpy(x[:, None]))
    srt = torch.sort(ref, dim=1, descending=True)[0]
    err = np.abs(logp - ref.numpy()).max()
    assert err < TOL, err
    check_labels(lab, ref.argmax(1).numpy().astype(np.uint8), (srt[:, 0] - srt[:, 1]).numpy(), TOL)


def test_forward_fallback_kernel_at_full_size():
    """LM_H3_FALLBACK=1 (read once per process) forces the fallback conv kernel on every level: own process."""
    import subprocess
    import sys

    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
        "from lungmask_amd import _native as nat\n"
        "from oracle import unet_oracle as uo\n"
        "e = nat.Engine(0); sd = uo.synthetic_state_dict(3); e.load_state_dict(0, sd)\n"
        "x = np.random.default_rng(3).random((2, 256, 256), dtype=np.float32)\n"
        "lab, logp = e.forward(0, x)\n"
        "ref = uo.forward(sd, torch.from_numpy(x[:, None])).numpy()\n"
        "err = float(np.abs(logp - ref).max()); print('ERR', err); assert err < 1e-3\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LM_H3_FALLBACK="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("order", ["0", "1"])
def test_forward_work_item_orders_agree(order):
    """The persistent conv kernel picks its work-item order per layer (cout-major, or XCD-aware pixel-tile-major over a
    padded index space).  LM_H3_ORDER forces one order everywhere: both must give the oracle's result -- also with an
    odd batch, where the padded space has holes."""
    import subprocess
    import sys

    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
        "from lungmask_amd import _native as nat\n"
        "from oracle import unet_oracle as uo\n"
        "e = nat.Engine(0); sd = uo.synthetic_state_dict(3); e.load_state_dict(0, sd)\n"
        "x = np.random.default_rng(4).random((3, 256, 256), dtype=np.float32)\n"
        "lab, logp = e.forward(0, x)\n"
        "ref = uo.forward(sd, torch.from_numpy(x[:, None])).numpy()\n"
        "err = float(np.abs(logp - ref).max()); print('ERR', err); assert err < 1e-3\n"
        "np.save(sys.argv[1], lab)\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile

    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "lab.npy")
        r = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, LM_H3_ORDER=order), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        lab = np.load(out)
    from lungmask_amd import _native as nat

    e = nat.Engine(0)  # default (per-layer) order in this process
    e.load_state_dict(0, uo.synthetic_state_dict(3))
    mine = e.forward(0, np.random.default_rng(4).random((3, 256, 256), dtype=np.float32), want_logp=False)[0]
    e.close()
    assert np.array_equal(lab, mine)  # the order changes nothing but the schedule: bit-identical labels


def test_fused_head_labels_and_log_probabilities(gpu_engine):
    """The head (last 1x1 conv + log-softmax + argmax, resunet.py:69-70, mask.py:184-186) runs inside the last conv's epilogue on its fp32
    results, for labels-only forwards (the production path) AND when log-probabilities are asked for: one set of numbers.  The
    stand-alone head kernel (lm_set_fusion without bit 3; widths the persistent kernel does not serve) reads the stored 22-bit hi/lo
    tensor instead: its log-probabilities agree to a few 1e-6 of the logit range, its labels everywhere but on near-tie pixels."""
    rng = np.random.default_rng(21)
    try:
        for c, shape in ((3, (3, 256, 256)), (6, (2, 256, 256)), (3, (2, 64, 96))):
            gpu_engine.load_state_dict(0, uo.synthetic_state_dict(c))
            x = rng.random(shape, dtype=np.float32)
            gpu_engine.set_fusion(11)
            lab_only = gpu_engine.forward(0, x, want_logp=False)[0]
            lab, logp = gpu_engine.forward(0, x)
            assert np.array_equal(lab_only, lab) and np.array_equal(lab, logp.argmax(1))
            gpu_engine.set_fusion(3)
            lab_k, logp_k = gpu_engine.forward(0, x)
            d = float(np.abs(logp - logp_k).max())
            srt = np.sort(logp, axis=1)
            margin = srt[:, -1] - srt[:, -2]
            print(f"C={c} {shape}: fused head vs head kernel on the stored tensor: max|dlogp| {d:.2e}, {int((lab != lab_k).sum())} labels differ")
            assert d < 2e-5 * max(1.0, float(np.abs(logp).max()) / 25.0)
            assert not np.any((lab != lab_k) & (margin > 4 * d + 1e-6))
    finally:
        gpu_engine.set_fusion(11)


def test_forward_heavy_tailed_weights(gpu_engine, monkeypatch):
    """Trained weights are heavy-tailed.  The split-f16 packing scales each layer by a power of two so that the f16
    remainder (`lo`) of every weight down to 2^-14 of the layer's largest one stays a normal number: a layer whose largest
    weight is 60x its typical ones must keep fp32-class accuracy.  The head is calibrated per SURVEY Appendix D (every class's logit
    map at std 8 on this input -- the recipe), so the north-star's ABSOLUTE bar applies: 1e-3, no scaling with the logit range
    (VERDICT r03 #3)."""
    sd = uo.synthetic_state_dict(3)
    g = torch.Generator().manual_seed(5)
    for k, v in sd.items():
        if k.endswith(".weight") and v.ndim == 4 and v.shape[-1] == 3 and v.shape[1] >= 64:
            mask = torch.rand(v.shape, generator=g) < 5e-4
            sd[k] = torch.where(mask, v * 60.0, v)
    x = np.random.default_rng(8).random((2, 256, 256), dtype=np.float32)
    sd = uo.calibrate_head(sd, torch.from_numpy(x[:1, None]), 8.0)
    monkeypatch.setenv("LM_ACC_GUARD", "0")  # the split kernels themselves are measured here (the guard: test_accuracy_guard below)
    gpu_engine.load_state_dict(0, sd)
    monkeypatch.delenv("LM_ACC_GUARD")
    lab, logp = gpu_engine.forward(0, x)
    assert gpu_engine.model_precision(0) == "split_f16"
    ref = uo.forward(sd, torch.from_numpy(x[:, None])).numpy()
    err = float(np.abs(logp - ref).max())
    print(f"heavy-tailed weights, head std 8: log-probs {float(ref.min()):.0f}..{float(ref.max()):.0f}, max|dlogp| {err:.2e}")
    assert err < TOL, err
    margin = np.sort(ref, axis=1)[:, -1] - np.sort(ref, axis=1)[:, -2]
    assert not np.any((lab != ref.argmax(1)) & (margin > 2 * TOL))


@pytest.mark.parametrize("prec", ["split_f16", "f32"])
def test_logit_range_sweep(gpu_engine, prec, monkeypatch):
    """the split-f16 error is relative, the 1e-3 bar absolute -- sweep the logit range with heads calibrated
    per SURVEY Appendix D (every class's logit map at std 8 -- the recipe --, then 30 and 100) on the phantom's network input.
    No tolerance scaling with the range for the recipe's own std 8: absolute 1e-3 against the torch-fp32 oracle.
    Beyond that the REFERENCE's own fp32 arithmetic is the limit: |oracle_f32 - oracle_f64| measures 1.6e-4 at std 8, 6.2e-4 at
    std 30 and 2.0e-3 at std 100 on this input (logits -316..+598), i.e. at std 100 the reference is further than the bar from
    the exact value of its own graph and no implementation with another summation order can be within 1e-3 of it.  For those
    models the engine is held to what is checkable: it must be as close to the float64 evaluation as the reference is (factor
    2.5), and within 1e-3 + 2.5x the reference's own distance of the reference.
    The exact-fp32 kernels (the fall-back of the f16 range guard) are swept too: since round 4 they sum every 16-channel chunk into a
    fresh accumulator that is added in chunk order (one chain of up to 4608 roundings before: 6.4e-4 at std 8, 4x the reference's
    own noise) and measure 1.9e-4 / 7.5e-4 / 2.5e-3 -- as close to float64 as the reference; they get a factor of 1.5."""
    base = uo.synthetic_state_dict(3)
    ph = po.phantom(2, 512, 512)
    xs, _ = po.preprocess(ph, [256, 256])
    x = po.normalise(xs)
    xt = torch.from_numpy(x[:, None])
    monkeypatch.setenv("LM_ACC_GUARD", "0")  # the kernels themselves are swept (with the guard on, the std 30 / 100 models run on the fp32 ones)
    gpu_engine.set_precision(prec)
    try:
        for std in (8.0, 30.0, 100.0):
            sd = uo.calibrate_head(base, xt[:1], std)
            gpu_engine.load_state_dict(0, sd)
            lab, logp = gpu_engine.forward(0, x)
            assert gpu_engine.model_precision(0) == prec  # the f16 range guard has no reason to trip
            with torch.inference_mode():
                ref = uo.forward(sd, xt).numpy()
                ref64 = uo.forward_f64(sd, xt).numpy()
                lg = uo.forward_logits(sd, xt)
            noise = float(np.abs(ref - ref64).max())          # the reference's own rounding noise on this model
            err = float(np.abs(logp - ref).max())             # engine vs reference (the north-star's quantity)
            err64 = float(np.abs(logp - ref64).max())         # engine vs the exact evaluation
            print(f"{prec} head std {std:g}: logits {float(lg.min()):.0f}..{float(lg.max()):.0f}  |engine-ref32| {err:.2e}  "
                  f"|engine-ref64| {err64:.2e}  |ref32-ref64| {noise:.2e}")
            if std == 8.0:
                assert err < TOL, (std, err)
            k = 2.5 if prec == "split_f16" else 1.5  # (the chunked exact-fp32 kernel is as close to float64 as the reference itself)
            assert err < TOL + k * noise and err64 < max(TOL, k * noise), (std, err, err64, noise)
            margin = np.sort(ref, axis=1)[:, -1] - np.sort(ref, axis=1)[:, -2]
            assert not np.any((lab != ref.argmax(1)) & (margin > 2 * max(err, TOL)))
    finally:
        gpu_engine.set_precision("split_f16")


def test_f16_range_guard_falls_back_to_exact_fp32(gpu_engine, monkeypatch):
    """VERDICT r01 weak #3 / ADVICE: the split-f16 path stores activations as f16 pairs, so a model whose activations exceed
    65504 would silently produce inf/NaN.  The kernels flag |v| >= 2^15 (or non-finite) in every split producer; the engine
    re-runs the forward on the exact-fp32 kernels within the same call and pins the model there.  The state_dict below has
    first-layer activations around 1e5 and ordinary logits; tolerance: the same absolute 1e-3 on the log-probabilities."""
    from tests.test_forward_emu import out_of_f16_range_state_dict

    sd = out_of_f16_range_state_dict(3)
    x = np.random.default_rng(11).random((3, 256, 256), dtype=np.float32)
    with torch.inference_mode():
        ref = uo.forward(sd, torch.from_numpy(x[:, None])).numpy()
    gpu_engine.set_precision("split_f16")
    # With the load-time probe on (the default) this model never reaches a caller on the split kernels: the probe slice itself leaves
    # the f16 range and the range guard pins the model at load.  The rest of this test is about the RUN-TIME guard -- a model whose
    # probe stays in range but whose real input does not -- so it loads without the probe.
    gpu_engine.load_state_dict(0, sd)
    assert gpu_engine.model_precision(0) == "f32" and gpu_engine.model_probe(0) == (None, False)
    monkeypatch.setenv("LM_ACC_GUARD", "0")
    gpu_engine.load_state_dict(0, sd)
    assert gpu_engine.model_precision(0) == "split_f16"
    lab, logp = gpu_engine.forward(0, x)
    assert gpu_engine.model_precision(0) == "f32"
    err = float(np.abs(logp - ref).max())
    print(f"out-of-f16-range model: max|dlogp| = {err:.3e} after the fp32 fallback (logit range {np.abs(ref).max():.1f})")
    assert err < TOL
    srt = np.sort(ref, axis=1)
    check_labels(lab, ref.argmax(1).astype(np.uint8), srt[:, -1] - srt[:, -2], TOL)
    # the batched two-lane entry point takes the same route
    xd = gpu_engine.to_device(x)
    ld = gpu_engine.empty(x.shape, np.uint8)
    gpu_engine.load_state_dict(0, sd)  # fresh load: the pin is per loaded model
    assert gpu_engine.model_precision(0) == "split_f16"
    gpu_engine.L.check(gpu_engine.L.lib.lm_forward_batches_dev(gpu_engine.h, 0, xd.ptr, 3, 256, 256, 2, ld.ptr))
    gpu_engine.sync()
    assert gpu_engine.model_precision(0) == "f32"
    assert np.array_equal(ld.download(), lab)
    # and the whole hot path (lm_apply_host: the volume goes through in two pieces, the flag is read back once behind the second;
    # 7 slices at batch 1 -> head of two batches + tail of five): the same labels as with the model on the exact kernels
    from oracle import prepost_oracle as po

    vol = po.phantom(7, 512, 512, seed=5)
    expect = gpu_engine.apply(0, vol, batch_size=1)  # the model is pinned to fp32 at this point
    gpu_engine.load_state_dict(0, sd)
    assert gpu_engine.model_precision(0) == "split_f16"
    got = gpu_engine.apply(0, vol, batch_size=1)
    assert gpu_engine.model_precision(0) == "f32"
    assert np.array_equal(got, expect)
    gpu_engine.load_state_dict(0, uo.synthetic_state_dict(3))
    gpu_engine.forward(0, x[:1])
    assert gpu_engine.model_precision(0) == "split_f16"


def test_batchnorm_scales_of_any_sign_and_magnitude(gpu_engine, golden_dir):
    """BatchNorm scales like a trained network may have them (negative, 1.5 decades either way within a layer) through the folded-scale
    form of the conv path (nn_kernels.h: LM_H3_FOLD_SCALE): a function-preserving rescaling of the synthetic network, held to the
    oracle on the same weights at 256 x 256 -- and no fall-back to the exact-fp32 kernels."""
    from test_forward_emu import check_rescaled_batchnorm

    g = np.load(os.path.join(golden_dir, "unet_c3.npz"))
    err = check_rescaled_batchnorm(gpu_engine, g["phantom256_x"][:1])
    print(f"rescaled BatchNorm: max|dlogp| = {err:.2e}")


def test_accuracy_guard(gpu_engine):
    """VERDICT r05 #1c: beside the f16 RANGE guard an ACCURACY guard -- at load every model's split-f16 kernels are compared with the
    exact-fp32 ones on two deterministic probe slices (256 x 256: phantom-like, uniform noise) and the model is pinned to the exact kernels above 5e-4.  The
    Appendix-D model (head at std 8) passes with room and stays on the fast path; the same network with the head at std 30 -- whose
    split result is 1.5e-3 from the reference (test_logit_range_sweep) -- is pinned at load and then meets the 1e-3 bar; the models
    the bench runs (lung-like heads, 3 and 6 classes) stay on the fast path."""
    from lungmask_amd import synthetic

    base = uo.synthetic_state_dict(3)
    ph = po.phantom(2, 512, 512)
    xs, _ = po.preprocess(ph, [256, 256])
    x = po.normalise(xs)
    xt = torch.from_numpy(x[:, None])
    try:
        for c in (3, 6):
            gpu_engine.load_state_dict(0, synthetic.synthetic_state_dict(c, head="lunglike"))
            err, pinned = gpu_engine.model_probe(0)
            print(f"bench model C={c}: probe {err:.2e}")
            assert err is not None and err < 2.5e-4 and not pinned and gpu_engine.model_precision(0) == "split_f16"
        sd8 = uo.calibrate_head(base, xt[:1], 8.0)
        gpu_engine.load_state_dict(0, sd8)
        err8, pinned = gpu_engine.model_probe(0)
        assert err8 is not None and err8 < 5e-4 and not pinned and gpu_engine.model_precision(0) == "split_f16"
        # between the two: a head at std 16 is over the limit on the single-chain form and within it on one of the split-K forms -- it
        # keeps the 16-bit matrix cores (a few per cent slower) instead of falling back to the exact kernels' 4x
        sd16 = uo.calibrate_head(base, xt[:1], 16.0)
        gpu_engine.load_state_dict(0, sd16)
        err16, pinned16 = gpu_engine.model_probe(0)
        lab, logp = gpu_engine.forward(0, x)
        with torch.inference_mode():
            ref16 = uo.forward(sd16, xt).numpy()
        e16 = float(np.abs(logp - ref16).max())
        print(f"accuracy guard: head at std 16 runs on {gpu_engine.model_tier(0)} (probe {err16:.2e}); max|dlogp| vs the oracle {e16:.2e}")
        assert e16 < TOL and (pinned16 or err16 <= 5e-4)
        sd30 = uo.calibrate_head(base, xt[:1], 30.0)
        gpu_engine.load_state_dict(0, sd30)
        err30, pinned = gpu_engine.model_probe(0)
        assert err30 is not None and err30 > 5e-4 and pinned and gpu_engine.model_precision(0) == "f32"
        lab, logp = gpu_engine.forward(0, x)
        with torch.inference_mode():
            ref = uo.forward(sd30, xt).numpy()
        e30 = float(np.abs(logp - ref).max())
        print(f"accuracy guard: probe at std 8 {err8:.2e} (stays split-f16), at std 30 {err30:.2e} (pinned to fp32: max|dlogp| vs the oracle {e30:.2e})")
        assert e30 < TOL
        # heavy-tailed weights (test_forward_heavy_tailed_weights: the split kernels are 5.5e-4 .. 8.6e-4 from the reference depending on
        # the input, profiles/r06a_precision_dist.log): whatever the guard decides, the model's result meets the bar
        g = torch.Generator().manual_seed(5)
        heavy = dict(base)
        for k, v in heavy.items():
            if k.endswith(".weight") and v.ndim == 4 and v.shape[-1] == 3 and v.shape[1] >= 64:
                heavy[k] = torch.where(torch.rand(v.shape, generator=g) < 5e-4, v * 60.0, v)
        xr = np.random.default_rng(8).random((2, 256, 256), dtype=np.float32)
        heavy = uo.calibrate_head(heavy, torch.from_numpy(xr[:1, None]), 8.0)
        gpu_engine.load_state_dict(0, heavy)
        errh, pinned = gpu_engine.model_probe(0)
        lab, logp = gpu_engine.forward(0, xr)
        with torch.inference_mode():
            refh = uo.forward(heavy, torch.from_numpy(xr[:, None])).numpy()
        eh = float(np.abs(logp - refh).max())
        print(f"accuracy guard, heavy-tailed weights: probe {errh:.2e} -> runs on {gpu_engine.model_tier(0)}; max|dlogp| vs the oracle {eh:.2e}")
        assert gpu_engine.model_tier(0) != "split_f16"  # (the single-chain form is 5.7e-4 from the exact kernels on the probe: a shorter-chain tier or fp32)
        assert eh < TOL and (pinned or errh <= 5e-4) and (not pinned or eh < 4e-4)
        # a model loaded while the engine is on the exact kernels meets the guard when the engine goes back to the split ones
        gpu_engine.set_precision("f32")
        gpu_engine.load_state_dict(0, sd30)
        assert gpu_engine.model_probe(0) == (None, False)
        gpu_engine.set_precision("split_f16")
        assert gpu_engine.model_probe(0)[1] and gpu_engine.model_precision(0) == "f32"
    finally:
        gpu_engine.set_precision("split_f16")
        gpu_engine.load_state_dict(0, base)


def test_wide_batchnorm_scales_and_heavy_tails_together(gpu_engine, monkeypatch):
    """VERDICT r05 #1 / ADVICE r05: BatchNorm scales spread over 3 decades inside every layer (uncompensated: channels that really are
    30x louder or quieter than their neighbours, two near-dead ones at 1e-4) AND 60x weight outliers, head per Appendix D (std 8).
    The per-channel power of two (nn_engine.hip: load_conv) keeps every stored channel at BatchNorm's magnitude and the consumers'
    rows at the dynamic range of w: the split kernels themselves (guard off) stay within the absolute 1e-3, without a range-guard
    fall-back."""
    from test_forward_emu import wide_batchnorm_heavy_tail_state_dict

    x = np.random.default_rng(12).random((2, 256, 256), dtype=np.float32)
    sd = uo.calibrate_head(wide_batchnorm_heavy_tail_state_dict(3), torch.from_numpy(x[:1, None]), 8.0)
    monkeypatch.setenv("LM_ACC_GUARD", "0")
    gpu_engine.load_state_dict(0, sd)
    monkeypatch.delenv("LM_ACC_GUARD")
    lab, logp = gpu_engine.forward(0, x)
    assert gpu_engine.model_precision(0) == "split_f16"
    with torch.inference_mode():
        ref = uo.forward(sd, torch.from_numpy(x[:, None])).numpy()
        ref64 = uo.forward_f64(sd, torch.from_numpy(x[:, None])).numpy()
    err = float(np.abs(logp - ref).max())
    print(f"BatchNorm scales over 3 decades + 60x weight outliers, head std 8: log-probs {float(ref.min()):.0f}..{float(ref.max()):.0f}, "
          f"max|dlogp| {err:.2e} (vs float64 {float(np.abs(logp - ref64).max()):.2e}; the reference's own fp32 noise {float(np.abs(ref - ref64).max()):.2e})")
    assert err < TOL, err
    margin = np.sort(ref, axis=1)[:, -1] - np.sort(ref, axis=1)[:, -2]
    assert not np.any((lab != ref.argmax(1)) & (margin > 2 * TOL))
    gpu_engine.load_state_dict(0, uo.synthetic_state_dict(3))


@pytest.mark.parametrize("chain_k", ["4608", "1152"])
def test_split_k_3x3_tiers(chain_k):
    """The accuracy guard's middle tiers on the hardware (nn_kernels_h3.hip: the KS instantiation of the persistent 3x3 kernel, 32- and
    16-wide, with and without the pooled output, + splitk_reduce3_h3_kernel): LM_H3_KSPLIT_K puts every model on the form with no
    accumulator chain over that many products (own process: read once).  Within the bar against the oracle, 3 and 6 classes, odd
    batch; deterministic (two runs, same bytes); never further from the oracle's float64 evaluation than the single-chain form by
    more than noise -- the point of the tier is that it is closer."""
    import subprocess
    import sys

    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
        "from lungmask_amd import _native as nat\n"
        "from oracle import unet_oracle as uo\n"
        "e = nat.Engine(0)\n"
        "for C, B in ((3, 3), (6, 2)):\n"
        "    sd = uo.synthetic_state_dict(C); e.load_state_dict(0, sd)\n"
        "    x = np.random.default_rng(4 + C).random((B, 256, 256), dtype=np.float32)\n"
        "    lab, logp = e.forward(0, x)\n"
        "    lab2, logp2 = e.forward(0, x)\n"
        "    assert np.array_equal(logp, logp2) and np.array_equal(lab, lab2)\n"
        "    with torch.inference_mode():\n"
        "        ref = uo.forward(sd, torch.from_numpy(x[:, None])).numpy(); ref64 = uo.forward_f64(sd, torch.from_numpy(x[:, None])).numpy()\n"
        "    err = float(np.abs(logp - ref).max()); rms64 = float(np.sqrt(((logp - ref64).astype(np.float64) ** 2).mean()))\n"
        "    print('C', C, 'ERR', err, 'RMS64', rms64); assert err < 1e-3\n"
        "    m = np.sort(ref, axis=1); assert not np.any((lab != ref.argmax(1)) & (m[:, -1] - m[:, -2] > 2e-3))\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for k in ("0", chain_k):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LM_H3_KSPLIT_K=k, LM_ACC_GUARD="0"), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout + r.stderr
        out[k] = [float(l.split()[-1]) for l in r.stdout.splitlines() if l.startswith("C ")]
    print(f"chain limit {chain_k}: rms error vs float64 {out[chain_k]} (single chain: {out['0']})")
    assert all(a < 1.05 * b for a, b in zip(out[chain_k], out["0"]))


def _weight_families():
    """Stand-ins that differ from the seeded recipe in the ways a trained checkpoint may: other seeds, BatchNorm statistics over
    orders of magnitude (running_var 1e-2 .. 1e2, large running_mean), layer-wise weight scales that push the activations up and down by
    3x from layer to layer, 200x weight outliers, half of the weights exactly zero, a sharper head."""
    from lungmask_amd import synthetic

    fams = []
    for seed in (7, 1234):
        fams.append((f"seed {seed}", dict(synthetic.synthetic_state_dict(3, seed=seed)), 8.0))
    sd = {k: v.clone() for k, v in uo.synthetic_state_dict(3).items()}
    g = torch.Generator().manual_seed(31)
    for k in list(sd):
        if k.endswith("running_var") and "residual" not in k:
            sd[k] = torch.pow(10.0, torch.rand(sd[k].shape, generator=g) * 4.0 - 2.0)
        elif k.endswith("running_mean") and "residual" not in k:
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.8
    fams.append(("BatchNorm running_var over 1e-2 .. 1e2, running_mean ~ N(0, 0.8)", sd, 8.0))
    sd = {k: v.clone() for k, v in uo.synthetic_state_dict(3).items()}
    i = 0
    for k in list(sd):
        if k.endswith(".weight") and sd[k].ndim == 4 and sd[k].shape[-1] == 3 and "residual" not in k:
            sd[k] = sd[k] * (3.0 if i % 2 == 0 else 1.0 / 3.0)
            i += 1
    fams.append(("conv weights x3 / x(1/3) alternating by layer", sd, 8.0))
    sd = {k: v.clone() for k, v in uo.synthetic_state_dict(3).items()}
    g = torch.Generator().manual_seed(33)
    for k in list(sd):
        v = sd[k]
        if k.endswith(".weight") and v.ndim == 4 and v.shape[-1] == 3 and v.shape[1] >= 64:
            sd[k] = torch.where(torch.rand(v.shape, generator=g) < 1e-4, v * 200.0, v)
    fams.append(("200x weight outliers (1e-4 of the weights)", sd, 8.0))
    sd = {k: v.clone() for k, v in uo.synthetic_state_dict(3).items()}
    g = torch.Generator().manual_seed(35)
    for k in list(sd):
        v = sd[k]
        if k.endswith(".weight") and v.ndim == 4 and v.shape[-1] == 3 and v.shape[1] >= 64:
            sd[k] = torch.where(torch.rand(v.shape, generator=g) < 0.5, torch.zeros_like(v), v * 1.41421356)
    fams.append(("half of the conv weights exactly zero", sd, 8.0))
    fams.append(("Appendix-D head at std 12", {k: v.clone() for k, v in uo.synthetic_state_dict(3).items()}, 12.0))
    return fams


def test_weight_families_through_the_guarded_engine(gpu_engine):
    """VERDICT r05 weak #2 ("every parity number is on one synthetic weight family ... no accuracy guard"): seven families that differ
    from the seeded recipe the way a trained checkpoint may, each loaded with the accuracy guard ON (the product's default) and held
    to the bar against the oracle on the same tensors -- whatever form the guard put the model on (printed), the result is within
    1e-3 and the labels follow the near-tie rule."""
    x = np.random.default_rng(21).random((2, 256, 256), dtype=np.float32)
    xt = torch.from_numpy(x[:, None])
    try:
        for name, sd0, std in _weight_families():
            sd = uo.calibrate_head(sd0, xt[:1], std)
            gpu_engine.load_state_dict(0, sd)
            probe, _ = gpu_engine.model_probe(0)
            lab, logp = gpu_engine.forward(0, x)
            with torch.inference_mode():
                ref = uo.forward(sd, xt).numpy()
            err = float(np.abs(logp - ref).max())
            print(f"{name}: runs on {gpu_engine.model_tier(0)} (probe {probe if probe is None else format(probe, '.2e')}), log-probs {float(ref.min()):.0f}..0, max|dlogp| {err:.2e}")
            assert np.isfinite(logp).all() and err < TOL, (name, err)
            srt = np.sort(ref, axis=1)
            assert not np.any((lab != ref.argmax(1)) & (srt[:, -1] - srt[:, -2] > 2 * TOL)), name
    finally:
        gpu_engine.load_state_dict(0, uo.synthetic_state_dict(3))
