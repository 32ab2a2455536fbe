"""GPU parity of the pre/post-processing kernels through the C ABI: bit-exact
against goldens produced by the reference's own utils.py, plus differential
tests against the CPU oracle on seeded volumes."""
import os

import numpy as np
import pytest

import prepost_cases as cases
from oracle import prepost_oracle as po

pytestmark = pytest.mark.gpu


def test_preprocess_bit_exact(gpu_engine):
    assert cases.check_preprocess(gpu_engine) >= 8


def test_preprocess_float_volumes(gpu_engine):
    cases.check_preprocess_float(gpu_engine)


def test_reshape_mask_bit_exact(gpu_engine):
    assert cases.check_reshape(gpu_engine) >= 10


def test_postprocessing_bit_exact(gpu_engine):
    assert cases.check_postprocess(gpu_engine) >= 20


def test_postprocessing_random_differential(gpu_engine):
    cases.check_postprocess_random(gpu_engine, seeds=range(12))
    cases.check_postprocess_random(gpu_engine, seeds=range(100, 104), shape=(24, 96, 80), nlab=5)


def test_postprocessing_diagonal_adversarial(gpu_engine):
    cases.check_postprocess_diagonal_adversarial(gpu_engine, n_iter=160)


def test_postprocessing_wide_rows(gpu_engine):
    cases.check_postprocess_wide_rows(gpu_engine)


def test_postprocessing_noise_volume(gpu_engine):
    cases.check_postprocess_noise(gpu_engine)


def test_empty_inputs(gpu_engine):
    cases.check_empty_inputs(gpu_engine)
    from oracle import unet_oracle as uo

    gpu_engine.load_state_dict(0, uo.synthetic_state_dict(3))
    assert gpu_engine.apply(0, np.zeros((0, 512, 512), np.int16)).shape == (0, 512, 512)


def test_bbox_3d_and_keep_largest(gpu_engine):
    """the reference-generated `klc*` goldens and tests/test_utils.py:58-63 through the HIP path."""
    cases.check_bbox_klc(gpu_engine)
    # a production-size mask: the phantom's lungs at 300 x 256 x 256 against the oracle
    from lungmask_amd import utils

    zz, yy, xx = np.ogrid[:300, :256, :256]
    m = (((zz - 150) / 135.0) ** 2 + ((yy - 128) / 55.0) ** 2 + ((xx - 75) / 40.0) ** 2 < 1) | \
        (((zz - 150) / 120.0) ** 2 + ((yy - 128) / 50.0) ** 2 + ((xx - 181) / 38.0) ** 2 < 1)
    m[7, 3:5, 250:253] = True
    utils.set_engine(gpu_engine)
    try:
        assert np.array_equal(utils.bbox_3D(m), po.bbox_3D(m))
        assert np.array_equal(utils.keep_largest_connected_component(m), po.keep_largest_connected_component(m))
    finally:
        utils.set_engine(None)


def test_reference_utils_tests(gpu_engine):
    """tests/test_utils.py of the reference, through the `lungmask_amd.utils` mirror."""
    from reference_utils_cases import check_reference_utils_tests

    check_reference_utils_tests(gpu_engine)


def test_fusion(gpu_engine):
    cases.check_fuse(gpu_engine)


def test_reorient(gpu_engine):
    cases.check_reorient(gpu_engine)


def test_preprocess_phantom_slices_vs_oracle(gpu_engine):
    vol = po.phantom(6, 512, 512)
    xi, xf, bb, _ = gpu_engine.preprocess(vol)
    ref_x, ref_bb = po.preprocess(vol, [256, 256])
    assert np.array_equal(bb, np.asarray(ref_bb, dtype=np.int32))
    assert np.array_equal(xi, ref_x)
    assert np.array_equal(xf, po.normalise(ref_x))


def test_postprocess_idempotent_and_deterministic_large(gpu_engine):
    """Size-independent properties at a BASELINE-sized label volume (300x256x256): running twice gives the
    same bytes, and post-processing its own output changes nothing but removes nothing new."""
    from oracle.make_golden import random_blobs

    rng = np.random.default_rng(9)
    small = random_blobs(rng, (30, 64, 64), 2, 40, 0.3)
    lab = np.kron(small, np.ones((10, 4, 4), dtype=np.uint8))  # 300 x 256 x 256
    a = gpu_engine.postprocess(lab)
    b = gpu_engine.postprocess(lab)
    assert np.array_equal(a, b)
    c = gpu_engine.postprocess(a)
    assert np.array_equal(a, c)  # one component per label without holes is a fixed point
    for v in np.unique(a)[1:]:
        assert po.sk_label(a == v).max() == 1


def test_slab_sharded_postprocessing(gpu_engine):
    """lm_slab_* with 1-4 in-process ranks (separate engines on the one GPU): small structured cases against the oracle."""
    from lungmask_amd import _native as nat

    extra = [nat.Engine(0) for _ in range(3)]
    try:
        assert cases.check_slab_postprocess([gpu_engine] + extra) >= 60
    finally:
        for e in extra:
            e.close()


def test_slab_protocol_adversarial_cuts(gpu_engine):
    """The slab protocol on diagonal / no-background / noise volumes cut at every position, three in-process ranks on the one GPU,
    against the oracle; then the same in the region-graph form (LM_SLAB_GRAPH=1, own process: the switch is read once)."""
    import subprocess
    import sys

    from lungmask_amd import _native as nat

    extra = [nat.Engine(0, gpu_engine.L) for _ in range(2)]
    try:
        cases.check_slab_postprocess_diagonal_adversarial([gpu_engine] + extra, n_iter=40)
    finally:
        for e in extra:
            e.close()
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import prepost_cases as cases\n"
            "from lungmask_amd import _native as nat\n"
            "from lungmask_amd.pipeline import postprocess_slabs_in_process\n"
            "engs = [nat.Engine(0) for _ in range(3)]\n"
            "assert cases.check_slab_postprocess(engs) >= 60\n"
            "assert postprocess_slabs_in_process.last_rounds == 4\n"
            "cases.check_slab_postprocess_diagonal_adversarial(engs, n_iter=40)\n"
            "print('graph form ok')\n") % (os.path.dirname(here), here)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LM_SLAB_GRAPH="1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "graph form ok" in r.stdout, r.stdout[-2000:]


def test_slab_sharded_postprocessing_full_size(gpu_engine):
    """A 120 x 256 x 256 label volume as the network produces it on the phantom (noisy: ~10^4 regions, ~10^5 boundary
    records), cut into 4 and 8 slabs (ragged): bit-identical to the whole-volume path, which the other tests pin to the oracle."""
    from lungmask_amd import _native as nat
    from lungmask_amd.pipeline import postprocess_slabs_in_process, shard_bounds
    from oracle import unet_oracle as uo

    gpu_engine.load_state_dict(0, uo.synthetic_state_dict(3))
    vol = po.phantom(120, 512, 512, seed=9)
    xf = gpu_engine.preprocess(vol)[1]
    lab = gpu_engine.forward(0, xf[:, None], want_logp=False)[0]
    whole = gpu_engine.postprocess(lab)
    info = gpu_engine.postprocess_info()
    assert info["regions"] > 100 and (whole != lab).any()
    extra = [nat.Engine(0) for _ in range(7)]
    try:
        engines = [gpu_engine] + extra
        for world, bounds in ((4, shard_bounds(120, 4)), (8, shard_bounds(120, 8)), (3, [0, 1, 119, 120]), (7, shard_bounds(120, 7))):
            out = postprocess_slabs_in_process(engines[:world], lab, bounds)
            assert np.array_equal(out, whole), (world, int((out != whole).sum()))
    finally:
        for e in extra:
            e.close()


def test_postprocessing_table_growth_paths_gpu():
    """As tests/test_prepost_emu.py::test_postprocessing_table_growth_paths_emulated, on the device: LM_POST_SMALL_TABLES=1 makes the
    speculative region / record tables and read-back guesses tiny, so the golden, random and noise cases take the
    grow-and-repeat and the second-copy paths (own process: the hook is read once per process)."""
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import prepost_cases as cases\n"
            "from lungmask_amd import _native as nat\n"
            "eng = nat.Engine(0)\n"
            "assert eng.L.is_gpu and cases.check_postprocess(eng) >= 20\n"
            "cases.check_postprocess_random(eng, seeds=range(6))\n"
            "cases.check_postprocess_noise(eng)\n"
            "cases.check_postprocess_wide_rows(eng)\n"
            "print('small tables ok', eng.postprocess_info())\n") % (os.path.dirname(here), here)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LM_POST_SMALL_TABLES="1"), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "small tables ok" in r.stdout, r.stdout[-2000:]


def test_apply_host_failure_leaves_output_untouched(gpu_engine):
    cases.check_apply_host_failure_leaves_output_untouched(gpu_engine)
