"""CPU suite: pre-processing / un-crop kernel SOURCES executed through the tests/emu
emulator, bit-exact against goldens produced by the reference's utils.py."""
import prepost_cases as cases


def test_preprocess_emulated_bit_exact(emu_engine):
    assert cases.check_preprocess(emu_engine) >= 8


def test_preprocess_emulated_float_volumes(emu_engine):
    cases.check_preprocess_float(emu_engine)


def test_reshape_mask_emulated_bit_exact(emu_engine):
    assert cases.check_reshape(emu_engine) >= 10


def test_postprocessing_emulated_bit_exact(emu_engine):
    assert cases.check_postprocess(emu_engine) >= 20


def test_postprocessing_emulated_random_differential(emu_engine):
    cases.check_postprocess_random(emu_engine, seeds=range(4))


def test_postprocessing_emulated_diagonal_adversarial(emu_engine):
    cases.check_postprocess_diagonal_adversarial(emu_engine)


def test_postprocessing_emulated_wide_rows(emu_engine):
    cases.check_postprocess_wide_rows(emu_engine)


def test_postprocessing_emulated_noise_volume(emu_engine):
    cases.check_postprocess_noise(emu_engine)


def test_empty_inputs_emulated(emu_engine):
    cases.check_empty_inputs(emu_engine)


def test_reference_utils_tests_emulated(emu_engine):
    """The reference's own tests of utils.preprocess / simple_bodymask / crop_and_resize / reshape_mask / postprocessing,
    through the `lungmask_amd.utils` mirror."""
    from reference_utils_cases import check_reference_utils_tests

    check_reference_utils_tests(emu_engine)


def test_bbox_3d_and_keep_largest_emulated(emu_engine):
    cases.check_bbox_klc(emu_engine)


def test_fusion_emulated(emu_engine):
    cases.check_fuse(emu_engine)


def test_reorient_emulated(emu_engine):
    cases.check_reorient(emu_engine)


def test_slab_sharded_postprocessing_emulated(emu_engine):
    """The multi-GPU form of the post-processing with up to 4 in-process ranks on the emulator."""
    from lungmask_amd import _native as nat

    extra = [nat.Engine(0, emu_engine.L) for _ in range(3)]
    try:
        assert cases.check_slab_postprocess([emu_engine] + extra) >= 60
    finally:
        for e in extra:
            e.close()


def test_slab_protocol_region_graph_adversarial_emulated(emu_engine):
    from lungmask_amd import _native as nat

    extra = [nat.Engine(0, emu_engine.L) for _ in range(2)]
    try:
        cases.check_slab_postprocess_diagonal_adversarial([emu_engine] + extra, n_iter=12)
    finally:
        for e in extra:
            e.close()


def test_slab_protocol_region_graph_form_emulated():
    """LM_SLAB_GRAPH=1: the four-exchange form of the slab protocol (second labelling on the region graph inside the first table
    merge: exact, measured slower than the voxel form from four ranks on, hence opt-in) stays exact.  Own process: the switch is read once."""
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import prepost_cases as cases\n"
            "from lungmask_amd import _native as nat\n"
            "from lungmask_amd.build import build_emu\n"
            "from lungmask_amd.pipeline import postprocess_slabs_in_process\n"
            "L = nat.Library(build_emu(), allow_emulation=True)\n"
            "engs = [nat.Engine(0, L) for _ in range(3)]\n"
            "assert cases.check_slab_postprocess(engs, seeds=range(2)) >= 40\n"
            "assert postprocess_slabs_in_process.last_rounds == 4\n"
            "cases.check_slab_postprocess_diagonal_adversarial(engs, n_iter=24)\n"
            "print('graph form ok')\n") % (os.path.dirname(here), here)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LM_SLAB_GRAPH="1", OMP_NUM_THREADS="4"),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "graph form ok" in r.stdout, r.stdout[-2000:]


def test_postprocessing_table_growth_paths_emulated():
    """The post-processing sizes its region / record tables from the previous volume and fetches a guessed prefix of them in the
    same read-back as the counts (post_engine.hip).  With LM_POST_SMALL_TABLES=1 every table and guess starts tiny, so the golden and
    random cases run through the grow-and-repeat and the second-copy paths -- results must not change.  (Own process: the hook is
    read once per process.)"""
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import prepost_cases as cases\n"
            "from lungmask_amd import _native as nat\n"
            "from lungmask_amd.build import build_emu\n"
            "eng = nat.Engine(0, nat.Library(build_emu(), allow_emulation=True))\n"
            "assert cases.check_postprocess(eng) >= 20\n"
            "cases.check_postprocess_random(eng, seeds=range(2))\n"
            "cases.check_postprocess_noise(eng)\n"
            "print('small tables ok', eng.postprocess_info())\n") % (os.path.dirname(here), here)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LM_POST_SMALL_TABLES="1", OMP_NUM_THREADS="4"),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "small tables ok" in r.stdout, r.stdout[-2000:]


def test_apply_host_failure_leaves_output_untouched_emulated(emu_engine):
    cases.check_apply_host_failure_leaves_output_untouched(emu_engine)
