"""Shared parity checks of the pre/post kernels against the reference-generated
goldens (tests/golden/prepost.npz) -- used by both the emulator (CPU) and the
GPU test modules so the two suites read the same."""
import os

import numpy as np

from oracle import prepost_oracle as po

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "prepost.npz")


def check_preprocess(eng, max_pixels=None):
    g = np.load(GOLD)
    n_checked = 0
    for i in range(int(g["n_pre"])):
        vol = g[f"pre{i}_vol"]
        res = [int(x) for x in g[f"pre{i}_res"]]
        if max_pixels and vol[0].size > max_pixels:
            continue
        xi, xf, bb, bm = eng.preprocess(vol, resolution=res, want_bmask=True)
        assert np.array_equal(bb, g[f"pre{i}_box"]), (i, bb.tolist(), g[f"pre{i}_box"].tolist())
        assert np.array_equal(np.packbits(bm.astype(bool), axis=-1), g[f"pre{i}_bmask"]), i
        assert np.array_equal(xi, g[f"pre{i}_x"]), (i, int((xi != g[f"pre{i}_x"]).sum()))
        assert np.array_equal(xf, po.normalise(g[f"pre{i}_x"])), i
        n_checked += 1
    return n_checked


def check_preprocess_float(eng):
    """float32 / float64 volumes (numpy mode of mask.py:153-155 accepts any dtype): differential vs the oracle."""
    rng = np.random.default_rng(5)
    base = po.phantom(2, 300, 420, seed=9).astype(np.float64)
    base += rng.normal(0, 0.37, size=base.shape)  # genuinely fractional HU values
    for dt in (np.float32, np.float64):
        vol = base.astype(dt)
        xi, xf, bb, _ = eng.preprocess(vol)
        ref_x, ref_bb = po.preprocess(vol, [256, 256])
        assert xi is None and ref_x.dtype == dt
        assert np.array_equal(bb, np.asarray(ref_bb, dtype=np.int32)), dt
        assert np.array_equal(xf, po.normalise(ref_x)), (dt, float(np.abs(xf - po.normalise(ref_x)).max()))


def check_reshape(eng):
    g = np.load(GOLD)
    for i in range(int(g["n_rs"])):
        out = eng.reshape_mask(g[f"rs{i}_mask"], g[f"rs{i}_box"], tuple(int(x) for x in g[f"rs{i}_osz"]))[0]
        assert np.array_equal(out, g[f"rs{i}_out"]), i
    # stacks of slices with a box each, against the oracle: the row-tiled kernel (width a multiple of 4; 100 rows = 6 full row
    # groups + one of 4) and the voxel-wise one
    rng = np.random.default_rng(17)
    for osz in ((100, 132), (50, 90)):
        masks = rng.integers(0, 4, size=(5, 64, 64)).astype(np.uint8)
        boxes = []
        for _ in range(5):
            r0, c0 = int(rng.integers(0, osz[0] // 2)), int(rng.integers(0, osz[1] // 2))
            boxes.append([r0, c0, int(rng.integers(r0 + 2, osz[0] + 1)), int(rng.integers(c0 + 2, osz[1] + 1))])
        boxes[0] = [0, 0, osz[0], osz[1]]
        out = eng.reshape_mask(masks, np.asarray(boxes), osz)
        for i in range(5):
            assert np.array_equal(out[i], po.reshape_mask(masks[i], boxes[i], osz)), (osz, i, boxes[i])
    return int(g["n_rs"])


def check_postprocess(eng, max_voxels=None):
    g = np.load(GOLD)
    n_checked = 0
    for i in range(int(g["n_post"])):
        lab = g[f"post{i}_lab"]
        if max_voxels and lab.size > max_voxels:
            continue
        spare = [int(x) for x in g[f"post{i}_spare"]]
        out = eng.postprocess(lab, spare=spare, skip_below=int(g[f"post{i}_skip"]))
        assert np.array_equal(out, g[f"post{i}_out"]), (i, lab.shape, spare, int((out != g[f"post{i}_out"]).sum()))
        n_checked += 1
    return n_checked


def check_postprocess_random(eng, seeds, shape=(7, 40, 36), nlab=4):
    """Differential test against the oracle on seeded random blob volumes."""
    from oracle.make_golden import random_blobs

    for seed in seeds:
        rng = np.random.default_rng(seed)
        lab = random_blobs(rng, shape, nlab, 18, 0.3)
        for spare, skip in (((), 3), ((nlab,), 3), ((), 1)):
            out = eng.postprocess(lab, spare=spare, skip_below=skip)
            ref = po.postprocessing(lab.copy(), spare=list(spare), skip_below=skip)
            assert np.array_equal(out, ref), (seed, spare, skip, int((out != ref).sum()))


def check_postprocess_wide_rows(eng):
    """Rows longer than one 256-voxel piece of the row-wise labelling kernels (ccl_*_rows_kernel): unions across the piece
    boundaries, a last piece of 8 voxels, and a width that is not a multiple of 4 (the voxel-wise kernels)."""
    from oracle.make_golden import random_blobs

    for seed, shape in ((21, (3, 6, 520)), (22, (2, 5, 768)), (23, (3, 6, 258))):
        rng = np.random.default_rng(seed)
        lab = random_blobs(rng, shape, 3, 14, 0.2)
        lab[:, 1:4, 200:330] = 2  # a solid bar through the boundaries at x = 256 (and 512 where it reaches)
        lab[0, 2, 250:262] = 0    # ... with a hole on one of them
        lab[-1, :, -9:] = 1
        for spare, skip in (((), 3), ((), 1)):
            out = eng.postprocess(lab, spare=spare, skip_below=skip)
            ref = po.postprocessing(lab.copy(), spare=list(spare), skip_below=skip)
            assert np.array_equal(out, ref), (shape, skip, int((out != ref).sum()))


def check_fuse(eng):
    from oracle.make_golden import random_blobs

    rng = np.random.default_rng(11)
    res_l = random_blobs(rng, (6, 40, 40), 5, 20, 0.3)
    res_r = (random_blobs(rng, (6, 40, 40), 2, 14, 0.35)).astype(np.uint8)
    fused, spare = eng.fuse(res_l, res_r)
    ref = res_l.copy()
    sv = ref.max() + 1
    ref[np.logical_and(ref == 0, res_r > 0)] = sv
    ref[res_r == 0] = 0
    assert spare == int(sv) and np.array_equal(fused, ref)
    out = eng.postprocess(fused, spare=[spare])
    assert np.array_equal(out, po.fuse(res_l, res_r))


def check_reorient(engine):
    """lm_reorient_dev == a.transpose(axes) with per-axis flips (the device form of sitk.DICOMOrient, mask.py:156-164,204-208)."""
    import itertools

    from lungmask_amd import volume_io as vio

    rng = np.random.default_rng(5)
    for dt in (np.uint8, np.int16, np.float32, np.float64):
        a = rng.integers(0, 250, (3, 5, 7)).astype(dt)
        d = engine.to_device(a)
        for axes in itertools.permutations(range(3)):
            for flips in itertools.product((False, True), repeat=3):
                o = engine.reorient_dev(d, axes, flips)
                engine.sync()
                got = o.download()
                o.free()
                assert np.array_equal(got, vio.apply_transform(a, axes, flips)), (dt, axes, flips)
                assert np.array_equal(vio.apply_transform(got, *vio.inverse_transform(axes, flips)), a)
        d.free()


def check_slab_postprocess(engines, shapes=((9, 24, 20), (5, 16, 16)), seeds=range(3), golden_max_voxels=40000):
    """Slab-sharded post-processing (lm_slab_*, csrc/slab_engine.hip) with 1..len(engines) in-process ranks and every kind of
    cut (ragged, one-slice slabs) == the oracle == the whole-volume path.  Also the reference-generated goldens."""
    from lungmask_amd.pipeline import postprocess_slabs_in_process, shard_bounds
    from oracle.make_golden import random_blobs

    n_checked = 0
    for shape in shapes:
        for seed in seeds:
            rng = np.random.default_rng(100 + seed)
            nlab = 4
            lab = random_blobs(rng, shape, nlab, 10, 0.3)
            for spare, skip in (((), 3), ((nlab,), 3), ((), 1)):
                ref = po.postprocessing(lab.copy(), spare=list(spare), skip_below=skip)
                for world in range(1, len(engines) + 1):
                    cuts = [shard_bounds(shape[0], world)]
                    if world == 2:
                        cuts += [[0, 1, shape[0]], [0, shape[0] - 1, shape[0]]]
                    if world == 3:
                        cuts += [[0, 1, 2, shape[0]]]
                    for b in cuts:
                        out = postprocess_slabs_in_process(engines[:world], lab, b, spare, skip)
                        assert np.array_equal(out, ref), (shape, seed, spare, skip, b, int((out != ref).sum()))
                        n_checked += 1
    # salt-and-pepper volumes: thousands of tiny regions, most boundary records cross a slab face
    for seed in seeds:
        rng = np.random.default_rng(200 + seed)
        lab = rng.integers(0, 4, (6, 14, 12)).astype(np.uint8)
        lab[rng.random(lab.shape) < 0.35] = 0
        ref = po.postprocessing(lab.copy())
        for world in range(2, len(engines) + 1):
            out = postprocess_slabs_in_process(engines[:world], lab, shard_bounds(6, world))
            assert np.array_equal(out, ref), (seed, world, int((out != ref).sum()))
            n_checked += 1
    # a hollow shell cut by the slab faces: the cavity is a hole only when all its slab pieces are enclosed
    lab = np.zeros((8, 12, 12), np.uint8)
    lab[1:7, 2:10, 2:10] = 1
    lab[2:6, 4:8, 4:8] = 0
    lab2 = lab.copy()
    lab2[4, 5, 2:5] = 0  # a tunnel from the cavity to the outside, inside ONE slab
    for v in (lab, lab2):
        ref = po.postprocessing(v.copy())
        for world in range(2, len(engines) + 1):
            out = postprocess_slabs_in_process(engines[:world], v, shard_bounds(8, world))
            assert np.array_equal(out, ref), (world, int((out != ref).sum()))
            n_checked += 1
    assert po.postprocessing(lab.copy())[3, 5, 5] == 1 and po.postprocessing(lab2.copy())[3, 5, 5] == 0
    # nothing at all (no atoms, no labels: every exchange is empty), and foreground confined to ONE slab
    empty = np.zeros((6, 10, 10), np.uint8)
    single = empty.copy()
    single[4, 2:8, 2:8] = 2
    single[4, 4:6, 4:6] = 0
    for v in (empty, single):
        ref = po.postprocessing(v.copy())
        for world in range(2, len(engines) + 1):
            out = postprocess_slabs_in_process(engines[:world], v, shard_bounds(6, world))
            assert np.array_equal(out, ref), (world, int((out != ref).sum()))
            n_checked += 1
    g = np.load(GOLD)
    for i in range(int(g["n_post"])):
        lab = g[f"post{i}_lab"]
        if lab.shape[0] < 2 or lab.size > golden_max_voxels:
            continue
        spare = [int(x) for x in g[f"post{i}_spare"]]
        world = min(len(engines), lab.shape[0])
        out = postprocess_slabs_in_process(engines[:world], lab, shard_bounds(lab.shape[0], world), spare, int(g[f"post{i}_skip"]))
        assert np.array_equal(out, g[f"post{i}_out"]), (i, lab.shape, spare)
        n_checked += 1
    return n_checked


def check_postprocess_noise(eng):
    """(1) dense label noise (a few percolating regions); (2) a lattice in which EVERY voxel is a region of its own with
    four foreign face neighbours: one distinct boundary record per voxel, so the per-workgroup record table of
    boundary_records_kernel (512 slots per 2048 voxels) overflows and most records take the bypass path."""
    rng = np.random.default_rng(3)
    lab = rng.integers(0, 4, size=(6, 48, 48)).astype(np.uint8)
    assert np.array_equal(eng.postprocess(lab), po.postprocessing(lab.copy()))
    lat = np.zeros((5, 32, 32), np.uint8)
    yy, xx = np.mgrid[0:32, 0:32]
    for z in (0, 2, 4):
        lat[z] = 1 + (xx % 2) + 2 * (yy % 2)
    lat[2, 8:12, 8:12] = 1  # one larger region for the small ones to merge into
    for skip in (1, 3):
        out = eng.postprocess(lat, skip_below=skip)
        info = eng.postprocess_info()
        assert info["regions"] > 3000 and info["boundary_records"] > 512 * ((lat.size + 2047) // 2048), info
        assert np.array_equal(out, po.postprocessing(lat.copy(), skip_below=skip)), skip


def check_empty_inputs(eng):
    """Zero-slice volumes through every stage (the reference's loops simply do not iterate): empty outputs, no error."""
    assert eng.postprocess(np.zeros((0, 8, 8), np.uint8)).shape == (0, 8, 8)
    xi, xf, bb, _ = eng.preprocess(np.zeros((0, 64, 64), np.int16), resolution=(32, 32))
    assert xi.shape == (0, 32, 32) and xf.shape == (0, 32, 32) and bb.shape == (0, 4)
    assert eng.reshape_mask(np.zeros((0, 32, 32), np.uint8), np.zeros((0, 4), np.int32), (64, 64)).shape == (0, 64, 64)
    fused, spare = eng.fuse(np.zeros((0, 8, 8), np.uint8), np.zeros((0, 8, 8), np.uint8))
    assert fused.shape == (0, 8, 8) and spare == 1


def check_apply_host_failure_leaves_output_untouched(eng):
    """lm_apply_host faults the caller's output pages in on a helper thread while the network runs; a call that fails (here: an
    empty model slot, detected after the helper has started) must leave the array exactly as it was, and the engine usable."""
    from lungmask_amd._native import LMError

    vol = po.phantom(45, 96, 80, seed=3)  # more than two batches of 20: the split (head / tail) path
    out = np.full(vol.shape, 0xAB, dtype=np.uint8)
    try:
        eng.apply(3, vol, out=out)
    except LMError as ex:
        assert "slot 3 is empty" in str(ex)
    else:
        raise AssertionError("an empty model slot must be an error")
    assert (out == 0xAB).all()
    eng.sync()


def check_postprocess_diagonal_adversarial(eng, n_iter=60):
    """The second labelling runs on the region graph (csrc/post_engine.hip: graph_components): components of the mapped volume are
    unions of first-pass regions joined by 6-adjacency (boundary records) and by the diagonal rest of the 26-adjacency
    (diag_pairs kernels, row-wise and voxel-wise).  Volumes made to stress exactly that -- pure label noise, blobs with noise,
    volumes without a background voxel, diagonal chains -- against the oracle, with and without a spare label, widths that are
    and are not multiples of 4."""
    from oracle.make_golden import random_blobs

    rng = np.random.default_rng(123)
    shapes = [(5, 12, 12), (4, 10, 14), (6, 9, 11), (3, 16, 8), (2, 7, 9)]
    for it in range(n_iter):
        shape = shapes[it % 5]
        kind = it % 4
        if kind == 0:
            lab = rng.integers(0, 4, shape).astype(np.uint8)
        elif kind == 1:
            lab = random_blobs(rng, shape, 3, 6, 0.4)
            lab[rng.random(shape) < 0.15] = rng.integers(1, 4)
        elif kind == 2:
            lab = rng.integers(1, 4, shape).astype(np.uint8)
        else:
            lab = np.zeros(shape, np.uint8)
            zz, yy, xx = np.indices(shape)
            m = (zz + yy + xx) % 2 == 0
            lab[m] = (1 + ((zz + 2 * yy + 3 * xx) % 3 == 0))[m]
            lab[rng.random(shape) < 0.1] = 3
        for spare, skip in (((), 3), ((3,), 3), ((), 1)):
            out = eng.postprocess(lab, spare=list(spare), skip_below=skip)
            ref = po.postprocessing(lab.copy(), spare=list(spare), skip_below=skip)
            assert np.array_equal(out, ref), (it, shape, kind, spare, skip, int((out != ref).sum()))


def check_slab_postprocess_diagonal_adversarial(engines, n_iter=24):
    """The slab protocol in either form -- the default (second labelling as voxel passes) and, with LM_SLAB_GRAPH=1, the region-graph
    form (csrc/slab_engine.hip: the second labelling on the region graph -- 6-adjacency from the boundary records incl. halo
    neighbours, diagonal pairs inside a slab, all 26-adjacent atom pairs across a slab face): the
    volumes of check_postprocess_diagonal_adversarial (label noise, volumes without a background voxel, diagonal lattices) cut into
    slabs at every position, with and without a spare label, against the oracle."""
    import os

    from lungmask_amd.pipeline import postprocess_slabs_in_process, shard_bounds
    from oracle.make_golden import random_blobs

    rng = np.random.default_rng(321)
    shapes = [(5, 12, 12), (4, 10, 14), (6, 9, 11), (3, 16, 8), (2, 7, 9)]
    for it in range(n_iter):
        shape = shapes[it % 5]
        kind = it % 4
        if kind == 0:
            lab = rng.integers(0, 4, shape).astype(np.uint8)
        elif kind == 1:
            lab = random_blobs(rng, shape, 3, 6, 0.4)
            lab[rng.random(shape) < 0.15] = rng.integers(1, 4)
        elif kind == 2:
            lab = rng.integers(1, 4, shape).astype(np.uint8)
        else:
            lab = np.zeros(shape, np.uint8)
            zz, yy, xx = np.indices(shape)
            m = (zz + yy + xx) % 2 == 0
            lab[m] = (1 + ((zz + 2 * yy + 3 * xx) % 3 == 0))[m]
            lab[rng.random(shape) < 0.1] = 3
        for spare, skip in (((), 3), ((3,), 3), ((), 1)):
            ref = po.postprocessing(lab.copy(), spare=list(spare), skip_below=skip)
            for world in range(2, min(len(engines), shape[0]) + 1):
                cuts = [shard_bounds(shape[0], world)]
                if world == 2:
                    cuts += [[0, c, shape[0]] for c in range(1, shape[0]) if [0, c, shape[0]] != cuts[0]]
                for b in cuts:
                    out = postprocess_slabs_in_process(engines[:world], lab, b, spare, skip)
                    assert np.array_equal(out, ref), (it, shape, kind, spare, skip, b, int((out != ref).sum()))
                    assert postprocess_slabs_in_process.last_rounds == (4 if os.environ.get("LM_SLAB_GRAPH") == "1" else 6)


def check_bbox_klc(eng):
    """utils.bbox_3D (utils.py:361-387) and utils.keep_largest_connected_component (utils.py:390-404) as calls of their own:
    every `klc*` golden (outputs of the reference's own functions, oracle/_ref_runner.py), the reference's test vector
    (tests/test_utils.py:58-63), differential cases against the oracle, the equal-area tie, the empty mask."""
    from lungmask_amd import utils

    utils.set_engine(eng)
    try:
        g = np.load(GOLD)
        for i in range(int(g["n_klc"])):
            m = g[f"klc{i}_mask"]
            out = utils.keep_largest_connected_component(m)
            assert out.dtype == bool and out.shape == m.shape
            assert np.array_equal(np.packbits(out), g[f"klc{i}_out"]), i
            bb = utils.bbox_3D(m)
            assert np.array_equal(bb, g[f"klc{i}_bbox"]), (i, bb.tolist())
        # tests/test_utils.py:58-63
        m = np.zeros((10, 10, 10), dtype=np.uint8)
        m[2:8, 3:7, 4:6] = 1
        assert tuple(utils.bbox_3D(m, margin=2)) == (0, 10, 1, 9, 2, 8)
        assert tuple(utils.bbox_3D(m, margin=0)) == (2, 8, 3, 7, 4, 6)
        # differential: ragged shapes (rows that are no multiple of the 8-voxel words), label maps, 2-D masks
        rng = np.random.default_rng(23)
        for shape in ((5, 13, 11), (3, 32, 40), (1, 21, 19), (9, 7, 5)):
            for p in (0.02, 0.3, 0.55):
                m = rng.random(shape) < p
                if not m.any():
                    continue
                for margin in (0, 2, 5):
                    assert np.array_equal(utils.bbox_3D(m, margin=margin), po.bbox_3D(m, margin=margin)), (shape, p, margin)
                ref = po.keep_largest_connected_component(m)
                areas = np.bincount(po.sk_label(m).ravel())[1:]
                if (areas == areas.max()).sum() == 1:  # (ties: below)
                    assert np.array_equal(utils.keep_largest_connected_component(m), ref), (shape, p)
            lab = rng.integers(0, 4, size=shape).astype(np.uint8)  # different non-zero values are different regions
            areas = np.bincount(po.sk_label(lab).ravel())[1:]
            if (areas == areas.max()).sum() == 1:
                assert np.array_equal(utils.keep_largest_connected_component(lab), po.keep_largest_connected_component(lab)), shape
        m2 = rng.random((17, 23)) < 0.4
        assert np.array_equal(utils.bbox_3D(m2), po.bbox_3D(m2)) and utils.bbox_3D(m2).shape == (4,)
        assert np.array_equal(utils.keep_largest_connected_component(m2), po.keep_largest_connected_component(m2))
        # equal areas: the region whose first voxel comes LAST in raster order (the stable reading of np.argsort(...)[-1]; the
        # reference leaves ties to numpy's unstable default sort) -- also across different label values
        t = np.zeros((4, 12, 12), np.uint8)
        t[0, 1:3, 1:3] = 1
        t[2, 5:7, 5:7] = 1
        t[3, 9:11, 1:3] = 1
        want = np.zeros_like(t, bool)
        want[3, 9:11, 1:3] = True
        assert np.array_equal(utils.keep_largest_connected_component(t), want)
        assert np.array_equal(po.keep_largest_connected_component(t), want)
        t[3, 9:11, 1:3] = 0
        t[1, 9:11, 9:11] = 2
        want[:] = False
        want[2, 5:7, 5:7] = True
        assert np.array_equal(utils.keep_largest_connected_component(t), want)
        # no region at all: IndexError, as utils.py:377 / :402
        for fn in (utils.bbox_3D, utils.keep_largest_connected_component):
            try:
                fn(np.zeros((3, 8, 8), np.uint8))
                raise AssertionError("an empty mask must raise IndexError like the reference")
            except IndexError:
                pass
    finally:
        utils.set_engine(None)
