"""Generates tests/golden/*.npz from the REFERENCE implementation (dev
container only; needs /root/reference and /opt/conda/bin/python3.9).

  python oracle/make_golden.py [unet] [prepost] [testvol] [--out DIR]

(--out DIR: write there instead of tests/golden -- tests/test_oracle.py::test_goldens_regenerate uses it to check that the
recipe still reproduces the committed fixtures array for array, bit for bit, whenever /root/reference is mounted.)

* unet_c{3,6}.npz  : reference `resunet.UNet` (mask.py:58-65 configuration)
                     on `oracle.unet_oracle.synthetic_state_dict(C)`;
                     log-probs (subsampled) + labels for seeded inputs.
* prepost.npz      : reference utils.py (unmodified, run by conda py3.9 via
                     oracle/_ref_runner.py) on seeded inputs.
TEST INFRASTRUCTURE ONLY.
"""
import importlib.util
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
GOLD = os.path.join(ROOT, "tests", "golden")


def random_blobs(rng, shape, nlab, nblob, rmax):
    """Seeded label volume: overlapping ellipsoid blobs + specks."""
    lab = np.zeros(shape, dtype=np.uint8)
    grids = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
    for _ in range(nblob):
        c = [rng.uniform(0, s) for s in shape]
        r = [rng.uniform(1.0, max(1.5, rmax * s)) for s in shape]
        d = sum(((g - ci) / ri) ** 2 for g, ci, ri in zip(grids, c, r))
        lab[d <= 1.0] = rng.integers(1, nlab + 1)
    nspeck = int(0.01 * lab.size)
    idx = tuple(rng.integers(0, s, nspeck) for s in shape)
    lab[idx] = rng.integers(0, nlab + 1, nspeck)
    return lab


def ct_like(rng, h, w, n=1):
    """Seeded CT-like slices: body ellipse, two lungs, table line, noise."""
    from oracle.prepost_oracle import phantom

    vol = phantom(n, h, w, seed=int(rng.integers(1 << 30))).astype(np.int32)
    # a patient table below the body and a detached object (exercise largest-CC/bbox)
    vol[:, int(0.93 * h) : int(0.95 * h), int(0.1 * w) : int(0.9 * w)] = 200
    vol[:, 2:6, 2:9] = 300
    return np.clip(vol, -2048, 3071).astype(np.int16)


def make_unet():
    import torch
    from oracle import unet_oracle as uo

    spec = importlib.util.spec_from_file_location("ref_resunet", "/root/reference/lungmask/resunet.py")
    R = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(R)
    from oracle.prepost_oracle import phantom, preprocess, normalise

    for C in (3, 6):
        net = R.UNet(n_classes=C, padding=True, depth=5, up_mode="upsample", batch_norm=True, residual=False)
        sd = uo.synthetic_state_dict(C)
        net.load_state_dict(sd)  # strict: same keys as the real .pth
        net.eval()
        out = {}
        g = torch.Generator().manual_seed(1000 + C)
        cases = {
            "rand64": torch.rand(2, 1, 64, 64, generator=g),
            "rand32": torch.rand(3, 1, 32, 32, generator=g),
        }
        ph = phantom(2, 512, 512)
        xs, _ = preprocess(ph, resolution=[256, 256])
        cases["phantom256"] = torch.from_numpy(normalise(xs)[:, None])
        with torch.inference_mode():
            for name, x in cases.items():
                y = net(x)
                out[f"{name}_x"] = x.numpy()
                # full log-probs for the small cases, 4x4-subsampled for 256x256
                out[f"{name}_logp"] = (y if x.shape[-1] <= 64 else y[:, :, ::4, ::4]).numpy()
                out[f"{name}_lab"] = torch.max(y, 1)[1].numpy().astype(np.uint8)
                srt = torch.sort(y, dim=1, descending=True)[0]
                out[f"{name}_margin"] = (srt[:, 0] - srt[:, 1]).numpy().astype(np.float16)
        np.savez_compressed(os.path.join(GOLD, f"unet_c{C}.npz"), **out)
        print("unet golden", C, {k: v.shape for k, v in out.items()})


def make_prepost():
    rng = np.random.default_rng(77)
    inp = {}
    pre = []
    # the reference's own synthetic case (tests/test_utils.py:91-99)
    img = np.full((2, 10, 10), dtype=np.int16, fill_value=-1000)
    img[:, 2:8, 3:7] = 1
    img[:, 9, 9] = 1
    pre.append((img, [20, 20]))
    pre.append((ct_like(rng, 512, 512, 2), [256, 256]))
    pre.append((ct_like(rng, 300, 420, 1), [256, 256]))
    pre.append((ct_like(rng, 100, 90, 1), [256, 256]))
    pre.append((ct_like(rng, 768, 640, 1), [256, 256]))
    # real CT slice of the reference's fixture (raw int16 at byte offset 910)
    raw = np.frombuffer(open("/root/reference/tests/testdata/0.dcm", "rb").read()[910 : 910 + 512 * 512 * 2], dtype="<i2")
    pre.append((raw.reshape(1, 512, 512).copy(), [256, 256]))
    # empty slice (no body) and all-body slice
    pre.append((np.full((1, 64, 64), -1000, dtype=np.int16), [256, 256]))
    pre.append((np.full((1, 200, 200), 50, dtype=np.int16), [256, 256]))
    inp["n_pre"] = len(pre)
    for i, (v, r) in enumerate(pre):
        inp[f"pre{i}_vol"] = v
        inp[f"pre{i}_res"] = np.asarray(r)
    post = []
    li = np.zeros((1, 6, 6), dtype=np.uint8)
    li[0] = np.asarray([[0, 0, 0, 0, 0, 0], [0, 1, 1, 2, 2, 0], [0, 2, 0, 3, 1, 0], [0, 4, 4, 4, 0, 0], [0, 4, 0, 4, 0, 0], [0, 4, 4, 4, 0, 0]])
    t = np.tile(li, (2, 1, 1))
    post += [(t, [], 1), (t, [3], 1), (t, [3], 3)]  # tests/test_utils.py:124-159
    for shape, nlab, nblob in (((6, 24, 20), 2, 10), ((10, 40, 36), 5, 25), ((16, 48, 48), 3, 30), ((5, 64, 64), 6, 40), ((12, 32, 32), 2, 6)):
        lab = random_blobs(rng, shape, nlab, nblob, 0.3)
        post.append((lab, [], 3))
        post.append((lab, [], 1))
        post.append((lab, [nlab], 3))
    # hole-filling / largest-CC stress: hollow boxes
    hb = np.zeros((9, 20, 20), dtype=np.uint8)
    hb[1:8, 2:12, 2:12] = 1
    hb[3:6, 4:9, 4:9] = 0
    hb[2:5, 14:18, 14:18] = 2
    hb[1, 16, 1] = 1
    post.append((hb, [], 3))
    # single slice volume (area_closing path)
    ss = random_blobs(rng, (1, 64, 64), 3, 12, 0.3)
    post.append((ss, [], 3))
    # volumes WITHOUT a background voxel: `np.unique(outmask_mapped)[1:]` (utils.py:355) then drops the smallest LABEL, not the 0
    # (own generator: the draws of the cases below stay what they were)
    rng_nb = np.random.default_rng(78)
    nb = random_blobs(rng_nb, (6, 24, 24), 3, 14, 0.4)
    nb[nb == 0] = 1
    post += [(nb, [], 3), (nb, [3], 3)]
    nb2 = random_blobs(rng_nb, (5, 20, 28), 2, 9, 0.4)
    nb2[nb2 == 0] = 2
    post.append((nb2, [], 3))
    nb1 = random_blobs(rng_nb, (1, 40, 40), 3, 10, 0.4)
    nb1[nb1 == 0] = 3
    post.append((nb1, [], 3))
    inp["n_post"] = len(post)
    for i, (lab, spare, skip) in enumerate(post):
        inp[f"post{i}_lab"] = lab
        inp[f"post{i}_spare"] = np.asarray(spare, dtype=np.int64)
        inp[f"post{i}_skip"] = skip
    rs = []
    m = random_blobs(rng, (256, 256), 3, 20, 0.25)
    for (y0, x0, hh, ww, oh, ow) in ((2, 2, 20, 20, 30, 30), (95, 3, 414, 506, 512, 512), (0, 0, 512, 512, 512, 512), (5, 7, 23, 27, 40, 40),
                                     (1, 1, 32, 45, 64, 64), (3, 0, 53, 56, 60, 60), (0, 2, 62, 63, 70, 70), (10, 20, 82, 89, 128, 128), (0, 0, 300, 256, 300, 420)):
        rs.append((m, [y0, x0, y0 + hh, x0 + ww], (oh, ow)))
    rs.append((np.full((10, 10), 1, dtype=np.uint8), [2, 2, 22, 22], (30, 30)))  # tests/test_utils.py:102-107
    inp["n_rs"] = len(rs)
    for i, (mm, box, osz) in enumerate(rs):
        inp[f"rs{i}_mask"] = mm
        inp[f"rs{i}_box"] = np.asarray(box)
        inp[f"rs{i}_osz"] = np.asarray(osz)
    klc = [random_blobs(rng, (8, 30, 30), 1, 8, 0.2) > 0, random_blobs(rng, (4, 50, 40), 1, 14, 0.15) > 0]
    mm = np.zeros((10, 10, 10), dtype=np.uint8)
    mm[2:8, 3:7, 4:6] = 1
    klc.append(mm > 0)
    inp["n_klc"] = len(klc)
    for i, k in enumerate(klc):
        inp[f"klc{i}_mask"] = k
    ac = [(random_blobs(rng, (64, 64), 1, 14, 0.2) > 0).astype(np.uint8), (random_blobs(rng, (40, 90), 1, 25, 0.15) > 0).astype(np.uint8)]
    inp["n_ac"] = len(ac)
    for i, a in enumerate(ac):
        inp[f"ac{i}_img"] = a
    with tempfile.TemporaryDirectory() as td:
        fi, fo = os.path.join(td, "in.npz"), os.path.join(td, "out.npz")
        np.savez(fi, **inp)
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONWARNINGS="ignore")
        subprocess.run(["/opt/conda/bin/python3.9", os.path.join(ROOT, "oracle", "_ref_runner.py"), fi, fo], check=True, env=env)
        out = dict(np.load(fo))
    out.update(inp)
    np.savez_compressed(os.path.join(GOLD, "prepost.npz"), **out)
    print("prepost golden:", len(out), "arrays")


def make_testvol():
    """The reference's end-to-end fixture volume (tests/test_mask.py:12-14: read_dicoms(tests/testdata)[0], two 512x512 CT
    slices) as a small .npz, so that the golden voxel-count tests of tests/test_mask.py:36,47,59 can run on a GPU box
    (which has no /root/reference) as soon as the pretrained weights are provided.  pydicom / SimpleITK are absent here, so
    the slices are read by lungmask_amd.volume_io (the same series logic, utils.py:132-230); the raw pixel block of each file
    (int16 at byte offset 910, tests/test_utils.py:18-55 wrote them) is checked against what the reader returns."""
    from lungmask_amd import volume_io

    td = "/root/reference/tests/testdata"
    vol = volume_io.read_dicoms(td)[0]
    raws = [np.frombuffer(open(os.path.join(td, f), "rb").read()[910 : 910 + 512 * 512 * 2], dtype="<i2").reshape(512, 512) for f in ("0.dcm", "1.dcm")]
    assert vol.array.shape == (2, 512, 512)
    assert all(any(np.array_equal(vol.array[i], r) for r in raws) for i in range(2))  # rescale slope 1 / intercept 0
    np.savez_compressed(os.path.join(GOLD, "testvol.npz"), vol=vol.array, direction=np.asarray(vol.direction, dtype=np.float64),
                        spacing=np.asarray(vol.spacing, dtype=np.float64))
    print("testvol golden:", vol.array.shape, vol.array.dtype)


if __name__ == "__main__":
    argv = sys.argv[1:]
    if "--out" in argv:
        i = argv.index("--out")
        GOLD = os.path.abspath(argv[i + 1])
        os.makedirs(GOLD, exist_ok=True)
        del argv[i : i + 2]
    which = argv or ["unet", "prepost", "testvol"]
    if "testvol" in which:
        make_testvol()
    if "prepost" in which:
        make_prepost()
    if "unet" in which:
        make_unet()
