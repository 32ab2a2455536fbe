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


def check_reshape(eng):
    g = np.load(GOLD)
    for i in range(int(g["n_rs"])):
        out = eng.reshape_mask(g[f"rs{i}_mask"], g[f"rs{i}_box"], tuple(int(x) for x in g[f"rs{i}_osz"]))[0]
        assert np.array_equal(out, g[f"rs{i}_out"]), i
    return int(g["n_rs"])
