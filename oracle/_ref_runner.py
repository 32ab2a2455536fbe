"""Runs under /opt/conda/bin/python3.9 (scikit-image 0.18.3, scipy 1.7.1):
executes the reference's UNMODIFIED lungmask/utils.py on the inputs in an .npz
and writes its outputs.  Invoked only by oracle/make_golden.py in the dev
container (needs /root/reference).  TEST INFRASTRUCTURE ONLY."""
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np
from scipy import ndimage


def _install_stubs():
    # fill_voids: absent everywhere on this image; stand-in documented in
    # oracle/prepost_oracle.py.  pydicom / SimpleITK / torch: I/O only, unused
    # by the functions exercised here.
    fv = types.ModuleType("fill_voids")
    fv.fill = lambda x: ndimage.binary_fill_holes(x)
    sys.modules["fill_voids"] = fv
    for name in ("pydicom", "SimpleITK"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["SimpleITK"].Image = object  # only used in a type annotation (utils.py:233)
    t = types.ModuleType("torch")
    tu = types.ModuleType("torch.utils")
    tud = types.ModuleType("torch.utils.data")
    tud.Dataset = object
    t.utils = tu
    tu.data = tud
    sys.modules.update({"torch": t, "torch.utils": tu, "torch.utils.data": tud})
    try:
        import tqdm  # noqa: F401
    except ImportError:
        tq = types.ModuleType("tqdm")
        tq.tqdm = lambda it, **kw: it
        sys.modules["tqdm"] = tq
    lm = types.ModuleType("lungmask")
    lm.__path__ = []
    lg = types.ModuleType("lungmask.logger")
    import logging

    lg.logger = logging.getLogger("lungmask-ref")
    sys.modules["lungmask"] = lm
    sys.modules["lungmask.logger"] = lg


def main(inp, outp):
    sys.dont_write_bytecode = True
    _install_stubs()
    spec = importlib.util.spec_from_file_location("ref_utils", "/root/reference/lungmask/utils.py")
    U = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(U)
    d = np.load(inp, allow_pickle=True)
    out = {}
    n_pre = int(d["n_pre"])
    for i in range(n_pre):
        vol = d[f"pre{i}_vol"]
        res = [int(x) for x in d[f"pre{i}_res"]]
        xs, boxes = U.preprocess(vol, resolution=res)
        out[f"pre{i}_x"] = np.asarray(xs)
        out[f"pre{i}_box"] = np.asarray([list(map(int, b)) for b in boxes], dtype=np.int32)
        out[f"pre{i}_bmask"] = np.packbits(np.asarray([U.simple_bodymask(s) for s in np.clip(vol, -1024, 600)]).astype(bool), axis=-1)
    n_post = int(d["n_post"])
    for i in range(n_post):
        lab = d[f"post{i}_lab"]
        spare = [int(x) for x in d[f"post{i}_spare"]]
        skip = int(d[f"post{i}_skip"])
        out[f"post{i}_out"] = U.postprocessing(lab.copy(), spare=spare, disable_tqdm=True, skip_below=skip)
    n_rs = int(d["n_rs"])
    for i in range(n_rs):
        m = d[f"rs{i}_mask"]
        box = [int(x) for x in d[f"rs{i}_box"]]
        osz = tuple(int(x) for x in d[f"rs{i}_osz"])
        out[f"rs{i}_out"] = U.reshape_mask(m, box, osz).astype(np.uint8)
    n_klc = int(d["n_klc"])
    for i in range(n_klc):
        out[f"klc{i}_out"] = np.packbits(U.keep_largest_connected_component(d[f"klc{i}_mask"]))
        out[f"klc{i}_bbox"] = np.asarray(U.bbox_3D(d[f"klc{i}_mask"]), dtype=np.int32)
    # single-slice hole filler (utils.py:344-350)
    import skimage.morphology

    n_ac = int(d["n_ac"])
    for i in range(n_ac):
        out[f"ac{i}_out"] = skimage.morphology.area_closing(d[f"ac{i}_img"].astype(int), area_threshold=64)
    np.savez_compressed(outp, **out)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
