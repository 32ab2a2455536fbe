"""`lungmask.utils` for the functions that sit on the hot path, with the reference's names, arguments and return
conventions, executed by the MI355X engine (no CPU implementation here: every call goes through liblungmask_hip.so).

    preprocess            utils.py:32-52      lm_preprocess_dev
    simple_bodymask       utils.py:55-82      lm_preprocess_dev (body-mask test seam)
    crop_and_resize       utils.py:85-111     lm_preprocess_dev on one slice
    reshape_mask          utils.py:114-129    lm_reshape_mask_dev
    postprocessing        utils.py:272-358    lm_postprocess_dev
    bbox_3D               utils.py:361-387    lm_bbox3d_dev
    keep_largest_connected_component  utils.py:390-404  lm_keep_largest_dev
    load_input_image / read_dicoms  utils.py:132-269   volume_io (host I/O)

`bbox_3D` and `keep_largest_connected_component` are steps of `postprocessing` in the reference (inside
`lm_postprocess_dev` they run on the region graph); the two functions below are the same steps as calls of their own.

The engine is created on first use (`set_engine` injects another one, e.g. a specific device)."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _native
from .volume_io import load_input_image, read_dicoms  # noqa: F401  (utils.py:132-269)

_engine: Optional["_native.Engine"] = None


def set_engine(engine) -> None:
    global _engine
    _engine = engine


def _eng():
    global _engine
    if _engine is None:
        _engine = _native.Engine(0)  # raises without liblungmask_hip.so / a GPU: there is no CPU path
    return _engine


def _int_volume(img: np.ndarray) -> np.ndarray:
    img = np.asarray(img)
    if img.dtype.kind not in "iu":
        raise TypeError("lungmask_amd.utils works on integer HU images like the reference's pipeline does "
                        f"(got {img.dtype}); LMInferer.apply accepts float volumes")
    if img.dtype in (np.int16, np.int32, np.int64):
        return img
    return img.astype(np.int32 if img.dtype.itemsize < 4 else np.int64)


def preprocess(img: np.ndarray, resolution: Sequence[int] = (192, 192)) -> Tuple[np.ndarray, List[np.ndarray]]:
    """utils.py:32-52: clip to [-1024, 600], crop every slice to the body, resize (order 1) to `resolution`.
    Returns (slices [n, res0, res1] in the input's dtype, list of bounding boxes)."""
    src = np.asarray(img)
    xi, _, bb, _ = _eng().preprocess(_int_volume(src), resolution=(int(resolution[0]), int(resolution[1])))
    return xi.astype(src.dtype, copy=False), [b for b in bb]


def simple_bodymask(img: np.ndarray) -> np.ndarray:
    """utils.py:55-82 for one slice -> bool mask of the slice's shape."""
    src = _int_volume(np.asarray(img))
    assert src.ndim == 2, "simple_bodymask takes a single slice"
    return _eng().preprocess(src[None], resolution=(8, 8), want_bmask=True)[3][0].astype(bool)


def crop_and_resize(img: np.ndarray, width: int = 192, height: int = 192) -> Tuple[np.ndarray, np.ndarray]:
    """utils.py:85-111 for one slice.  The engine's kernel is `preprocess`'s (HU clip on read), so values outside
    [-1024, 600] -- which `utils.preprocess` never passes on -- are refused instead of silently clipped."""
    src = np.asarray(img)
    assert src.ndim == 2, "crop_and_resize takes a single slice"
    if src.size and (src.min() < -1024 or src.max() > 600):
        raise ValueError("crop_and_resize on the engine expects HU values already clipped to [-1024, 600] (as utils.preprocess passes them)")
    xs, boxes = preprocess(src[None], resolution=(width, height))
    return xs[0], boxes[0]


def reshape_mask(mask: np.ndarray, tbox: Sequence[int], origsize: Sequence[int]) -> np.ndarray:
    """utils.py:114-129: nearest-neighbour resize of `mask` to the bounding box, pasted into zeros(origsize)
    (float64 like the reference's `np.ones(origsize) * 0`)."""
    out = _eng().reshape_mask(np.asarray(mask), np.asarray(tbox, dtype=np.int32), (int(origsize[0]), int(origsize[1])))
    return out[0].astype(np.float64)


def postprocessing(label_image: np.ndarray, spare: Sequence[int] = (), disable_tqdm: bool = False, skip_below: int = 3) -> np.ndarray:
    """utils.py:272-358 (the progress bar argument is accepted and ignored)."""
    lab = np.asarray(label_image)
    assert lab.ndim == 3, "postprocessing takes a [n, h, w] label volume"
    return _eng().postprocess(lab, spare=[int(s) for s in spare], skip_below=int(skip_below)).astype(lab.dtype, copy=False)


def _as_3d(a: np.ndarray) -> np.ndarray:
    if a.ndim < 1 or a.ndim > 3:
        raise ValueError(f"the engine's volumes have 1 to 3 axes (got an array of shape {a.shape})")
    return a.reshape((1,) * (3 - a.ndim) + a.shape)


def bbox_3D(labelmap: np.ndarray, margin: int = 2) -> np.ndarray:
    """utils.py:361-387: `[zmin, zmax, ymin, ymax, xmin, xmax]` (two entries per axis of `labelmap`, maxima exclusive) of the
    non-zero voxels, grown by `margin` and clipped.  An all-zero labelmap raises IndexError, as the reference's
    `np.where(margin_label)[0][[0, -1]]` does."""
    lm_ = np.asarray(labelmap)
    bb = _eng().bbox_3d(_as_3d(lm_), margin=int(margin))
    if bb is None:
        raise IndexError("bbox_3D of a labelmap without a non-zero voxel (utils.py:377)")
    return np.asarray(bb[2 * (3 - lm_.ndim):], dtype=np.int64)


def keep_largest_connected_component(mask: np.ndarray) -> np.ndarray:
    """utils.py:390-404: bool mask of the largest region of `skimage.measure.label(mask)` (full connectivity; different non-zero
    values are different regions).  Equal areas: the region whose first voxel comes last in raster order (what numpy's stable
    insertion sort makes of `np.argsort(resizes)[-1]` up to 16 regions).  A mask without a region raises IndexError like :402."""
    m = np.asarray(mask)
    if m.dtype == bool:
        m8 = m.astype(np.uint8)
    elif m.dtype.kind in "iu" and m.size and 0 <= m.min() and m.max() <= 255:
        m8 = m.astype(np.uint8)
    elif m.size == 0:
        m8 = m.astype(np.uint8)
    else:
        raise TypeError("keep_largest_connected_component on the engine takes bool masks or label maps with values in [0, 255]")
    out, area = _eng().keep_largest(_as_3d(m8))
    if area == 0:
        raise IndexError("keep_largest_connected_component of a mask without a region (utils.py:402)")
    return out.reshape(m.shape)
