"""ctypes binding of liblungmask_hip.so (include/lungmask_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or no GPU is
visible, `load()` / `Engine()` raise.  (`tests/` may point `Library` at the
g++-built emulation of the same kernel sources; that library reports
`lm_is_gpu_build() == 0` and is refused here unless explicitly allowed.)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Sequence

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, "liblungmask_hip.so")

LM_DTYPES = {
    np.dtype(np.int16): 0,
    np.dtype(np.int32): 1,
    np.dtype(np.float32): 2,
    np.dtype(np.float64): 3,
    np.dtype(np.uint8): 4,
    np.dtype(np.uint16): 5,
    np.dtype(np.int64): 6,
}


class LMError(RuntimeError):
    pass


class _Tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.POINTER(C.c_float)), ("numel", C.c_int64)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("total_ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double)]


class LaunchSpan(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("lane", C.c_int), ("start_ms", C.c_double), ("end_ms", C.c_double)]


class Library:
    def __init__(self, path: Optional[str] = None, allow_emulation: bool = False):
        path = path or os.environ.get("LUNGMASK_HIP_LIB") or DEFAULT_LIB
        if not os.path.exists(path):
            raise LMError(
                f"{path} not found: build it with `python -m lungmask_amd.build` (hipcc, gfx950). "
                "lungmask_amd has no CPU fallback."
            )
        self.path = path
        self.lib = C.CDLL(path)
        L = self.lib
        L.lm_last_error.restype = C.c_char_p
        L.lm_version.restype = C.c_char_p
        L.lm_engine_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        L.lm_engine_destroy.argtypes = [C.c_void_p]
        L.lm_engine_destroy.restype = None
        L.lm_engine_sync.argtypes = [C.c_void_p]
        L.lm_dev_alloc.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t]
        L.lm_dev_free.argtypes = [C.c_void_p, C.c_void_p]
        L.lm_copy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.lm_copy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        if hasattr(L, "lm_host_alloc"):
            L.lm_host_alloc.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t]
            L.lm_host_free.argtypes = [C.c_void_p, C.c_void_p]
        L.lm_model_load.argtypes = [C.c_void_p, C.c_int, C.POINTER(_Tensor), C.c_int]
        L.lm_model_classes.argtypes = [C.c_void_p, C.c_int]
        if hasattr(L, "lm_engine_stream"):
            L.lm_engine_stream.argtypes = [C.c_void_p]
            L.lm_engine_stream.restype = C.c_void_p
        if hasattr(L, "lm_dist_init"):
            L.lm_dist_unique_id.argtypes = [C.c_void_p]
            L.lm_dist_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
            L.lm_dist_rank.argtypes = [C.c_void_p]
            L.lm_dist_world.argtypes = [C.c_void_p]
            L.lm_dist_all_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
            L.lm_dist_destroy.argtypes = [C.c_void_p]
        if hasattr(L, "lm_model_precision"):  # (absent from older builds that tools/ab_forward.py may load for comparison)
            L.lm_model_precision.argtypes = [C.c_void_p, C.c_int]
        if hasattr(L, "lm_model_probe_error"):
            L.lm_model_probe_error.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
            L.lm_model_chain_limit.argtypes = [C.c_void_p, C.c_int]
        L.lm_forward_dev.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.lm_set_precision.argtypes = [C.c_void_p, C.c_int]
        L.lm_set_streams.argtypes = [C.c_void_p, C.c_int]
        L.lm_set_fusion.argtypes = [C.c_void_p, C.c_int]
        L.lm_forward_batches_dev.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.lm_preprocess_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_int] * 5 + [C.c_void_p] * 4
        L.lm_reshape_mask_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]
        L.lm_reorient_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_int64] * 4
        L.lm_postprocess_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int]
        if hasattr(L, "lm_bbox3d_dev"):  # (absent from older builds that tools/ab_forward.py may load for comparison)
            L.lm_bbox3d_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32)]
            L.lm_keep_largest_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]
        L.lm_slab_begin.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [C.POINTER(C.c_int), C.c_int, C.c_int]
        L.lm_slab_pending.argtypes = [C.c_void_p]
        L.lm_slab_pending.restype = C.c_int64
        L.lm_slab_pending_uniform.argtypes = [C.c_void_p]
        L.lm_slab_emit.argtypes = [C.c_void_p, C.c_void_p]
        L.lm_slab_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.lm_postprocess_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.lm_fuse_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]
        L.lm_label_max_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]
        L.lm_fuse_spare_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.lm_apply_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p]
        L.lm_apply_host.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p]
        if hasattr(L, "lm_apply_host_ex"):  # (absent from older builds that tools/ab_forward.py may load for comparison)
            L.lm_apply_host_ex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p, C.c_uint]
        if hasattr(L, "lm_pipe_upload"):
            L.lm_pipe_upload.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
            L.lm_pipe_apply.argtypes = [C.c_void_p] + [C.c_int] * 9
            L.lm_pipe_download.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
            L.lm_pipe_wait.argtypes = [C.c_void_p, C.c_int]
        L.lm_profile_enable.argtypes = [C.c_void_p, C.c_int]
        L.lm_profile_reset.argtypes = [C.c_void_p]
        L.lm_profile_read.argtypes = [C.c_void_p, C.POINTER(KernelStat), C.c_int]
        if hasattr(L, "lm_profile_timeline"):
            L.lm_profile_timeline.argtypes = [C.c_void_p, C.POINTER(LaunchSpan), C.c_int]
        self.is_gpu = bool(L.lm_is_gpu_build())
        if not self.is_gpu and not allow_emulation:
            raise LMError(f"{path} is not a GPU build; refusing to run the product path on an emulation library")

    def check(self, status: int, what: str = ""):
        if status < 0:
            raise LMError(f"{what or 'lungmask_hip'} failed ({status}): {self.lib.lm_last_error().decode()}")
        return status


_default: Optional[Library] = None


def load() -> Library:
    global _default
    if _default is None:
        _default = Library()
    return _default


class DeviceArray:
    """A typed device allocation owned by an Engine (freed with it or via free())."""

    def __init__(self, eng: "Engine", shape, dtype):
        self.eng = eng
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_void_p()
        eng.L.check(eng.L.lib.lm_dev_alloc(eng.h, C.byref(p), self.nbytes), "lm_dev_alloc")
        self.ptr = p.value

    def upload(self, arr: np.ndarray) -> "DeviceArray":
        a = np.ascontiguousarray(arr, dtype=self.dtype)
        assert a.nbytes == self.nbytes, (a.shape, self.shape)
        self.eng.L.check(self.eng.L.lib.lm_copy_h2d(self.eng.h, self.ptr, a.ctypes.data, self.nbytes), "lm_copy_h2d")
        return self

    def download(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        self.eng.L.check(self.eng.L.lib.lm_copy_d2h(self.eng.h, out.ctypes.data, self.ptr, self.nbytes), "lm_copy_d2h")
        return out

    def free(self):
        if self.ptr:
            self.eng.L.lib.lm_dev_free(self.eng.h, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            if self.ptr and self.eng.h:
                self.free()
        except Exception:
            pass


class Engine:
    """One device + one HIP stream + workspaces (lm_engine)."""

    def __init__(self, device_id: int = 0, library: Optional[Library] = None):
        self.L = library or load()
        h = C.c_void_p()
        self.L.check(self.L.lib.lm_engine_create(C.byref(h), device_id), "lm_engine_create")
        self.h = h.value
        self.device_id = int(device_id)

    def close(self):
        if getattr(self, "h", None):
            self.L.lib.lm_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- memory
    def empty(self, shape, dtype) -> DeviceArray:
        return DeviceArray(self, shape, dtype)

    def to_device(self, arr: np.ndarray) -> DeviceArray:
        arr = np.ascontiguousarray(arr)
        return DeviceArray(self, arr.shape, arr.dtype).upload(arr)

    def sync(self):
        self.L.check(self.L.lib.lm_engine_sync(self.h), "lm_engine_sync")

    def host_alloc(self, nbytes: int) -> int:
        """Page-locked host memory (lm_host_alloc) -> address.  The caller owns it: free with host_free (also after close())."""
        p = C.c_void_p()
        self.L.check(self.L.lib.lm_host_alloc(self.h, C.byref(p), int(nbytes)), "lm_host_alloc")
        return int(p.value)

    def host_free(self, addr: int):
        self.L.check(self.L.lib.lm_host_free(self.h if getattr(self, "h", None) else None, C.c_void_p(addr)), "lm_host_free")

    # -- model
    def load_state_dict(self, slot: int, state_dict: Dict[str, "np.ndarray"]) -> int:
        """state_dict: name -> array-like (torch tensors are converted with .numpy())."""
        keep, names, arrs = [], [], []
        for k, v in state_dict.items():
            if hasattr(v, "detach"):
                v = v.detach().cpu().numpy()
            a = np.asarray(v)
            if a.dtype.kind != "f":
                continue  # num_batches_tracked
            a = np.ascontiguousarray(a, dtype=np.float32)
            names.append(k.encode())
            arrs.append(a)
        tens = (_Tensor * len(arrs))()
        for i, (n, a) in enumerate(zip(names, arrs)):
            tens[i].name = n
            tens[i].data = a.ctypes.data_as(C.POINTER(C.c_float))
            tens[i].numel = a.size
        keep.append((names, arrs))
        self.L.check(self.L.lib.lm_model_load(self.h, slot, tens, len(arrs)), "lm_model_load")
        return self.L.check(self.L.lib.lm_model_classes(self.h, slot))

    def stream_handle(self) -> int:
        """hipStream_t of the engine as an integer (0 under emulation): see lm_engine_stream in include/lungmask_hip.h."""
        if not hasattr(self.L.lib, "lm_engine_stream"):  # an older build of the library: the documented fall-back (host syncs)
            return 0
        return int(self.L.lib.lm_engine_stream(self.h) or 0)

    # -- one rank per engine: the RCCL communicator behind the C ABI (include/lungmask_hip.h: lm_dist_*)
    def dist_unique_id(self) -> bytes:
        """128 opaque bytes that rank 0 creates and every rank passes to dist_init (ncclUniqueId)."""
        buf = (C.c_uint8 * 128)()
        self.L.check(self.L.lib.lm_dist_unique_id(buf), "lm_dist_unique_id")
        return bytes(buf)

    def dist_init(self, rank: int, world: int, unique_id: Optional[bytes] = None):
        if unique_id is not None and len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of dist_unique_id()")
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id) if unique_id is not None else None
        self.L.check(self.L.lib.lm_dist_init(self.h, int(rank), int(world), buf), "lm_dist_init")

    def dist_all_gather(self, send_ptr: int, recv_ptr: int, nbytes: int):
        """Equal-size all-gather of device buffers, enqueued on the engine's stream (recv holds world * nbytes)."""
        self.L.check(self.L.lib.lm_dist_all_gather(self.h, send_ptr, recv_ptr, int(nbytes)), "lm_dist_all_gather")

    def dist_destroy(self):
        self.L.check(self.L.lib.lm_dist_destroy(self.h), "lm_dist_destroy")

    def n_classes(self, slot: int) -> int:
        return self.L.check(self.L.lib.lm_model_classes(self.h, slot), "lm_model_classes")

    def model_precision(self, slot: int) -> str:
        """'split_f16' or 'f32': what the next forward of `slot` runs on (a model whose activations left the f16 range is
        pinned to 'f32' by the engine's range guard)."""
        return "split_f16" if self.L.check(self.L.lib.lm_model_precision(self.h, slot), "lm_model_precision") == 1 else "f32"

    # -- network
    def model_probe(self, slot: int):
        """(max |delta log-prob| split-f16 vs exact fp32 on the load-time probe slice or None when no probe ran, pinned by it?)"""
        err = C.c_float()
        rc = self.L.lib.lm_model_probe_error(self.h, slot, C.byref(err))
        self.L.check(min(rc, 0), "lm_model_probe_error")
        return (None if err.value < 0 else float(err.value)), rc == 1

    def model_tier(self, slot: int) -> str:
        """'split_f16' (fast form), 'split_f16_chain<K>' (the accuracy guard's middle tiers: 3x3 convs split along K, no accumulator chain
        over K products) or 'f32'."""
        err = C.c_float()
        rc = self.L.lib.lm_model_probe_error(self.h, slot, C.byref(err))
        self.L.check(min(rc, 0), "lm_model_probe_error")
        if self.model_precision(slot) == "f32":
            return "f32"
        return f"split_f16_chain{self.L.lib.lm_model_chain_limit(self.h, slot)}" if rc == 2 else "split_f16"

    def set_precision(self, mode):
        """'f32' / 0: exact fp32 matrix ops;  'split_f16' / 1: 3-product split-f16."""
        m = {"f32": 0, "split_f16": 1}.get(mode, mode)
        self.L.check(self.L.lib.lm_set_precision(self.h, int(m)), "lm_set_precision")

    def set_streams(self, n: int):
        self.L.check(self.L.lib.lm_set_streams(self.h, int(n)), "lm_set_streams")

    def set_fusion(self, mask: int):
        """Bit 0: first conv inside conv 2's loader (bit-identical to the stand-alone kernel); bit 1: reserved (bilinear x2 inside the
        decoder conv's loader: not built); bit 2: split-K of the 16 x 16 / 32 x 32 decoder 1x1 convs (fixed order, last bits differ;
        measured slower -- off by default); bit 3: head inside the last conv's epilogue (fp32 instead of the stored 22-bit tensor:
        last bits differ).  Default 11 (A/B and test hook)."""
        self.L.check(self.L.lib.lm_set_fusion(self.h, int(mask)), "lm_set_fusion")

    def forward_dev(self, slot: int, x: DeviceArray, labels: Optional[DeviceArray] = None, logp: Optional[DeviceArray] = None):
        b, h, w = x.shape
        self.L.check(
            self.L.lib.lm_forward_dev(self.h, slot, x.ptr, b, h, w, labels.ptr if labels else None, logp.ptr if logp else None),
            "lm_forward_dev",
        )

    def forward(self, slot: int, x: np.ndarray, want_logp: bool = True):
        """x: f32 [b,h,w] (or [b,1,h,w]) host -> (labels u8 [b,h,w], logp f32 [b,C,h,w] or None)."""
        x = np.asarray(x, dtype=np.float32)
        if x.ndim == 4:
            x = x[:, 0]
        b, h, w = x.shape
        c = self.n_classes(slot)
        xd = self.to_device(x)
        ld = self.empty((b, h, w), np.uint8)
        pd = self.empty((b, c, h, w), np.float32) if want_logp else None
        self.forward_dev(slot, xd, ld, pd)
        self.sync()
        out = ld.download(), (pd.download() if pd else None)
        for d in (xd, ld, pd):
            if d is not None:
                d.free()
        return out

    # -- pre-processing
    def preprocess_dev(self, vol: DeviceArray, bbox: DeviceArray, x_f32: Optional[DeviceArray] = None,
                       x_i16: Optional[DeviceArray] = None, bmask: Optional[DeviceArray] = None, resolution=(256, 256)):
        n, h, w = vol.shape
        if vol.dtype not in LM_DTYPES:
            raise LMError(f"unsupported volume dtype {vol.dtype}")
        self.L.check(
            self.L.lib.lm_preprocess_dev(self.h, vol.ptr, LM_DTYPES[vol.dtype], n, h, w, int(resolution[0]), int(resolution[1]), bbox.ptr,
                                         x_f32.ptr if x_f32 else None, x_i16.ptr if x_i16 else None, bmask.ptr if bmask else None),
            "lm_preprocess_dev",
        )

    def preprocess(self, vol: np.ndarray, resolution=(256, 256), want_bmask: bool = False):
        """== utils.preprocess + normalisation: returns (x_i16 [n,oh,ow], x_f32, bbox int32 [n,4], bmask|None)."""
        vol = np.ascontiguousarray(vol)
        n, h, w = vol.shape
        vd = self.to_device(vol)
        bb = self.empty((n, 4), np.int32)
        xf = self.empty((n, resolution[0], resolution[1]), np.float32)
        xi = self.empty((n, resolution[0], resolution[1]), np.int16) if vol.dtype.kind == "i" else None
        bm = self.empty((n, h, w), np.uint8) if want_bmask else None
        self.preprocess_dev(vd, bb, xf, xi, bm, resolution)
        self.sync()
        out = (xi.download() if xi else None), xf.download(), bb.download(), (bm.download() if bm else None)
        for d in (vd, bb, xf, xi, bm):
            if d is not None:
                d.free()
        return out

    def reshape_mask_dev(self, mask: DeviceArray, bbox: DeviceArray, out: DeviceArray):
        n, mh, mw = mask.shape
        _, h, w = out.shape
        self.L.check(self.L.lib.lm_reshape_mask_dev(self.h, mask.ptr, bbox.ptr, n, mh, mw, h, w, out.ptr), "lm_reshape_mask_dev")

    def reshape_mask(self, mask: np.ndarray, bbox: np.ndarray, origsize) -> np.ndarray:
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        if mask.ndim == 2:
            mask, bbox = mask[None], np.asarray(bbox)[None]
        n = mask.shape[0]
        md = self.to_device(mask)
        bd = self.to_device(np.ascontiguousarray(bbox, dtype=np.int32).reshape(n, 4))
        od = self.empty((n, int(origsize[0]), int(origsize[1])), np.uint8)
        self.reshape_mask_dev(md, bd, od)
        self.sync()
        out = od.download()
        for d in (md, bd, od):
            d.free()
        return out

    # -- orientation
    def reorient_dev(self, src: DeviceArray, axes, flips) -> DeviceArray:
        """out = src.transpose(axes) with out-axis k reversed where flips[k] (device index transform)."""
        in_strides = [int(np.prod(src.shape[a + 1:], dtype=np.int64)) for a in range(3)]
        shape, strides, base = [], [], 0
        for k in range(3):
            a = int(axes[k])
            shape.append(src.shape[a])
            if flips[k]:
                strides.append(-in_strides[a])
                base += (src.shape[a] - 1) * in_strides[a]
            else:
                strides.append(in_strides[a])
        out = self.empty(tuple(shape), src.dtype)
        self.L.check(self.L.lib.lm_reorient_dev(self.h, src.ptr, out.ptr, np.dtype(src.dtype).itemsize, *shape, *strides, base),
                     "lm_reorient_dev")
        return out

    # -- post-processing
    def postprocess_dev(self, lab: DeviceArray, spare: Sequence[int] = (), skip_below: int = 3):
        n, h, w = lab.shape
        sp = (C.c_int * max(len(spare), 1))(*[int(s) for s in spare])
        self.L.check(self.L.lib.lm_postprocess_dev(self.h, lab.ptr, n, h, w, sp, len(spare), int(skip_below)), "lm_postprocess_dev")

    def postprocess(self, lab: np.ndarray, spare: Sequence[int] = (), skip_below: int = 3) -> np.ndarray:
        """== utils.postprocessing(label_image, spare, skip_below=...)."""
        ld = self.to_device(np.ascontiguousarray(lab, dtype=np.uint8))
        self.postprocess_dev(ld, spare, skip_below)
        self.sync()
        out = ld.download()
        ld.free()
        return out

    def bbox_3d(self, mask: np.ndarray, margin: int = 2):
        """== utils.bbox_3D(labelmap, margin) for a [n, h, w] volume (non-zero = set): 6 ints, or None for an empty mask."""
        md = self.to_device(np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8))
        n, h, w = md.shape
        bb = (C.c_int32 * 6)()
        try:
            self.L.check(self.L.lib.lm_bbox3d_dev(self.h, md.ptr, n, h, w, int(margin), bb), "lm_bbox3d_dev")
        finally:
            md.free()
        return None if bb[1] < 0 else [int(v) for v in bb]

    def keep_largest_dev(self, mask: DeviceArray) -> int:
        n, h, w = mask.shape
        area = C.c_int64()
        self.L.check(self.L.lib.lm_keep_largest_dev(self.h, mask.ptr, n, h, w, C.byref(area)), "lm_keep_largest_dev")
        return int(area.value)

    def keep_largest(self, mask: np.ndarray):
        """== utils.keep_largest_connected_component(mask) for a [n, h, w] u8 volume -> (bool volume, area); area 0 = no region."""
        md = self.to_device(np.ascontiguousarray(mask, dtype=np.uint8))
        try:
            area = self.keep_largest_dev(md)
            self.sync()
            out = md.download()
        finally:
            md.free()
        return out.astype(bool), area

    def postprocess_info(self) -> dict:
        buf = (C.c_int64 * 5)()
        self.L.check(self.L.lib.lm_postprocess_info(self.h, buf))
        return dict(regions=buf[0], boundary_records=buf[1], processed=buf[2], merged=buf[3], host_replay_ms=buf[4] / 1000.0)

    def fuse(self, res_l: np.ndarray, res_r: np.ndarray):
        """mask.py:228-230 -> (fused volume incl. spare label, spare value)."""
        ld = self.to_device(np.ascontiguousarray(res_l, dtype=np.uint8))
        rd = self.to_device(np.ascontiguousarray(res_r, dtype=np.uint8))
        sp = C.c_int()
        self.L.check(self.L.lib.lm_fuse_dev(self.h, ld.ptr, rd.ptr, ld.nbytes, C.byref(sp)), "lm_fuse_dev")
        self.sync()
        out = ld.download()
        ld.free()
        rd.free()
        return out, sp.value

    # -- the whole hot path
    def apply_dev(self, slot: int, vol: DeviceArray, out: DeviceArray, fill_slot: int = -1, batch_size: int = 20, volume_postprocessing: bool = True):
        n, h, w = vol.shape
        if vol.dtype not in LM_DTYPES:
            raise LMError(f"unsupported volume dtype {vol.dtype}")
        self.L.check(
            self.L.lib.lm_apply_dev(self.h, slot, fill_slot, vol.ptr, LM_DTYPES[vol.dtype], n, h, w, int(batch_size), int(bool(volume_postprocessing)), out.ptr),
            "lm_apply_dev",
        )

    # -- volumes queued through the engine (lm_pipe_*; LMInferer.apply_async drives them from two threads)
    def pipe_upload(self, k: int, vol: Optional[np.ndarray]):
        if vol is None:
            self.L.check(self.L.lib.lm_pipe_upload(self.h, k, None, 0), "lm_pipe_upload")
        else:
            self.L.check(self.L.lib.lm_pipe_upload(self.h, k, vol.ctypes.data, vol.nbytes), "lm_pipe_upload")

    def pipe_apply(self, k: int, slot: int, shape, dtype, fill_slot: int = -1, batch_size: int = 20, volume_postprocessing: bool = True):
        n, h, w = (int(v) for v in shape)
        self.L.check(self.L.lib.lm_pipe_apply(self.h, k, slot, fill_slot, LM_DTYPES[np.dtype(dtype)], n, h, w, int(batch_size), int(bool(volume_postprocessing))),
                     "lm_pipe_apply")

    def pipe_download(self, k: int, out: np.ndarray):
        self.L.check(self.L.lib.lm_pipe_download(self.h, k, out.ctypes.data, out.nbytes), "lm_pipe_download")

    def pipe_wait(self, k: int):
        self.L.check(self.L.lib.lm_pipe_wait(self.h, k), "lm_pipe_wait")

    def apply(self, slot: int, vol: np.ndarray, fill_slot: int = -1, batch_size: int = 20, volume_postprocessing: bool = True,
              out: Optional[np.ndarray] = None, out_scratch: bool = False) -> np.ndarray:
        """numpy -> numpy (lm_apply_host).  `out`: an optional caller-owned uint8 C-contiguous array of the volume's shape to
        receive the labels (a reused buffer spares the page faults and the unmapping of a fresh 79 MB array per volume).
        `out_scratch`: the contents of `out` are of no value (LM_APPLY_OUT_SCRATCH): it is zero-filled while the network runs and
        only the slab that carries labels is copied back; always so for an array allocated here."""
        vol = np.ascontiguousarray(vol)
        if vol.dtype not in LM_DTYPES:
            raise LMError(f"unsupported volume dtype {vol.dtype}")
        n, h, w = vol.shape
        if out is None:
            out, out_scratch = np.empty((n, h, w), dtype=np.uint8), True
        elif out.dtype != np.uint8 or out.shape != (n, h, w) or not out.flags.c_contiguous:
            raise LMError("apply(out=...): need a C-contiguous uint8 array of the volume's shape")
        args = (self.h, slot, fill_slot, vol.ctypes.data, LM_DTYPES[vol.dtype], n, h, w, int(batch_size), int(bool(volume_postprocessing)), out.ctypes.data)
        if out_scratch and hasattr(self.L.lib, "lm_apply_host_ex"):
            self.L.check(self.L.lib.lm_apply_host_ex(*args, 1), "lm_apply_host_ex")
        else:
            self.L.check(self.L.lib.lm_apply_host(*args), "lm_apply_host")
        return out

    # -- profiling
    def profile(self, on):
        """on: False/True, or 2 for one entry per conv layer shape."""
        self.L.check(self.L.lib.lm_profile_enable(self.h, int(on)))

    def profile_reset(self):
        self.L.check(self.L.lib.lm_profile_reset(self.h))

    def profile_timeline(self, cap: int = 4096):
        """After profile(4): [(name, lane, start_ms, end_ms)] of every launch since the last reset."""
        buf = (LaunchSpan * cap)()
        n = self.L.check(self.L.lib.lm_profile_timeline(self.h, buf, cap))
        return [(buf[i].name.decode(), buf[i].lane, buf[i].start_ms, buf[i].end_ms) for i in range(min(n, cap))]

    def profile_read(self):
        buf = (KernelStat * 96)()
        n = self.L.check(self.L.lib.lm_profile_read(self.h, buf, 96))
        return [
            dict(name=buf[i].name.decode(), launches=buf[i].launches, total_ms=buf[i].total_ms, flops=buf[i].flops, bytes=buf[i].bytes)
            for i in range(min(n, 96))
        ]
