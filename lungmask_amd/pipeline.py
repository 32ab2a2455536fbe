"""Slice-sharded multi-GPU form of `LMInferer.apply` (one process per GPU,
`torch.distributed` over RCCL/xGMI) built from the stage-level C-ABI calls.

Sharding: pre-processing, the network forward and the argmax are
slice-wise (utils.py:48-51, mask.py:173-187) -> contiguous slice blocks per rank,
weights replicated, NO collective.  The 3-D post-processing (utils.py:272-358) spans
the whole volume; two forms:

* slab-sharded (default from three ranks on): every rank post-processes its own slab and the slabs are tied
  together by six small all-gathers of face planes / atom tables (`lm_slab_*`,
  csrc/slab_engine.hip) -- the voxel passes scale with 1/world;
* gathered (`sharded_post=False`, the default with one or two ranks, and whenever a rank has no slice): ONE all-gather of the
  uint8 256x256 label shards (64 KiB/slice), then every rank runs the identical
  deterministic whole-volume post-processing (the serial fraction of weak scaling).

Each rank then un-crops its own slices and a final all-gather assembles the [n,h,w]
result on every rank (79 MB per rank at 300 slices/rank; xGMI is point-to-point and fully
connected inside a node, so it is single-step and per-link bound, well under a millisecond).

All buffers are torch tensors (device memory is torch's job here); the engine receives raw pointers.  The collectives come
either from torch.distributed or -- `NativeDist` -- from the engine's own RCCL communicator behind the C ABI (lm_dist_*), in
which case torch is a buffer allocator and nothing more.  On the GPU the engine's own HIP stream
(`lm_engine_stream`) is made torch's current stream for everything in here, so the RCCL
collectives are ordered against the engine's kernels by stream dependencies (torch's
process group waits for the current stream before a collective and lets it wait for the
collective afterwards): the host only blocks where it needs data -- the three table merges
of the slab protocol.  In the CPU test-suite the tensors are host tensors, the backend is
gloo and the engine is the test emulation.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import numpy as np
import torch


def shard_bounds(n: int, world: int) -> List[int]:
    """Contiguous slice blocks: rank r owns [b[r], b[r+1]); sizes differ by at most one."""
    base, rem = divmod(n, world)
    b = [0]
    for r in range(world):
        b.append(b[-1] + base + (1 if r < rem else 0))
    return b


class NativeDist:
    """The `dist` argument of ShardedPipeline WITHOUT torch.distributed: the engine's own RCCL communicator behind the C ABI
    (include/lungmask_hip.h: lm_dist_*).  The one collective of the pipeline, an equal-size all-gather of device buffers, is
    enqueued on the engine's stream.  `unique_id`: the 128 bytes rank 0 got from `engine.dist_unique_id()`, handed to the other
    ranks by any side channel (`exchange_id` uses a torch.distributed Store; a file or MPI do as well).  A world of one without
    an id involves no library."""

    def __init__(self, engine, rank: int, world: int, unique_id: bytes = None):
        engine.dist_init(rank, world, unique_id)
        self.e, self.rank, self.world = engine, int(rank), int(world)

    @staticmethod
    def exchange_id(engine, rank: int, store, key: str = "lungmask_amd/dist_id") -> bytes:
        """Rank 0 creates the id and publishes it in `store` (e.g. torch.distributed.TCPStore); the others read it."""
        if rank == 0:
            uid = engine.dist_unique_id()
            store.set(key, uid)
            return uid
        return bytes(store.get(key))

    def get_world_size(self) -> int:
        return self.world

    def get_rank(self) -> int:
        return self.rank

    def all_gather_into_tensor(self, out: torch.Tensor, mine: torch.Tensor):
        nbytes = mine.numel() * mine.element_size()
        assert out.is_contiguous() and mine.is_contiguous() and out.numel() * out.element_size() == self.world * nbytes
        self.e.dist_all_gather(mine.data_ptr(), out.data_ptr(), nbytes)

    def destroy(self):
        self.e.dist_destroy()


class ShardedPipeline:
    """engine: lungmask_amd._native.Engine; dist: the torch.distributed module (initialised), a NativeDist, or None;
    device: torch device that matches the engine's memory space ('cuda:<i>' or 'cpu' under emulation)."""

    def __init__(self, engine, slot: int = 0, batch_size: int = 20, volume_postprocessing: bool = True,
                 resolution: Sequence[int] = (256, 256), dist=None, device="cpu", sharded_post=None):
        self.e = engine
        self.slot = slot
        self.batch_size = int(batch_size)
        self.volume_postprocessing = volume_postprocessing
        self.res = tuple(int(r) for r in resolution)
        self.dist = dist
        self.device = torch.device(device)
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        # None: by world size.  The slab protocol's fixed part (six exchanges, three host table merges: 3.7 ms per rank with two
        # ranks of 300 slices) only pays from three ranks on -- the redundant whole-volume pass on the gathered labels costs 3.3 ms
        # for 600 slices, 6.5 for 1200, 12.4 for 2400 against 3.7 / 4.3 / 5.4 (tools/slab_timing.py, profiles/r04r_slab_timing.log)
        self.sharded_post = (self.world >= 3) if sharded_post is None else bool(sharded_post)
        self._buf = {}
        self._slab_caps = {}   # agreed capacity (ints) of the variable-length table exchange of every protocol round
        self.collectives = 0   # collectives issued for variable-length tables (tests / tools/slab_timing.py read it)
        # the engine's stream as torch's current stream (see the module docstring); None on the CPU / under emulation
        self._stream = None
        if self.device.type == "cuda" and engine.stream_handle():
            self._stream = torch.cuda.ExternalStream(engine.stream_handle(), device=self.device)

    def _on_engine_stream(self):
        import contextlib

        return torch.cuda.stream(self._stream) if self._stream is not None else contextlib.nullcontext()

    def _tensor(self, key, shape, dtype):
        t = self._buf.get(key)
        n = int(np.prod(shape))
        if t is None or t.numel() < n or t.dtype != dtype:
            t = torch.empty(max(n, 1), dtype=dtype, device=self.device)
            self._buf[key] = t
        return t[:n].view(*shape) if n else t[:0]

    def _sync_torch(self):
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def _all_gather(self, out: torch.Tensor, mine: torch.Tensor):
        # gloo must not alias input and output; RCCL gathers in place.  Called with the engine's stream current: no host
        # synchronisation here (without that stream, e.g. an engine of another build, fall back to a device sync)
        self.dist.all_gather_into_tensor(out, mine.clone() if self.device.type == "cpu" else mine)
        if self._stream is None:
            self._sync_torch()

    def postprocess_slab(self, lab_slab: torch.Tensor, z0: int, n_total: int, spare: Sequence[int] = (), skip_below: int = 3):
        """utils.postprocessing over a volume whose slices are spread over the ranks; `lab_slab` (uint8 [n_r,h,w], n_r >= 1,
        slices [z0, z0+n_r)) is processed IN PLACE.  Protocol of include/lungmask_hip.h (lm_slab_*)."""
        with self._on_engine_stream():
            return self._postprocess_slab(lab_slab, z0, n_total, spare, skip_below)

    def _postprocess_slab(self, lab_slab, z0, n_total, spare, skip_below):
        e, lib = self.e, self.e.L.lib
        n_r, h, w = (int(v) for v in lab_slab.shape)
        sp = (C.c_int * max(len(spare), 1))(*[int(v) for v in spare])
        e.L.check(lib.lm_slab_begin(e.h, lab_slab.data_ptr(), n_r, h, w, self.rank, self.world, int(z0), int(n_total), sp, len(spare), int(skip_below)),
                  "lm_slab_begin")
        rnd = 0
        while True:
            n = int(lib.lm_slab_pending(e.h))
            if n < 0:
                raise RuntimeError("lm_slab_pending: no slab post-processing in progress")
            hdr = 0  # ints in front of every rank's table inside the gathered buffer
            if self.dist is None:
                lens, stride = [n], n
                mine = self._tensor("slab_mine", (max(stride, 1),), torch.int32)
                gathered = mine
                e.L.check(lib.lm_slab_emit(e.h, mine.data_ptr()), "lm_slab_emit")
            elif lib.lm_slab_pending_uniform(e.h):
                lens, stride = [n] * self.world, n  # face planes: the same length on every rank, no need to exchange it
                mine = self._tensor("slab_mine", (max(stride, 1),), torch.int32)
                gathered = self._tensor("slab_all", (self.world * max(stride, 1),), torch.int32)
                e.L.check(lib.lm_slab_emit(e.h, mine.data_ptr()), "lm_slab_emit")
                if stride:
                    self._all_gather(gathered, mine)
            else:
                # variable-length tables: every rank sends [length | table | padding] of ONE agreed size, so the lengths travel
                # inside the table exchange (6 collectives per volume instead of 9).  The agreed capacity of round `rnd` is what
                # the previous volume needed plus a quarter -- every rank saw the same lengths, so every rank holds the same
                # number; the first volume (and a table that outgrows the capacity, which every rank notices at the same
                # time) exchanges the lengths first, as before.
                cap = self._slab_caps.get(rnd, 0)
                lens = lens_known = None
                if cap > 0:
                    mine = self._tensor("slab_mine", (cap + 1,), torch.int32)
                    gathered = self._tensor("slab_all", (self.world * (cap + 1),), torch.int32)
                    mine[:1].fill_(n)
                    if n <= cap:
                        e.L.check(lib.lm_slab_emit(e.h, mine.data_ptr() + 4), "lm_slab_emit")
                    self._all_gather(gathered, mine)
                    self.collectives += 1
                    got = [int(v) for v in gathered.view(self.world, cap + 1)[:, 0].cpu().tolist()]
                    if max(got) <= cap:
                        lens, stride, hdr = got, cap + 1, 1
                    else:
                        lens_known = got  # somebody's table did not fit: exchange again at the exact size
                if lens is None:
                    if lens_known is None:
                        lens_t = self._tensor("slab_lens", (self.world,), torch.int64)
                        self._all_gather(lens_t, torch.tensor([n], dtype=torch.int64, device=self.device))
                        lens_known = [int(v) for v in lens_t.cpu().tolist()]
                        self.collectives += 1
                    lens, stride = lens_known, max(lens_known)
                    mine = self._tensor("slab_mine", (max(stride, 1),), torch.int32)
                    gathered = self._tensor("slab_all", (self.world * max(stride, 1),), torch.int32)
                    e.L.check(lib.lm_slab_emit(e.h, mine.data_ptr()), "lm_slab_emit")
                    if stride:
                        self._all_gather(gathered, mine)
                        self.collectives += 1
                # (monotone: alternating small and large volumes must not fall back to the exact-size exchange every other time)
                self._slab_caps[rnd] = max(self._slab_caps.get(rnd, 0), -(-(max(lens) + max(lens) // 4 + 256) // 1024) * 1024)
            status = e.L.check(lib.lm_slab_step(e.h, gathered.data_ptr() + 4 * hdr, stride, (C.c_int64 * self.world)(*lens)), "lm_slab_step")
            rnd += 1
            if status == 1:
                return

    def apply(self, volume: np.ndarray) -> np.ndarray:
        """Convenience form of `LMInferer.apply` for a process group: every rank passes the SAME host volume [n,h,w] (int16),
        works on its own block of slices and returns the complete uint8 label volume."""
        volume = np.asarray(volume)
        if volume.dtype != np.int16:
            # this entry point shards int16 HU volumes (what DICOM / the bench phantom give); other integer types are accepted
            # when their values fit -- never wrapped -- and everything else belongs to LMInferer.apply (float volumes, fusion
            # with a fill model and image orientation are not part of the sharded form)
            if volume.dtype.kind not in "iu":
                raise TypeError(f"ShardedPipeline.apply: integer HU volume expected, got {volume.dtype} (use LMInferer.apply)")
            if volume.size and (volume.min() < -32768 or volume.max() > 32767):
                raise ValueError("ShardedPipeline.apply: values outside the int16 range (use LMInferer.apply)")
        vol = np.ascontiguousarray(volume, dtype=np.int16)
        n_total = int(vol.shape[0])
        b = shard_bounds(n_total, self.world)
        shard = torch.from_numpy(vol[b[self.rank] : b[self.rank + 1]]).to(self.device)
        return self.apply_shard(shard.contiguous(), n_total).cpu().numpy()

    def apply_local(self, shard: np.ndarray, n_total: int, gather: bool = True):
        """The rank-local form (config 5: 2400 slices over 8 GPUs): every rank passes ONLY its own contiguous block of slices
        `shard` [n_r,h,w] (int16; n_r = shard_bounds(n_total, world) of this rank) -- no rank ever holds the whole input volume.
        gather=True: returns the complete uint8 label volume [n_total,h,w] (the reference's result on every rank, one all-gather of
        the output shards); gather=False: only this rank's [n_r,h,w] block (no output collective; host memory per rank stays 1/world)."""
        shard = np.ascontiguousarray(shard)
        if shard.dtype != np.int16:
            if shard.dtype.kind not in "iu":
                raise TypeError(f"ShardedPipeline.apply_local: integer HU volume expected, got {shard.dtype}")
            if shard.size and (shard.min() < -32768 or shard.max() > 32767):
                raise ValueError("ShardedPipeline.apply_local: values outside the int16 range")
            shard = shard.astype(np.int16)
        b = shard_bounds(int(n_total), self.world)
        if shard.shape[0] != b[self.rank + 1] - b[self.rank]:
            raise ValueError(f"rank {self.rank} of {self.world} owns slices [{b[self.rank]}, {b[self.rank + 1]}) of {n_total}: got {shard.shape[0]} slices")
        out = self.apply_shard(torch.from_numpy(shard).to(self.device).contiguous(), int(n_total), gather=gather)
        return out.cpu().numpy()

    def shard_buffers(self, n_total: int):
        """(bounds, bbox [maxc,4] int32, lab_all [world*maxc,oh,ow] u8, lab_loc = this rank's part of lab_all)."""
        bounds = shard_bounds(n_total, self.world)
        maxc = max(bounds[r + 1] - bounds[r] for r in range(self.world))
        oh, ow = self.res
        bbox = self._tensor("bbox", (maxc, 4), torch.int32)
        lab_all = self._tensor("lab_all", (self.world * maxc, oh, ow), torch.uint8)
        lab_loc = lab_all[self.rank * maxc : (self.rank + 1) * maxc]
        return bounds, bbox, lab_all, lab_loc

    def apply_shard(self, vol_shard: torch.Tensor, n_total: int, gather: bool = True) -> torch.Tensor:
        """vol_shard: this rank's contiguous slice block [n_r,h,w] int16, resident in the engine's memory
        space.  Returns the FULL uint8 label volume [n_total,h,w] (same memory space), or with gather=False this rank's block."""
        e, lib = self.e, self.e.L.lib
        n_r, h, w = (int(s) for s in vol_shard.shape)
        bounds, bbox, _, lab_loc = self.shard_buffers(n_total)
        assert n_r == bounds[self.rank + 1] - bounds[self.rank], (n_r, bounds, self.rank)
        assert vol_shard.dtype == torch.int16 and vol_shard.is_contiguous()
        oh, ow = self.res
        xf = self._tensor("xf", (max(n_r, 1), oh, ow), torch.float32)
        self._sync_torch()  # the caller's shard (copied in on another stream) is complete
        # ---- sliced stages: no communication
        if n_r:
            e.L.check(lib.lm_preprocess_dev(e.h, vol_shard.data_ptr(), 0, n_r, h, w, oh, ow, bbox.data_ptr(), xf.data_ptr(), None, None), "lm_preprocess_dev")
            e.L.check(lib.lm_forward_batches_dev(e.h, self.slot, xf.data_ptr(), n_r, oh, ow, self.batch_size, lab_loc.data_ptr()), "lm_forward_batches_dev")
        if self._stream is None:
            e.sync()
        return self.assemble(n_total, h, w, gather=gather)

    def assemble(self, n_total: int, h: int, w: int, gather: bool = True) -> torch.Tensor:
        """Everything after the argmax: volume post-processing of the label shards in `shard_buffers(n_total)`, un-crop with
        the shard's bounding boxes, and the all-gather of the [n_total,h,w] result.  Safe to call on its own: it makes the
        engine's stream torch's current stream for its collectives (so they are ordered with the engine's kernels whatever
        stream the caller had current) and the result is complete when it returns."""
        with self._on_engine_stream():
            out = self._assemble(n_total, h, w, gather)
        if self._stream is not None:
            self._stream.synchronize()
        return out

    def _assemble(self, n_total: int, h: int, w: int, gather: bool = True) -> torch.Tensor:
        e, lib = self.e, self.e.L.lib
        bounds, bbox, lab_all, lab_loc = self.shard_buffers(n_total)
        counts = [bounds[r + 1] - bounds[r] for r in range(self.world)]
        n_r, maxc = counts[self.rank], max(counts)
        oh, ow = self.res
        # (a process group of ONE rank still runs the exchange protocol: that is how the RCCL calls are exercised on a 1-GPU box)
        slabs = self.sharded_post and self.dist is not None and min(counts) >= 1 and n_total > 1
        if slabs:
            # ---- post-processing on the own slab; six small exchanges inside (no label all-gather at all)
            mine_lab = lab_loc[:n_r]
            if self.volume_postprocessing:
                self.postprocess_slab(mine_lab, bounds[self.rank], n_total)
        else:
            # ---- exchange #1: 256^2 label shards -> whole label volume on every rank (RCCL all-gather, in place)
            if self.dist is not None:
                self._all_gather(lab_all.view(-1), lab_loc.reshape(-1))
                if any(c != maxc for c in counts):  # ragged tail: compact the padded shards
                    full = torch.cat([lab_all[r * maxc : r * maxc + counts[r]] for r in range(self.world)]).contiguous()
                else:
                    full = lab_all
            else:
                full = lab_all[:n_r]
            if self._stream is None:
                self._sync_torch()
            if self.volume_postprocessing and n_total:
                e.L.check(lib.lm_postprocess_dev(e.h, full.data_ptr(), n_total, oh, ow, None, 0, 3), "lm_postprocess_dev")
            mine_lab = full[bounds[self.rank] : bounds[self.rank + 1]]
        # ---- un-crop own slices
        out_all = self._tensor("out_all", (self.world * maxc, h, w), torch.uint8)
        out_loc = out_all[self.rank * maxc : (self.rank + 1) * maxc]
        if n_r:
            e.L.check(lib.lm_reshape_mask_dev(e.h, mine_lab.data_ptr(), bbox.data_ptr(), n_r, oh, ow, h, w, out_loc.data_ptr()), "lm_reshape_mask_dev")
        if self._stream is None:
            e.sync()
        # ---- exchange #2: output shards
        if not gather:
            return out_loc[:n_r]
        if self.dist is not None:
            self._all_gather(out_all.view(-1), out_loc.reshape(-1))
            if any(c != maxc for c in counts):
                return torch.cat([out_all[r * maxc : r * maxc + counts[r]] for r in range(self.world)]).contiguous()
            return out_all
        return out_all[:n_r]


def postprocess_slabs_in_process(engines, lab: np.ndarray, bounds: Sequence[int], spare: Sequence[int] = (), skip_below: int = 3) -> np.ndarray:
    """The lm_slab_* protocol with every "rank" in THIS process: engines[r] (all on one device / memory space) owns slices
    [bounds[r], bounds[r+1]) of `lab`; the exchanges are plain device copies.  Used to exercise the multi-rank code on
    one GPU (tests) -- same calls, same tables, no torch.distributed."""
    world = len(engines)
    assert len(bounds) == world + 1 and all(bounds[r + 1] > bounds[r] for r in range(world))
    n_total, h, w = lab.shape
    sp = (C.c_int * max(len(spare), 1))(*[int(v) for v in spare])
    slabs = []
    for r, e in enumerate(engines):
        d = e.to_device(np.ascontiguousarray(lab[bounds[r] : bounds[r + 1]], dtype=np.uint8))
        slabs.append(d)
        e.L.check(e.L.lib.lm_slab_begin(e.h, d.ptr, bounds[r + 1] - bounds[r], h, w, r, world, bounds[r], n_total, sp, len(spare), int(skip_below)),
                  "lm_slab_begin")
    rounds = 0
    while True:
        lens = [int(e.L.lib.lm_slab_pending(e.h)) for e in engines]
        assert min(lens) >= 0
        stride = max(lens)
        gathered = engines[0].empty((world * max(stride, 1),), np.int32)
        for r, e in enumerate(engines):
            e.L.check(e.L.lib.lm_slab_emit(e.h, gathered.ptr + 4 * r * stride), "lm_slab_emit")
        status = [e.L.check(e.L.lib.lm_slab_step(e.h, gathered.ptr, stride, (C.c_int64 * world)(*lens)), "lm_slab_step") for e in engines]
        gathered.free()
        rounds += 1
        assert len(set(status)) == 1, status
        if status[0] == 1:
            break
    assert rounds == 6, rounds
    out = np.concatenate([d.download() for d in slabs])
    for d in slabs:
        d.free()
    return out
