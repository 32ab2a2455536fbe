"""Slice-sharded multi-GPU form of `LMInferer.apply` (one process per GPU,
`torch.distributed` over RCCL/xGMI) built from the stage-level C-ABI calls.

Sharding: pre-processing, the network forward and the argmax are
slice-wise (utils.py:48-51, mask.py:173-187) -> contiguous slice blocks per rank,
weights replicated, NO collective.  The 3-D post-processing (utils.py:272-358) spans
the whole volume; two forms:

* slab-sharded (default from four ranks on): every rank post-processes its own slab and the slabs are tied
  together by six small all-gathers of face planes / atom tables (`lm_slab_*`; four in the region-graph form, LM_SLAB_GRAPH=1,
  csrc/slab_engine.hip) -- the voxel passes scale with 1/world;
* gathered (`sharded_post=False`, the default below four ranks, and whenever a rank has no slice): ONE all-gather of the
  uint8 256x256 label shards (64 KiB/slice), then every rank runs the identical
  deterministic whole-volume post-processing (the serial fraction of weak scaling).

The fused LTRCLobes_R231 mode (mask.py:223-232; SURVEY.md section 8e row 4, `fill_slot`) runs both networks on the rank's slice
block, post-processes and un-crops both label shards, agrees on `spare = res_l.max() + 1` over ALL ranks (one 4-byte all-gather),
fuses the own slab (lm_fuse_spare_dev) and runs the full-resolution post-processing in the same two forms.

Besides one process per GPU there is a single-process form: `InProcessGroup` -- N engines (one per device) driven by N threads
of this process, the exchanges as peer copies between the engines' buffers.  `LMInferer(device_ids=[...])` uses it.

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
import threading
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


class InProcessGroup:
    """N ranks as N threads of ONE process (one engine per rank, normally one device per engine): the `dist` of rank r is
    `member(r, engine_r)`.  An all-gather is a rendezvous of the ranks' threads and world x (world - 1) plain copies between the
    engines' buffers (peer copies over xGMI when the engines sit on different devices) -- no RCCL, no torch.distributed.  The
    host synchronises each engine's stream around the rendezvous; results are those of any other `dist`.  A rank that fails
    breaks the barrier, so the other ranks raise instead of waiting for ever."""

    def __init__(self, world: int):
        self.world = int(world)
        self.barrier = threading.Barrier(self.world)
        self.slots = [None] * self.world

    def member(self, rank: int, engine):
        return _InProcessMember(self, int(rank), engine)

    def abort(self):
        self.barrier.abort()

    def reset(self):
        self.barrier.reset()


class _InProcessMember:
    def __init__(self, group: InProcessGroup, rank: int, engine):
        self.g, self.rank, self.e = group, rank, engine

    def get_world_size(self) -> int:
        return self.g.world

    def get_rank(self) -> int:
        return self.rank

    def all_gather_into_tensor(self, out: torch.Tensor, mine: torch.Tensor):
        g, n = self.g, mine.numel()
        assert out.is_contiguous() and mine.is_contiguous() and out.numel() == g.world * n
        self.e.sync()  # this rank's contribution is complete (the engine's stream produced it)
        g.slots[self.rank] = mine
        g.barrier.wait()
        flat = out.view(-1)
        for r in range(g.world):
            src, dst = g.slots[r].view(-1), flat[r * n : (r + 1) * n]
            assert src.numel() == n and src.dtype == mine.dtype
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)  # (on the engine's stream when the pipeline made it torch's current stream)
        if out.device.type == "cuda":
            torch.cuda.current_stream(out.device).synchronize()
        g.barrier.wait()  # nobody's `mine` is overwritten before every rank has read it
        g.slots[self.rank] = None


class _Marks:
    """Time stamps of ONE diagnostic call (bench.py's dist_breakdown): `mark(name)` waits for the engine's stream and closes the
    segment `name` -- everything since the previous mark, the host's share included (the slab protocol alternates host merges and
    kernels) -- on the host clock.  The extra synchronisations only exist in this one diagnostic call.  (A 40-55 ms first
    post-processing segment that round 6 chased through two forms of this class was the interpreter's garbage collector: bench.py
    now keeps it off during the measured passes.)"""

    def __init__(self, stream, device):
        import time

        self.stream, self.device, self.clock = stream, device, time.perf_counter
        self.names, self.stamps, self.extra = [], [], {}
        self._stamp()

    def _stamp(self):
        if self.stream is not None:
            self.stream.synchronize()
        self.stamps.append(self.clock())

    def mark(self, name: str):
        self.names.append(name)
        self._stamp()

    def add(self, key: str, ms: float):
        self.extra[key] = self.extra.get(key, 0.0) + ms

    def segments(self):
        """[(name, ms)] in order."""
        return [(n, (self.stamps[i + 1] - self.stamps[i]) * 1e3) for i, n in enumerate(self.names)]


class ShardedPipeline:
    """engine: lungmask_amd._native.Engine; dist: the torch.distributed module (initialised), a NativeDist, an InProcessGroup
    member, or None; device: torch device that matches the engine's memory space ('cuda:<i>' or 'cpu' under emulation);
    fill_slot: model slot of the fill model of the fused LTRCLobes_R231 mode (mask.py:223-232), -1 = none."""

    _TORCH_DTYPES = {torch.int16: 0, torch.int32: 1, torch.float32: 2, torch.float64: 3, torch.int64: 6}  # include/lungmask_hip.h: LM_I16 ...

    def __init__(self, engine, slot: int = 0, batch_size: int = 20, volume_postprocessing: bool = True,
                 resolution: Sequence[int] = (256, 256), dist=None, device="cpu", sharded_post=None, fill_slot: int = -1):
        self.e = engine
        self.slot = slot
        self.fill_slot = int(fill_slot)
        self.batch_size = int(batch_size)
        self.volume_postprocessing = volume_postprocessing
        self.res = tuple(int(r) for r in resolution)
        self.dist = dist
        self.device = torch.device(device)
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        # None: by world size.  The slab protocol's fixed part (six exchanges, three host table merges) only pays from four ranks
        # on -- below the crossover the redundant whole-volume pass on the gathered labels is cheaper (tools/slab_timing.py)
        # (round 5, lung-like labels of ONE 300 w-slice volume, profiles/history/r05d_slab_timing_lunglike.log: whole-volume pass 2.7 / 5.0 / 9.1 ms
        # at 600 / 1200 / 2400 slices against 3.7 / 4.7 / 6.5 ms per rank for the protocol at 2 / 4 / 8 ranks: the crossover is at four)
        self.sharded_post = (self.world >= 4) if sharded_post is None else bool(sharded_post)
        self._buf = {}
        self._slab_caps = {}   # agreed capacity (ints) of the variable-length table exchange of every protocol round, per volume geometry
        self._slab_key = None
        self.collectives = 0   # collectives issued for variable-length tables (tests / tools/slab_timing.py read it)
        # breakdown of ONE call (bench.py's N > 1 line; `want_breakdown = True` before apply_shard, `breakdown()` after): where the
        # step's time goes on the engine's stream -- sliced stages, every collective with its bytes, post-processing, un-crop --
        # and how long the host sat in the slab protocol's table merges
        self.want_breakdown = False
        self._marks = None
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

    def _all_gather(self, out: torch.Tensor, mine: torch.Tensor, name: str = "tables"):
        # gloo must not alias input and output; RCCL gathers in place.  Called with the engine's stream current: no host
        # synchronisation here (without that stream, e.g. an engine of another build, fall back to a device sync)
        if self._marks is not None:
            self._marks.mark("compute")  # (whatever ran on the stream since the last mark; breakdown() sorts it into stages)
        self.dist.all_gather_into_tensor(out, mine.clone() if self.device.type == "cpu" else mine)
        if self._stream is None:
            self._sync_torch()
        if self._marks is not None:
            self._marks.mark(f"collective:{name}:{mine.numel() * mine.element_size()}")

    def _stage(self, name: str):
        if self._marks is not None:
            self._marks.mark("stage:" + name)

    def breakdown(self):
        """The call's time line as bench.py reports it (dist_breakdown), or None when `want_breakdown` was off.  Stream time between
        two marks belongs to the stage that closes the run of segments (`stage:` marks); collectives are listed one by one with
        the bytes this rank contributed; host_merge_ms = wall time the host spent inside lm_slab_step (table merges + the waits
        for the device data they need)."""
        m = self._marks
        if m is None:
            return None
        stages, coll, pend = {}, [], 0.0
        for name, ms in m.segments():
            if name.startswith("collective:"):
                _, what, nbytes = name.split(":")
                coll.append({"name": what, "bytes_per_rank": int(nbytes), "ms": round(ms, 4)})
            elif name.startswith("stage:"):
                stages[name[6:]] = round(stages.get(name[6:], 0.0) + pend + ms, 4)
                pend = 0.0
            else:
                pend += ms
        out = {k + "_ms": v for k, v in stages.items()}
        out["segments"] = [[n, round(ms, 4)] for n, ms in m.segments()]  # the raw time line, in stream order
        out["host_merge_ms"] = round(m.extra.get("host_merge", 0.0), 4)
        out["collectives"] = coll
        out["collectives_ms"] = round(sum(c["ms"] for c in coll), 4)
        return out

    def postprocess_slab(self, lab_slab: torch.Tensor, z0: int, n_total: int, spare: Sequence[int] = (), skip_below: int = 3):
        """utils.postprocessing over a volume whose slices are spread over the ranks; `lab_slab` (uint8 [n_r,h,w], n_r >= 1,
        slices [z0, z0+n_r)) is processed IN PLACE.  Protocol of include/lungmask_hip.h (lm_slab_*)."""
        # (the agreed table capacities are remembered per volume geometry: the fused mode runs the protocol at the network's
        # resolution and at full resolution within one volume)
        self._slab_key = (tuple(int(v) for v in lab_slab.shape[1:]), len(spare))
        if self._marks is not None:
            self._marks.mark("compute:before_slab_begin")
        with self._on_engine_stream():
            return self._postprocess_slab(lab_slab, z0, n_total, spare, skip_below)

    def _postprocess_slab(self, lab_slab, z0, n_total, spare, skip_below):
        e, lib = self.e, self.e.L.lib
        n_r, h, w = (int(v) for v in lab_slab.shape)
        sp = (C.c_int * max(len(spare), 1))(*[int(v) for v in spare])
        e.L.check(lib.lm_slab_begin(e.h, lab_slab.data_ptr(), n_r, h, w, self.rank, self.world, int(z0), int(n_total), sp, len(spare), int(skip_below)),
                  "lm_slab_begin")
        if self._marks is not None:
            self._marks.mark("compute:slab_begin")
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
                    self._all_gather(gathered, mine, f"slab_planes_round{rnd}")
            else:
                # variable-length tables: every rank sends [length | table | padding] of ONE agreed size, so the lengths travel
                # inside the table exchange (6 collectives per volume instead of 9).  The agreed capacity of round `rnd` is what
                # the previous volume needed plus a quarter -- every rank saw the same lengths, so every rank holds the same
                # number; the first volume (and a table that outgrows the capacity, which every rank notices at the same
                # time) exchanges the lengths first, as before.
                cap = self._slab_caps.get((self._slab_key, rnd), 0)
                lens = lens_known = None
                if cap > 0:
                    mine = self._tensor("slab_mine", (cap + 1,), torch.int32)
                    gathered = self._tensor("slab_all", (self.world * (cap + 1),), torch.int32)
                    mine[:1].fill_(n)
                    if n <= cap:
                        e.L.check(lib.lm_slab_emit(e.h, mine.data_ptr() + 4), "lm_slab_emit")
                    self._all_gather(gathered, mine, f"slab_table_round{rnd}")
                    self.collectives += 1
                    got = [int(v) for v in gathered.view(self.world, cap + 1)[:, 0].cpu().tolist()]
                    if max(got) <= cap:
                        lens, stride, hdr = got, cap + 1, 1
                    else:
                        lens_known = got  # somebody's table did not fit: exchange again at the exact size
                if lens is None:
                    if lens_known is None:
                        lens_t = self._tensor("slab_lens", (self.world,), torch.int64)
                        self._all_gather(lens_t, torch.tensor([n], dtype=torch.int64, device=self.device), f"slab_lengths_round{rnd}")
                        lens_known = [int(v) for v in lens_t.cpu().tolist()]
                        self.collectives += 1
                    lens, stride = lens_known, max(lens_known)
                    mine = self._tensor("slab_mine", (max(stride, 1),), torch.int32)
                    gathered = self._tensor("slab_all", (self.world * max(stride, 1),), torch.int32)
                    e.L.check(lib.lm_slab_emit(e.h, mine.data_ptr()), "lm_slab_emit")
                    if stride:
                        self._all_gather(gathered, mine, f"slab_table_round{rnd}")
                        self.collectives += 1
                # (monotone: alternating small and large volumes must not fall back to the exact-size exchange every other time)
                self._slab_caps[(self._slab_key, rnd)] = max(self._slab_caps.get((self._slab_key, rnd), 0), -(-(max(lens) + max(lens) // 4 + 256) // 1024) * 1024)
            if self._marks is not None:
                import time

                t0 = time.perf_counter()
            status = e.L.check(lib.lm_slab_step(e.h, gathered.data_ptr() + 4 * hdr, stride, (C.c_int64 * self.world)(*lens)), "lm_slab_step")
            if self._marks is not None:
                self._marks.add("host_merge", (time.perf_counter() - t0) * 1e3)
            rnd += 1
            if status == 1:
                return

    @staticmethod
    def _as_volume(volume: np.ndarray, what: str) -> np.ndarray:
        """The dtypes the engine pre-processes on the device (lm_preprocess_dev): int16 / int32 / int64 / float32 / float64."""
        volume = np.ascontiguousarray(volume)
        if volume.dtype not in (np.int16, np.int32, np.int64, np.float32, np.float64):
            raise TypeError(f"ShardedPipeline.{what}: int16/int32/int64/float32/float64 volume expected, got {volume.dtype} (LMInferer.apply widens the others)")
        return volume

    def apply(self, volume: np.ndarray) -> np.ndarray:
        """Convenience form of `LMInferer.apply` for a process group: every rank passes the SAME host volume [n,h,w],
        works on its own block of slices and returns the complete uint8 label volume."""
        vol = self._as_volume(volume, "apply")
        n_total = int(vol.shape[0])
        b = shard_bounds(n_total, self.world)
        shard = torch.from_numpy(vol[b[self.rank] : b[self.rank + 1]]).to(self.device)
        return np.array(self.apply_shard(shard.contiguous(), n_total).cpu().numpy(), copy=True)  # (never a view of a cached buffer)

    def apply_local(self, shard: np.ndarray, n_total: int, gather: bool = True, out: np.ndarray = None):
        """The rank-local form (config 5: 2400 slices over 8 GPUs): every rank passes ONLY its own contiguous block of slices
        `shard` [n_r,h,w] (n_r = shard_bounds(n_total, world) of this rank) -- no rank ever holds the whole input volume.
        gather=True: returns the complete uint8 label volume [n_total,h,w] (the reference's result on every rank, one all-gather of
        the output shards); gather=False: only this rank's [n_r,h,w] block (no output collective; host memory per rank stays 1/world).
        `out`: an optional uint8 C-contiguous host array of the result's shape that receives it."""
        shard = self._as_volume(shard, "apply_local")
        b = shard_bounds(int(n_total), self.world)
        if shard.shape[0] != b[self.rank + 1] - b[self.rank]:
            raise ValueError(f"rank {self.rank} of {self.world} owns slices [{b[self.rank]}, {b[self.rank + 1]}) of {n_total}: got {shard.shape[0]} slices")
        res = self.apply_shard(torch.from_numpy(shard).to(self.device).contiguous(), int(n_total), gather=gather)
        if out is not None:
            if out.dtype != np.uint8 or tuple(out.shape) != tuple(res.shape) or not out.flags.c_contiguous:
                raise ValueError("apply_local(out=...): need a C-contiguous uint8 array of the result's shape")
            torch.from_numpy(out).copy_(res)  # device -> the caller's array, no intermediate
            return out
        # a copy: `res` may be a view of a cached buffer that the next call overwrites (on the CPU .numpy() does not copy)
        return np.array(res.cpu().numpy(), copy=True)

    def shard_buffers(self, n_total: int, key: str = "lab_all"):
        """(bounds, bbox [maxc,4] int32, lab_all [world*maxc,oh,ow] u8, lab_loc = this rank's part of lab_all)."""
        bounds = shard_bounds(n_total, self.world)
        maxc = max(bounds[r + 1] - bounds[r] for r in range(self.world))
        oh, ow = self.res
        bbox = self._tensor("bbox", (maxc, 4), torch.int32)
        lab_all = self._tensor(key, (self.world * maxc, oh, ow), torch.uint8)
        lab_loc = lab_all[self.rank * maxc : (self.rank + 1) * maxc]
        return bounds, bbox, lab_all, lab_loc

    def apply_shard(self, vol_shard: torch.Tensor, n_total: int, gather: bool = True) -> torch.Tensor:
        """vol_shard: this rank's contiguous slice block [n_r,h,w] (int16/int32/int64/float32/float64), resident in the engine's
        memory space and produced on torch's CURRENT stream.  Returns the FULL uint8 label volume [n_total,h,w] (same memory
        space), or with gather=False this rank's block.  The result lives in a buffer of this object: valid until the next call."""
        e, lib = self.e, self.e.L.lib
        n_r, h, w = (int(s) for s in vol_shard.shape)
        bounds, bbox, _, lab_loc = self.shard_buffers(n_total)
        assert n_r == bounds[self.rank + 1] - bounds[self.rank], (n_r, bounds, self.rank)
        if vol_shard.dtype not in self._TORCH_DTYPES or not vol_shard.is_contiguous():
            raise TypeError(f"apply_shard: contiguous int16/int32/int64/float32/float64 tensor expected, got {vol_shard.dtype}")
        dtype = self._TORCH_DTYPES[vol_shard.dtype]
        oh, ow = self.res
        xf = self._tensor("xf", (max(n_r, 1), oh, ow), torch.float32)
        # the caller's shard is complete when its stream reaches this point: the engine's stream waits for exactly that -- an event,
        # not a device-wide synchronisation (other engines / lanes on this device keep running)
        if self._stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._stream.wait_event(ev)
        else:
            self._sync_torch()
        self._marks = _Marks(self._stream, self.device) if self.want_breakdown else None
        # ---- sliced stages: no communication
        if n_r:
            e.L.check(lib.lm_preprocess_dev(e.h, vol_shard.data_ptr(), dtype, n_r, h, w, oh, ow, bbox.data_ptr(), xf.data_ptr(), None, None), "lm_preprocess_dev")
            e.L.check(lib.lm_forward_batches_dev(e.h, self.slot, xf.data_ptr(), n_r, oh, ow, self.batch_size, lab_loc.data_ptr()), "lm_forward_batches_dev")
        if self.fill_slot < 0:
            if self._stream is None:
                e.sync()
            self._stage("forward")
            return self.assemble(n_total, h, w, gather=gather)
        # ---- fused mode: the fill model on the same pre-processed slices (the reference recomputes the identical pre-processing)
        _, _, _, lab_fill = self.shard_buffers(n_total, "lab_fill")
        if n_r:
            e.L.check(lib.lm_forward_batches_dev(e.h, self.fill_slot, xf.data_ptr(), n_r, oh, ow, self.batch_size, lab_fill.data_ptr()), "lm_forward_batches_dev")
        if self._stream is None:
            e.sync()
        self._stage("forward")
        return self.assemble_fused(n_total, h, w, gather=gather)

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

    def assemble_fused(self, n_total: int, h: int, w: int, gather: bool = True) -> torch.Tensor:
        """`assemble` for the fused mode: the base model's label shards in `shard_buffers(n_total)`, the fill model's in
        `shard_buffers(n_total, "lab_fill")`."""
        with self._on_engine_stream():
            out = self._assemble_fused(n_total, h, w, gather)
        if self._stream is not None:
            self._stream.synchronize()
        return out

    def _counts(self, n_total):
        bounds = shard_bounds(n_total, self.world)
        return bounds, [bounds[r + 1] - bounds[r] for r in range(self.world)]

    def _use_slabs(self, counts, n_total) -> bool:
        # (a process group of ONE rank still runs the exchange protocol: that is how the RCCL calls are exercised on a 1-GPU box)
        return bool(self.sharded_post and self.dist is not None and min(counts) >= 1 and n_total > 1)

    def _compact(self, all_t: torch.Tensor, counts, maxc) -> torch.Tensor:
        """[world * maxc, ...] with every rank's shard padded to maxc slices -> [n_total, ...]."""
        if any(c != maxc for c in counts):
            return torch.cat([all_t[r * maxc : r * maxc + counts[r]] for r in range(self.world)]).contiguous()
        return all_t

    def _post_lowres(self, n_total: int, key: str) -> torch.Tensor:
        """Volume post-processing (mask.py:191-194) of the network-resolution label shards in buffer `key`; returns this rank's
        post-processed slices [n_r,oh,ow]."""
        e, lib = self.e, self.e.L.lib
        bounds, _, lab_all, lab_loc = self.shard_buffers(n_total, key)
        _, counts = self._counts(n_total)
        n_r, maxc = counts[self.rank], max(counts)
        oh, ow = self.res
        if self._use_slabs(counts, n_total):
            # ---- post-processing on the own slab; six small exchanges inside (no label all-gather at all)
            mine_lab = lab_loc[:n_r]
            if self.volume_postprocessing:
                self.postprocess_slab(mine_lab, bounds[self.rank], n_total)
            return mine_lab
        # ---- exchange #1: 256^2 label shards -> whole label volume on every rank (RCCL all-gather, in place)
        if self.dist is not None:
            if self.volume_postprocessing:  # (without the volume pass every rank only needs its own slices)
                self._all_gather(lab_all.view(-1), lab_loc.reshape(-1), "label_shards")
                full = self._compact(lab_all, counts, maxc)
            else:
                return lab_loc[:n_r]
        else:
            full = lab_all[:n_r]
        if self._stream is None:
            self._sync_torch()
        if self.volume_postprocessing and n_total:
            e.L.check(lib.lm_postprocess_dev(e.h, full.data_ptr(), n_total, oh, ow, None, 0, 3), "lm_postprocess_dev")
        return full[bounds[self.rank] : bounds[self.rank + 1]]

    def _uncrop(self, mine_lab: torch.Tensor, out_loc: torch.Tensor, n_r: int, h: int, w: int):
        e, lib = self.e, self.e.L.lib
        oh, ow = self.res
        if n_r:  # (the bounding boxes of the own slices: written by the pre-processing into the "bbox" buffer)
            e.L.check(lib.lm_reshape_mask_dev(e.h, mine_lab.data_ptr(), self._buf["bbox"].data_ptr(), n_r, oh, ow, h, w, out_loc.data_ptr()), "lm_reshape_mask_dev")
        if self._stream is None:
            e.sync()

    def _gather_out(self, out_all: torch.Tensor, counts, gather: bool) -> torch.Tensor:
        n_r, maxc = counts[self.rank], max(counts)
        out_loc = out_all[self.rank * maxc : (self.rank + 1) * maxc]
        if not gather:
            return out_loc[:n_r]
        if self.dist is not None:
            self._all_gather(out_all.view(-1), out_loc.reshape(-1), "output_shards")
            res = self._compact(out_all, counts, maxc)
            self._stage("output_assembly")
            return res
        return out_all[:n_r]

    def _assemble(self, n_total: int, h: int, w: int, gather: bool = True) -> torch.Tensor:
        _, counts = self._counts(n_total)
        n_r, maxc = counts[self.rank], max(counts)
        mine_lab = self._post_lowres(n_total, "lab_all")
        self._stage("post")
        # ---- un-crop own slices
        out_all = self._tensor("out_all", (self.world * maxc, h, w), torch.uint8)
        self._uncrop(mine_lab, out_all[self.rank * maxc : (self.rank + 1) * maxc], n_r, h, w)
        self._stage("uncrop")
        # ---- exchange #2: output shards
        return self._gather_out(out_all, counts, gather)

    def _assemble_fused(self, n_total: int, h: int, w: int, gather: bool = True) -> torch.Tensor:
        """mask.py:223-232 on slice blocks: res_l / res_r = post-processed + un-cropped labels of the two models on the own slab;
        spare = max over ALL slabs + 1; fusion on the own slab; full-resolution post-processing with the spare label."""
        e, lib = self.e, self.e.L.lib
        bounds, counts = self._counts(n_total)
        n_r, maxc = counts[self.rank], max(counts)
        out_all = self._tensor("out_all", (self.world * maxc, h, w), torch.uint8)
        out_loc = out_all[self.rank * maxc : (self.rank + 1) * maxc]
        res_r = self._tensor("res_r", (max(n_r, 1), h, w), torch.uint8)
        self._uncrop(self._post_lowres(n_total, "lab_all"), out_loc, n_r, h, w)   # res_l (mask.py:222)
        self._uncrop(self._post_lowres(n_total, "lab_fill"), res_r, n_r, h, w)     # res_r (mask.py:227)
        self._stage("post")
        # ---- spare = res_l.max() + 1 over the whole volume (mask.py:228): every rank's maximum, one 4-byte all-gather
        mx = C.c_int(0)
        e.L.check(lib.lm_label_max_dev(e.h, out_loc.data_ptr(), n_r * h * w, C.byref(mx)), "lm_label_max_dev")
        top = mx.value
        if self.dist is not None and self.world > 1:
            mine = self._tensor("mx_mine", (1,), torch.int32)
            mx_all = self._tensor("mx_all", (self.world,), torch.int32)
            mine.fill_(top)
            self._all_gather(mx_all, mine, "spare_label_max")
            top = max(int(v) for v in mx_all.cpu().tolist())
        spare = (top + 1) & 0xff  # uint8 arithmetic, as the reference's numpy expression
        e.L.check(lib.lm_fuse_spare_dev(e.h, out_loc.data_ptr(), res_r.data_ptr(), n_r * h * w, spare), "lm_fuse_spare_dev")  # mask.py:229-230
        # ---- postprocessing(res_l, spare=[spare]) at full resolution (mask.py:232; not conditional on volume_postprocessing)
        self._stage("fusion")
        if self._use_slabs(counts, n_total):
            self.postprocess_slab(out_loc[:n_r], bounds[self.rank], n_total, spare=(spare,))
            self._stage("post_fullres")
            return self._gather_out(out_all, counts, gather)
        if self.dist is not None:
            self._all_gather(out_all.view(-1), out_loc.reshape(-1), "fused_slabs")
            full = self._compact(out_all, counts, maxc)
        else:
            full = out_all[:n_r]
        if self._stream is None:
            self._sync_torch()
        if n_total:
            sp = (C.c_int * 1)(spare)
            e.L.check(lib.lm_postprocess_dev(e.h, full.data_ptr(), n_total, h, w, sp, 1, 3), "lm_postprocess_dev")
        if self._stream is None:
            e.sync()
        self._stage("post_fullres")
        # every rank now holds the complete result: no output collective in this form
        return full if gather else full[bounds[self.rank] : bounds[self.rank + 1]]


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
    assert rounds in (4, 6), rounds  # six exchanges (the default), four in the region-graph form (LM_SLAB_GRAPH=1)
    postprocess_slabs_in_process.last_rounds = rounds
    out = np.concatenate([d.download() for d in slabs])
    for d in slabs:
        d.free()
    return out
