"""Slice-sharded multi-GPU form of `LMInferer.apply` (one process per GPU,
`torch.distributed` over RCCL/xGMI) built from the stage-level C-ABI calls.

Sharding: pre-processing, the network forward and the argmax are
slice-wise (utils.py:48-51, mask.py:173-187) -> contiguous slice blocks per rank,
weights replicated, NO collective.  The 3-D post-processing (utils.py:272-358) spans
the whole volume -> ONE all-gather of the uint8 256x256 label shards (64 KiB/slice);
every rank then runs the identical deterministic post-processing, un-crops its own
slices, and a second all-gather assembles the [n,h,w] result on every rank.  xGMI is
point-to-point and fully connected inside a node, so both all-gathers are single-step
and per-link bound (shard_bytes / ~153 GB/s): 19.7 MB + 78.6 MB per rank at 300
slices/rank, well under a millisecond each.

All buffers are torch tensors (device memory + collectives are torch's job here,
nothing else); the engine receives raw pointers.  In the CPU test-suite the tensors
are host tensors, the backend is gloo and the engine is the test emulation.
"""
from __future__ import annotations

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


class ShardedPipeline:
    """engine: lungmask_amd._native.Engine; dist: the torch.distributed module (initialised) or None;
    device: torch device that matches the engine's memory space ('cuda:<i>' or 'cpu' under emulation)."""

    def __init__(self, engine, slot: int = 0, batch_size: int = 20, volume_postprocessing: bool = True,
                 resolution: Sequence[int] = (256, 256), dist=None, device="cpu"):
        self.e = engine
        self.slot = slot
        self.batch_size = int(batch_size)
        self.volume_postprocessing = volume_postprocessing
        self.res = tuple(int(r) for r in resolution)
        self.dist = dist
        self.device = torch.device(device)
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self._buf = {}

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

    def apply_shard(self, vol_shard: torch.Tensor, n_total: int) -> torch.Tensor:
        """vol_shard: this rank's contiguous slice block [n_r,h,w] int16, resident in the engine's memory
        space.  Returns the FULL uint8 label volume [n_total,h,w] (same memory space)."""
        e, lib = self.e, self.e.L.lib
        n_r, h, w = (int(s) for s in vol_shard.shape)
        bounds = shard_bounds(n_total, self.world)
        counts = [bounds[r + 1] - bounds[r] for r in range(self.world)]
        assert n_r == counts[self.rank], (n_r, counts, self.rank)
        assert vol_shard.dtype == torch.int16 and vol_shard.is_contiguous()
        maxc = max(counts)
        oh, ow = self.res
        bbox = self._tensor("bbox", (maxc, 4), torch.int32)
        xf = self._tensor("xf", (maxc, oh, ow), torch.float32)
        lab_all = self._tensor("lab_all", (self.world * maxc, oh, ow), torch.uint8)
        lab_loc = lab_all[self.rank * maxc : (self.rank + 1) * maxc] if self.world > 1 else lab_all
        self._sync_torch()
        # ---- sliced stages: no communication
        if n_r:
            e.L.check(lib.lm_preprocess_dev(e.h, vol_shard.data_ptr(), 0, n_r, h, w, oh, ow, bbox.data_ptr(), xf.data_ptr(), None, None), "lm_preprocess_dev")
            e.L.check(lib.lm_forward_batches_dev(e.h, self.slot, xf.data_ptr(), n_r, oh, ow, self.batch_size, lab_loc.data_ptr()), "lm_forward_batches_dev")
        e.sync()
        # ---- exchange #1: 256^2 label shards -> whole label volume on every rank (RCCL all-gather, in place)
        if self.world > 1:
            self.dist.all_gather_into_tensor(lab_all.view(-1), lab_loc.reshape(-1).clone() if self.device.type == "cpu" else lab_loc.reshape(-1))
            self._sync_torch()
            if any(c != maxc for c in counts):  # ragged tail: compact the padded shards
                full = torch.cat([lab_all[r * maxc : r * maxc + counts[r]] for r in range(self.world)]).contiguous()
            else:
                full = lab_all
        else:
            full = lab_all[:n_r]
        self._sync_torch()
        if self.volume_postprocessing and n_total:
            e.L.check(lib.lm_postprocess_dev(e.h, full.data_ptr(), n_total, oh, ow, None, 0, 3), "lm_postprocess_dev")
        # ---- un-crop own slices
        out_all = self._tensor("out_all", (self.world * maxc, h, w), torch.uint8)
        out_loc = out_all[self.rank * maxc : (self.rank + 1) * maxc] if self.world > 1 else out_all
        if n_r:
            mine = full[bounds[self.rank] : bounds[self.rank + 1]]
            e.L.check(lib.lm_reshape_mask_dev(e.h, mine.data_ptr(), bbox.data_ptr(), n_r, oh, ow, h, w, out_loc.data_ptr()), "lm_reshape_mask_dev")
        e.sync()
        # ---- exchange #2: output shards
        if self.world > 1:
            self.dist.all_gather_into_tensor(out_all.view(-1), out_loc.reshape(-1).clone() if self.device.type == "cpu" else out_loc.reshape(-1))
            self._sync_torch()
            if any(c != maxc for c in counts):
                return torch.cat([out_all[r * maxc : r * maxc + counts[r]] for r in range(self.world)]).contiguous()
            return out_all
        return out_all[:n_r]
