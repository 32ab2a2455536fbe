"""CPU suite: the N>1 slice-sharded path (lungmask_amd/pipeline.py) with
world_size=2 over gloo.  The kernels run through the tests/emu emulator at a
tiny resolution; what is under test is the sharding, the two all-gathers
(ragged shards included) and that the result does not depend on the world size."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="4")
    import torch.distributed as dist

    from lungmask_amd import _native as nat
    from lungmask_amd.build import build_emu
    from lungmask_amd.pipeline import ShardedPipeline, shard_bounds
    from oracle import prepost_oracle as po
    from oracle import unet_oracle as uo

    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = nat.Engine(0, nat.Library(build_emu(), allow_emulation=True))
    eng.load_state_dict(0, uo.synthetic_state_dict(3))
    vol = po.phantom(n_total, 96, 80, seed=3)
    b = shard_bounds(n_total, world)
    shard = torch.from_numpy(vol[b[rank] : b[rank + 1]].copy())
    pipe = ShardedPipeline(eng, slot=0, batch_size=2, resolution=(32, 32), dist=dist, device="cpu")
    out = pipe.apply_shard(shard, n_total).numpy().copy()
    np.save(os.path.join(outdir, f"out{rank}.npy"), out)
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


def test_shard_bounds():
    from lungmask_amd.pipeline import shard_bounds

    assert shard_bounds(2400, 8) == [300 * i for i in range(9)]
    assert shard_bounds(3, 2) == [0, 2, 3]
    assert shard_bounds(1, 4) == [0, 1, 1, 1, 1]


@pytest.mark.slow
def test_two_rank_gloo_matches_single_rank(emu_engine, tmp_path):
    from lungmask_amd.build import build_emu
    from lungmask_amd.pipeline import ShardedPipeline
    from oracle import prepost_oracle as po
    from oracle import unet_oracle as uo

    build_emu()
    n_total = 3  # ragged: rank 0 gets 2 slices, rank 1 gets 1
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_total, str(tmp_path)), nprocs=2, join=True)
    out0 = np.load(tmp_path / "out0.npy")
    out1 = np.load(tmp_path / "out1.npy")
    assert out0.shape == (n_total, 96, 80) and np.array_equal(out0, out1)  # every rank holds the full result
    # single-rank reference through the same stage calls
    emu_engine.load_state_dict(0, uo.synthetic_state_dict(3))
    vol = po.phantom(n_total, 96, 80, seed=3)
    single = ShardedPipeline(emu_engine, slot=0, batch_size=2, resolution=(32, 32)).apply_shard(torch.from_numpy(vol), n_total).numpy()
    assert np.array_equal(out0, single)
    # and the stages around the network against the oracle
    xs, boxes = po.preprocess(vol, [32, 32])
    lab = emu_engine.forward(0, po.normalise(xs)[:, None], want_logp=False)[0]
    post = po.postprocessing(lab.copy())
    expect = np.asarray([po.reshape_mask(post[i], boxes[i], vol.shape[1:]) for i in range(n_total)], dtype=np.uint8)
    assert np.array_equal(single, expect)
