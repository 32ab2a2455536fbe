"""Worker of tests/test_gpu_apply.py::test_two_ranks_on_one_gpu_over_rccl: one rank of a 2-rank RCCL process group whose ranks
BOTH use cuda:0 (all a 1-GPU box offers).  Runs the slice-sharded pipeline on a ragged volume in both post-processing forms and
compares with the single-engine result.  Prints RCCL2_OK, or RCCL2_UNSUPPORTED: <reason> when RCCL refuses two ranks on one
device (the caller then skips)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist

    from lungmask_amd import _native as nat
    from lungmask_amd import synthetic as syn
    from lungmask_amd.pipeline import ShardedPipeline

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        t = torch.full((4,), float(rank), device="cuda:0")
        out = torch.empty(4 * world, device="cuda:0")
        dist.all_gather_into_tensor(out, t)  # the first collective creates the communicator
        torch.cuda.synchronize()
    except Exception as ex:  # noqa: BLE001  (RCCL reports duplicate devices as a generic DistBackendError)
        print("RCCL2_UNSUPPORTED:", str(ex).replace("\n", " ")[:400], flush=True)
        return 0
    eng = nat.Engine(0)
    eng.load_state_dict(0, syn.synthetic_state_dict(3))
    vol = syn.phantom(300, 512, 512, seed=2024, z0=140, z1=147)  # 7 slices: shards of 4 and 3
    expect = eng.apply(0, vol, batch_size=20)
    ok = True
    for sharded_post in (True, False):
        pipe = ShardedPipeline(eng, slot=0, batch_size=20, dist=dist, device="cuda:0", sharded_post=sharded_post)
        got = pipe.apply(vol)
        same = bool(np.array_equal(got, expect))
        print(f"rank {rank} sharded_post={sharded_post}: identical to the single-engine result: {same}", flush=True)
        ok = ok and same
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL2_OK" if ok else "RCCL2_MISMATCH", flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
