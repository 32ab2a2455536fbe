"""Where does ShardedPipeline.apply_shard spend its time compared with lm_apply_dev?  (world of one)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from lungmask_amd import synthetic as uo; po = uo
from lungmask_amd import _native as nat
from lungmask_amd.pipeline import ShardedPipeline
import torch.distributed as dist

use_dist = len(sys.argv) > 1 and sys.argv[1] == "dist"
if use_dist:
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29544", rank=0, world_size=1, device_id=torch.device("cuda", 0))
eng = nat.Engine(0)
eng.load_state_dict(0, uo.synthetic_state_dict(3))
vol = po.phantom(300, 512, 512, seed=2024)
vd = eng.to_device(vol); od = eng.empty(vol.shape, np.uint8)
def T(f, n=3):
    f(); eng.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    eng.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("lm_apply_dev            %.1f ms" % T(lambda: eng.apply_dev(0, vd, od)))
vt = torch.from_numpy(vol).to("cuda:0")
for sharded in (True, False):
    pipe = ShardedPipeline(eng, dist=dist if use_dist else None, device="cuda:0", sharded_post=sharded)
    print(f"apply_shard sharded_post={sharded}  %.1f ms" % T(lambda: pipe.apply_shard(vt, 300)))
    e, lib = eng, eng.L.lib
    bounds, bbox, lab_all, lab_loc = pipe.shard_buffers(300)
    xf = pipe._tensor("xf", (300, 256, 256), torch.float32)
    def pre(): e.L.check(lib.lm_preprocess_dev(e.h, vt.data_ptr(), 0, 300, 512, 512, 256, 256, bbox.data_ptr(), xf.data_ptr(), None, None))
    def fwd(): e.L.check(lib.lm_forward_batches_dev(e.h, 0, xf.data_ptr(), 300, 256, 256, 20, lab_loc.data_ptr()))
    print("   preprocess %.1f ms   forward_batches %.1f ms   assemble %.1f ms" % (T(pre), T(fwd), T(lambda: pipe.assemble(300, 512, 512))))
x = eng.empty((300, 256, 256), np.float32); lab = eng.empty((300, 256, 256), np.uint8)
print("forward_batches on engine-allocated buffers %.1f ms" % T(lambda: eng.L.check(eng.L.lib.lm_forward_batches_dev(eng.h, 0, x.ptr, 300, 256, 256, 20, lab.ptr))))
if use_dist: dist.destroy_process_group()
