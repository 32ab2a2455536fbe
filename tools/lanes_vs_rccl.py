"""Does an initialised RCCL process group cost the two forward lanes their overlap?
argv[1]: none | init (init_process_group only) | coll (init + one collective) ; argv[2]: engine 'first' or 'last'"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from lungmask_amd import synthetic as uo
from lungmask_amd import _native as nat
import torch.distributed as dist
mode, order = sys.argv[1], sys.argv[2]
def init():
    if mode == "none": return
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29545", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    if mode == "coll":
        t = torch.ones(4, device="cuda:0"); dist.all_reduce(t); torch.cuda.synchronize()
    if mode == "tensor":
        t = torch.ones(4, device="cuda:0"); torch.cuda.synchronize()
if order == "last": init()
eng = nat.Engine(0); eng.load_state_dict(0, uo.synthetic_state_dict(3))
if order == "first": init()
x = eng.empty((300, 256, 256), np.float32); lab = eng.empty((300, 256, 256), np.uint8)
f = lambda: eng.L.check(eng.L.lib.lm_forward_batches_dev(eng.h, 0, x.ptr, 300, 256, 256, 20, lab.ptr))
def T():
    f(); eng.sync(); t0 = time.perf_counter()
    for _ in range(3): f()
    eng.sync(); return (time.perf_counter() - t0) / 3 * 1e3
t2 = T(); eng.set_streams(1); t1 = T()
print(f"{mode:6s} engine {order:5s} prio={os.environ.get('LM_LANE2_PRIORITY')}: two lanes {t2:.1f} ms, one lane {t1:.1f} ms", flush=True)
if mode != "none" and dist.is_initialized(): dist.destroy_process_group()
