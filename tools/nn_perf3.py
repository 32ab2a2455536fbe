"""Scratch probe: slices/s of lm_forward_batches_dev for several internal batch sizes (two lanes)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lungmask_amd import _native as nat
from lungmask_amd import synthetic as uo
eng = nat.Engine(0); eng.load_state_dict(0, uo.synthetic_state_dict(3))
n = 300
x = eng.to_device(np.random.default_rng(0).random((n, 256, 256), dtype=np.float32)); lab = eng.empty((n, 256, 256), np.uint8)
lib = eng.L.lib
for bs in (20, 30, 40, 50, 60, 75, 100, 150, 20):
    def f(): eng.L.check(lib.lm_forward_batches_dev(eng.h, 0, x.ptr, n, 256, 256, bs, lab.ptr))
    f(); eng.sync(); t = time.perf_counter()
    for _ in range(3): f()
    eng.sync(); dt = (time.perf_counter() - t) / 3
    print(f"batch {bs:4d}: {dt*1e3:7.2f} ms / 300 slices  {n/dt:8.1f} slices/s")
