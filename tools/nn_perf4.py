"""Scratch probe: does an internal batch that makes the work-item count a multiple of the CU count help?
(items per conv launch: 128B @256^2, 64B @128^2, 32B @64^2, 16B @32^2, 8B @16^2 -> B = 32 fills 256 CUs exactly)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lungmask_amd import _native as nat
from lungmask_amd import synthetic as uo
eng = nat.Engine(0); eng.load_state_dict(0, uo.synthetic_state_dict(3))
lib = eng.L.lib
for n in (300, 320):
    x = eng.to_device(np.random.default_rng(0).random((n, 256, 256), dtype=np.float32)); lab = eng.empty((n, 256, 256), np.uint8)
    for streams in (1, 2):
        eng.set_streams(streams)
        for bs in (20, 32, 64, 96, 160, 20):
            def f(): eng.L.check(lib.lm_forward_batches_dev(eng.h, 0, x.ptr, n, 256, 256, bs, lab.ptr))
            f(); eng.sync(); t = time.perf_counter()
            for _ in range(3): f()
            eng.sync(); dt = (time.perf_counter() - t) / 3
            print(f"n {n} lanes {streams} batch {bs:4d}: {dt*1e3:7.2f} ms  {n/dt:8.1f} slices/s", flush=True)
    x.free(); lab.free()
