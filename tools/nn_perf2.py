"""Scratch probe: two engines (two streams, two workspaces) fed alternately -- does inter-batch overlap pay?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lungmask_amd import _native as nat
from lungmask_amd import synthetic as uo
B = 20; iters = 12
sd = uo.synthetic_state_dict(3)
engs = [nat.Engine(0) for _ in range(3)]
xs, labs = [], []
for e in engs:
    e.load_state_dict(0, sd); e.set_precision("split_f16"); e.set_streams(1)
    xs.append(e.to_device(np.random.default_rng(0).random((B, 256, 256), dtype=np.float32)))
    labs.append(e.empty((B, 256, 256), np.uint8))
for n_eng in (1, 2, 3, 1, 2, 3):
    for e, x, l in zip(engs, xs, labs): e.forward_dev(0, x, l)
    for e in engs: e.sync()
    t = time.time()
    for i in range(iters):
        k = i % n_eng
        engs[k].forward_dev(0, xs[k], labs[k])
    for e in engs: e.sync()
    dt = (time.time() - t) / iters
    print(f"{n_eng} stream(s): {dt*1e3:.2f} ms/batch  {B/dt:.1f} slices/s")
