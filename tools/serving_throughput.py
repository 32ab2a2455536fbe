"""Serving-style throughput: several volumes in flight on ONE GPU, numpy in -> numpy out.
K engine handles (distinct handles are independent: own streams and workspaces), one Python thread each (ctypes
releases the GIL); while one handle post-processes / copies, the other's network keeps the matrix cores busy."""
import sys, time, threading
import numpy as np
sys.path.insert(0, ".")
from lungmask_amd import synthetic as uo; po = uo
from lungmask_amd import _native as nat

sd = uo.synthetic_state_dict(3)
vol = po.phantom(300, 512, 512, seed=2024)
ref = None
for K in (1, 2, 3):
    engines = [nat.Engine(0) for _ in range(K)]
    for e in engines:
        e.load_state_dict(0, sd)
        e.apply(0, vol)  # warm-up: workspaces
    n_vol = 6
    outs = [None] * K
    def work(i):
        for _ in range(n_vol):
            outs[i] = engines[i].apply(0, vol)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(K)]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    if ref is None: ref = outs[0]
    same = all(np.array_equal(o, ref) for o in outs)
    print(f"{K} handle(s): {K * n_vol * 300 / dt:8.1f} slices/s host-to-host aggregate ({dt / (K * n_vol) * 1e3:.1f} ms per volume amortised), identical results: {same}", flush=True)
    for e in engines: e.close()
