"""A/B of the loader-side fusions (lm_set_fusion masks, one process, one box): network forward of 300 slices, two lanes and one
lane, labels compared across masks; then per-layer HIP-event times of the kernels a fusion touches.  argv: masks (default 0 1)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lungmask_amd import _native as nat
from lungmask_amd import synthetic as uo

masks = [int(a) for a in sys.argv[1:]] or [0, 1]
eng = nat.Engine(0)
eng.load_state_dict(0, uo.synthetic_state_dict(3))
x_h = np.random.default_rng(0).random((300, 256, 256), dtype=np.float32)
x = eng.to_device(x_h); lab = eng.empty((300, 256, 256), np.uint8)
f = lambda: eng.L.check(eng.L.lib.lm_forward_batches_dev(eng.h, 0, x.ptr, 300, 256, 256, 20, lab.ptr))
outs = {}
for rep in range(2):
    for m in masks:
        eng.set_fusion(m)
        res = []
        for lanes in (2, 1):
            eng.set_streams(lanes)
            f(); eng.sync(); t = time.perf_counter()
            for _ in range(4): f()
            eng.sync(); res.append((time.perf_counter() - t) / 4 * 1e3)
        outs.setdefault(m, lab.download())
        print(f"fusion mask {m}: two lanes {res[0]:6.2f} ms   one lane {res[1]:6.2f} ms", flush=True)
print("labels identical across masks:", all(np.array_equal(outs[masks[0]], o) for o in outs.values()))
eng.set_streams(1)
xb = eng.to_device(x_h[:20]); lb = eng.empty((20, 256, 256), np.uint8)
for m in masks:
    eng.set_fusion(m)
    eng.forward_dev(0, xb, lb); eng.sync()
    eng.profile(2); eng.profile_reset()
    for _ in range(5): eng.forward_dev(0, xb, lb)
    eng.sync()
    tot = 0.0
    print(f"-- per layer, mask {m} (one lane, batch 20)")
    for s in eng.profile_read():
        ms = s['total_ms'] / max(s['launches'], 1); tot += s['total_ms'] / 5
        print(f"   {s['name']:40s} n={s['launches']:3d} avg={ms:7.3f} ms  {s['flops']/max(s['total_ms'],1e-9)/1e9:7.1f} TFLOP/s")
    print(f"   sum of kernels per batch: {tot:.3f} ms")
    eng.profile(0)
