"""Scratch perf probe of the network forward (not the bench)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lungmask_amd import _native as nat
from lungmask_amd import synthetic as uo

B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
eng = nat.Engine(0)
eng.load_state_dict(0, uo.synthetic_state_dict(3))
eng.set_precision(sys.argv[3] if len(sys.argv) > 3 else "f32")
x = eng.to_device(np.random.default_rng(0).random((B, 256, 256), dtype=np.float32))
lab = eng.empty((B, 256, 256), np.uint8)
for _ in range(2):
    eng.forward_dev(0, x, lab)
eng.sync()
t = time.time()
for _ in range(iters):
    eng.forward_dev(0, x, lab)
eng.sync()
dt = (time.time() - t) / iters
print(f"B={B}: {dt*1e3:.2f} ms/batch  {B/dt:.1f} slices/s  {B*96.2e9/dt/1e12:.1f} TFLOP/s")
eng.profile(2); eng.profile_reset()
for _ in range(iters):
    eng.forward_dev(0, x, lab)
eng.sync()
for s in eng.profile_read():
    ms = s['total_ms']/max(s['launches'],1)
    print(f"{s['name']:24s} n={s['launches']:4d} avg={ms:8.3f} ms  total={s['total_ms']:9.2f} ms  {s['flops']/max(s['total_ms'],1e-9)/1e9:8.1f} TFLOP/s  {s['bytes']/max(s['total_ms'],1e-9)/1e6:8.1f} GB/s")
