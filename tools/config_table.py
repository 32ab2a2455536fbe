"""Throughput of BASELINE.json configs 2-4 (volume resident in HBM, labels left in HBM) -- fills the table of BASELINE.md section 5."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lungmask_amd import _native as nat
from lungmask_amd import synthetic as uo; po = uo
eng = nat.Engine(0)
eng.load_state_dict(0, uo.synthetic_state_dict(3)); eng.load_state_dict(1, uo.synthetic_state_dict(6))
n = 300
vol = po.phantom(n, 512, 512)
vd = eng.to_device(vol); out = eng.empty(vol.shape, np.uint8)
def T(f, reps=3):
    f(); eng.sync(); t = time.perf_counter()
    for _ in range(reps): f()
    eng.sync(); return (time.perf_counter() - t) / reps
for name, f, flop in (("cfg2 R231 512x512x300 b20", lambda: eng.apply_dev(0, vd, out), 96.200556544e9),
                      ("cfg3 LTRCLobes 512x512x300 b20", lambda: eng.apply_dev(1, vd, out), 96.225722368e9),
                      ("cfg4 LTRCLobes_R231 fused 512x512x300 b20", lambda: eng.apply_dev(1, vd, out, fill_slot=0), 192.426e9)):
    dt = T(f)
    print(f"{name:44s} {n/dt:8.1f} slices/s  {dt*1e3:7.1f} ms  {n*flop/dt/1e12:6.1f} TFLOP/s  post: {eng.postprocess_info()}")
