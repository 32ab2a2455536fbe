"""Host-boundary cost: LMInferer-style apply(numpy) -> numpy vs the device-resident apply_dev (bench.py's `value`)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from lungmask_amd import synthetic as uo; po = uo
from lungmask_amd import _native

eng = _native.Engine(0)
eng.load_state_dict(0, uo.synthetic_state_dict(3))
vol = po.phantom(300, 512, 512, seed=2024)
d = eng.to_device(vol); o = eng.empty(vol.shape, np.uint8)
for _ in range(2):
    eng.apply_dev(0, d, o); eng.sync()
t = []
for _ in range(5):
    t0 = time.perf_counter(); eng.apply_dev(0, d, o); eng.sync(); t.append(time.perf_counter() - t0)
print("device-resident  ms:", [round(x * 1e3, 1) for x in t])
for _ in range(2):
    eng.apply(0, vol)
t = []
for _ in range(5):
    t0 = time.perf_counter(); r = eng.apply(0, vol); t.append(time.perf_counter() - t0)
print("host to host     ms:", [round(x * 1e3, 1) for x in t])
t0 = time.perf_counter(); d.upload(vol); eng.sync(); t1 = time.perf_counter(); x = o.download(); t2 = time.perf_counter()
print("H2D 157MB ms %.1f   D2H 79MB ms %.1f" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
