"""numpy -> numpy (lm_apply_host) against device-resident (lm_apply_dev) on the bench volume, with the C side's own breakdown
of one host call (LM_HOST_TIMING=1)."""
import os, sys, time
os.environ["LM_HOST_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lungmask_amd import _native as nat
from lungmask_amd import synthetic as syn
eng = nat.Engine(0); eng.load_state_dict(0, syn.synthetic_state_dict(3))
vol = syn.phantom(300, 512, 512)
vd = eng.to_device(vol); od = eng.empty(vol.shape, np.uint8)
for _ in range(2): eng.apply_dev(0, vd, od)
eng.sync(); t = time.perf_counter()
for _ in range(5): eng.apply_dev(0, vd, od)
eng.sync(); res = (time.perf_counter() - t) / 5 * 1e3
out = eng.apply(0, vol)
t = time.perf_counter()
for _ in range(5): eng.apply(0, vol, out=out)  # caller-owned output array, reused
host = (time.perf_counter() - t) / 5 * 1e3
t = time.perf_counter()
for _ in range(5): fresh = eng.apply(0, vol)   # a fresh 79 MB array per call: page faults + unmapping on the Python side
host_fresh = (time.perf_counter() - t) / 5 * 1e3
print(f"device-resident {res:.2f} ms   numpy->numpy {host:.2f} ms (+{host - res:.2f} ms, {100 * (host / res - 1):.1f} %) with a reused output array, "
      f"{host_fresh:.2f} ms with a fresh one   identical: {np.array_equal(out, od.download()) and np.array_equal(fresh, out)}")
