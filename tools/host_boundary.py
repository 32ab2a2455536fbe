"""numpy -> numpy (lm_apply_host) against device-resident (lm_apply_dev) on the bench volume, with the C side's own breakdown
of one host call (LM_HOST_TIMING=1)."""
import os, sys, time
os.environ["LM_HOST_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lungmask_amd import _native as nat
from lungmask_amd import synthetic as syn
eng = nat.Engine(0); eng.load_state_dict(0, syn.synthetic_state_dict(3, head="lunglike"))  # (a lung-like label volume: the slab that is copied back is what a mask's would be)
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
# the result arrays of the binding (what LMInferer.apply hands out): page-locked, zero-filled beside the forward, only the labelled slab copied back
from lungmask_amd.mask import LMInferer
inf = LMInferer(state_dict=syn.synthetic_state_dict(3, head="lunglike"), engine=eng)
r = inf.apply(vol); r = inf.apply(vol)
t = time.perf_counter()
for _ in range(5): r = inf.apply(vol)
lmi = (time.perf_counter() - t) / 5 * 1e3
pinned = inf._result_array(vol.shape)
t = time.perf_counter()
for _ in range(5): eng.apply(0, vol, out=pinned)  # the same page-locked block WITHOUT the scratch flag: the whole volume comes back
full = (time.perf_counter() - t) / 5 * 1e3
print(f"LMInferer.apply {lmi:.2f} ms (+{lmi - res:.2f} ms, {100 * (lmi / res - 1):.1f} %); page-locked block, whole-volume copy-back {full:.2f} ms   identical: {np.array_equal(r, out) and np.array_equal(pinned, out)}")
