"""What do the per-launch HIP events of lm_profile_enable cost in the timed region of bench.py?"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from lungmask_amd import synthetic as sy, _native as nat
eng = nat.Engine(0); eng.load_state_dict(0, sy.synthetic_state_dict(3))
vol = sy.phantom(300, 512, 512, seed=2024)
d = eng.to_device(vol); o = eng.empty(vol.shape, np.uint8)
def T(n=4):
    eng.apply_dev(0, d, o); eng.sync(); t0 = time.perf_counter()
    for _ in range(n): eng.apply_dev(0, d, o)
    eng.sync(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(2):
    eng.profile(False); a = T()
    eng.profile(True); eng.profile_reset(); b = T(); eng.profile_read()
    print(f"profile off {a:.2f} ms   on {b:.2f} ms", flush=True)
