"""Per-layer times of the decoder's 1x1 + upsample stage (needs a -DLM_LAB_HOOKS library as argv[1]): the fused kernel, the fused
kernel without its main loop / without its output phase (LM_LAB_UPVAR=1/2, wrong results, timing only), single forward lane."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lungmask_amd import _native as nat
from lungmask_amd import synthetic as sy
eng = nat.Engine(0, nat.Library(sys.argv[1]))
eng.load_state_dict(0, sy.synthetic_state_dict(3))
x = eng.to_device(np.random.default_rng(0).random((20, 256, 256), dtype=np.float32))
lab = eng.empty((20, 256, 256), np.uint8)
for var in ("0", "1", "2"):
    os.environ["LM_LAB_UPVAR"] = var
    for _ in range(2): eng.forward_dev(0, x, lab)
    eng.sync(); eng.profile(2); eng.profile_reset()
    for _ in range(5): eng.forward_dev(0, x, lab)
    eng.sync()
    print("variant", var)
    for s in eng.profile_read():
        if "1x1" in s["name"] or "upsample" in s["name"]:
            print(f"  {s['name']:34s} avg={s['total_ms'] / max(s['launches'], 1):7.3f} ms  {s['bytes'] / max(s['total_ms'], 1e-9) / 1e6:8.1f} GB/s")
    eng.profile(0)
