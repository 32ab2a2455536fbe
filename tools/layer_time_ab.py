"""Per-layer HIP-event time of ONE layer shape across builds (one lane, batch 20): argv = substring of the layer name, then library paths."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lungmask_amd import _native as nat
from lungmask_amd import synthetic as uo
key, libs = sys.argv[1], sys.argv[2:]
sd = uo.synthetic_state_dict(3)
x_h = np.random.default_rng(0).random((20, 256, 256), dtype=np.float32)
for rep in range(2):
    for path in libs:
        eng = nat.Engine(0, nat.Library(path)); eng.load_state_dict(0, sd); eng.set_streams(1)
        xb = eng.to_device(x_h); lb = eng.empty((20, 256, 256), np.uint8)
        eng.forward_dev(0, xb, lb); eng.sync()
        eng.profile(2); eng.profile_reset()
        for _ in range(10): eng.forward_dev(0, xb, lb)
        eng.sync()
        st = eng.profile_read()
        tot = sum(s['total_ms'] for s in st) / 10
        sel = [s for s in st if key in s['name']]
        print(f"{os.path.basename(path):28s} " + "  ".join(f"{s['name']} n={s['launches']} avg={s['total_ms']/s['launches']:.4f} ms" for s in sel) + f"   sum of kernels {tot:.3f} ms/batch", flush=True)
        eng.close()
