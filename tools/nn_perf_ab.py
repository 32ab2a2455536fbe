"""Per-layer A/B of several builds of the library: HIP-event time per conv shape of the network forward (one lane, batch 20,
the network's own activations), side by side.  argv = library paths (first = baseline)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lungmask_amd import _native as nat
from lungmask_amd import synthetic as uo

sd = uo.synthetic_state_dict(3)
x_h = np.random.default_rng(0).random((20, 256, 256), dtype=np.float32)
rows, order = {}, []
for rep in range(2):
    for li, path in enumerate(sys.argv[1:]):
        eng = nat.Engine(0, nat.Library(path)); eng.load_state_dict(0, sd); eng.set_streams(1)
        x = eng.to_device(x_h); lab = eng.empty((20, 256, 256), np.uint8)
        for _ in range(3): eng.forward_dev(0, x, lab)
        eng.sync(); eng.profile(2); eng.profile_reset()
        for _ in range(10): eng.forward_dev(0, x, lab)
        eng.sync()
        for s in eng.profile_read():
            if s["name"] not in rows: rows[s["name"]] = {}; order.append(s["name"])
            rows[s["name"]].setdefault(li, []).append(s["total_ms"] / 10)
        eng.close()
n = len(sys.argv) - 1
print(f"{'ms per batch of 20':36s}" + "".join(f"{os.path.basename(p)[-22:]:>24s}" for p in sys.argv[1:]))
tot = [0.0] * n
for k in order:
    v = [min(rows[k].get(i, [0.0])) for i in range(n)]
    tot = [a + b for a, b in zip(tot, v)]
    print(f"{k:36s}" + "".join(f"{t:16.4f} {100 * (t / v[0] - 1) if v[0] else 0:+6.1f}%" for t in v))
print(f"{'sum':36s}" + "".join(f"{t:16.4f} {100 * (t / tot[0] - 1):+6.1f}%" for t in tot))
