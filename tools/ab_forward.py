"""A/B of two builds of the library on the network forward (300 slices, batch 20): argv = library paths."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lungmask_amd import _native as nat
from lungmask_amd import synthetic as uo
sd = uo.synthetic_state_dict(3)
x_h = np.random.default_rng(0).random((300, 256, 256), dtype=np.float32)
outs = []
for rep in range(2):
    for path in sys.argv[1:]:
        eng = nat.Engine(0, nat.Library(path)); eng.load_state_dict(0, sd)
        x = eng.to_device(x_h); lab = eng.empty((300, 256, 256), np.uint8)
        f = lambda: eng.L.check(eng.L.lib.lm_forward_batches_dev(eng.h, 0, x.ptr, 300, 256, 256, 20, lab.ptr))
        res = []
        for lanes in (2, 1):
            eng.set_streams(lanes)
            f(); eng.sync(); t = time.perf_counter()
            for _ in range(4): f()
            eng.sync(); res.append((time.perf_counter() - t) / 4 * 1e3)
        outs.append(lab.download())
        print(f"{os.path.basename(path):32s} two lanes {res[0]:6.2f} ms   one lane {res[1]:6.2f} ms", flush=True)
        eng.close()
print("labels identical across builds:", all(np.array_equal(outs[0], o) for o in outs))
