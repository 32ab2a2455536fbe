"""How much of the step do the bandwidth-bound kernels of the network (first conv, the four 1x1 convs, the four bilinear
upsamples) cost INSIDE the two-lane pipeline, where they overlap with the other lane's matrix kernels?  Upper bound for what
fusing them into their neighbours could gain.  Needs a library built with -DLM_LAB_HOOKS (argv[1]): LM_ABL_SKIP is a bit mask of
kernels that are simply not launched (1 first conv, 2 1x1 convs, 4 upsamples).  The buffers keep the values of the last complete
forward of the same input, so the matrix kernels see the same operands (their power, and so their clock, depends on the data)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lungmask_amd import _native as nat
from lungmask_amd import synthetic as sy
eng = nat.Engine(0, nat.Library(sys.argv[1]))
eng.load_state_dict(0, sy.synthetic_state_dict(3))
vol = sy.phantom(300, 512, 512, seed=2024)
xf = eng.preprocess(vol)[1]
x = eng.to_device(xf); lab = eng.empty((300, 256, 256), np.uint8)
f = lambda: eng.L.check(eng.L.lib.lm_forward_batches_dev(eng.h, 0, x.ptr, 300, 256, 256, 20, lab.ptr))
def timed(n=4):
    f(); eng.sync(); t = time.perf_counter()
    for _ in range(n): f()
    eng.sync(); return (time.perf_counter() - t) / n * 1e3
for lanes in (2, 1):
    eng.set_streams(lanes)
    for rep in range(2):
        row = []
        for mask in (0, 4, 6, 0):  # 1 and 2 alone would feed stale buffers of OTHER layers to their consumers (the workspaces are reused)
            os.environ["LM_ABL_SKIP"] = str(mask)
            if mask == 0: f(); eng.sync()  # refresh the buffers with a complete forward
            row.append((mask, timed()))
        print(f"{lanes} lane(s): " + "  ".join(f"skip={m}: {t:6.2f} ms" for m, t in row), flush=True)
