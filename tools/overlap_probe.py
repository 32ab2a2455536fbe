"""What cross-volume overlap could buy: engine A loops the network forward of 300 slices (two lanes), engine B -- a second handle on the
same GPU, its own thread -- loops pre-processing + 3-D post-processing + un-crop of another volume.  Forward time alone vs beside B, B's
time alone vs beside A."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lungmask_amd import _native as nat
from lungmask_amd import synthetic as uo

A = nat.Engine(0); B = nat.Engine(0)
sd = uo.synthetic_state_dict(3)
A.load_state_dict(0, sd); B.load_state_dict(0, sd)
vol = uo.phantom(300, 512, 512)
# a realistic label volume for B: A's own result of the hot path without volume post-processing
vd = A.to_device(vol); od = A.empty(vol.shape, np.uint8)
x = A.to_device(np.random.default_rng(0).random((300, 256, 256), dtype=np.float32)); lab = A.empty((300, 256, 256), np.uint8)
fA = lambda: A.L.check(A.L.lib.lm_forward_batches_dev(A.h, 0, x.ptr, 300, 256, 256, 20, lab.ptr))
fA(); A.sync()
workB = B.to_device(lab.download())
def fB():  # (re-labels its own output after the first call: same voxel passes, fewer regions to replay)
    B.postprocess_dev(workB); B.sync()
def timeit(f, n):
    f(); t = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t) / n * 1e3
def fAs():
    fA(); A.sync()
tA = timeit(fAs, 6); tB = timeit(fB, 20)
print(f"alone: forward {tA:.2f} ms, post-processing {tB:.2f} ms")
stop = [False]; cnt = [0]; tb = [0.0]
def worker():
    while not stop[0]:
        t = time.perf_counter(); fB(); tb[0] += time.perf_counter() - t; cnt[0] += 1
th = threading.Thread(target=worker); th.start()
tA2 = timeit(fAs, 10)
stop[0] = True; th.join()
print(f"together: forward {tA2:.2f} ms (+{tA2 - tA:.2f}), post-processing {tb[0] / max(cnt[0], 1) * 1e3:.2f} ms per call, {cnt[0]} calls beside 11 forwards "
      f"= {cnt[0] / 11:.2f} per forward")
print("=> cost of ONE post-processing per forward when overlapped: +%.2f ms (sequential today: +%.2f ms)" % ((tA2 - tA) / max(cnt[0] / 11, 1e-9), tB))
