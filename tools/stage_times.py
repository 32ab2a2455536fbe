"""Scratch: wall time per stage of the hot path with a sync after each (300-slice phantom)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
from lungmask_amd import _native as nat
from lungmask_amd import synthetic as uo; po = uo
eng = nat.Engine(0); eng.load_state_dict(0, uo.synthetic_state_dict(3))
n = 300
vol = po.phantom(n, 512, 512)
vd = eng.to_device(vol); bb = eng.empty((n, 4), np.int32); xf = eng.empty((n, 256, 256), np.float32)
lab = eng.empty((n, 256, 256), np.uint8); out = eng.empty((n, 512, 512), np.uint8)
lib = eng.L.lib
def T(f, reps=3):
    f(); eng.sync(); t = time.perf_counter()
    for _ in range(reps): f()
    eng.sync(); return (time.perf_counter() - t) / reps * 1e3
print("preprocess   %.2f ms" % T(lambda: eng.preprocess_dev(vd, bb, xf)))
print("forward x15  %.2f ms" % T(lambda: eng.L.check(lib.lm_forward_batches_dev(eng.h, 0, xf.ptr, n, 256, 256, 20, lab.ptr))))
keep = lab.download()
def post():
    lab.upload(keep); eng.postprocess_dev(lab)
def up(): lab.upload(keep)
print("upload only  %.2f ms" % T(up))
print("post+upload  %.2f ms" % T(post), eng.postprocess_info())
print("reshape      %.2f ms" % T(lambda: eng.reshape_mask_dev(lab, bb, out)))
print("apply_dev    %.2f ms" % T(lambda: eng.apply_dev(0, vd, out)))
print("apply_dev no post %.2f ms" % T(lambda: eng.apply_dev(0, vd, out, volume_postprocessing=False)))
