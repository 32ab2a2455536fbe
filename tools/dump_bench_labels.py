import sys, numpy as np
sys.path.insert(0, ".")
from lungmask_amd import synthetic as sy, _native as nat
eng = nat.Engine(0)
eng.load_state_dict(0, sy.synthetic_state_dict(3))
vol = sy.phantom(300, 512, 512, seed=2024)
xf = eng.preprocess(vol)[1]
x = eng.to_device(xf); lab = eng.empty((300, 256, 256), np.uint8)
eng.L.check(eng.L.lib.lm_forward_batches_dev(eng.h, 0, x.ptr, 300, 256, 256, 20, lab.ptr)); eng.sync()
l = lab.download()
np.savez_compressed("gpurun_out/bench_labels_c3.npz", lab=l)
print(l.shape, np.bincount(l.ravel()))
