"""Where the wall time of the 3-D post-processing goes (LM_POST_TIMING=1 prints the C side's timestamps): the lung-like label volume of
the bench workload, post-processed five times."""
import sys, os
os.environ["LM_POST_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lungmask_amd import _native as nat
from lungmask_amd import synthetic as uo
eng = nat.Engine(0)
eng.load_state_dict(0, uo.synthetic_state_dict(3, head=sys.argv[1] if len(sys.argv) > 1 else "lunglike"))
vol = uo.phantom(300, 512, 512)
out = eng.apply(0, vol, volume_postprocessing=False)  # labels of the network, un-cropped: only to report the histogram
print("un-post-processed label histogram:", np.bincount(out.ravel()).tolist())
for _ in range(5):
    eng.apply(0, vol)
print("post-processing info:", eng.postprocess_info())
