"""Where the wall time of a whole hot-path step goes that is NOT the network: timeline of two consecutive lm_apply_dev calls on the
bench volume (lm_profile_enable(e, 4): every profiled launch's span relative to the first), reduced to
  * the union of the forward launches' spans (two lanes), the gap between the last forward launch and the first post-processing launch,
  * the post-processing's wall time and how much of it no profiled kernel covers (read-backs, the host merge replay, launch gaps),
  * the un-crop, and the gap from one call's last launch to the next call's first.
(In this mode the whole volume is pre-processed in front of the forward -- engine.h: the profiling pass keeps everything on the main
stream -- so "pre" here is 0.33 ms that the timed region hides beside the head's forward.)  argv: [n_slices=300]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from lungmask_amd import _native as nat
from lungmask_amd import synthetic as po

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
eng = nat.Engine(0)
eng.load_state_dict(0, po.synthetic_state_dict(3, head="lunglike"))
vol = po.phantom(300, 512, 512, z0=150 - n // 2, z1=150 - n // 2 + n)
vd = eng.to_device(vol)
od = eng.empty(vol.shape, np.uint8)
for _ in range(3):
    eng.apply_dev(0, vd, od)
eng.sync()
eng.profile(4)
eng.profile_reset()
import time

t0 = time.perf_counter()
eng.apply_dev(0, vd, od)
eng.apply_dev(0, vd, od)
eng.sync()
wall2 = (time.perf_counter() - t0) * 1e3
tl = eng.profile_timeline(8192)
eng.profile(0)
# split the launches into the two calls at the second bodymask launch
starts = [i for i, t in enumerate(tl) if t[0] == "bodymask_bbox"]
calls = [tl[starts[0]:starts[1]], tl[starts[1]:]] if len(starts) >= 2 else [tl]


def union(spans):
    tot, cur_a, cur_b = 0.0, None, None
    for a, b in sorted(spans):
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                tot += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    return tot + (cur_b - cur_a if cur_b is not None else 0.0)


print(f"two consecutive lm_apply_dev calls, {n} slices: host wall {wall2:.2f} ms ({wall2 / 2:.2f} per call; events around every launch cost ~1 %)")
prev_end = None
for ci, c in enumerate(calls):
    pre = [t for t in c if t[0] in ("bodymask_bbox", "resample_norm")]
    fwd = [t for t in c if t[0].startswith("conv") or t[0] in ("upsample2x", "first_conv", "head_argmax")]
    post = [t for t in c if t[0].startswith("post_")]
    rs = [t for t in c if t[0] == "reshape_mask"]
    a0 = min(t[2] for t in c)
    b1 = max(t[3] for t in c)
    f_a, f_b = min(t[2] for t in fwd), max(t[3] for t in fwd)
    p_a, p_b = min(t[2] for t in post), max(t[3] for t in post)
    print(f"call {ci}: first launch -> last launch end {b1 - a0:.3f} ms" + (f"; gap from the previous call's last launch {a0 - prev_end:.3f} ms" if prev_end is not None else ""))
    print(f"  pre-processing {max(t[3] for t in pre) - min(t[2] for t in pre):.3f} ms (kernels {sum(t[3] - t[2] for t in pre):.3f})")
    print(f"  forward: first -> last launch end {f_b - f_a:.3f} ms, covered by launches {union([(t[2], t[3]) for t in fwd]):.3f} ms")
    print(f"  forward end -> first post-processing launch {p_a - f_b:.3f} ms")
    print(f"  post-processing: {p_b - p_a:.3f} ms of wall, {union([(t[2], t[3]) for t in post]):.3f} ms covered by its profiled kernels "
          f"({len(post)} launches; the rest: read-backs, host merge replay, unprofiled helpers, launch gaps)")
    # the largest uncovered gaps inside the post-processing
    sp = sorted((t[2], t[3], t[0]) for t in post)
    gaps = sorted(((sp[i + 1][0] - max(s[1] for s in sp[:i + 1]), sp[i][2], sp[i + 1][2]) for i in range(len(sp) - 1)), reverse=True)[:4]
    print("  largest gaps inside it: " + "; ".join(f"{g:.3f} ms after {a} before {b}" for g, a, b in gaps))
    if rs:
        print(f"  post end -> un-crop start {rs[0][2] - p_b:.3f} ms; un-crop {rs[0][3] - rs[0][2]:.3f} ms")
    prev_end = b1
