"""Two-lane timeline of the network forward (lm_profile_enable(e, 4)): per layer shape, the launch durations with one lane and with
two lanes (where the other lane's kernels run beside it), and how much of the two-lane wall time each lane's launches cover.
Evidence for "the second lane fills the 320 / 160-item levels" (DESIGN.md 3.5).  argv: [n_slices=120]"""
import sys, os, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lungmask_amd import _native as nat
from lungmask_amd import synthetic as uo

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
eng = nat.Engine(0)
eng.load_state_dict(0, uo.synthetic_state_dict(3))
x = eng.to_device(np.random.default_rng(0).random((n, 256, 256), dtype=np.float32)); lab = eng.empty((n, 256, 256), np.uint8)
f = lambda: eng.L.check(eng.L.lib.lm_forward_batches_dev(eng.h, 0, x.ptr, n, 256, 256, 20, lab.ptr))
res = {}
for lanes in (1, 2):
    eng.set_streams(lanes)
    f(); eng.sync()
    eng.profile(4); eng.profile_reset()
    f(); eng.sync()
    tl = eng.profile_timeline(); eng.profile(0)
    wall = max(t[3] for t in tl) - min(t[2] for t in tl)
    per = collections.defaultdict(list)
    for name, lane, a, b in tl: per[name].append(b - a)
    res[lanes] = (wall, per, tl)
    print(f"{lanes} lane(s): {len(tl)} launches, wall {wall:.2f} ms for {n} slices ({wall / n * 300:.2f} ms per 300)")
w1, p1, _ = res[1]; w2, p2, tl2 = res[2]
print(f"{'layer':42s} {'1 lane ms':>10s} {'2 lanes ms':>10s} {'ratio':>6s}  (a launch's span with two lanes includes the time it shares the CUs)")
for name in p1:
    a, b = np.mean(p1[name]), np.mean(p2[name])
    print(f"{name:42s} {a:10.3f} {b:10.3f} {b / a:6.2f}")
s1 = sum(sum(v) for v in p1.values()); s2 = sum(sum(v) for v in p2.values())
print(f"sum of launch spans: 1 lane {s1:.2f} ms (wall {w1:.2f}), 2 lanes {s2:.2f} ms (wall {w2:.2f}): mean concurrency {s2 / w2:.2f}")
# how much of the two-lane wall time has 0 / 1 / 2 launches in flight
ev = sorted([(a, 1) for _, _, a, b in tl2] + [(b, -1) for _, _, a, b in tl2])
cur, last, hist = 0, ev[0][0], collections.Counter()
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
print("two lanes, share of wall time with k launches in flight:", {k: round(v / w2, 3) for k, v in sorted(hist.items())})
