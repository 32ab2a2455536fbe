"""Cost of the slab-sharded post-processing per rank vs the whole-volume path (one GPU, in-process ranks run one
after the other; the exchanges are device copies here).  Per-step wall time of rank 0 is printed."""
import sys, time, ctypes as C
import numpy as np
sys.path.insert(0, ".")
from lungmask_amd import synthetic as uo; po = uo
from lungmask_amd import _native as nat
from lungmask_amd.pipeline import shard_bounds

HEAD = sys.argv[1] if len(sys.argv) > 1 else "lunglike"  # (round 4 measured on the random head's 899-region volume: VERDICT r04 #10)
eng = nat.Engine(0)
eng.load_state_dict(0, uo.synthetic_state_dict(3, head=HEAD))


def network_labels(n_total):
    """argmax labels (256 x 256) of the n_total-slice bench phantom -- for world w the ONE volume of 300 w slices that config 5 shards
    (its lungs span the ranks' slabs), not w copies of the 300-slice one."""
    labs = []
    for i in range(0, n_total, 100):
        vol = po.phantom(n_total, 512, 512, seed=2024, z0=i, z1=min(i + 100, n_total))
        xf = eng.preprocess(vol)[1]
        labs.append(eng.forward(0, xf[:, None], want_logp=False)[0])
    return np.concatenate(labs)


extra = [nat.Engine(0) for _ in range(7)]
for world in (1, 2, 4, 8):
    lab = network_labels(300 * world)  # weak scaling: 300 slices per rank
    N = lab.shape[0]
    d = eng.to_device(lab)
    for _ in range(2):
        d.upload(lab); eng.sync(); t0 = time.perf_counter(); eng.postprocess_dev(d); eng.sync(); t1 = time.perf_counter()
    whole = d.download(); d.free()
    print(f"world {world}: whole-volume ({N} slices) {(t1 - t0) * 1e3:.1f} ms  {eng.postprocess_info()}")
    if world == 1:
        continue
    engines = [eng] + extra[: world - 1]
    b = shard_bounds(N, world)
    for rep in range(2):
        slabs = [e.to_device(lab[b[r]:b[r + 1]]) for r, e in enumerate(engines)]
        tb = []
        for r, e in enumerate(engines):
            t0 = time.perf_counter()
            e.L.check(e.L.lib.lm_slab_begin(e.h, slabs[r].ptr, b[r + 1] - b[r], 256, 256, r, world, b[r], N, None, 0, 3)); e.sync()
            tb.append(time.perf_counter() - t0)
        steps = [("begin", tb[0])]
        while True:
            lens = [int(e.L.lib.lm_slab_pending(e.h)) for e in engines]
            stride = max(lens)
            g = eng.empty((world * max(stride, 1),), np.int32)
            t0 = time.perf_counter()
            for r, e in enumerate(engines):
                e.L.check(e.L.lib.lm_slab_emit(e.h, g.ptr + 4 * r * stride))
            t_emit = (time.perf_counter() - t0) / world
            ts, st = [], []
            for e in engines:
                t0 = time.perf_counter()
                st.append(e.L.check(e.L.lib.lm_slab_step(e.h, g.ptr, stride, (C.c_int64 * world)(*lens)))); e.sync()
                ts.append(time.perf_counter() - t0)
            g.free()
            steps.append((f"emit {sum(lens) * 4 / 1e6:.1f}MB", t_emit)); steps.append(("step", max(ts)))
            if st[0] == 1:
                break
        out = np.concatenate([s.download() for s in slabs])
        for s in slabs: s.free()
    tot = sum(t for _, t in steps)
    print(f"   slab path per rank: {tot * 1e3:.1f} ms = " + " | ".join(f"{n} {t * 1e3:.2f}" for n, t in steps) + f"   equal={np.array_equal(out, whole)}")
