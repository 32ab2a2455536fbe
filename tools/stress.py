"""Race screen: the same inputs through the whole hot path many times -- every output must be bit-identical to the first
(a half-landed LDS-DMA chunk, an unordered staging read or a lost union shows up as a differing byte once in 10^4-10^6
tiles).  argv[1]: repetitions of the 300-slice volume (default 300)."""
import sys, time, zlib
import numpy as np
sys.path.insert(0, ".")
from lungmask_amd import synthetic as sy, _native as nat
from lungmask_amd.pipeline import postprocess_slabs_in_process, shard_bounds

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
eng = nat.Engine(0)
eng.load_state_dict(0, sy.synthetic_state_dict(3, head="lunglike"))  # (2 553 regions, 341 merges: the post-processing's region graph has work to do)
vol = sy.phantom(300, 512, 512, seed=2024)
d = eng.to_device(vol); o = eng.empty(vol.shape, np.uint8)
eng.apply_dev(0, d, o); eng.sync()
ref = o.download(); ref_crc = zlib.crc32(ref.tobytes())
t0 = time.time(); bad = 0
for i in range(reps):
    eng.apply_dev(0, d, o); eng.sync()
    if zlib.crc32(o.download().tobytes()) != ref_crc:
        bad += 1; print("MISMATCH in apply repetition", i, int((o.download() != ref).sum()), flush=True)
print(f"apply: {reps} repetitions of 300 slices, {bad} differing outputs, {time.time() - t0:.1f} s", flush=True)
# the host path (three-piece copy-in, scratch result arrays: zero-fill beside the forward, labelled slab copied back)
from lungmask_amd.mask import LMInferer
inf = LMInferer(state_dict=sy.synthetic_state_dict(3, head="lunglike"), engine=eng)
t0 = time.time(); bad = 0; keep = None
for i in range(max(10, reps // 3)):
    r = inf.apply(vol)
    if zlib.crc32(r.tobytes()) != ref_crc:
        bad += 1; print("MISMATCH in LMInferer.apply repetition", i, int((r != ref).sum()), flush=True)
    if i % 7 == 0: keep = r  # (held results force other pool blocks)
print(f"LMInferer.apply: {max(10, reps // 3)} repetitions, {bad} differing outputs, {time.time() - t0:.1f} s", flush=True)
# the same volumes through the queue (apply_async: two host threads, copy-in / copy-back beside the hot path), results held and dropped at random
t0 = time.time(); bad = 0; pend = []; held = []
n_async = max(12, reps // 3)
for i in range(n_async + 1):
    if i < n_async:
        pend.append(inf.apply_async(vol))
    while len(pend) > (2 if i < n_async else 0):
        r = pend.pop(0).result()
        bad += zlib.crc32(r.tobytes()) != ref_crc
        if i % 5 == 0: held.append(r)
        if len(held) > 2: held.pop(0)
print(f"LMInferer.apply_async: {n_async} queued volumes, {bad} differing outputs, {time.time() - t0:.1f} s", flush=True)
del held
# labels of the network (before post-processing) with one and two lanes, small and odd batch sizes
xf = eng.preprocess(vol[:100])[1]
x = eng.to_device(xf); lab = eng.empty((100, 256, 256), np.uint8)
lib = eng.L.lib
def fwd(bs):
    eng.L.check(lib.lm_forward_batches_dev(eng.h, 0, x.ptr, 100, 256, 256, bs, lab.ptr)); eng.sync(); return zlib.crc32(lab.download().tobytes())
base = fwd(20); bad = 0; n = 0
for lanes in (2, 1):
    eng.set_streams(lanes)
    for bs in (20, 7, 33, 1 if lanes == 2 else 50):
        for _ in range(max(2, reps // 30)):
            n += 1; bad += fwd(bs) != base
eng.set_streams(2)
print(f"forward: {n} runs over lanes x batch sizes, {bad} differing outputs", flush=True)
# slab protocol, 4 in-process ranks
extra = [nat.Engine(0) for _ in range(3)]
labv = lab.download()
whole = eng.postprocess(labv)
bad = 0
for _ in range(max(3, reps // 20)):
    bad += not np.array_equal(postprocess_slabs_in_process([eng] + extra, labv, shard_bounds(100, 4)), whole)
print(f"slab protocol: {max(3, reps // 20)} runs, {bad} differing outputs", flush=True)
