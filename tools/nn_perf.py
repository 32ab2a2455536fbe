"""Perf probes of the network forward (not the bench).  One script, three modes:

  nn_perf.py [layers] [B=20] [iters=5] [precision=split_f16]   per-layer HIP-event times of one batch (what gpu_round_check.sh records)
  nn_perf.py engines                                           1 / 2 / 3 engines (streams + workspaces) fed alternately: does inter-batch overlap pay?
  nn_perf.py batches                                           lm_forward_batches_dev over internal batch sizes, one and two lanes, 300 and 320 slices
                                                               (320: every level's work-item count is a multiple of the 256 CUs at batch 32)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from lungmask_amd import _native as nat
from lungmask_amd import synthetic as uo

args = sys.argv[1:]
mode = args.pop(0) if args and args[0] in ("layers", "engines", "batches") else "layers"
sd = uo.synthetic_state_dict(3)

if mode == "layers":
    B = int(args[0]) if len(args) > 0 else 20
    iters = int(args[1]) if len(args) > 1 else 5
    eng = nat.Engine(0)
    eng.load_state_dict(0, sd)
    eng.set_precision(args[2] if len(args) > 2 else "split_f16")
    x = eng.to_device(np.random.default_rng(0).random((B, 256, 256), dtype=np.float32))
    lab = eng.empty((B, 256, 256), np.uint8)
    for _ in range(2):
        eng.forward_dev(0, x, lab)
    eng.sync()
    t = time.time()
    for _ in range(iters):
        eng.forward_dev(0, x, lab)
    eng.sync()
    dt = (time.time() - t) / iters
    print(f"B={B}: {dt*1e3:.2f} ms/batch  {B/dt:.1f} slices/s  {B*96.2e9/dt/1e12:.1f} TFLOP/s")
    eng.profile(2)
    eng.profile_reset()
    for _ in range(iters):
        eng.forward_dev(0, x, lab)
    eng.sync()
    for s in eng.profile_read():
        ms = s["total_ms"] / max(s["launches"], 1)
        print(f"{s['name']:24s} n={s['launches']:4d} avg={ms:8.3f} ms  total={s['total_ms']:9.2f} ms  {s['flops']/max(s['total_ms'],1e-9)/1e9:8.1f} TFLOP/s  "
              f"{s['bytes']/max(s['total_ms'],1e-9)/1e6:8.1f} GB/s")
elif mode == "engines":
    B, iters = 20, 12
    engs = [nat.Engine(0) for _ in range(3)]
    xs, labs = [], []
    for e in engs:
        e.load_state_dict(0, sd)
        e.set_streams(1)
        xs.append(e.to_device(np.random.default_rng(0).random((B, 256, 256), dtype=np.float32)))
        labs.append(e.empty((B, 256, 256), np.uint8))
    for n_eng in (1, 2, 3, 1, 2, 3):
        for e, x, l in zip(engs, xs, labs):
            e.forward_dev(0, x, l)
        for e in engs:
            e.sync()
        t = time.time()
        for i in range(iters):
            engs[i % n_eng].forward_dev(0, xs[i % n_eng], labs[i % n_eng])
        for e in engs:
            e.sync()
        dt = (time.time() - t) / iters
        print(f"{n_eng} engine(s): {dt*1e3:.2f} ms/batch  {B/dt:.1f} slices/s")
else:
    eng = nat.Engine(0)
    eng.load_state_dict(0, sd)
    lib = eng.L.lib
    for n in (300, 320):
        x = eng.to_device(np.random.default_rng(0).random((n, 256, 256), dtype=np.float32))
        lab = eng.empty((n, 256, 256), np.uint8)
        for streams in (1, 2):
            eng.set_streams(streams)
            for bs in (20, 32, 40, 64, 100, 160, 20):
                def f():
                    eng.L.check(lib.lm_forward_batches_dev(eng.h, 0, x.ptr, n, 256, 256, bs, lab.ptr))
                f()
                eng.sync()
                t = time.perf_counter()
                for _ in range(3):
                    f()
                eng.sync()
                dt = (time.perf_counter() - t) / 3
                print(f"n {n} lanes {streams} batch {bs:4d}: {dt*1e3:7.2f} ms  {n/dt:8.1f} slices/s", flush=True)
        x.free()
        lab.free()
