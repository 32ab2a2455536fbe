"""Where the per-volume time of LMInferer.apply_async goes: the hot path on a resident volume (lm_apply_dev) against lm_pipe_apply alone,
with the copy-back enqueued behind it, with the next volume's copy-in on a second thread, and the binding itself."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lungmask_amd import _native as nat, synthetic
from lungmask_amd.mask import LMInferer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
e = nat.Engine(0)
sd = synthetic.synthetic_state_dict(3, head="lunglike")
e.load_state_dict(0, sd)
vol = synthetic.phantom(300, 512, 512)
vd, od = e.to_device(vol), e.empty(vol.shape, np.uint8)


def timed(fn, n=N, warm=2):
    for _ in range(warm):
        fn()
    e.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    e.sync()
    return (time.perf_counter() - t0) / n * 1e3


print(f"lm_apply_dev (resident volume)                  {timed(lambda: e.apply_dev(0, vd, od)):.3f} ms", flush=True)
e.pipe_upload(0, vol); e.pipe_upload(1, vol)
k = [0]
def a():
    e.pipe_apply(k[0], 0, vol.shape, vol.dtype); k[0] ^= 1
print(f"lm_pipe_apply alone (buffers 0 / 1 alternating)   {timed(a):.3f} ms", flush=True)
res = [np.empty(vol.shape, np.uint8), np.empty(vol.shape, np.uint8)]
pin = [e.host_alloc(vol.size), e.host_alloc(vol.size)]
import ctypes
pres = [np.ndarray(vol.shape, np.uint8, buffer=(ctypes.c_uint8 * vol.size).from_address(p)) for p in pin]
def b():
    e.pipe_apply(k[0], 0, vol.shape, vol.dtype); e.pipe_download(k[0], pres[k[0]]); k[0] ^= 1
print(f"  + copy-back enqueued (page-locked result)        {timed(b):.3f} ms", flush=True)
stop = threading.Event()
def uploader():
    kk = 0
    while not stop.is_set():
        # (protocol violated on purpose: uploads race the hot path's reads -- timing only)
        e.L.lib.lm_pipe_upload(e.h, 2 - 2 + kk, vol.ctypes.data, vol.nbytes); kk ^= 1
        time.sleep(0.03)
th = threading.Thread(target=uploader, daemon=True); th.start()
print(f"  + copy-in of a pageable volume on a second thread  {timed(b):.3f} ms", flush=True)
stop.set(); th.join()
inf = LMInferer(state_dict=sd, engine=e)
box = [None]
def c():
    box[0] = inf.apply(vol)
print(f"LMInferer.apply                                   {timed(c):.3f} ms", flush=True)
def pipelined(n):
    pend = []
    for _ in range(n):
        pend.append(inf.apply_async(vol))
        if len(pend) > 1:
            box[0] = pend.pop(0).result()
    while pend:
        box[0] = pend.pop(0).result()
pipelined(3)
t0 = time.perf_counter(); pipelined(N); dt = (time.perf_counter() - t0) / N * 1e3
print(f"LMInferer.apply_async, two in flight              {dt:.3f} ms", flush=True)
def pipelined3(n):
    pend = []
    for _ in range(n):
        pend.append(inf.apply_async(vol))
        if len(pend) > 2:
            box[0] = pend.pop(0).result()
    while pend:
        box[0] = pend.pop(0).result()
t0 = time.perf_counter(); pipelined3(N); dt = (time.perf_counter() - t0) / N * 1e3
print(f"LMInferer.apply_async, three submitted ahead      {dt:.3f} ms", flush=True)
inf.close()
