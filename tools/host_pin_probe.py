"""Probe for the host boundary of lm_apply_host: how fast does a 78.6 MB label volume come back into a caller-owned pageable
numpy array -- one pageable copy, the same split over 2-4 threads/streams, or through hipHostRegister (cost of registering and
unregistering included)?  Uses the HIP runtime directly (ctypes); no engine involved."""
import ctypes as C, time, threading, sys
import numpy as np
hip = C.CDLL("libamdhip64.so")
def chk(r, what=""):
    if r != 0: raise RuntimeError(f"hip error {r} {what}")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
D2H, H2D = 2, 1
n = 300 * 512 * 512
d = C.c_void_p(); chk(hip.hipMalloc(C.byref(d), n))
src = np.random.randint(0, 3, n, dtype=np.uint8)
chk(hip.hipMemcpy(d, src.ctypes.data, n, H2D))
streams = []
for _ in range(4):
    s = C.c_void_p(); chk(hip.hipStreamCreateWithFlags(C.byref(s), 1)); streams.append(s)
def ms(t): return (time.perf_counter() - t) * 1e3
def one(out):
    t = time.perf_counter(); chk(hip.hipMemcpy(out.ctypes.data, d, n, D2H)); return ms(t)
def split(out, k):
    step = (n // k + 4095) // 4096 * 4096
    def work(i):
        lo = i * step; hi = min(n, lo + step)
        chk(hip.hipMemcpyAsync(out.ctypes.data + lo, d.value + lo, hi - lo, D2H, streams[i])); chk(hip.hipStreamSynchronize(streams[i]))
    t = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(1, k)]
    for x in th: x.start()
    work(0)
    for x in th: x.join()
    return ms(t)
def registered(out):
    t = time.perf_counter(); chk(hip.hipHostRegister(out.ctypes.data, n, 0), "register"); a = ms(t)
    t = time.perf_counter(); chk(hip.hipMemcpy(out.ctypes.data, d, n, D2H)); b = ms(t)
    t = time.perf_counter(); chk(hip.hipHostUnregister(out.ctypes.data)); c = ms(t)
    return a, b, c
for fresh in (True, False):
    print("output array:", "freshly allocated, pages touched" if fresh else "reused")
    out = np.empty(n, np.uint8); out[::4096] = 0
    for rep in range(3):
        if fresh and rep: out = np.empty(n, np.uint8); out[::4096] = 0
        r = [one(out)]; assert np.array_equal(out, src)
        for k in (2, 3, 4):
            out[:] = 0; r.append(split(out, k)); assert np.array_equal(out, src)
        out[:] = 0; reg = registered(out); assert np.array_equal(out, src)
        print(f"  pageable 1 thread {r[0]:.2f} ms ({n / r[0] / 1e6:.1f} GB/s) | 2/3/4 threads {r[1]:.2f} {r[2]:.2f} {r[3]:.2f} ms | "
              f"register {reg[0]:.2f} + copy {reg[1]:.2f} ({n / reg[1] / 1e6:.1f} GB/s) + unregister {reg[2]:.2f} ms", flush=True)
# the same for the input side: 157 MB of int16
vin = np.random.randint(-1000, 1000, n, dtype=np.int16); dv = C.c_void_p(); chk(hip.hipMalloc(C.byref(dv), 2 * n))
for rep in range(2):
    t = time.perf_counter(); chk(hip.hipMemcpy(dv, vin.ctypes.data, 2 * n, H2D)); a = ms(t)
    t = time.perf_counter(); chk(hip.hipHostRegister(vin.ctypes.data, 2 * n, 0)); b = ms(t)
    t = time.perf_counter(); chk(hip.hipMemcpy(dv, vin.ctypes.data, 2 * n, H2D)); c = ms(t)
    t = time.perf_counter(); chk(hip.hipHostUnregister(vin.ctypes.data)); e = ms(t)
    print(f"input 157 MB: pageable H2D {a:.2f} ms ({2 * n / a / 1e6:.1f} GB/s) | register {b:.2f} + H2D {c:.2f} ({2 * n / c / 1e6:.1f} GB/s) + unregister {e:.2f} ms")
