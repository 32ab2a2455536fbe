"""What the chip does while the two-lane forward runs back to back: rocm-smi clock / power samples from the main thread while a worker
thread loops the 300-slice forward (ctypes releases the GIL).  argv: [seconds=6] [lanes=2]"""
import sys, os, time, threading, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lungmask_amd import _native as nat
from lungmask_amd import synthetic as uo

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
eng = nat.Engine(0)
eng.load_state_dict(0, uo.synthetic_state_dict(3))
eng.set_streams(lanes)
x = eng.to_device(np.random.default_rng(0).random((300, 256, 256), dtype=np.float32)); lab = eng.empty((300, 256, 256), np.uint8)
f = lambda: eng.L.check(eng.L.lib.lm_forward_batches_dev(eng.h, 0, x.ptr, 300, 256, 256, 20, lab.ptr))
f(); eng.sync()
stop, count = [False], [0]
def worker():
    while not stop[0]:
        f(); eng.sync(); count[0] += 1
def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out); c = d[sorted(d)[0]]
        return {k: v for k, v in c.items() if any(s in k.lower() for s in ("sclk", "mclk", "power", "junction", "hotspot"))}
    except Exception as ex:
        return {"error": str(ex)}
print("idle:", smi(), flush=True)
th = threading.Thread(target=worker); t0 = time.time(); th.start()
while time.time() - t0 < secs:
    time.sleep(0.5)
    print(f"t={time.time() - t0:4.1f}s", smi(), flush=True)
stop[0] = True; th.join()
dt = time.time() - t0
print(f"{lanes} lane(s): {count[0]} forwards in {dt:.2f} s = {dt / max(count[0], 1) * 1e3:.2f} ms per 300 slices")
