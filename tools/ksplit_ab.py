"""The split-K ("precise") tier of the split-f16 path at several chain limits: LM_H3_KSPLIT_K is read once per process, so this script
runs itself once per limit (argv: limits, 0 = the single-chain form) and prints, per limit, the log-prob error against the oracle on
the Appendix-D model and the heavy-tailed one (max / rms, vs the reference's fp32 result and vs float64) and the time of the
300-slice forward with two lanes and with one."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    for k in sys.argv[1:]:
        r = subprocess.run([sys.executable, __file__, "--child"], env=dict(os.environ, LM_H3_KSPLIT_K=k, LM_ACC_GUARD="0"), capture_output=True, text=True)
        print(f"--- LM_H3_KSPLIT_K={k}\n" + "\n".join(l for l in (r.stdout + r.stderr).splitlines() if "amdgpu.ids" not in l), flush=True)
    sys.exit(0)
import numpy as np, torch
from lungmask_amd import _native as nat, synthetic
from oracle import unet_oracle as uo
torch.set_num_threads(min(32, os.cpu_count() or 8))
e = nat.Engine(0)
base = uo.synthetic_state_dict(3)
g = torch.Generator().manual_seed(5)
heavy = dict(base)
for k, v in heavy.items():
    if k.endswith(".weight") and v.ndim == 4 and v.shape[-1] == 3 and v.shape[1] >= 64:
        heavy[k] = torch.where(torch.rand(v.shape, generator=g) < 5e-4, v * 60.0, v)
x = np.random.default_rng(8).random((2, 256, 256), dtype=np.float32)
xt = torch.from_numpy(x[:, None])
for name, sd0 in (("appendixD_std8", base), ("heavy_tailed", heavy)):
    sd = uo.calibrate_head(sd0, xt[:1], 8.0)
    with torch.inference_mode():
        ref = uo.forward(sd, xt).numpy(); ref64 = uo.forward_f64(sd, xt).numpy()
    e.load_state_dict(0, sd)
    lab, logp = e.forward(0, x)
    d32, d64 = np.abs(logp - ref), np.abs(logp - ref64)
    print(f"{name:16s} vs ref32: max {d32.max():.2e} rms {np.sqrt((d32.astype(np.float64)**2).mean()):.2e} | vs f64: max {d64.max():.2e} rms {np.sqrt((d64.astype(np.float64)**2).mean()):.2e} | "
          f"labels != ref argmax: {int((lab != ref.argmax(1)).sum())}", flush=True)
e.load_state_dict(0, synthetic.synthetic_state_dict(3, head="lunglike"))
xs = np.random.default_rng(1).random((300, 256, 256), dtype=np.float32)
xd, ld = e.to_device(xs), e.empty(xs.shape, np.uint8)
for lanes in (2, 1):
    e.set_streams(lanes)
    for _ in range(2):
        e.L.check(e.L.lib.lm_forward_batches_dev(e.h, 0, xd.ptr, 300, 256, 256, 20, ld.ptr)); e.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        e.L.check(e.L.lib.lm_forward_batches_dev(e.h, 0, xd.ptr, 300, 256, 256, 20, ld.ptr))
    e.sync()
    print(f"forward of 300 slices, {lanes} lane(s): {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms", flush=True)
