"""(test-side tool: the oracle is the checker)  Distribution of the split-f16 log-prob error, not just its maximum, for several builds of
the library (argv: library paths; none = the product library) on three models x several inputs: the Appendix-D head (std 8), the
heavy-tailed weights of test_forward_heavy_tailed_weights, BatchNorm scales over 3 decades + heavy tails.  The maximum over 4e5 values
is an extreme-value statistic that moves by +-15 % from one rounding pattern to the next; mean / rms / p99.99 say whether an
arithmetic change moved the distribution."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ["LM_ACC_GUARD"] = "0"  # the kernels themselves
import numpy as np, torch
from lungmask_amd import _native as nat
from oracle import unet_oracle as uo, prepost_oracle as po
from test_forward_emu import wide_batchnorm_heavy_tail_state_dict
torch.set_num_threads(min(32, os.cpu_count() or 8))
libs = sys.argv[1:] or [None]


def heavy(sd):
    g = torch.Generator().manual_seed(5)
    sd = dict(sd)
    for k, v in sd.items():
        if k.endswith(".weight") and v.ndim == 4 and v.shape[-1] == 3 and v.shape[1] >= 64:
            sd[k] = torch.where(torch.rand(v.shape, generator=g) < 5e-4, v * 60.0, v)
    return sd


xs, _ = po.preprocess(po.phantom(2, 512, 512), [256, 256])
inputs = {"phantom": po.normalise(xs)}
for seed in (8, 12):
    inputs[f"rand{seed}"] = np.random.default_rng(seed).random((2, 256, 256), dtype=np.float32)
models = {"appendixD_std8": uo.synthetic_state_dict(3), "heavy_tailed": heavy(uo.synthetic_state_dict(3)), "wideBN_heavy": wide_batchnorm_heavy_tail_state_dict(3)}
engines = [(os.path.basename(p) if p else "product", nat.Engine(0, nat.Library(p)) if p else nat.Engine(0)) for p in libs]
for mname, base in models.items():
    for iname, x in inputs.items():
        xt = torch.from_numpy(x[:, None])
        sd = uo.calibrate_head(base, xt[:1], 8.0)
        with torch.inference_mode():
            ref = uo.forward(sd, xt).numpy()
            ref64 = uo.forward_f64(sd, xt).numpy()
        n32 = np.abs(ref - ref64)
        print(f"{mname} / {iname}: reference fp32 vs float64: max {n32.max():.2e} rms {np.sqrt((n32 ** 2).mean()):.2e}", flush=True)
        for name, e in engines:
            for prec in ("split_f16", "f32"):
                e.set_precision(prec)
                e.load_state_dict(0, sd)
                lab, logp = e.forward(0, x)
                d32, d64 = np.abs(logp - ref), np.abs(logp - ref64)
                print(f"   {name:22s} {prec:9s} vs ref32: max {d32.max():.2e} p99.99 {np.quantile(d32, 0.9999):.2e} rms {np.sqrt((d32 ** 2).mean()):.2e} mean {d32.mean():.2e} | "
                      f"vs f64: max {d64.max():.2e} rms {np.sqrt((d64.astype(np.float64) ** 2).mean()):.2e} | ran on {e.model_precision(0)}", flush=True)
            e.set_precision("split_f16")
