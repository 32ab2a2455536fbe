"""(test-side tool: the oracle is the checker)  The logit-range sweep of test_logit_range_sweep as a table, for several builds of
the library (argv: library paths; none = the product library): heads calibrated per SURVEY Appendix D at std 8 / 30 / 100."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lungmask_amd import _native as nat
from oracle import unet_oracle as uo, prepost_oracle as po
torch.set_num_threads(16)
libs = sys.argv[1:] or [None]
base = uo.synthetic_state_dict(3)
xs, _ = po.preprocess(po.phantom(2, 512, 512), [256, 256])
x = po.normalise(xs); xt = torch.from_numpy(x[:, None])
for std in (8.0, 30.0, 100.0):
    sd = uo.calibrate_head(base, xt[:1], std)
    with torch.inference_mode():
        ref = uo.forward(sd, xt).numpy(); ref64 = uo.forward_f64(sd, xt).numpy()
    print(f"head std {std:g}: |ref32-ref64| {np.abs(ref - ref64).max():.2e}", flush=True)
    for path in libs:
        e = nat.Engine(0, nat.Library(path)) if path else nat.Engine(0)
        for prec in ("split_f16", "f32"):
            e.set_precision(prec); e.load_state_dict(0, sd)
            lab, logp = e.forward(0, x)
            print(f"   {os.path.basename(path) if path else 'product':24s} {prec:9s} |engine-ref32| {np.abs(logp - ref).max():.2e}  |engine-ref64| {np.abs(logp - ref64).max():.2e}  "
                  f"labels != ref argmax: {(lab != ref.argmax(1)).sum()}", flush=True)
        e.close()
