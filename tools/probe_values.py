"""What the load-time accuracy guard (lm_model_probe_error) measures for the models of the test-suite and the bench:
max |delta log-prob| split-f16 vs exact fp32 on the engine's probe slice, and whether the model was pinned (limit: LM_ACC_GUARD, default 5e-4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from lungmask_amd import _native as nat, synthetic
from oracle import unet_oracle as uo
from test_forward_emu import wide_batchnorm_heavy_tail_state_dict

e = nat.Engine(0)
x0 = torch.from_numpy(np.random.default_rng(8).random((1, 1, 256, 256), dtype=np.float32))
base = uo.synthetic_state_dict(3)
g = torch.Generator().manual_seed(5)
heavy = dict(base)
for k, v in heavy.items():
    if k.endswith(".weight") and v.ndim == 4 and v.shape[-1] == 3 and v.shape[1] >= 64:
        heavy[k] = torch.where(torch.rand(v.shape, generator=g) < 5e-4, v * 60.0, v)
models = [("bench R231 (lung-like head)", synthetic.synthetic_state_dict(3, head="lunglike")), ("bench LTRCLobes (lung-like head)", synthetic.synthetic_state_dict(6, head="lunglike")),
          ("seeded random head C=3", base), ("seeded random head C=6", uo.synthetic_state_dict(6))]
for std in (8.0, 16.0, 30.0, 100.0):
    models.append((f"Appendix-D head std {std:g}", uo.calibrate_head(base, x0, std)))
models.append(("heavy-tailed (60x outliers), head std 8", uo.calibrate_head(heavy, x0, 8.0)))
models.append(("BatchNorm scales over 3 decades + heavy tails, head std 8", uo.calibrate_head(wide_batchnorm_heavy_tail_state_dict(3), x0, 8.0)))
for name, sd in models:
    e.load_state_dict(0, sd)
    err, pinned = e.model_probe(0)
    print(f"{name:62s} probe {err if err is None else format(err, '.2e')}  pinned {pinned}  runs on {e.model_tier(0)}", flush=True)
