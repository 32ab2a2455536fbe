"""(test-side tool: it uses the oracle as the checker)  max |delta log-prob| of the engine vs the torch-fp32 reference forward on random inputs (argv: library paths)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
import numpy as np, torch
from lungmask_amd import _native as nat
from oracle import unet_oracle as uo
torch.set_num_threads(16)
for C in (3, 6):
    sd = uo.synthetic_state_dict(C)
    x = np.random.default_rng(5).random((2, 256, 256), dtype=np.float32)
    ref = uo.forward(sd, torch.from_numpy(x[:, None])).numpy()
    for path in sys.argv[1:]:
        e = nat.Engine(0, nat.Library(path)); e.load_state_dict(0, sd)
        lab, logp = e.forward(0, x)
        print(f"C={C} {os.path.basename(path):28s} max|dlogp| = {np.abs(logp - ref).max():.3e}  mean = {np.abs(logp - ref).mean():.3e}  label mismatches vs ref argmax: {(lab != ref.argmax(1)).sum()}", flush=True)
        e.close()
