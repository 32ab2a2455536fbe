import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from lungmask_amd import _native as nat
from lungmask_amd import synthetic as uo
eng = nat.Engine(0)
eng.load_state_dict(0, uo.synthetic_state_dict(3))
g = np.load('/root/repo/tests/golden/unet_c3.npz')
for case in ('rand32','rand64'):
    x = g[case+'_x']; ref = g[case+'_logp']
    eng.set_precision('split_f16'); lab, logp = eng.forward(0, x)
    err = np.abs(logp-ref).max(axis=1)  # [b,h,w]
    print(case, 'max', err.max(), 'frac bad', (err>1e-3).mean())
    for b in range(err.shape[0]):
        bad = err[b] > 1e-3
        print(' b',b,'bad rows', np.nonzero(bad.any(1))[0][:40].tolist(), 'bad cols', np.nonzero(bad.any(0))[0][:70].tolist())
