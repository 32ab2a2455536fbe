"""TEST / WORKLOAD INFRASTRUCTURE -- fits the 1x1 head of the synthetic stand-in network so that the HU phantom's label volume looks
like a lung segmentation (the random head calls 60 % of the volume one class; post-processing time and the merge
replay are data dependent).

The pretrained weights are not available offline, so the 23 conv layers stay the seeded random ones of
`lungmask_amd.synthetic.synthetic_state_dict`; only `last.weight` / `last.bias` (resunet.py:55,69) are replaced by a ridge
regression of the network's own final 64-channel features (torch-fp32 oracle forward) onto +-6 logit targets derived from the
phantom's geometry: lungs = dark regions of the pre-processed slice that do not touch the crop border, left / right by column
(3 classes), additionally cut into lobes by row thirds / halves (6 classes).  A linear read-out of random features segments the
lungs only roughly -- two lung-sized components plus specks and ragged borders, which is the point.

    python oracle/make_lunglike_head.py        # writes lungmask_amd/data/lunglike_head_c{3,6}.npz (committed, ~2 KB each)
"""
import os
import sys

import numpy as np
import torch
from scipy import ndimage

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lungmask_amd import synthetic  # noqa: E402
from oracle import prepost_oracle as po  # noqa: E402
from oracle import unet_oracle as uo  # noqa: E402

ZS = (20, 60, 100, 140, 150, 190, 230, 280)


def targets(x: np.ndarray, n_classes: int) -> np.ndarray:
    """x: pre-processed, normalised slice [256,256] -> class map."""
    low = x < 0.3
    lab, n = ndimage.label(low)
    t = np.zeros(x.shape, np.int64)
    for i in range(1, n + 1):
        m = lab == i
        ys, xs = np.nonzero(m)
        if ys.min() == 0 or xs.min() == 0 or ys.max() == x.shape[0] - 1 or xs.max() == x.shape[1] - 1 or m.sum() < 50:
            continue  # air around the body / specks
        left = xs.mean() < x.shape[1] / 2
        if n_classes == 3:
            t[m] = 1 if left else 2
        else:  # lobes: left lung two (rows), right lung three
            y0, y1 = ys.min(), ys.max() + 1
            rows = np.arange(x.shape[0])[:, None] * np.ones((1, x.shape[1]), np.int64)
            if left:
                t[m & (rows < (y0 + y1) // 2)] = 1
                t[m & (rows >= (y0 + y1) // 2)] = 2
            else:
                a, b = y0 + (y1 - y0) // 3, y0 + 2 * (y1 - y0) // 3
                t[m & (rows < a)] = 3
                t[m & (rows >= a) & (rows < b)] = 4
                t[m & (rows >= b)] = 5
    return t


def main():
    torch.set_num_threads(16)
    vol = synthetic.phantom(300, 512, 512)[list(ZS)]
    xs, _ = po.preprocess(vol, [256, 256])
    x = po.normalise(xs)
    for C in (3, 6):
        sd = synthetic.synthetic_state_dict(C)
        with torch.inference_mode():
            _, feats = uo.forward_logits(sd, torch.from_numpy(x[:, None]), return_features=True)
        F = feats["up3"].numpy()  # [n, 64, 256, 256]
        T = np.stack([targets(x[i], C) for i in range(len(x))])
        Fm = F.transpose(0, 2, 3, 1).reshape(-1, 64)[::7].astype(np.float64)
        Tm = T.reshape(-1)[::7]
        Y = -6.0 * np.ones((len(Tm), C))
        Y[np.arange(len(Tm)), Tm] = 6.0
        A = np.concatenate([Fm, np.ones((len(Fm), 1))], axis=1)
        lam = 1e-3 * len(A)
        Wb = np.linalg.solve(A.T @ A + lam * np.eye(65), A.T @ Y)  # ridge
        W, b = Wb[:64].T.astype(np.float32), Wb[64].astype(np.float32)
        pred = (F.transpose(0, 2, 3, 1) @ W.T + b).argmax(-1)
        acc = (pred == T).mean()
        hist = np.bincount(pred.reshape(-1), minlength=C) / pred.size
        print(f"C={C}: fit on {len(A)} pixels, train agreement {acc:.3f}, predicted class shares {np.round(hist, 3)}, target shares "
              f"{np.round(np.bincount(T.reshape(-1), minlength=C) / T.size, 3)}")
        out = os.path.join(ROOT, "lungmask_amd", "data", f"lunglike_head_c{C}.npz")
        np.savez(out, weight=W, bias=b, seed=np.int64(231), zs=np.asarray(ZS))
        print("wrote", out)


if __name__ == "__main__":
    main()
