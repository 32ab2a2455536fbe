"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the U-Net forward pass.

Plain torch-fp32 (CPU) restatement of `lungmask/resunet.py:58-70`
(`UNet.forward`) exactly as `lungmask/mask.py:58-65` instantiates it:
``UNet(n_classes=C, padding=True, depth=5, up_mode='upsample',
batch_norm=True, residual=False)``.  It works on a *state_dict* only, so it
travels to the GPU box where `/root/reference` does not exist.

Parity pin: `tests/golden/unet_*.npz` were produced by running the reference's
own `resunet.UNet` class (imported from /root/reference) on the same seeded
state_dict -- see `oracle/make_golden.py`; `tests/test_oracle.py`
checks this restatement against them.

Floating point: the tolerance of the path is 1e-3 absolute on the
log-softmax output (BASELINE.json north_star).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from lungmask_amd.synthetic import DEPTH, WF, channel_plan, state_dict_keys, synthetic_state_dict  # noqa: E402,F401  (generators live with the package)


def _conv_block(sd, prefix, x):
    """resunet.py:93-105 with residual=False, batch_norm=True, padding=1:
    Conv3x3 -> ReLU -> BN(eval) -> Conv3x3 -> ReLU -> BN(eval)."""
    for conv, bn in ((0, 2), (3, 5)):
        x = F.conv2d(x, sd[f"{prefix}.block.{conv}.weight"], sd[f"{prefix}.block.{conv}.bias"], padding=1)
        x = F.relu(x)
        x = F.batch_norm(
            x,
            sd[f"{prefix}.block.{bn}.running_mean"],
            sd[f"{prefix}.block.{bn}.running_var"],
            sd[f"{prefix}.block.{bn}.weight"],
            sd[f"{prefix}.block.{bn}.bias"],
            training=False,
            eps=1e-5,
        )
    return x


def forward_logits(sd, x: torch.Tensor, return_features: bool = False):
    """Pre-softmax logits of resunet.py:58-69. x: f32 [B,1,H,W], H,W % 16 == 0."""
    feats = {}
    blocks = []
    for i in range(DEPTH):
        x = _conv_block(sd, f"down_path.{i}", x)  # resunet.py:60-61
        feats[f"down{i}"] = x
        if i != DEPTH - 1:
            blocks.append(x)
            x = F.avg_pool2d(x, 2)  # resunet.py:64
    for i in range(DEPTH - 1):
        p = f"up_path.{i}"
        bridge = blocks[-i - 1]
        up = F.interpolate(x, scale_factor=2, mode="bilinear")  # nn.Upsample, resunet.py:132
        up = F.conv2d(up, sd[f"{p}.up.1.weight"], sd[f"{p}.up.1.bias"])  # resunet.py:133
        x = torch.cat([up, bridge], 1)  # resunet.py:147 (center_crop is a no-op with padding=True)
        x = _conv_block(sd, f"{p}.conv_block", x)
        feats[f"up{i}"] = x
    logits = F.conv2d(x, sd["last.weight"], sd["last.bias"])  # resunet.py:69
    if return_features:
        return logits, feats
    return logits


def calibrate_head(sd, x0: torch.Tensor, std: float):
    """SURVEY.md Appendix D's head calibration: a copy of `sd` whose 1x1 head is rescaled so that, on the input x0, every class's
    logit map has mean 0 and standard deviation `std` (`last.weight *= std/std_c(L)`, `last.bias = -mean_c(L)*std/std_c(L)`).
    The recipe's std 8 gives logit ranges of a few tens up to +-100; tests sweep larger values."""
    sd = OrderedDict(sd)
    with torch.inference_mode():
        sd["last.bias"] = torch.zeros_like(sd["last.bias"])
        L = forward_logits(sd, x0)
        s, m = L.std(dim=(0, 2, 3)), L.mean(dim=(0, 2, 3))
        sd["last.weight"] = sd["last.weight"] * (std / s)[:, None, None, None]
        sd["last.bias"] = -m * std / s
    return sd


def forward_f64(sd, x: torch.Tensor) -> torch.Tensor:
    """The same graph evaluated in float64: what the reference's fp32 arithmetic itself is an approximation of.  Tests use
    |forward - forward_f64| as the reference's own rounding noise for a given model (it grows with the logit range)."""
    sd64 = {k: (v.double() if torch.is_floating_point(v) else v) for k, v in sd.items()}
    return F.log_softmax(forward_logits(sd64, x.double()), dim=1)


def forward(sd, x: torch.Tensor) -> torch.Tensor:
    """== UNet.forward: LogSoftmax(dim=1) of the logits (resunet.py:70)."""
    return F.log_softmax(forward_logits(sd, x), dim=1)


def predict_labels(sd, x: torch.Tensor) -> np.ndarray:
    """mask.py:183-186: torch.max(prediction, 1)[1] -> uint8 (first index on ties)."""
    with torch.inference_mode():
        pred = forward(sd, x)
        return torch.max(pred, 1)[1].cpu().numpy().astype(np.uint8)


from lungmask_amd.synthetic import FLOP_PER_SLICE  # noqa: E402,F401
