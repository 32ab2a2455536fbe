"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the U-Net forward pass.

Plain torch-fp32 (CPU) restatement of `lungmask/resunet.py:58-70`
(`UNet.forward`) exactly as `lungmask/mask.py:58-65` instantiates it:
``UNet(n_classes=C, padding=True, depth=5, up_mode='upsample',
batch_norm=True, residual=False)``.  It works on a *state_dict* only, so it
travels to the GPU box where `/root/reference` does not exist.

Parity pin: `tests/golden/unet_*.npz` were produced by running the reference's
own `resunet.UNet` class (imported from /root/reference) on the same seeded
state_dict -- see `oracle/make_golden.py`; `tests/test_oracle_unet.py`
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

DEPTH = 5
WF = 6


def channel_plan():
    """[(in, out)] for the 5 encoder blocks and 4 decoder blocks.
    resunet.py:38-53 with depth=5, wf=6."""
    down, prev = [], 1
    for i in range(DEPTH):
        down.append((prev, 2 ** (WF + i)))
        prev = 2 ** (WF + i)
    up = []
    for i in reversed(range(DEPTH - 1)):
        up.append((prev, 2 ** (WF + i)))
        prev = 2 ** (WF + i)
    return down, up


def state_dict_keys(n_classes: int):
    """Key order and shapes of the reference state_dict (resunet.py:73-135).
    Includes the always-constructed-but-unused residual_* tensors
    (resunet.py:81-82,125-126) and num_batches_tracked."""
    down, up = channel_plan()
    keys = []

    def conv_block(prefix, cin, cout):
        # residual_input_conv / residual_batchnorm are registered FIRST
        keys.append((f"{prefix}.residual_input_conv.weight", (cout, cin, 1, 1)))
        keys.append((f"{prefix}.residual_input_conv.bias", (cout,)))
        for nm in ("weight", "bias", "running_mean", "running_var"):
            keys.append((f"{prefix}.residual_batchnorm.{nm}", (cout,)))
        keys.append((f"{prefix}.residual_batchnorm.num_batches_tracked", ()))
        keys.append((f"{prefix}.block.0.weight", (cout, cin, 3, 3)))
        keys.append((f"{prefix}.block.0.bias", (cout,)))
        for nm in ("weight", "bias", "running_mean", "running_var"):
            keys.append((f"{prefix}.block.2.{nm}", (cout,)))
        keys.append((f"{prefix}.block.2.num_batches_tracked", ()))
        keys.append((f"{prefix}.block.3.weight", (cout, cout, 3, 3)))
        keys.append((f"{prefix}.block.3.bias", (cout,)))
        for nm in ("weight", "bias", "running_mean", "running_var"):
            keys.append((f"{prefix}.block.5.{nm}", (cout,)))
        keys.append((f"{prefix}.block.5.num_batches_tracked", ()))

    for i, (cin, cout) in enumerate(down):
        conv_block(f"down_path.{i}", cin, cout)
    for i, (cin, cout) in enumerate(up):
        p = f"up_path.{i}"
        keys.append((f"{p}.residual_input_conv.weight", (cout, cin, 1, 1)))
        keys.append((f"{p}.residual_input_conv.bias", (cout,)))
        for nm in ("weight", "bias", "running_mean", "running_var"):
            keys.append((f"{p}.residual_batchnorm.{nm}", (cout,)))
        keys.append((f"{p}.residual_batchnorm.num_batches_tracked", ()))
        keys.append((f"{p}.up.1.weight", (cout, cin, 1, 1)))
        keys.append((f"{p}.up.1.bias", (cout,)))
        conv_block(f"{p}.conv_block", cin, cout)
    keys.append(("last.weight", (n_classes, 64, 1, 1)))
    keys.append(("last.bias", (n_classes,)))
    return keys


def synthetic_state_dict(n_classes: int = 3, seed: int = 231) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic, non-degenerate stand-in for the pretrained .pth files
    (no network here).  Same keys/shapes/order as the
    reference state_dict so `mask.py:56` (n_classes = len(last tensor)) holds.

    Conv weights ~ U(-b, b) with b = sqrt(6/fan_in)/sqrt(3)... (Kaiming-uniform
    like torch's default), BN stats perturbed so BN is not the identity, head
    scaled so that logits are O(10)."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for name, shape in state_dict_keys(n_classes):
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.tensor(0, dtype=torch.int64)
        elif name.endswith("running_var"):
            sd[name] = 0.75 + 0.5 * torch.rand(shape, generator=g)
        elif name.endswith("running_mean"):
            sd[name] = 0.1 * torch.randn(shape, generator=g)
        elif ".block.2." in name or ".block.5." in name or "residual_batchnorm" in name:
            if name.endswith("weight"):
                sd[name] = 0.75 + 0.5 * torch.rand(shape, generator=g)
            else:
                sd[name] = 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("weight"):
            fan_in = shape[1] * shape[2] * shape[3]
            # gain chosen so activations keep O(1) variance through ReLU+BN
            bound = math.sqrt(6.0 / fan_in)
            sd[name] = (torch.rand(shape, generator=g) * 2 - 1) * bound
        else:  # conv bias
            sd[name] = 0.05 * torch.randn(shape, generator=g)
    # head: make logits O(10) with class-dependent offsets (argmax diversity)
    sd["last.weight"] = sd["last.weight"] * 12.0
    sd["last.bias"] = torch.linspace(-1.0, 1.0, n_classes)
    return sd


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


def forward(sd, x: torch.Tensor) -> torch.Tensor:
    """== UNet.forward: LogSoftmax(dim=1) of the logits (resunet.py:70)."""
    return F.log_softmax(forward_logits(sd, x), dim=1)


def predict_labels(sd, x: torch.Tensor) -> np.ndarray:
    """mask.py:183-186: torch.max(prediction, 1)[1] -> uint8 (first index on ties)."""
    with torch.inference_mode():
        pred = forward(sd, x)
        return torch.max(pred, 1)[1].cpu().numpy().astype(np.uint8)


FLOP_PER_SLICE = {3: 96.200556544e9, 6: 96.225722368e9}
