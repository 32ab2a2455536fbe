"""(test-side probe: it uses the oracle's torch restatement as the thing measured, never the product)

SURVEY.md section 8(d), "optional secondary baseline": the reference's own compute path -- PyTorch eager, fp32, NCHW, what
`LMInferer(force_cpu=False)` runs on a GPU (mask.py:118-134, :173-187) -- on the SAME MI355X through PyTorch-ROCm (MIOpen /
rocBLAS), timed beside the engine on the same slices.  Two questions:

  * how fast is the batch loop of mask.py:173-187 (float32 batch of 20 to the device, model, torch.max, labels back to the host) in
    slices/s, network only -- the number the engine's network-only rate compares with on equal hardware;
  * how far is the reference's GPU result from the reference's CPU result (the parity bar is defined against the CPU path; the
    reference's own GPU path has a different summation order too).

    python tests/torch_gpu_baseline.py [n_slices=100] [repeats=3]

Prints one JSON line.  Not part of the test suites (no GPU assertion depends on MIOpen's kernel choice).
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
import numpy as np
import torch

from lungmask_amd import _native as nat
from oracle import prepost_oracle as po
from oracle import unet_oracle as uo


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    batch = 20
    dev = torch.device("cuda", 0)
    sd = uo.synthetic_state_dict(3, head="lunglike")
    z0 = 150 - n // 2
    vol = po.phantom(300, 512, 512, z0=z0, z1=z0 + n)
    xs, _ = po.preprocess(vol, [256, 256])
    x = po.normalise(xs)[:, None].astype(np.float32)  # [n,1,256,256], what mask.py:167-168 hands to the DataLoader
    sd_dev = {k: v.to(dev) for k, v in sd.items()}

    def ref_loop(xin, sdd, autocast=None):
        """mask.py:173-187 on the device: per batch float32 -> device, forward, torch.max(...)[1] -> uint8 on the host."""
        out = np.empty((0, 256, 256), dtype=np.uint8)
        parts = []
        with torch.inference_mode():
            for b0 in range(0, len(xin), batch):
                xb = torch.from_numpy(xin[b0:b0 + batch]).float().to(dev)
                if autocast is not None:
                    with torch.autocast("cuda", dtype=autocast):
                        pred = uo.forward(sdd, xb)
                else:
                    pred = uo.forward(sdd, xb)
                parts.append(torch.max(pred, 1)[1].detach().cpu().numpy().astype(np.uint8))
        return np.vstack([out] + parts)

    res = {"probe": "reference compute path (PyTorch eager) on the same MI355X, network only (mask.py:173-187), batch 20", "n_slices": n,
           "torch": torch.__version__, "device": torch.cuda.get_device_name(0)}
    variants = [("torch_fp32_nchw", None, False), ("torch_fp32_miopen_benchmark", None, True), ("torch_autocast_f16", torch.float16, False),
                ("torch_autocast_bf16", torch.bfloat16, False)]
    labels_fp32 = None
    for name, ac, bench in variants:
        torch.backends.cudnn.benchmark = bench
        try:
            t0 = time.perf_counter()
            lab = ref_loop(x[:batch], sd_dev, ac)  # warm-up (kernel selection / compilation)
            torch.cuda.synchronize()
            warm = time.perf_counter() - t0
            best = None
            for _ in range(reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                lab = ref_loop(x, sd_dev, ac)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            res[name] = {"slices_per_s": round(n / best, 1), "ms_per_batch_of_20": round(best / (n / batch) * 1e3, 2), "warmup_s": round(warm, 1)}
            if name == "torch_fp32_nchw":
                labels_fp32 = lab
            elif labels_fp32 is not None:
                res[name]["labels_differing_from_torch_fp32_gpu"] = int((lab != labels_fp32).sum())
        except Exception as e:  # a variant MIOpen cannot serve must not take the probe down
            res[name] = {"error": repr(e)[:200]}
    # the reference's GPU result against the reference's CPU result (8 slices: the CPU forward is the slow part)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    xs8 = torch.from_numpy(x[n // 2 - 4:n // 2 + 4])
    with torch.inference_mode():
        cpu = uo.forward(sd, xs8)
        gpu = uo.forward(sd_dev, xs8.to(dev)).cpu()
    res["torch_gpu_vs_torch_cpu_fp32"] = {"max_abs_dlogp": float((gpu - cpu).abs().max()), "labels_differing": int((gpu.argmax(1) != cpu.argmax(1)).sum()), "slices": 8}
    # the engine on the same slices, network only, numpy in -> labels out is not what is timed here: device-resident forward_batches
    eng = nat.Engine(0)
    eng.load_state_dict(0, sd)
    xd = eng.to_device(x[:, 0])
    ld = eng.empty((n, 256, 256), np.uint8)
    for lanes in (2, 1):
        eng.set_streams(lanes)
        eng.L.check(eng.L.lib.lm_forward_batches_dev(eng.h, 0, xd.ptr, n, 256, 256, batch, ld.ptr))
        eng.sync()
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            eng.L.check(eng.L.lib.lm_forward_batches_dev(eng.h, 0, xd.ptr, n, 256, 256, batch, ld.ptr))
            eng.sync()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        res[f"engine_split_f16_{lanes}_lane"] = {"slices_per_s": round(n / best, 1), "ms_per_batch_of_20": round(best / (n / batch) * 1e3, 2)}
    lab_e = ld.download()
    with torch.inference_mode():
        logp8 = eng.forward(0, x[n // 2 - 4:n // 2 + 4, 0])[1]
    res["engine_vs_torch_cpu_fp32"] = {"max_abs_dlogp": float(np.abs(logp8 - cpu.numpy()).max()), "slices": 8}
    if labels_fp32 is not None:
        res["engine_labels_differing_from_torch_fp32_gpu"] = int((lab_e != labels_fp32).sum())
    print(json.dumps(res))


if __name__ == "__main__":
    main()
