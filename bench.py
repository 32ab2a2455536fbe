#!/usr/bin/env python
"""bench.py -- CT slices/s of the lungmask hot path (`LMInferer.apply`) on MI355X.

One "step" = one pass of the whole hot path (body-mask crop + 256x256 resample ->
U-Net forward in batches of 20 -> argmax -> 3-D connected-component clean-up ->
un-crop) over a synthetic 512x512 int16 HU volume that is already resident in
HBM; the uint8 label volume is left in HBM.  Workload = BASELINE.json configs[1]
(R231, 512x512x300, batchsize=20) per GPU; with N GPUs the volume has 300*N
slices, slice-sharded, with the RCCL all-gathers of lungmask_amd/pipeline.py
(weak scaling).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: Peak FP32 (matrix)
PEAK_F16_MFMA_TFLOPS = 2500.0  # same table: Peak BF16/FP16 MFMA, dense
# What the matrix pipes SUSTAIN on this network's operand statistics (tools/ubench/mfma_power.hip, the conv kernel's own
# instruction pattern on full-range split operands, no memory traffic at all): the chip clocks to its power budget, 2.37 GHz
# on zeros but ~1.6 GHz on real data.  profiles/history/r02d_mfma_power.log
SUSTAINED_F16_MFMA_TFLOPS = 1737.0
# algorithmic FLOP per slice, SURVEY.md Appendix A (256x256): R231 (3 classes), LTRCLobes (6 classes)
FLOP_PER_SLICE = {3: 96.200556544e9, 6: 96.225722368e9}
CONFIGS = {  # BASELINE.json configs[1..3]
    2: ("R231 U-Net", 3, None),
    3: ("LTRCLobes 6-class U-Net", 6, None),
    4: ("LTRCLobes_R231 fused mode (LTRCLobes + R231 fill model, label fusion, full-resolution post-processing)", 6, 3),
}


PMC_PIN = os.path.join(ROOT, "profiles", "pmc_current.json")  # written by tools/summarize_prof.py --pin


def _sha256(path):
    import hashlib

    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of the dominant kernel, launch-weighted over every instantiation whose name contains
    `kernel_substr`, from the ONE committed rocprofv3 PMC summary that profiles/pmc_current.json names (file, the commit
    it was measured on, sha256 of the kernel sources at that time).  It is read from that file, not measured in this run.
    Returns (bytes | None, provenance string): None when there is no pinned summary or when the conv kernel's sources
    have changed since it was taken (a stale number is refused, not reported)."""
    if not os.path.exists(PMC_PIN):
        return None, "no pinned PMC summary (profiles/pmc_current.json)"
    pin = json.load(open(PMC_PIN))
    path = os.path.join(ROOT, "profiles", pin["file"])
    if not os.path.exists(path):
        return None, f"pinned PMC summary profiles/{pin['file']} is missing"
    for rel, want in pin.get("sources_sha256", {}).items():
        if not os.path.exists(os.path.join(ROOT, rel)) or _sha256(os.path.join(ROOT, rel)) != want:
            return None, (f"REFUSED: {rel} has changed since profiles/{pin['file']} was measured (commit {pin.get('commit', '?')}); "
                          "re-run the PMC passes (tools/gpu_round_check.sh TAG pmc; tools/summarize_prof.py ... --pin)")
    d = json.load(open(path))
    tot = n = 0.0
    for k, v in d.get("kernels", {}).items():
        if kernel_substr in k and "hbm_bytes_per_launch_corrected" in v:
            w = float(v.get("launches_FETCH_SIZE", 1))
            tot += v["hbm_bytes_per_launch_corrected"] * w
            n += w
    return (tot / n if n else None), f"profiles/{pin['file']} (measured on commit {pin.get('commit', '?')}, kernel sources unchanged since)"


def cpu_baseline_config1(sd):
    """BASELINE.json configs[0] itself: the 512 x 512 x 32 synthetic volume, batch 1, through the CPU restatement of the reference
    (oracle/, kind = 'port'; the reference cannot be imported on the GPU box) -- the whole 32-slice volume, not a sample."""
    import torch

    from oracle import prepost_oracle as po
    from oracle import unet_oracle as uo

    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    vol = po.phantom(32, 512, 512)
    t = time.perf_counter()
    po.inference(vol, lambda xb: uo.predict_labels(sd, torch.from_numpy(np.ascontiguousarray(xb))), batch_size=1)
    dt = time.perf_counter() - t
    return {"value": round(32 / dt, 3), "unit": "slices/s", "cores": cores, "kind": "port",
            "sample": f"BASELINE.json configs[0]: the whole 512x512x32 phantom, batch 1, torch-CPU fp32 forward on {cores} threads + scipy pre/post, {dt:.1f} s"}


def cpu_baseline(n_sample, sd):
    """The reference algorithm restated on the CPU (oracle/, kind='port'), timed on this host's cores on a
    bounded sample: n_sample central slices of the same phantom, batch 1 (what the reference's --cpu forces)."""
    import torch

    from oracle import prepost_oracle as po
    from oracle import unet_oracle as uo

    # torch's intra-op pool degrades badly beyond a few dozen threads at batch 1; use what actually helps
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    z0 = 150 - n_sample // 2
    vol = po.phantom(300, 512, 512, z0=z0, z1=z0 + n_sample)
    t_nn = [0.0]

    def predict(xb):
        t = time.perf_counter()
        r = uo.predict_labels(sd, torch.from_numpy(np.ascontiguousarray(xb)))
        t_nn[0] += time.perf_counter() - t
        return r

    predict(np.zeros((1, 1, 256, 256), np.float32))  # warm-up
    t_nn[0] = 0.0
    t = time.perf_counter()
    po.inference(vol, predict, batch_size=1)
    dt = time.perf_counter() - t
    out = {
        "value": n_sample / dt,
        "unit": "slices/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{n_sample} central slices (z={z0}..{z0 + n_sample - 1}) of the 512x512x300 phantom, batch 1 (what --cpu forces), "
                  f"torch-CPU fp32 forward on {cores} threads ({t_nn[0]:.1f} s) + single-threaded scipy pre/post ({dt - t_nn[0]:.1f} s), {dt:.1f} s total",
    }
    # Beside the CPU number: the same restated forward on THIS GPU through PyTorch-ROCm (eager fp32 NCHW, MIOpen / rocBLAS) -- what the
    # reference runs without --cpu (mask.py:118-134, batch loop :173-187; SURVEY 8d's optional secondary baseline), network only, on
    # 60 pre-processed slices in batches of 20.  A reported baseline like `value` above, never the thing `bench.py` measures; any
    # failure of that stack is recorded, not raised.
    try:
        xs, _ = po.preprocess(vol[:60], [256, 256])
        x = po.normalise(xs)[:, None].astype(np.float32)
        dev = torch.device("cuda", torch.cuda.current_device())
        sdd = {k: v.to(dev) for k, v in sd.items()}

        def loop():
            with torch.inference_mode():
                for b0 in range(0, len(x), 20):
                    pred = uo.forward(sdd, torch.from_numpy(x[b0:b0 + 20]).float().to(dev))
                    torch.max(pred, 1)[1].detach().cpu().numpy().astype(np.uint8)

        t = time.perf_counter()
        loop()
        torch.cuda.synchronize()
        warm = time.perf_counter() - t
        best = None
        for _ in range(2):
            t = time.perf_counter()
            loop()
            torch.cuda.synchronize()
            best = min(best or 1e9, time.perf_counter() - t)
        out["same_gpu_reference_path"] = {
            "value": round(len(x) / best, 1), "unit": "slices/s (network only)", "ms_per_batch_of_20": round(best / (len(x) / 20) * 1e3, 2),
            "what": f"the restated UNet.forward in PyTorch {torch.__version__} eager fp32 on this GPU (MIOpen / rocBLAS), the batch loop of mask.py:173-187 "
                    f"(batch to the device, forward, torch.max, labels to the host) over {len(x)} slices; first pass {warm:.1f} s (kernel selection); "
                    "compare with this line's value / end_to_end_tflops, which include pre- and post-processing"}
        del sdd
        torch.cuda.empty_cache()
    except Exception as e:  # noqa: BLE001 -- a baseline must not take the benchmark down
        out["same_gpu_reference_path"] = {"error": repr(e)[:300]}
    return out


class ChipSampler:
    """Shader clock and socket power of one GPU, sampled by a thread WHILE the timed repetitions run, so that a reader of the line
    can tell a slow box from a slow tree (boxes of this pool differ by +-3 % in the clock they sustain under this load).  Source:
    the amdgpu hwmon files (one small read per value), else `rocm-smi --json` (a process per sample: coarse).  Failures are
    recorded in the line, never raised; nothing here touches the GPU's queues."""

    def __init__(self, index, period=0.05, pci_bus_id=None):
        import glob
        import threading

        self.period, self.samples, self.err, self.source = period, [], None, None
        self._stop = threading.Event()
        self._th = None
        self.files = None
        try:
            cards = []
            for c in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
                if os.path.basename(c).count("-"):
                    continue
                try:
                    if open(os.path.join(c, "device/vendor")).read().strip() != "0x1002":
                        continue
                except OSError:
                    continue
                cards.append((os.path.realpath(os.path.join(c, "device")), c))
            cards.sort()
            # the card of THIS process's device: by PCI address when the runtime gives one (a box may show more cards in sysfs than
            # the container has devices: r05e sampled an idle neighbour at 114 MHz), else by position
            if pci_bus_id:
                hit = [i for i, c in enumerate(cards) if c[0].lower().endswith(pci_bus_id.lower())]
                index = hit[0] if hit else len(cards)
            if index < len(cards):
                hw = sorted(glob.glob(os.path.join(cards[index][1], "device/hwmon/hwmon*")))
                if hw:
                    pw = next((os.path.join(hw[0], f) for f in ("power1_average", "power1_input") if os.path.exists(os.path.join(hw[0], f))), None)
                    fq = os.path.join(hw[0], "freq1_input") if os.path.exists(os.path.join(hw[0], "freq1_input")) else None
                    if pw or fq:
                        self.files = (pw, fq)
                        self.source = f"sysfs hwmon ({cards[index][0].rsplit('/', 1)[-1]}" + (", matched by PCI address)" if pci_bus_id else ", by position)")
        except Exception as e:  # noqa: BLE001
            self.err = repr(e)[:200]
        self.index = index

    def _read(self):
        if self.files is not None:
            pw, fq = self.files
            p = float(open(pw).read()) * 1e-6 if pw else None  # microwatts
            f = float(open(fq).read()) * 1e-6 if fq else None  # Hz -> MHz
            return f, p
        import subprocess

        out = subprocess.run(["rocm-smi", "-d", str(self.index), "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out)
        c = d[sorted(d)[0]]
        f = p = None
        for k, v in c.items():
            kl = k.lower()
            if "sclk clock speed" in kl:
                f = float(str(v).strip("()").lower().replace("mhz", ""))
            elif "power" in kl and "(w)" in kl:
                p = float(v)
        self.source = "rocm-smi --showclocks --showpower --json (one process per sample)"
        return f, p

    def start(self):
        import threading

        def loop():
            while not self._stop.is_set():
                try:
                    self.samples.append((time.perf_counter(),) + tuple(self._read()))
                except Exception as e:  # noqa: BLE001
                    self.err = repr(e)[:200]
                    return
                self._stop.wait(self.period)

        self._th = threading.Thread(target=loop, daemon=True)
        self._th.start()

    def stop(self):
        self._stop.set()
        if self._th is not None:
            self._th.join(30)

    def summary(self, t_begin, t_end):
        def stat(vals):
            v = sorted(x for x in vals if x is not None)
            return None if not v else {"min": round(v[0], 1), "median": round(v[len(v) // 2], 1), "max": round(v[-1], 1)}

        inside = [s for s in self.samples if t_begin <= s[0] <= t_end]
        return {"sclk_mhz": stat(s[1] for s in inside), "socket_power_w": stat(s[2] for s in inside), "samples": len(inside), "period_s": self.period,
                "source": self.source, "error": self.err,
                "note": "sampled by a host thread while the timed repetitions ran (the contract's timed region and the extra ones); the chip "
                        "clocks to its power budget, and boxes of this pool differ by a few per cent in the clock they sustain under this load"}


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, emu):
    """`python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU, LOCAL_RANK = device),
    exactly the environment `python -m torch.distributed.run --nproc-per-node N` would give them.  Rank 0 inherits stdout
    (the JSON line); the exit status is the worst of the ranks'."""
    import subprocess

    if not emu:
        import torch

        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible on this machine -- refusing to report a {have}-GPU number as a "
                             f"{n}-GPU one (BASELINE config 5 needs an {n}-GPU node)")
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LM_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rcs = [p.wait() for p in procs]
    raise SystemExit(max(abs(rc) for rc in rcs))


class NativeGroup:
    """`--dist native`: every collective of the job -- the pipeline's all-gathers AND the bench's own barrier / max-over-ranks
    -- goes through the engine's RCCL communicator behind the C ABI (lm_dist_*); torch.distributed is not initialised, a
    TCPStore only hands the ncclUniqueId to the ranks."""

    def __init__(self, eng, rank, world, device):
        import datetime

        import torch
        import torch.distributed as td

        from lungmask_amd.pipeline import NativeDist

        self.torch, self.dev = torch, device
        store = td.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29533")), world, rank == 0,
                            timeout=datetime.timedelta(seconds=120))
        uid = NativeDist.exchange_id(eng, rank, store) if world > 1 else None
        self.nd = NativeDist(eng, rank, world, uid)
        self.eng = eng
        self._one = torch.zeros(1, dtype=torch.float64, device=device)
        self._all = torch.zeros(world, dtype=torch.float64, device=device)

    def gather_f64(self, v):
        self._one.fill_(float(v))
        self.torch.cuda.synchronize()
        self.nd.all_gather_into_tensor(self._all, self._one)
        self.eng.sync()
        return [float(x) for x in self._all.cpu().tolist()]

    def barrier(self):
        self.gather_f64(0.0)

    def max(self, v):
        return max(self.gather_f64(v))

    def world_size(self):
        return int(self.eng.L.lib.lm_dist_world(self.eng.h))

    def destroy(self):
        self.nd.destroy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--slices", type=int, default=300, help="slices per GPU")
    ap.add_argument("--batch", type=int, default=20)
    ap.add_argument("--cpu-sample", type=int, default=96)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="split_f16", choices=["split_f16", "f32"])
    ap.add_argument("--head", default="lunglike", choices=["lunglike", "random"],
                    help="1x1 head of the synthetic stand-in weights: fitted so that the phantom's lungs are labelled as lungs (default; a label volume "
                         "like production's for the 3-D post-processing), or the seeded random one of rounds 1-3")
    ap.add_argument("--post", default="auto", choices=["auto", "slab", "gathered"],
                    help="N>1 post-processing: slab-sharded, or label all-gather + redundant whole-volume pass; auto (default) = the pipeline's own "
                         "choice by world size: slab from four ranks on (profiles/history/r05d_slab_timing_lunglike.log)")
    ap.add_argument("--dist", default="torch", choices=["torch", "native"],
                    help="N>1 collectives: torch.distributed (backend nccl = RCCL) or the engine's own RCCL communicator behind the C ABI (lm_dist_*)")
    ap.add_argument("--streams", type=int, default=2, choices=[1, 2], help="forward lanes (2: consecutive batches overlap on two streams)")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4], help="BASELINE.json configuration: 2 R231 (the headline), 3 LTRCLobes, 4 LTRCLobes_R231 fused")
    ap.add_argument("--repeat", type=int, default=2, help="extra repetitions of the timed region, reported as min / median / max beside `value`")
    ap.add_argument("--host-steps", type=int, default=-1,
                    help="steps of the two numpy -> numpy measurements beside `value` (lm_apply_host and LMInferer.apply); default: --steps, 0 to skip")
    args = ap.parse_args()
    cfg_name, n_classes, fill_classes = CONFIGS[args.config]
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    # (BASELINE.json quotes configs 3 and 4 on one GPU; the sharded pipeline runs them on N as well -- SURVEY 8e rows 1-4)
    if args.host_steps < 0:
        args.host_steps = args.steps
    # TEST HOOK (tests/test_bench_spawn.py, CPU suite): LM_BENCH_EMU=1 runs this same script on the g++ emulation of the kernels
    # over gloo with a tiny volume, to drive the spawn / sharding / reporting logic where there is no GPU.  The line it prints
    # says so in `data`; nothing else reads the variable.
    emu = os.environ.get("LM_BENCH_EMU") == "1"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus, emu)  # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with `python bench.py --gpus N` or "
                         f"`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")

    import torch

    from lungmask_amd import _native as nat
    from lungmask_amd import synthetic as po  # phantom + seeded stand-in weights (no oracle code in the timed job)
    uo = po

    if not emu and (not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank):
        raise SystemExit(f"rank {rank}: no GPU cuda:{local_rank} on this machine ({torch.cuda.device_count() if torch.cuda.is_available() else 0} visible); "
                         "lungmask_amd has no CPU path")
    dist = None    # torch.distributed, when it carries the collectives
    group = None   # NativeGroup, when the engine's own communicator does
    # LM_BENCH_FORCE_DIST=1: take the multi-GPU code path (RCCL communicator, ShardedPipeline) with a world of one,
    # which is how that path is smoke-tested on a 1-GPU box
    use_dist = world > 1 or os.environ.get("LM_BENCH_FORCE_DIST") == "1"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if use_dist and args.dist == "torch":
        import torch.distributed as dist

        if emu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    if emu:
        from lungmask_amd.build import build_emu

        eng = nat.Engine(0, nat.Library(build_emu(), allow_emulation=True))
    else:
        eng = nat.Engine(local_rank)  # raises if liblungmask_hip.so or the GPU is missing: no fallback
    dev = torch.device("cpu") if emu else torch.device("cuda", local_rank)
    if use_dist and args.dist == "native":
        if emu:
            raise SystemExit("--dist native needs RCCL (no emulation of it)")
        torch.cuda.set_device(local_rank)
        group = NativeGroup(eng, rank, world, dev)
    wd = os.environ.get("LUNGMASK_WEIGHTS_DIR")
    pth = {3: "unet_r231-d5d2fc3d.pth", 6: "unet_ltrclobes-3a07043d.pth"}

    def load(c):
        if wd and os.path.exists(os.path.join(wd, pth[c])):
            return torch.load(os.path.join(wd, pth[c]), map_location="cpu"), "pretrained " + pth[c]
        return (uo.synthetic_state_dict(c, head=args.head),
                f"synthetic (lungmask_amd.synthetic.synthetic_state_dict({c}, head='{args.head}'), seed 231"
                + ("; 1x1 head fitted so that the phantom's lungs are labelled as lungs, oracle/make_lunglike_head.py)" if args.head == "lunglike" else ")"))

    sd, weights = load(n_classes)
    eng.load_state_dict(0, sd)
    fill_slot, sd_fill = -1, None
    if fill_classes is not None:
        sd_fill, w2 = load(fill_classes)
        eng.load_state_dict(1, sd_fill)
        weights += " + fill model: " + w2
        fill_slot = 1
    eng.set_precision(args.precision)
    eng.set_streams(args.streams)
    # what the load-time accuracy guard saw (include/lungmask_hip.h: lm_model_probe_error) and what the models therefore run on
    guard = None
    if hasattr(eng.L.lib, "lm_model_probe_error"):
        guard = {"limit": os.environ.get("LM_ACC_GUARD", "5e-4 (default)"), "models": []}
        for slot in ([0] if fill_slot < 0 else [0, 1]):
            err, pinned = eng.model_probe(slot)
            guard["models"].append({"slot": slot, "probe_max_abs_dlogp_split_vs_exact_fp32": err, "pinned_to_fp32_by_the_guard": pinned,
                                    "runs_on": eng.model_tier(slot) if hasattr(eng, "model_tier") else eng.model_precision(slot)})
        guard["note"] = ("lm_model_load ran two deterministic 256 x 256 probe slices (phantom-like, uniform noise) through the split-f16 and the exact-fp32 kernels of each model; a model "
                         "above the limit is pinned to the exact kernels (the timed region below runs on whatever `runs_on` says)")

    n_local, n_total = args.slices, args.slices * world
    hw, res = ((96, 80), (32, 32)) if emu else ((512, 512), (256, 256))
    vol = po.phantom(n_total, hw[0], hw[1], z0=rank * n_local, z1=(rank + 1) * n_local)

    def barrier():
        if dist is not None:
            dist.barrier()
        elif group is not None:
            group.barrier()

    def sync_all():
        eng.sync()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None or group is not None:
            barrier()
            if torch.cuda.is_available():
                torch.cuda.synchronize()

    if not use_dist:
        vd = eng.to_device(vol)
        od = eng.empty(vol.shape, np.uint8)

        def step():
            eng.apply_dev(0, vd, od, fill_slot=fill_slot, batch_size=args.batch)
    else:
        from lungmask_amd.pipeline import ShardedPipeline

        vt = torch.from_numpy(vol).to(dev)
        pipe = ShardedPipeline(eng, slot=0, batch_size=args.batch, resolution=res, dist=dist if dist is not None else group.nd, device=dev,
                               sharded_post=None if args.post == "auto" else args.post == "slab", fill_slot=fill_slot)
        args.post = "slab" if pipe.sharded_post else "gathered"  # (what the line reports)

        def step():
            pipe.apply_shard(vt, n_total)

    for _ in range(args.warmup):
        step()
    eng.profile(3)  # HIP events around the dominant kernel's launches only (events around every launch cost ~1 %)
    eng.profile_reset()
    def over_ranks(v):
        if dist is not None:
            tt = torch.tensor([v], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        return group.max(v) if group is not None else v

    sampler = None
    if rank == 0 and not emu:
        pci = None
        try:  # "dddd:bb:dd.f" of the HIP device (hipDeviceGetPCIBusId through the runtime torch already loaded)
            import ctypes

            buf = ctypes.create_string_buffer(64)
            hip = ctypes.CDLL("libamdhip64.so")
            if hip.hipDeviceGetPCIBusId(buf, 64, local_rank) == 0:
                pci = buf.value.decode()
        except Exception:  # noqa: BLE001
            pci = None
        sampler = ChipSampler(local_rank, pci_bus_id=pci)
    if sampler is not None:
        sampler.start()
    # The interpreter's cyclic garbage collector stays out of the measured passes: a generation-2 sweep over this process's object graph
    # (torch is loaded) is a 40-55 ms pause that lands, deterministically for a given command line, wherever the allocation counter
    # happens to run over -- round 6 found one inside the slab protocol's first stage of the breakdown step (profiles/r06y_*).  Collected
    # once here, disabled until the measurements are over; nothing the steps do depends on it.
    import gc

    gc.collect()
    gc.disable()
    # ---- the contract's timed region: exactly --steps steps between barrier + synchronise on both sides, max over ranks
    sync_all()
    t_first = t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    dt = over_ranks(time.perf_counter() - t0)
    stats = eng.profile_read()
    eng.profile(False)
    # ---- the same region again, --repeat more times (reported under `repetitions`; `value` / `ms_per_step` stay the first region's)
    rep_ms = [dt / args.steps * 1e3]
    for _ in range(max(args.repeat, 0)):
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync_all()
        rep_ms.append(over_ranks(time.perf_counter() - t0) / args.steps * 1e3)
    t_last = time.perf_counter()
    chip = None
    if sampler is not None:
        sampler.stop()
        chip = sampler.summary(t_first, t_last)
    post_info = eng.postprocess_info()
    if not use_dist and rank == 0:  # what the label volume of this workload looks like (post-processing cost is data dependent)
        fin = od.download()
        post_info["label_histogram_final_volume"] = np.bincount(fin.ravel(), minlength=n_classes).tolist()
        del fin
    # One extra, untimed pass with a single forward lane and events around EVERY launch: the per-stage table, and the
    # dominant kernel's duration when it has the GPU to itself (in the timed region two batches' kernels overlap on two
    # streams, which inflates per-kernel times).  Every rank runs it (the multi-GPU step contains collectives).
    solo = []
    if not emu:  # (the emulation test hook only drives the launch / sharding / reporting logic: one step is enough)
        eng.set_streams(1)
        eng.profile(True)
        eng.profile_reset()
        step()
        sync_all()
        solo = eng.profile_read()
        eng.profile(False)
        eng.set_streams(args.streams)
    if args.streams == 1:
        stats = stats or solo
    # N > 1 (and the forced world of one): where ONE step's time goes on the engine's stream -- sliced stages, every collective with
    # the bytes a rank contributes, post-processing, un-crop, output assembly -- and the host's share of the slab protocol; the host clock
    # behind a stream synchronisation at every boundary, maximum over the ranks per entry (every rank walks the same sequence)
    dist_breakdown = None
    if use_dist:
        pipe.want_breakdown = True
        sync_all()
        step()
        sync_all()
        bd = pipe.breakdown()
        pipe.want_breakdown = False
        if bd is not None:
            dist_breakdown = {k: round(over_ranks(v), 4) for k, v in bd.items() if isinstance(v, float)}
            dist_breakdown["collectives"] = [{"name": c["name"], "bytes_per_rank": c["bytes_per_rank"], "ms": round(over_ranks(c["ms"]), 4)} for c in bd["collectives"]]
            dist_breakdown["collectives_ms"] = round(sum(c["ms"] for c in dist_breakdown["collectives"]), 4)  # (of the per-entry maxima)
            dist_breakdown["segments_rank0"] = bd.get("segments")
            dist_breakdown["note"] = ("one extra untimed step with a stream synchronisation at every stage / collective boundary (host clock; the engine's stream is the one "
                                      "kernels AND collectives are enqueued on), max over ranks per entry; *_ms = the stage without its collectives; host_merge_ms = wall time inside lm_slab_step "
                                      "(the slab protocol's table merges incl. their waits for device data; 0 in the gathered form)")

    # numpy in -> numpy out: what a caller of the drop-in sees (SURVEY.md section 8d defines the metric on a host volume).  Two forms,
    # both over --steps steps like `value`, both reported beside `value`, never as `value` (the contract's timed region starts with
    # the volume resident in HBM): the C-ABI call lm_apply_host with a caller-owned, reused output array, and LMInferer.apply
    # itself -- the class a user of the reference calls -- with the reference's semantics (a FRESH result array per call) and
    # with reuse_output=True.
    host = lmi = None
    if not use_dist and args.host_steps > 0 and not emu:
        ref_labels = od.download()
        n_rep = 1 + max(args.repeat, 0)

        def timed_passes(fn):
            """--host-steps calls of fn, 1 + --repeat times back to back (like the timed region): ms per call of every pass."""
            out_ms = []
            for _ in range(n_rep):
                t0h = time.perf_counter()
                for _ in range(args.host_steps):
                    fn()
                out_ms.append((time.perf_counter() - t0h) / args.host_steps * 1e3)
            return out_ms

        def leg(ms):
            med = sorted(ms)[len(ms) // 2]
            return {"value": round(n_total / med * 1e3, 2), "ms_per_step": round(med, 3),
                    "passes_ms_per_step": [round(v, 3) for v in ms], "min": round(min(ms), 3), "median": round(med, 3), "max": round(max(ms), 3)}

        res_h = eng.apply(0, vol, fill_slot=fill_slot, batch_size=args.batch)
        eng.sync()
        host = leg(timed_passes(lambda: eng.apply(0, vol, fill_slot=fill_slot, batch_size=args.batch, out=res_h)))  # caller-owned output buffer, reused
        host.update({"unit": "slices/s", "steps": args.host_steps,
                     "note": "numpy int16 volume in pageable host memory -> uint8 numpy labels in a caller-owned, reused host array (lm_apply_host), PCIe copies included; "
                             f"`value` = the median of {n_rep} passes of --host-steps calls; identical labels: " + str(bool(np.array_equal(res_h, ref_labels)))})
        from lungmask_amd.mask import LMInferer

        lmi = {}
        for key, reuse in (("fresh_output_per_call", False), ("reuse_output", True)):
            inf = LMInferer(modelname="R231" if n_classes == 3 else "LTRCLobes", fillmodel="R231" if fill_classes else None, batch_size=args.batch,
                            device_id=local_rank, precision=args.precision, reuse_output=reuse, state_dict=sd, fill_state_dict=sd_fill, engine=eng)
            box = [inf.apply(vol)]
            box[0] = inf.apply(vol)  # (steady state: the second result block of the pool exists before the clock starts)

            def call(inf=inf, box=box):
                box[0] = inf.apply(vol)

            lmi[key] = leg(timed_passes(call))
            lmi[key]["identical_labels"] = bool(np.array_equal(box[0], ref_labels))
            if not reuse and hasattr(inf, "apply_async"):
                # volumes queued through apply_async: the copy-back of volume i and the copy-in of volume i + 1 run beside the hot path
                # of their neighbours ("multi-volume queueing"); two volumes in flight, results consumed in order
                # ONE continuously fed queue of 3 + n_rep * host_steps + 1 volumes, two in flight; a pass = host_steps consecutive results
                # (time between the arrival of result p * host_steps and of result (p + 1) * host_steps): the rate of the queue
                # itself -- filling it (the first volume's copy-in, ~5 ms) and draining it (the last copy-back) happen once per
                # stream of volumes, not once per volume, and lie outside the passes
                warm_q = 3  # results in front of the first pass (the other legs run two calls before their clock starts)
                total = warm_q + n_rep * args.host_steps + 1
                stamps, pend = [], []
                t_sub = time.perf_counter()
                for i in range(total + 1):
                    if i < total:
                        pend.append(inf.apply_async(vol))
                    if len(pend) > 2 or i == total:
                        while pend and (len(pend) > 2 or i == total):
                            box[0] = pend.pop(0).result()
                            stamps.append(time.perf_counter())
                ms = [(stamps[warm_q + (p + 1) * args.host_steps] - stamps[warm_q + p * args.host_steps]) / args.host_steps * 1e3 for p in range(n_rep)]
                lmi["async_pipelined"] = leg(ms)
                lmi["async_pipelined"]["identical_labels"] = bool(np.array_equal(box[0], ref_labels))
                lmi["async_pipelined"]["first_result_after_ms"] = round((stamps[0] - t_sub) * 1e3, 3)
            box[0] = None
        # the reference's cost model for comparison: a brand-new pageable numpy array per call (page faults + unmapping), results kept alive
        keep = []
        t0h = time.perf_counter()
        for _ in range(min(args.host_steps, 4)):
            keep.append(eng.apply(0, vol, fill_slot=fill_slot, batch_size=args.batch, out=np.empty(vol.shape, np.uint8)))
        dth = (time.perf_counter() - t0h) / max(len(keep), 1)
        lmi["new_pageable_array_per_call"] = {"value": round(n_total / dth, 2), "ms_per_step": round(dth * 1e3, 3), "steps": len(keep)}
        del keep
        lmi.update({"unit": "slices/s", "steps": args.host_steps,
                    "note": f"every leg: `value` = the median of {n_rep} back-to-back passes of --host-steps calls (passes_ms_per_step).  "
                            "lungmask_amd.LMInferer.apply(ndarray int16 [300,512,512]) -> ndarray uint8, the drop-in call itself (mask.py:212-232): "
                            "fresh_output_per_call = the reference's semantics, a result array of the caller's own per call -- a root array over a "
                            "page-locked block of the inferer's pool, given back by a finalizer when the result and all its views are gone (the loop "
                            "drops each result, so two blocks alternate); async_pipelined = LMInferer.apply_async, a continuously fed queue of volumes "
                            "(lm_pipe_*: copy-in of volume i + 1 and copy-back of volume i beside the hot path of volume i; two in flight; the rate between "
                            "consecutive results, queue fill / drain outside the passes); reuse_output = LMInferer(reuse_output=True), one pageable array for every "
                            "call; new_pageable_array_per_call = np.empty per call with the results kept alive (what a fresh allocation costs)"})

    gc.enable()
    if rank == 0:
        value = n_total * args.steps / dt
        h3 = args.precision == "split_f16"
        kname = "conv3x3_igemm_h3" if h3 else "conv3x3_igemm_f32"
        conv = next((s for s in stats if s["name"] == kname), None)
        roof = None
        overlapped = None
        if conv and conv["total_ms"] > 0 and solo and args.streams > 1:
            sc = next((s for s in solo if s["name"] == kname), None)
            if sc and sc["total_ms"] > 0:
                overlapped = {"achieved": round(conv["flops"] / (conv["total_ms"] * 1e-3) / 1e12, 2),
                              "avg_launch_ms": round(conv["total_ms"] / conv["launches"], 4), "launches": conv["launches"],
                              "note": "the same launches inside the timed region, where two batches' kernels share the GPU on two streams "
                                      "(per-kernel durations include that sharing)"}
                conv = sc
        if conv and conv["total_ms"] > 0:
            ach = conv["flops"] / (conv["total_ms"] * 1e-3) / 1e12
            peak = PEAK_F16_MFMA_TFLOPS if h3 else PEAK_F32_MFMA_TFLOPS
            traffic, traffic_src = pmc_traffic("conv_igemm_h3p<9" if h3 else "conv_igemm_f32<9")
            roof = {
                "bound": "mfma",
                "kernel": kname + (" (v_mfma_f32_32x32x16_f16, 3-product split-f16: 3 executed MFMA FLOPs per algorithmic FLOP)" if h3
                                   else " (v_mfma_f32_32x32x2_f32, exact fp32)"),
                "achieved": round(ach, 2),
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": round(ach / peak, 4),
                "executed_mfma_tflops": round(ach * (3 if h3 else 1), 2),
                "executed_frac_of_peak": round(ach * (3 if h3 else 1) / peak, 4),
                "sustained_mfma_ceiling_tflops": SUSTAINED_F16_MFMA_TFLOPS if h3 else None,
                "executed_frac_of_sustained_ceiling": round(ach * 3 / SUSTAINED_F16_MFMA_TFLOPS, 4) if h3 else None,
                "ceiling_note": "the data sheet peak assumes 2.4 GHz; on full-range operands the matrix pipes alone (no memory traffic) sustain "
                                "1737 TFLOP/s at ~1.6 GHz under the power budget (tools/ubench/mfma_power.hip, profiles/history/r02d_mfma_power.log)" if h3 else None,
                "traffic": traffic,
                "traffic_unit": "HBM bytes/launch, launch-weighted over the same 17 launches per batch as algorithmic_bytes_per_launch; NOT measured in this "
                                f"run: read from the committed rocprofv3 PMC summary {traffic_src} (separate FETCH_SIZE / WRITE_SIZE passes of "
                                "`bench.py --streams 1`, gfx950-corrected)" if traffic is not None else traffic_src,
                "algorithmic_bytes_per_launch": conv["bytes"] / conv["launches"],
                "launches": conv["launches"],
                "avg_launch_ms": round(conv["total_ms"] / conv["launches"], 4),
                "algorithmic_flop_per_launch": conv["flops"] / conv["launches"],
                "note": "aggregate over the 17 conv3x3 launches of every batch of one volume (rank 0), HIP events on the launching stream; "
                        + ("measured in a dedicated single-lane pass of the same step right after the timed region (kernel alone on the GPU; "
                           "`bench.py --streams 1` times the whole run that way and is the command profiled under profiles/)" if overlapped
                           else "measured over the timed region"),
                "overlapped": overlapped,
            }
        out = {
            "metric": "CT slices/sec (whole node), R231 512x512 volume" if args.config == 2 else f"CT slices/sec, {cfg_name}, 512x512 volume",
            "value": round(value, 2),
            "unit": "slices/s",
            "n_gpus": world,
            "collective_world": (dist.get_world_size() if dist is not None else group.world_size()) if use_dist else 1,
            "collectives": (("torch.distributed/" + ("gloo" if emu else "nccl(RCCL)")) if dist is not None else "lm_dist_* (engine-owned RCCL communicator)") if use_dist else None,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16x3-split (hi/lo f16 operands, f32 accumulate; fp32-class)" if h3 else "f32",
            "data": "synthetic" if not emu else "EMULATION TEST HOOK (LM_BENCH_EMU=1): g++ emulation of the kernels on the CPU, tiny volume -- not a measurement",
            "config": {
                "workload": f"BASELINE.json configs[{args.config - 1}]: {cfg_name}, {hw[0]}x{hw[1]}x{n_local} int16 HU phantom per GPU ({n_total} slices total), "
                            f"batchsize={args.batch}, LMInferer.apply device-resident (pre + forward + argmax + 3-D post + un-crop"
                            + (" + second model + fusion + full-resolution post" if fill_slot >= 0 else "") + ")",
                "weights": weights,
                "parallelism": "single GPU" if not use_dist else (f"slice-sharded x{world}: slab-local post-processing + 6 small RCCL table all-gathers, 1 all-gather of the output" if args.post == "slab" else f"slice-sharded x{world}: RCCL all-gather of the labels, redundant whole-volume post-processing, all-gather of the output"),
            },
            "repetitions": {"ms_per_step": [round(v, 3) for v in rep_ms], "min": round(min(rep_ms), 3), "median": round(sorted(rep_ms)[len(rep_ms) // 2], 3),
                            "max": round(max(rep_ms), 3), "steps_each": args.steps,
                            "note": "the timed region repeated back to back; the first entry is the contract's region (`value`, `ms_per_step`)"},
            "chip_during_timed_region": chip,
            "end_to_end_tflops": round(value * (FLOP_PER_SLICE[n_classes] + (FLOP_PER_SLICE[fill_classes] if fill_classes else 0.0)) / 1e12, 2),
            "value_definition": "`value` = device-resident (contract: inputs in HBM when the timed region starts); value_host_to_host / "
                                "value_lminferer_apply = the same step numpy -> numpy over the same number of steps (SURVEY 8d's definition)",
            "value_host_to_host": host,
            "value_lminferer_apply": lmi,
            "roofline": roof,
            "stages_ms_per_step": {s["name"]: round(s["total_ms"], 3) for s in solo},
            "stages_note": "HIP-event time per kernel kind of ONE step, from the untimed single-lane pass after the timed region (in that pass the whole "
                           "volume is pre-processed on the main stream; in the timed region the tail's pre-processing runs beside the head's forward).  "
                           "There is no first_conv / head_argmax row: the first conv runs inside the loader of down_path.0's second conv and the head "
                           "inside the last conv's epilogue (lm_set_fusion, default 11); both are part of conv3x3_igemm_h3",
            "postprocessing": post_info,
            "dist_breakdown": dist_breakdown,
            "accuracy_guard": guard,
        }
        if world == 1 and not args.no_cpu_baseline and not emu:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, sd)
            out["cpu_baseline_config1"] = cpu_baseline_config1(sd)
    # The JSON line must be the LAST line of the job's stdout: RCCL prints a version banner through C stdio, which is
    # block-buffered on a pipe and would otherwise be flushed at exit, after Python's own output -- on every rank.
    import ctypes

    libc = ctypes.CDLL(None)
    if dist is not None:
        libc.fflush(None)
        dist.barrier()
        dist.destroy_process_group()
    elif group is not None:
        libc.fflush(None)
        group.barrier()
        group.destroy()
    libc.fflush(None)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
