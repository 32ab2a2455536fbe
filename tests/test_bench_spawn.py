"""CPU suite: `bench.py --gpus N` launches its own ranks and never reports an N-GPU line from fewer ranks.

The spawn / sharding / reporting logic is driven end to end with the LM_BENCH_EMU=1 test hook of bench.py (g++ emulation of
the kernels, gloo, a tiny volume); the refusals are checked without it."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, drop=("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "LM_BENCH_EMU")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)


def test_gpus_2_refused_on_a_box_with_fewer_gpus():
    import torch

    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have >= 2:
        import pytest

        pytest.skip("this machine has two GPUs")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert r.returncode != 0
    assert f"only {have} GPU(s) visible" in r.stderr and "refusing" in r.stderr
    assert r.stdout.strip() == ""  # no JSON line that could be mistaken for a measurement


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "2"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in r.stderr
    r = _run(["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 1 but WORLD_SIZE=2" in r.stderr


def _check_line(r, n_local, repeat=0):
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 only
    out = json.loads(lines[-1])
    assert out["n_gpus"] == 2 and out["collective_world"] == 2 and out["collectives"].startswith("torch.distributed/gloo")
    assert out["steps"] == 1 and out["scaling"] == "weak" and out["value"] > 0
    assert "EMULATION TEST HOOK" in out["data"]
    assert f"96x80x{n_local} int16 HU phantom per GPU ({2 * n_local} slices total)" in out["config"]["workload"]
    assert "cpu_baseline" not in out
    # the timed region repeated: the first entry is the contract's region
    rep = out["repetitions"]
    assert len(rep["ms_per_step"]) == 1 + repeat and rep["ms_per_step"][0] == out["ms_per_step"] and rep["min"] <= rep["median"] <= rep["max"]
    assert out["chip_during_timed_region"] is None  # (no chip to sample under emulation)
    # the N > 1 line explains itself: stages, every collective with its bytes, the host's share of the slab protocol
    bd = out["dist_breakdown"]
    assert bd["forward_ms"] > 0 and bd["post_ms"] >= 0 and bd["uncrop_ms"] >= 0 and "host_merge_ms" in bd
    names = [c["name"] for c in bd["collectives"]]
    assert names[0] == "label_shards" and names[-1] == "output_shards" and all(c["bytes_per_rank"] > 0 and c["ms"] >= 0 for c in bd["collectives"])
    assert abs(bd["collectives_ms"] - sum(c["ms"] for c in bd["collectives"])) < 1e-2


def test_self_spawned_two_ranks_over_gloo_and_the_emulator():
    from lungmask_amd.build import build_emu

    build_emu()  # once, before two ranks race to build it
    r = _run(["--gpus", "2", "--slices", "2", "--steps", "1", "--warmup", "0", "--batch", "2", "--no-cpu-baseline", "--repeat", "1"], {"LM_BENCH_EMU": "1"})
    _check_line(r, 2, repeat=1)


def test_two_ranks_under_torch_distributed_run():
    """The way the driver launches N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher's environment)."""
    import socket

    from lungmask_amd.build import build_emu

    build_emu()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(LM_BENCH_EMU="1", OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2", "--slices", "1", "--steps", "1", "--warmup", "0", "--batch", "2",
                        "--no-cpu-baseline", "--repeat", "0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    _check_line(r, 1)
