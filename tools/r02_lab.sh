# Round-2 check run: distributed-path tests on the GPU (stream-ordered collectives, 2 ranks on 1 GPU), host boundary timing
TAG=${1:-r02f}
timeout 900 python -m pytest tests/test_gpu_apply.py -m gpu -q -s -k "rccl or sharded or lminferer or cli or force_cpu" > gpurun_out/${TAG}_pytest_dist.log 2>&1; grep -E "passed|failed|SKIP|skipped" gpurun_out/${TAG}_pytest_dist.log | tail -3; grep -i "RCCL2\|refuses" gpurun_out/${TAG}_pytest_dist.log | head -5
timeout 300 python -m pytest tests/test_gpu_apply.py -m gpu -q -rs -k "two_ranks" 2>&1 | tail -5
timeout 300 python bench.py --no-cpu-baseline 2>gpurun_out/${TAG}_bench_err.log | tail -1 > gpurun_out/${TAG}_bench.json; python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench.json"))
print("config 2:", d["value"], d["ms_per_step"], "host:", d.get("value_host_to_host"))
PY
LM_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 3 --no-cpu-baseline 2>gpurun_out/${TAG}_bench_dist_err.log | tail -1 > gpurun_out/${TAG}_bench_forced_dist.json; cut -c1-200 gpurun_out/${TAG}_bench_forced_dist.json
