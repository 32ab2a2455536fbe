# Round-2 check run: GPU parity suite (incl. the full-size configurations and the f16 range guard) and the three bench configurations
TAG=${1:-r02e}
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/${TAG}_pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/${TAG}_pytest_gpu.log | tail -2; grep -E "differing voxels|forward mismatches|out-of-f16|oracle pre" gpurun_out/${TAG}_pytest_gpu.log
for c in 2 3 4; do timeout 300 python bench.py --config $c 2>gpurun_out/${TAG}_bench_c${c}_err.log | tail -1 > gpurun_out/${TAG}_bench_c$c.json; python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench_c$c.json"))
print("config $c:", d["value"], d["ms_per_step"], "host:", d.get("value_host_to_host"))
PY
done
