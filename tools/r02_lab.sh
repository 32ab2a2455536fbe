TAG=${1:-r02x}
timeout 1200 python -m pytest tests/test_gpu_prepost.py tests/test_gpu_apply.py tests/test_gpu_fullsize.py -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1; grep -E "passed|failed" gpurun_out/${TAG}_pytest.log
for c in 2 3 4; do timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_c$c.json; python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench_c$c.json"))
print("config $c:", d["value"], d["ms_per_step"], "host", d["value_host_to_host"]["value"], {k:v for k,v in d["stages_ms_per_step"].items() if k.startswith("post")})
PY
done
