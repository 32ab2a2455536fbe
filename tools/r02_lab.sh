TAG=${1:-r02v}
timeout 900 python -m pytest tests/test_gpu_prepost.py tests/test_gpu_apply.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -3
for c in 2 4; do timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_c$c.json; python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench_c$c.json"))
print("config $c:", d["value"], d["ms_per_step"], {k:v for k,v in d["stages_ms_per_step"].items() if k.startswith("post")})
PY
done
