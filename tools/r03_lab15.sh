T=${1:-r03s}; mkdir -p gpurun_out/$T
timeout 1200 python -m pytest tests/test_gpu_apply.py tests/test_gpu_prepost.py -m gpu -x -q > gpurun_out/$T/pytest.log 2>&1; grep -E "passed|failed|error" gpurun_out/$T/pytest.log | tail -3
python bench.py --steps 10 --no-cpu-baseline 2>gpurun_out/$T/bench_err.log | tail -1 > gpurun_out/$T/bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03s/bench.json').read())
print(d['value'], d['ms_per_step'], d['value_host_to_host']['value'], d['value_host_to_host']['ms_per_step'], d['value_lminferer_apply']['fresh_output_per_call'], d['value_lminferer_apply']['reuse_output'])
PY
LM_HOST_TIMING=1 python tools/host_boundary.py 2>&1 | grep -v amdgpu.ids | tail -4
python tools/stress.py 40 2>&1 | grep -v amdgpu.ids | tail -4
