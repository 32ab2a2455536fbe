T=${1:-r03n}; mkdir -p gpurun_out/$T
timeout 1200 python -m pytest tests/test_gpu_prepost.py tests/test_gpu_apply.py -m gpu -x -q > gpurun_out/$T/pytest.log 2>&1; grep -E "passed|failed|error" gpurun_out/$T/pytest.log | tail -3
timeout 300 python bench.py --steps 10 2>gpurun_out/$T/bench_err.log | tail -1 > gpurun_out/$T/bench.json; python - <<'PY'
import json,sys
d=json.loads(open('gpurun_out/'+sys.argv[1] if len(sys.argv)>1 else 'gpurun_out/r03n/bench.json').read()) if False else json.loads(open('gpurun_out/r03n/bench.json').read())
print(d['value'], d['ms_per_step'], d['value_host_to_host'], d['value_lminferer_apply'])
print(d['stages_ms_per_step'])
PY
