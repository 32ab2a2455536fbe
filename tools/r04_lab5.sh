# round-4 lab session 5: fp32 head (fused, with log-probs), chunked fp32 fall-back, pinned result blocks: parity suites + precision sweep + bench
mkdir -p gpurun_out/r04e
O=gpurun_out/r04e
timeout 1200 python -m pytest tests/test_gpu_forward.py tests/test_gpu_apply.py -x -q -rs -s > $O/pytest_fwd_apply.log 2>&1; tail -3 $O/pytest_fwd_apply.log
timeout 600 python tests/precision_sweep.py 2>&1 | grep -v amdgpu.ids > $O/precision_sweep.log; cat $O/precision_sweep.log
timeout 300 python bench.py --steps 10 2>$O/bench_err.log | tail -1 > $O/bench.json; python -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'],d['ms_per_step'],d['value_host_to_host']['value'],d['value_lminferer_apply'])"
LM_HOST_TIMING=1 timeout 200 python tools/host_boundary.py 2>&1 | grep -v amdgpu.ids | tail -6 > $O/host_boundary.log; cat $O/host_boundary.log
