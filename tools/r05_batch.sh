mkdir -p gpurun_out
for T in 0 200 400 1300 0 200; do
  echo "== LM_H3_NT1_MAXITEMS=$T"
  LM_H3_NT1_MAXITEMS=$T timeout 200 python tools/ab_forward.py lungmask_amd/liblungmask_hip.so 2>&1 | grep -v amdgpu.ids | grep "two lanes"
  LM_H3_NT1_MAXITEMS=$T timeout 200 python tools/nn_perf.py 20 5 split_f16 2>&1 | grep -E "^B=|conv1x1"
done > gpurun_out/r05g_nt1_ab.log 2>&1; cat gpurun_out/r05g_nt1_ab.log
timeout 900 python -m pytest tests/test_gpu_forward.py -m gpu -x -q -rs > gpurun_out/r05g_pytest_gpu_forward.log 2>&1; grep -E "passed|failed|error" gpurun_out/r05g_pytest_gpu_forward.log | tail -2
