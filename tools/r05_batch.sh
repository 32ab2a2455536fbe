mkdir -p gpurun_out
LIBS="lungmask_amd/_ab/lib_nodpp.so lungmask_amd/liblungmask_hip.so"
timeout 400 python tools/ab_forward.py $LIBS 2>&1 | grep -v amdgpu.ids > gpurun_out/r05b_ab_forward.log; cat gpurun_out/r05b_ab_forward.log
timeout 400 python tools/nn_perf_ab.py $LIBS 2>&1 | grep -v amdgpu.ids > gpurun_out/r05b_nn_perf_ab.log; grep -E "H256|sum|ms per" gpurun_out/r05b_nn_perf_ab.log
timeout 300 python tools/host_boundary.py 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/r05b_host_boundary.log; cat gpurun_out/r05b_host_boundary.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/r05b_bench_err.log | tail -1 > gpurun_out/r05b_bench.json; cut -c1-300 gpurun_out/r05b_bench.json
timeout 900 python -m pytest tests/test_gpu_apply.py tests/test_gpu_forward.py -m gpu -x -q -rs > gpurun_out/r05b_pytest_gpu.log 2>&1; tail -4 gpurun_out/r05b_pytest_gpu.log
