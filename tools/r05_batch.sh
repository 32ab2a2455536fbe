mkdir -p gpurun_out
timeout 300 python tools/host_boundary.py 2>&1 | grep -v amdgpu.ids | tail -13 > gpurun_out/r05e_host_boundary.log; tail -8 gpurun_out/r05e_host_boundary.log
timeout 300 python tools/post_timing.py 2>&1 | grep -E "lm_postprocess|info" | tail -3 > gpurun_out/r05e_post_timing.log; cat gpurun_out/r05e_post_timing.log
timeout 400 python bench.py --steps 20 --warmup 5 2>gpurun_out/r05e_bench_err.log | tail -1 > gpurun_out/r05e_bench.json; cut -c1-200 gpurun_out/r05e_bench.json
timeout 1500 python -m pytest tests/test_gpu_apply.py tests/test_gpu_fullsize.py -m gpu -x -q -rs > gpurun_out/r05e_pytest_gpu.log 2>&1; grep -E "passed|failed|error" gpurun_out/r05e_pytest_gpu.log | tail -3
