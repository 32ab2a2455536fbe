mkdir -p gpurun_out
(LM_POST_SIDE=0 timeout 300 python tools/post_timing.py 2>&1 | grep -E "lm_postprocess|info" | tail -3; timeout 300 python tools/post_timing.py 2>&1 | grep -E "lm_postprocess|info" | tail -3) > gpurun_out/r05f_post_timing_side_stream.log; cat gpurun_out/r05f_post_timing_side_stream.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --host-steps 0 2>gpurun_out/r05f_bench_err.log | tail -1 > gpurun_out/r05f_bench.json; cut -c1-200 gpurun_out/r05f_bench.json
LM_POST_SIDE=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --host-steps 0 2>>gpurun_out/r05f_bench_err.log | tail -1 > gpurun_out/r05f_bench_no_side.json; cut -c1-200 gpurun_out/r05f_bench_no_side.json
timeout 900 python -m pytest tests/test_gpu_prepost.py tests/test_gpu_fullsize.py -m gpu -x -q -rs > gpurun_out/r05f_pytest_gpu.log 2>&1; grep -E "passed|failed|error" gpurun_out/r05f_pytest_gpu.log | tail -3
timeout 300 python tools/stress.py 60 2>&1 | grep -v amdgpu.ids | tail -4
