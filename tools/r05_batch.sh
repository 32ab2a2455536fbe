mkdir -p gpurun_out
(LM_POST_GRAPH=0 timeout 300 python tools/post_timing.py 2>&1 | grep -E "lm_postprocess|info" | tail -4; timeout 300 python tools/post_timing.py 2>&1 | grep -E "lm_postprocess|info" | tail -4) > gpurun_out/r05c_post_timing.log; cat gpurun_out/r05c_post_timing.log
timeout 300 python tools/step_timeline.py 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/r05c_step_timeline.log; tail -6 gpurun_out/r05c_step_timeline.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --host-steps 0 2>gpurun_out/r05c_bench_err.log | tail -1 > gpurun_out/r05c_bench.json; cut -c1-200 gpurun_out/r05c_bench.json
LM_POST_GRAPH=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --host-steps 0 2>>gpurun_out/r05c_bench_err.log | tail -1 > gpurun_out/r05c_bench_voxel_form.json; cut -c1-200 gpurun_out/r05c_bench_voxel_form.json
timeout 300 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --host-steps 0 2>>gpurun_out/r05c_bench_err.log | tail -1 > gpurun_out/r05c_bench_config4.json; cut -c1-200 gpurun_out/r05c_bench_config4.json
LM_POST_GRAPH=0 timeout 300 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --host-steps 0 2>>gpurun_out/r05c_bench_err.log | tail -1 > gpurun_out/r05c_bench_config4_voxel_form.json; cut -c1-200 gpurun_out/r05c_bench_config4_voxel_form.json
timeout 1200 python -m pytest tests/test_gpu_prepost.py tests/test_gpu_fullsize.py -m gpu -x -q -rs > gpurun_out/r05c_pytest_gpu.log 2>&1; grep -E "passed|failed|error" gpurun_out/r05c_pytest_gpu.log | tail -3
