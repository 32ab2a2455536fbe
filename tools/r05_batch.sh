mkdir -p gpurun_out
(LM_POST_GRAPH=0 timeout 300 python tools/post_timing.py 2>&1 | grep -E "lm_postprocess|info" | tail -3; timeout 300 python tools/post_timing.py 2>&1 | grep -E "lm_postprocess|info" | tail -3) > gpurun_out/r05d_post_timing.log; cat gpurun_out/r05d_post_timing.log
timeout 300 python tools/step_timeline.py 2>&1 | grep -v amdgpu.ids | tail -7 > gpurun_out/r05d_step_timeline.log; cat gpurun_out/r05d_step_timeline.log
timeout 600 python tools/slab_timing.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05d_slab_timing.log; cat gpurun_out/r05d_slab_timing.log
timeout 900 python -m pytest tests/test_gpu_prepost.py tests/test_gpu_fullsize.py -m gpu -x -q -rs > gpurun_out/r05d_pytest_gpu.log 2>&1; grep -E "passed|failed|error" gpurun_out/r05d_pytest_gpu.log | tail -3
