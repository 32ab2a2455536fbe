TAG=${1:-r02r}
timeout 120 python tools/nn_perf.py 20 5 split_f16 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_nn_perf.log; grep "B=20\|first\|upsample" gpurun_out/${TAG}_nn_perf.log
timeout 600 python -m pytest tests/test_gpu_forward.py -m gpu -q -x 2>&1 | tail -2
timeout 300 python tools/ab_forward.py lungmask_amd/_ab/liblungmask_hip_r02a.so lungmask_amd/liblungmask_hip.so 2>&1 | grep "two lanes"
