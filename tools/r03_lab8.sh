T=${1:-r03j}; mkdir -p gpurun_out/$T
python tools/ab_forward.py lungmask_amd/_ab/liblungmask_hip_base.so lungmask_amd/liblungmask_hip.so 2>&1 | grep -v amdgpu.ids > gpurun_out/$T/ab.log; cat gpurun_out/$T/ab.log
python tools/nn_perf_ab.py lungmask_amd/_ab/liblungmask_hip_base.so lungmask_amd/liblungmask_hip.so 2>&1 | grep -v amdgpu.ids > gpurun_out/$T/layers.log; cat gpurun_out/$T/layers.log
timeout 600 python -m pytest tests/test_gpu_forward.py -m gpu -x -q -s 2>&1 | grep -E "head std|passed|failed" | tail -12
