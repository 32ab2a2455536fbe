T=${1:-r03x}; mkdir -p gpurun_out/$T
python tools/nn_perf_ab.py lungmask_amd/_ab/liblungmask_hip_base.so lungmask_amd/liblungmask_hip.so 2>&1 | grep -v amdgpu.ids | grep -E "ms per|upsample|first|sum" > gpurun_out/$T/first_conv.log; cat gpurun_out/$T/first_conv.log
python tools/ab_forward.py lungmask_amd/_ab/liblungmask_hip_base.so lungmask_amd/liblungmask_hip.so 2>&1 | grep -v amdgpu.ids > gpurun_out/$T/ab.log; cat gpurun_out/$T/ab.log
