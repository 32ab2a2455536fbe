T=${1:-r03o}; mkdir -p gpurun_out/$T
python tools/ab_forward.py lungmask_amd/_ab/liblungmask_hip_base.so lungmask_amd/liblungmask_hip.so 2>&1 | grep -v amdgpu.ids > gpurun_out/$T/ab.log; cat gpurun_out/$T/ab.log
python tools/nn_perf_ab.py lungmask_amd/_ab/liblungmask_hip_base.so lungmask_amd/liblungmask_hip.so 2>&1 | grep -v amdgpu.ids | grep -E "upsample|first|sum|ms per" > gpurun_out/$T/layers.log; cat gpurun_out/$T/layers.log
python tools/stage_times.py 2>&1 | grep -v amdgpu.ids | tail -25 > gpurun_out/$T/stage_times.log; cat gpurun_out/$T/stage_times.log
