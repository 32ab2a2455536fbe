T=${1:-r03v}; mkdir -p gpurun_out/$T
for k in 2 4 8 16; do echo "LM_BW_WGS_PER_CU=$k"; LM_BW_WGS_PER_CU=$k python tools/nn_perf_ab.py lungmask_amd/_ab/liblungmask_hip_base.so lungmask_amd/liblungmask_hip.so 2>&1 | grep -v amdgpu.ids | grep -E "upsample|first|sum"; done > gpurun_out/$T/bw_grid.log 2>&1; cat gpurun_out/$T/bw_grid.log
python tools/ab_forward.py lungmask_amd/_ab/liblungmask_hip_base.so lungmask_amd/liblungmask_hip.so 2>&1 | grep -v amdgpu.ids > gpurun_out/$T/ab.log; cat gpurun_out/$T/ab.log
