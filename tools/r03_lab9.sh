T=${1:-r03k}; mkdir -p gpurun_out/$T
python tools/ab_forward.py lungmask_amd/_ab/liblungmask_hip_base.so lungmask_amd/liblungmask_hip.so 2>&1 | grep -v amdgpu.ids > gpurun_out/$T/ab.log; cat gpurun_out/$T/ab.log
for sb in 10 5 4; do echo "LM_L0_SUB=$sb"; LM_L0_SUB=$sb python tools/ab_forward.py lungmask_amd/liblungmask_hip.so 2>&1 | grep -v amdgpu.ids | head -1; done > gpurun_out/$T/l0sub.log 2>&1; cat gpurun_out/$T/l0sub.log
python tools/nn_perf_ab.py lungmask_amd/_ab/liblungmask_hip_base.so lungmask_amd/liblungmask_hip.so 2>&1 | grep -v amdgpu.ids > gpurun_out/$T/layers.log; cat gpurun_out/$T/layers.log
LM_L0_SUB=5 python tools/nn_perf_ab.py lungmask_amd/liblungmask_hip.so 2>&1 | grep -v amdgpu.ids > gpurun_out/$T/layers_sub5.log; cat gpurun_out/$T/layers_sub5.log
