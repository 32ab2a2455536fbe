T=${1:-r03m}; mkdir -p gpurun_out/$T
LM_LAB_VERIFY=1 ./tools/ubench/conv_lab_plain 4 1 2>&1 | grep verify | cut -c1-120 > gpurun_out/$T/verify.log; cat gpurun_out/$T/verify.log
python tools/ab_forward.py lungmask_amd/_ab/liblungmask_hip_base.so lungmask_amd/liblungmask_hip.so 2>&1 | grep -v amdgpu.ids > gpurun_out/$T/ab.log; cat gpurun_out/$T/ab.log
python tools/nn_perf_ab.py lungmask_amd/_ab/liblungmask_hip_base.so lungmask_amd/liblungmask_hip.so 2>&1 | grep -v amdgpu.ids > gpurun_out/$T/layers.log; cat gpurun_out/$T/layers.log
LM_TL_DIR=gpurun_out/$T ./tools/ubench/conv_lab_tl 20 4 > gpurun_out/$T/conv_lab_tl.log 2>&1
python tools/tl_epi.py gpurun_out/$T > gpurun_out/$T/epi_phases.txt; grep "wave4" gpurun_out/$T/epi_phases.txt
timeout 600 python -m pytest tests/test_gpu_forward.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -3
