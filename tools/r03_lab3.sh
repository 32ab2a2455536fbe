T=${1:-r03c}
mkdir -p gpurun_out/$T
export LM_TL_DIR=gpurun_out/$T
timeout 600 python -m pytest tests/test_gpu_forward.py -m gpu -x -q > gpurun_out/$T/pytest_forward.log 2>&1; tail -3 gpurun_out/$T/pytest_forward.log
./tools/ubench/conv_lab_plain 20 6 > gpurun_out/$T/conv_lab_new.log 2>&1
./tools/ubench/conv_lab_tl 20 4 > gpurun_out/$T/conv_lab_tl.log 2>&1
python tools/ab_forward.py lungmask_amd/_ab/liblungmask_hip_r02.so lungmask_amd/liblungmask_hip.so 2>&1 | grep -v amdgpu.ids > gpurun_out/$T/ab.log
awk '{print $1, $2, $3}' gpurun_out/$T/conv_lab_new.log
cat gpurun_out/$T/ab.log
