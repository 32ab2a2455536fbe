TAG=${1:-r02o}
timeout 900 python -m pytest tests/test_gpu_forward.py -m gpu -q -x 2>&1 | tail -3
echo "== conv_lab 16x16x32"; timeout 120 tools/ubench/conv_lab > gpurun_out/${TAG}_conv_lab_mma16.log 2>&1; cat gpurun_out/${TAG}_conv_lab_mma16.log
echo "== conv_lab 32x32x16"; LM_H3_MMA=32 timeout 120 tools/ubench/conv_lab > gpurun_out/${TAG}_conv_lab_mma32.log 2>&1; cut -c1-44 gpurun_out/${TAG}_conv_lab_mma32.log
for m in 16 32 16 32; do echo "LM_H3_MMA=$m"; LM_H3_MMA=$m timeout 120 python tools/ab_forward.py lungmask_amd/liblungmask_hip.so 2>&1 | grep "two lanes" | head -1; done
