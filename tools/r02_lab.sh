TAG=${1:-r02k}
timeout 300 python tools/ab_forward.py lungmask_amd/_ab/liblungmask_hip_r02a.so lungmask_amd/liblungmask_hip.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_ab.log
echo "LM_H3_DEFER_SHIFT=0"; LM_H3_DEFER_SHIFT=0 timeout 120 python tools/ab_forward.py lungmask_amd/liblungmask_hip.so 2>&1 | grep "two lanes" | head -2
timeout 900 python -m pytest tests/test_gpu_forward.py -m gpu -q -x 2>&1 | tail -2
