for g in 0 248 240 224 208 0; do echo "LM_H3_GRID=$g"; LM_H3_GRID=$g python tools/ab_forward.py lungmask_amd/liblungmask_hip.so 2>&1 | grep -v amdgpu.ids | head -1; done
