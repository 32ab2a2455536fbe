for g in 0 16 32 48 0; do echo "LM_BW_CUS=$g"; LM_BW_CUS=$g python tools/ab_forward.py lungmask_amd/liblungmask_hip.so 2>&1 | grep -v amdgpu.ids | head -1; done
