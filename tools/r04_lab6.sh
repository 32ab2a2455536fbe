mkdir -p gpurun_out/r04f
O=gpurun_out/r04f
LM_H3_FUSE_FIRST=0 timeout 200 python tools/layer_time_ab.py H256_Ci64_Co64 lungmask_amd/liblungmask_hip.so 2>&1 | grep -v amdgpu.ids > $O/fc_unfused.log; cat $O/fc_unfused.log
timeout 300 python tools/layer_time_ab.py H256_Ci64_Co64 lungmask_amd/liblungmask_hip.so lungmask_amd/_ab/lib_fcabl1.so lungmask_amd/_ab/lib_fcabl2.so 2>&1 | grep -v amdgpu.ids > $O/fc_ablation.log; cat $O/fc_ablation.log
