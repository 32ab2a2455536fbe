# round-4 lab session 2: pipelined first-conv producer (six micro-steps), slot placement variants
mkdir -p gpurun_out/r04b
O=gpurun_out/r04b
timeout 300 python tools/ab_fusion.py 0 1 2>&1 | grep -v amdgpu.ids > $O/ab_fusion.log; head -6 $O/ab_fusion.log; grep -E "H256_Ci64_Co64|first_conv|sum of" $O/ab_fusion.log
timeout 400 python tools/ab_forward.py lungmask_amd/liblungmask_hip.so lungmask_amd/_ab/lib_s0.so lungmask_amd/_ab/lib_s2.so 2>&1 | grep -v amdgpu.ids > $O/ab_slots.log; cat $O/ab_slots.log
