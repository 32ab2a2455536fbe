mkdir -p gpurun_out/r04i
O=gpurun_out/r04i
timeout 400 python tools/ab_forward.py lungmask_amd/liblungmask_hip.so lungmask_amd/_ab/lib_epid.so 2>&1 | grep -v amdgpu.ids > $O/ab_epi_direct.log; cat $O/ab_epi_direct.log
timeout 300 python tools/nn_perf_ab.py lungmask_amd/liblungmask_hip.so lungmask_amd/_ab/lib_epid.so 2>&1 | grep -v amdgpu.ids > $O/nn_perf_ab_epi_direct.log; cat $O/nn_perf_ab_epi_direct.log
