mkdir -p gpurun_out/r04j
O=gpurun_out/r04j
LM_STREAM_OUT_MB=1000000 timeout 300 python tools/nn_perf_ab.py lungmask_amd/liblungmask_hip.so lungmask_amd/_ab/lib_epid.so 2>&1 | grep -v amdgpu.ids > $O/nn_perf_ab_epi_direct_no_nt.log; cat $O/nn_perf_ab_epi_direct_no_nt.log
