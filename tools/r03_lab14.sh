T=${1:-r03q}; mkdir -p gpurun_out/$T
for mb in 1e9 128 60 10; do echo "LM_STREAM_OUT_MB=$mb"; LM_STREAM_OUT_MB=$mb python tools/ab_forward.py lungmask_amd/liblungmask_hip.so 2>&1 | grep -v amdgpu.ids | head -1; done > gpurun_out/$T/stream_out.log 2>&1; cat gpurun_out/$T/stream_out.log
for mb in 1e9 128 60; do echo "LM_STREAM_OUT_MB=$mb"; LM_STREAM_OUT_MB=$mb python tools/nn_perf_ab.py lungmask_amd/liblungmask_hip.so 2>&1 | grep -v amdgpu.ids | grep -E "H256|H128|sum"; done > gpurun_out/$T/stream_layers.log 2>&1; cat gpurun_out/$T/stream_layers.log
