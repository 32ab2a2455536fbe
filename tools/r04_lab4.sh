mkdir -p gpurun_out/r04d
O=gpurun_out/r04d
timeout 300 python tools/overlap_probe.py 2>&1 | grep -v amdgpu.ids > $O/overlap_probe.log; cat $O/overlap_probe.log
