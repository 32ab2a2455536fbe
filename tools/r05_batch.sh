mkdir -p gpurun_out
timeout 900 python tools/stress.py 150 2>&1 | grep -v amdgpu.ids > gpurun_out/r05z_stress.log; cat gpurun_out/r05z_stress.log
