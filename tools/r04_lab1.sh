# round-4 lab session 1: naive fused first conv (correctness + A/B), LM_LO_BITS variants (time + precision), first bench line
mkdir -p gpurun_out/r04a
O=gpurun_out/r04a
timeout 900 python -m pytest tests/test_gpu_forward.py -x -q -rs > $O/pytest_forward.log 2>&1; tail -3 $O/pytest_forward.log
timeout 300 python tools/ab_fusion.py 0 1 2>&1 | grep -v amdgpu.ids > $O/ab_fusion.log; head -6 $O/ab_fusion.log
timeout 400 python tools/ab_forward.py lungmask_amd/liblungmask_hip.so lungmask_amd/_ab/lib_lo8.so lungmask_amd/_ab/lib_lo6.so 2>&1 | grep -v amdgpu.ids > $O/ab_lobits.log; cat $O/ab_lobits.log
timeout 600 python tests/precision_sweep.py lungmask_amd/liblungmask_hip.so lungmask_amd/_ab/lib_lo8.so lungmask_amd/_ab/lib_lo6.so 2>&1 | grep -v amdgpu.ids > $O/precision_sweep.log; cat $O/precision_sweep.log
timeout 300 python bench.py --steps 10 2>$O/bench_err.log | tail -1 > $O/bench.json; cut -c1-300 $O/bench.json
