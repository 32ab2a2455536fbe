# r03i: split via v_fma_mix (bit-equivalence ubench), compile-time BN in the persistent kernel: parity + A/B against the r03h tree
T=r03i; mkdir -p gpurun_out/$T
./tools/ubench/split_mix > gpurun_out/$T/split_mix.log 2>&1; tail -2 gpurun_out/$T/split_mix.log
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_apply.py -m gpu -x -q > gpurun_out/$T/pytest.log 2>&1; grep -E "passed|failed|error" gpurun_out/$T/pytest.log | tail -3
python tools/ab_forward.py lungmask_amd/_ab/liblungmask_hip_base.so lungmask_amd/liblungmask_hip.so 2>&1 | grep -v amdgpu.ids > gpurun_out/$T/ab.log; cat gpurun_out/$T/ab.log
