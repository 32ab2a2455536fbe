mkdir -p gpurun_out
LIBS="lungmask_amd/liblungmask_hip.so lungmask_amd/_ab/lib_scalarfma.so lungmask_amd/_ab/lib_fold.so lungmask_amd/_ab/lib_foldscalar.so"
timeout 400 python tools/ab_forward.py $LIBS 2>&1 | grep -v amdgpu.ids > gpurun_out/r05a_ab_forward.log; cat gpurun_out/r05a_ab_forward.log
timeout 400 python tools/nn_perf_ab.py $LIBS 2>&1 | grep -v amdgpu.ids > gpurun_out/r05a_nn_perf_ab.log; tail -25 gpurun_out/r05a_nn_perf_ab.log
timeout 300 python tests/precision_check.py lungmask_amd/liblungmask_hip.so lungmask_amd/_ab/lib_fold.so 2>&1 | grep -v amdgpu.ids > gpurun_out/r05a_precision.log; cat gpurun_out/r05a_precision.log
timeout 300 python bench.py --steps 10 --warmup 3 2>gpurun_out/r05a_bench_err.log | tail -1 > gpurun_out/r05a_bench.json; cut -c1-400 gpurun_out/r05a_bench.json
timeout 1500 python -m pytest tests -m gpu -x -q -rs > gpurun_out/r05a_pytest_gpu.log 2>&1; tail -5 gpurun_out/r05a_pytest_gpu.log
