T=r03zz; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$T.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_gpu_$T.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1
python bench.py --steps 20 2>gpurun_out/bench_err.log | tail -1 > gpurun_out/bench_$T.json; cut -c1-160 gpurun_out/bench_$T.json
