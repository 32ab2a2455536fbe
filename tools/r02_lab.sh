# Round-2 lab run (measurement only, not the product): operand-pattern power probe, stream ablations of the conv kernel,
# the GPU parity suite and the bench line.
TAG=${1:-r02b}
timeout 120 tools/ubench/mfma_power > gpurun_out/${TAG}_mfma_power.log 2>&1; cat gpurun_out/${TAG}_mfma_power.log
for v in plain abl1 abl2; do echo "== conv_lab_$v"; timeout 120 tools/ubench/conv_lab_$v > gpurun_out/${TAG}_conv_lab_$v.log 2>&1; cut -c1-44 gpurun_out/${TAG}_conv_lab_$v.log; done
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; tail -4 gpurun_out/${TAG}_pytest_gpu.log
timeout 300 python bench.py 2>gpurun_out/${TAG}_bench_err.log | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-400 gpurun_out/${TAG}_bench.json
