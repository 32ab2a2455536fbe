# Round-end evidence run: parity suite, smoke, distributed path with a world of one, bench (3 configs), rocprofv3 stats + PMC + MFMA utilisation
TAG=${1:-r02}
bash tools/gpu_round_check.sh $TAG 2>&1 | tail -12
for c in 3 4; do timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_${TAG}_config$c.json; done
TAG=$TAG bash tools/pmc_mfma.sh 2>&1 | tail -3
timeout 120 python tools/nn_perf.py 20 5 split_f16 2>&1 | grep -v amdgpu > gpurun_out/nn_perf_${TAG}.log
timeout 60 python tools/host_boundary.py > gpurun_out/host_boundary_${TAG}.log 2>&1; tail -1 gpurun_out/host_boundary_${TAG}.log
