# Round-end GPU check: full parity suite, smoke, the multi-GPU code path with a world of one, the bench lines of configs 2-4, the host
# boundary, and the rocprofv3 evidence (kernel stats with one forward lane; add "pmc" as a second argument for the separate PMC
# passes -- FETCH_SIZE / WRITE_SIZE are only needed again when the conv kernel changes).
# Usage: gpurun -- 'bash tools/gpu_round_check.sh r02zz [pmc]'
TAG=${1:-rXX}
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$TAG.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_gpu_$TAG.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
LM_BENCH_FORCE_DIST=1 python bench.py --steps 3 --no-cpu-baseline 2>gpurun_out/bench_dist_err.log > gpurun_out/bench_forced_dist_$TAG.txt; tail -1 gpurun_out/bench_forced_dist_$TAG.txt | cut -c1-200
python bench.py 2>gpurun_out/bench_err.log | tail -1 > gpurun_out/bench_$TAG.json; cut -c1-200 gpurun_out/bench_$TAG.json
for c in 3 4; do python bench.py --config $c --steps 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_${TAG}_config$c.json; cut -c1-160 gpurun_out/bench_${TAG}_config$c.json; done
LM_HOST_TIMING=1 python tools/host_boundary.py 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/host_boundary_$TAG.log; tail -1 gpurun_out/host_boundary_$TAG.log
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG/trace -- python $R/bench.py --streams 1 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_bench.json 2>$R/gpurun_out/prof_$TAG.log
if [ "$2" = "pmc" ]; then
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_$TAG/pmc_fetch -- python $R/bench.py --streams 1 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>>$R/gpurun_out/prof_$TAG.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_$TAG/pmc_write -- python $R/bench.py --streams 1 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>>$R/gpurun_out/prof_$TAG.log
fi
ls $R/gpurun_out/prof_$TAG
