# round-3 baseline of the tree as restored: GPU parity suite, bench line, rocprofv3 kernel stats
TAG=${1:-r03h}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$TAG.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_gpu_$TAG.log | tail -3
timeout 300 python bench.py 2>gpurun_out/bench_err.log | tail -1 > gpurun_out/bench_$TAG.json; cut -c1-200 gpurun_out/bench_$TAG.json
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG/trace -- python $R/bench.py --streams 1 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_bench.json 2>$R/gpurun_out/prof_$TAG.log
ls $R/gpurun_out/prof_$TAG
