set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
LM_BENCH_FORCE_DIST=1 python bench.py --steps 3 --no-cpu-baseline 2>gpurun_out/bench_dist_err.log | tail -1 > gpurun_out/bench_forced_dist.json; cat gpurun_out/bench_forced_dist.json | cut -c1-400
LM_BENCH_FORCE_DIST=1 python bench.py --steps 3 --no-cpu-baseline --post gathered 2>>gpurun_out/bench_dist_err.log | tail -1 > gpurun_out/bench_forced_dist_gathered.json; cat gpurun_out/bench_forced_dist_gathered.json | cut -c1-300
python bench.py 2>gpurun_out/bench_err.log | tail -1 > gpurun_out/bench_r01e.json; cat gpurun_out/bench_r01e.json | cut -c1-600
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01e/trace -- python $R/bench.py --streams 1 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_r01e_bench.json 2>$R/gpurun_out/prof_r01e.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_r01e/pmc_fetch -- python $R/bench.py --streams 1 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>>$R/gpurun_out/prof_r01e.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_r01e/pmc_write -- python $R/bench.py --streams 1 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>>$R/gpurun_out/prof_r01e.log
ls $R/gpurun_out/prof_r01e/*
