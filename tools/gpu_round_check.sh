# Round-end GPU check: full parity suite, smoke, the multi-GPU code path with a world of one (torch and native collectives), the
# bench lines of configs 2-4, the host boundary, the slab protocol's timing with in-process ranks, and the rocprofv3 evidence
# (kernel stats with one forward lane; "pmc" as a second argument adds the separate PMC passes: HBM bytes and MFMA utilisation).
# Usage: gpurun -- 'bash tools/gpu_round_check.sh r03z [pmc]'
TAG=${1:-rXX}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -rs -s > gpurun_out/pytest_gpu_$TAG.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_gpu_$TAG.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
LM_BENCH_FORCE_DIST=1 python bench.py --steps 3 --no-cpu-baseline --host-steps 0 2>gpurun_out/bench_dist_err.log | tail -1 > gpurun_out/bench_forced_dist_$TAG.txt; cut -c1-200 gpurun_out/bench_forced_dist_$TAG.txt
LM_BENCH_FORCE_DIST=1 python bench.py --steps 3 --no-cpu-baseline --host-steps 0 --dist native 2>>gpurun_out/bench_dist_err.log | tail -1 > gpurun_out/bench_forced_dist_native_$TAG.txt; cut -c1-200 gpurun_out/bench_forced_dist_native_$TAG.txt
LM_BENCH_FORCE_DIST=1 python bench.py --steps 3 --no-cpu-baseline --host-steps 0 --post slab 2>>gpurun_out/bench_dist_err.log | tail -1 > gpurun_out/bench_forced_dist_slab_$TAG.txt; cut -c1-200 gpurun_out/bench_forced_dist_slab_$TAG.txt
LM_BENCH_FORCE_DIST=1 python bench.py --config 4 --steps 2 --no-cpu-baseline --host-steps 0 --dist native --post slab 2>>gpurun_out/bench_dist_err.log | tail -1 > gpurun_out/bench_forced_dist_fused_$TAG.txt; cut -c1-200 gpurun_out/bench_forced_dist_fused_$TAG.txt
python bench.py --gpus 2 > gpurun_out/bench_gpus2_refusal_$TAG.txt 2>&1; tail -1 gpurun_out/bench_gpus2_refusal_$TAG.txt | cut -c1-200
python bench.py --steps 20 --warmup 5 2>gpurun_out/bench_err.log | tail -1 > gpurun_out/bench_$TAG.json; cut -c1-200 gpurun_out/bench_$TAG.json
for c in 3 4; do python bench.py --config $c --steps 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_${TAG}_config$c.json; cut -c1-160 gpurun_out/bench_${TAG}_config$c.json; done
LM_HOST_TIMING=1 python tools/host_boundary.py 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/host_boundary_$TAG.log; tail -1 gpurun_out/host_boundary_$TAG.log
python tools/slab_timing.py 2>&1 | grep -v amdgpu.ids > gpurun_out/slab_timing_$TAG.log; tail -4 gpurun_out/slab_timing_$TAG.log
python tools/probe_values.py 2>&1 | grep -v amdgpu.ids > gpurun_out/probe_values_$TAG.log; tail -3 gpurun_out/probe_values_$TAG.log
LM_ASYNC_TIMING=0 python tools/async_probe.py 12 2>&1 | grep -v amdgpu.ids > gpurun_out/async_probe_$TAG.log; tail -3 gpurun_out/async_probe_$TAG.log
python tools/stress.py 60 2>&1 | grep -v amdgpu.ids > gpurun_out/stress_$TAG.log; tail -2 gpurun_out/stress_$TAG.log
python tools/nn_perf.py 20 5 split_f16 2>&1 | grep -v amdgpu.ids > gpurun_out/nn_perf_$TAG.log; head -1 gpurun_out/nn_perf_$TAG.log
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG/trace -- python $R/bench.py --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --host-steps 0 > $R/gpurun_out/prof_${TAG}_bench.json 2>$R/gpurun_out/prof_$TAG.log
if [ "$2" = "pmc" ]; then
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_$TAG/pmc_fetch -- python $R/bench.py --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --host-steps 0 > /dev/null 2>>$R/gpurun_out/prof_$TAG.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_$TAG/pmc_write -- python $R/bench.py --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --host-steps 0 > /dev/null 2>>$R/gpurun_out/prof_$TAG.log
echo "python bench.py --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --host-steps 0 (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes)" > $R/gpurun_out/prof_$TAG/pmc_workload.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/prof_$TAG/pmc_sq -- python $R/bench.py --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --host-steps 0 > /dev/null 2>>$R/gpurun_out/prof_$TAG.log
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $R/gpurun_out/prof_$TAG/pmc_grbm -- python $R/bench.py --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --host-steps 0 > /dev/null 2>>$R/gpurun_out/prof_$TAG.log
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_$TAG/trace2 -- python $R/bench.py --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --host-steps 0 > /dev/null 2>>$R/gpurun_out/prof_$TAG.log
fi
ls $R/gpurun_out/prof_$TAG
