# Separate rocprofv3 passes for the MFMA-utilisation counters (SQ block, GRBM block) + a plain kernel trace.  Usage: TAG=r01j bash tools/pmc_mfma.sh
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/prof_${TAG:-r01j}/pmc_sq -- python $R/bench.py --streams 1 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>$R/gpurun_out/prof_${TAG:-r01j}_sq.log
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $R/gpurun_out/prof_${TAG:-r01j}/pmc_grbm -- python $R/bench.py --streams 1 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>$R/gpurun_out/prof_${TAG:-r01j}_grbm.log
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG:-r01j}/trace2 -- python $R/bench.py --streams 1 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>>$R/gpurun_out/prof_${TAG:-r01j}_grbm.log
ls $R/gpurun_out/prof_${TAG:-r01j}/pmc_sq/* $R/gpurun_out/prof_${TAG:-r01j}/pmc_grbm/* | head; tail -3 $R/gpurun_out/prof_${TAG:-r01j}_sq.log
