T=${1:-r03l}; mkdir -p gpurun_out/$T
LM_LAB_VERIFY=1 ./tools/ubench/conv_lab_plain 4 1 2>&1 | grep verify > gpurun_out/$T/verify.log; cat gpurun_out/$T/verify.log
./tools/ubench/conv_lab 20 6 > gpurun_out/$T/conv_lab_trace.log 2>&1; cat gpurun_out/$T/conv_lab_trace.log
LM_TL_DIR=gpurun_out/$T ./tools/ubench/conv_lab_tl 20 4 > gpurun_out/$T/conv_lab_tl.log 2>&1
python tools/tl_epi.py gpurun_out/$T > gpurun_out/$T/epi_phases.txt; cat gpurun_out/$T/epi_phases.txt
