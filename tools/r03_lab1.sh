# round-3 lab batch 1: timelines of the conv kernel (two builds), lane/grid experiment
mkdir -p gpurun_out/r03a
export LM_TL_DIR=gpurun_out/r03a
./tools/ubench/conv_lab_tl 20 4 > gpurun_out/r03a/conv_lab_tl.log 2>&1
mkdir -p gpurun_out/r03a/prio; LM_TL_DIR=gpurun_out/r03a/prio ./tools/ubench/conv_lab_tl_prio 20 4 > gpurun_out/r03a/conv_lab_tl_prio.log 2>&1
for g in 0 128 192 96; do echo "LM_H3_GRID=$g"; LM_H3_GRID=$g python tools/ab_forward.py lungmask_amd/liblungmask_hip.so 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r03a/grid_lanes.log 2>&1
tail -5 gpurun_out/r03a/conv_lab_tl.log; cat gpurun_out/r03a/grid_lanes.log
