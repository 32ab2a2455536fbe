mkdir -p gpurun_out/r03b
export LM_TL_DIR=gpurun_out/r03b
./tools/ubench/conv_lab_base 20 6 > gpurun_out/r03b/conv_lab_base.log 2>&1
./tools/ubench/conv_lab_plain 20 6 > gpurun_out/r03b/conv_lab_new.log 2>&1
./tools/ubench/conv_lab_base 20 6 > gpurun_out/r03b/conv_lab_base2.log 2>&1
./tools/ubench/conv_lab_plain 20 6 > gpurun_out/r03b/conv_lab_new2.log 2>&1
./tools/ubench/conv_lab_tl 20 4 > gpurun_out/r03b/conv_lab_tl.log 2>&1
python tools/ab_forward.py lungmask_amd/_ab/liblungmask_hip_r02.so lungmask_amd/liblungmask_hip.so 2>&1 | grep -v amdgpu.ids > gpurun_out/r03b/ab.log
paste <(awk '{print $1, $2, $3}' gpurun_out/r03b/conv_lab_base.log) <(awk '{print $2, $3}' gpurun_out/r03b/conv_lab_new.log) <(awk '{print $2,$3}' gpurun_out/r03b/conv_lab_base2.log) <(awk '{print $2, $3}' gpurun_out/r03b/conv_lab_new2.log)
cat gpurun_out/r03b/ab.log
