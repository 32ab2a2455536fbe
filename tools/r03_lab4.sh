T=r03f; mkdir -p gpurun_out/$T/prio
for r in 1 2; do for b in plain skew skewprio; do ./tools/ubench/conv_lab_$b 20 6 > gpurun_out/$T/conv_lab_$b.$r.log 2>&1; done; done
LM_TL_DIR=gpurun_out/$T/prio ./tools/ubench/conv_lab_tl_skewprio 20 4 > gpurun_out/$T/tl.log 2>&1
paste <(awk '{print $1,$3}' gpurun_out/$T/conv_lab_plain.1.log) <(awk '{print $3}' gpurun_out/$T/conv_lab_skew.1.log) <(awk '{print $3}' gpurun_out/$T/conv_lab_skewprio.1.log) <(awk '{print $3}' gpurun_out/$T/conv_lab_plain.2.log) <(awk '{print $3}' gpurun_out/$T/conv_lab_skew.2.log) <(awk '{print $3}' gpurun_out/$T/conv_lab_skewprio.2.log)
