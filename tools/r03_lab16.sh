T=${1:-r03u}; mkdir -p gpurun_out/$T
./tools/ubench/write_bw > gpurun_out/$T/write_bw.log 2>&1; cat gpurun_out/$T/write_bw.log
python bench.py --steps 20 2>/dev/null | tail -1 > gpurun_out/$T/bench.json; python -c "
import json; d=json.loads(open('gpurun_out/$T/bench.json').read()); print(d['value'], d['ms_per_step'], d['value_host_to_host']['value'], d['value_lminferer_apply']['fresh_output_per_call']['value'], d['roofline']['traffic'], d['roofline']['frac'])"
