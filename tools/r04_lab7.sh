mkdir -p gpurun_out/r04g
O=gpurun_out/r04g
timeout 200 python tools/post_timing.py lunglike 2>&1 | grep -v amdgpu.ids > $O/post_timing_lunglike.log; cat $O/post_timing_lunglike.log
timeout 200 python tools/post_timing.py random 2>&1 | grep -v amdgpu.ids > $O/post_timing_random.log; tail -3 $O/post_timing_random.log
timeout 300 python bench.py --steps 10 --no-cpu-baseline 2>$O/bench_err.log | tail -1 > $O/bench_lunglike.json; python -c "
import json;d=json.load(open('$O/bench_lunglike.json'));print(d['value'],d['ms_per_step'],d['postprocessing'],d['stages_ms_per_step'])"
timeout 300 python bench.py --steps 10 --no-cpu-baseline --host-steps 0 --head random 2>>$O/bench_err.log | tail -1 > $O/bench_random.json; python -c "
import json;d=json.load(open('$O/bench_random.json'));print(d['value'],d['ms_per_step'],d['postprocessing'])"
