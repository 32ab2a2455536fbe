# round-4 lab session 3: clocks / power under the two-lane forward, two-lane per-layer timeline, slot placement variant s2 vs product
mkdir -p gpurun_out/r04c
O=gpurun_out/r04c
timeout 120 python tools/power_probe.py 6 2 2>&1 | grep -v amdgpu.ids > $O/power_two_lanes.log; cat $O/power_two_lanes.log
timeout 120 python tools/power_probe.py 5 1 2>&1 | grep -v amdgpu.ids > $O/power_one_lane.log; tail -4 $O/power_one_lane.log
timeout 200 python tools/lane_timeline.py 300 2>&1 | grep -v amdgpu.ids > $O/lane_timeline.log; cat $O/lane_timeline.log
timeout 300 python tools/ab_forward.py lungmask_amd/liblungmask_hip.so lungmask_amd/_ab/lib_s2.so 2>&1 | grep -v amdgpu.ids > $O/ab_s2.log; cat $O/ab_s2.log
