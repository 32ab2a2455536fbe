# Round-2 lab run (measurement only, not the product)
TAG=${1:-r02c}
timeout 200 python tools/nn_perf3.py > gpurun_out/${TAG}_nn_perf3.log 2>&1; grep batch gpurun_out/${TAG}_nn_perf3.log
timeout 300 python tools/nn_perf4.py > gpurun_out/${TAG}_nn_perf4.log 2>&1; grep batch gpurun_out/${TAG}_nn_perf4.log
for b in 40 80; do echo "== conv_lab_plain B=$b"; timeout 120 tools/ubench/conv_lab_plain $b 3 > gpurun_out/${TAG}_conv_lab_B$b.log 2>&1; cut -c1-44 gpurun_out/${TAG}_conv_lab_B$b.log; done
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/${TAG}_pytest_gpu.log | tail -2
