cd /root/repo; mkdir -p gpurun_out
( TNR_TEST_MMA=bf16x3 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "wgrad" 2>&1 | tail -3 ) > gpurun_out/r03aa_wgrad_tests.log 2>&1
cat gpurun_out/r03aa_wgrad_tests.log
timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --detail > gpurun_out/r03aa_bench.json.log 2> gpurun_out/r03aa_per_shape_table.txt
python - <<P
import json
j=json.loads(open("gpurun_out/r03aa_bench.json.log").read().strip().split("\n")[-1])
print(j["value"], j["ms_per_step"], j["roofline"]["kernel_ms_per_step"])
P
grep "wgrad_tile" gpurun_out/r03aa_per_shape_table.txt | head -24
