cd /root/repo; mkdir -p gpurun_out
timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r03w_bench.json.log 2>gpurun_out/r03w_bench.err
python - <<P
import json
j=json.loads(open("gpurun_out/r03w_bench.json.log").read().strip().split("\n")[-1])
print(j["value"], j["ms_per_step"], j["roofline"]["achieved"], j["roofline"]["frac"], j["roofline"]["kernel_ms_per_step"])
P
