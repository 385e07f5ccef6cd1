cd /root/repo; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py -m gpu -q -x 2>&1 | tail -4 ) > gpurun_out/r03z_tests.log 2>&1
cat gpurun_out/r03z_tests.log
for o in 3 2; do TNR_WG_X3_OCC=$o timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r03z_bench_occ$o.json.log 2>/dev/null
python - <<P
import json
j=json.loads(open("gpurun_out/r03z_bench_occ$o.json.log").read().strip().split("\n")[-1])
print("occ$o", j["value"], j["ms_per_step"], j["roofline"]["kernel_ms_per_step"])
P
done
