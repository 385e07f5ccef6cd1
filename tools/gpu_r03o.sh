# round 3: bf16x3 weight gradient with half-height tiles and two workgroups per CU (TNR_WG_X3_OCC=2, default) against one (=1)
cd /root/repo; mkdir -p gpurun_out
( TNR_TEST_MMA=bf16x3 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "wgrad or bf16x3 or weight" 2>&1 | tail -5 ) > gpurun_out/r03o_wgrad_tests.log 2>&1
cat gpurun_out/r03o_wgrad_tests.log
for o in 2 1; do TNR_WG_X3_OCC=$o timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r03o_bench_occ$o.json.log 2>gpurun_out/r03o_bench_occ$o.err
python - <<P
import json
j=json.loads(open("gpurun_out/r03o_bench_occ$o.json.log").read().strip().split("\n")[-1])
print("occ$o", j["value"], j["ms_per_step"], j["roofline"]["kernel_ms_per_step"])
P
done
tail -5 gpurun_out/r03o_bench_occ2.err
