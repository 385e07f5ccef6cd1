# round 3: the GPU suite in TNR_MMA=bf16x3 with tnr_conv_sweep + the conflict-free LDS swizzle, then the bench in both modes
cd /root/repo; mkdir -p gpurun_out
( time TNR_MMA=bf16x3 timeout 900 python -m pytest tests -m gpu -q -x ) > gpurun_out/r03f_pytest_gpu_mma_bf16x3.log 2>&1
tail -4 gpurun_out/r03f_pytest_gpu_mma_bf16x3.log
timeout 300 python bench.py --mma bf16x3 --steps 8 --warmup 3 --no-cpu-baseline --detail > gpurun_out/r03f_bench_mma_bf16x3.json.log 2> gpurun_out/r03f_per_shape_table_mma_bf16x3.txt
tail -1 gpurun_out/r03f_bench_mma_bf16x3.json.log | cut -c1-400
TNR_CONV_SWEEP=0 timeout 300 python bench.py --mma bf16x3 --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r03f_bench_mma_bf16x3_nosweep.json.log 2>/dev/null
tail -1 gpurun_out/r03f_bench_mma_bf16x3_nosweep.json.log | cut -c1-300
head -30 gpurun_out/r03f_per_shape_table_mma_bf16x3.txt | cut -c1-100
