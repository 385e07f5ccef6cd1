# last validation of the round: the GPU suite with TNR_MMA=bf16x3 (8-wave kernel on), the default bench line (headline + variant), the x3 per-shape table
TAG=${1:-r02g}
cd /root/repo; mkdir -p gpurun_out
( time TNR_MMA=bf16x3 timeout 1200 python -m pytest tests -m gpu -q ) > gpurun_out/${TAG}_pytest_gpu_mma_bf16x3.log 2>&1
tail -3 gpurun_out/${TAG}_pytest_gpu_mma_bf16x3.log
( time timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py -m gpu -q ) > gpurun_out/${TAG}_pytest_gpu_kernels_step.log 2>&1
tail -3 gpurun_out/${TAG}_pytest_gpu_kernels_step.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench_n1.json.log 2>/dev/null; tail -1 gpurun_out/${TAG}_bench_n1.json.log | cut -c1-200
timeout 300 python bench.py --mma bf16x3 --steps 6 --warmup 2 --no-cpu-baseline --detail > gpurun_out/${TAG}_bench_mma_bf16x3.json.log 2> gpurun_out/${TAG}_per_shape_table_mma_bf16x3.txt
tail -1 gpurun_out/${TAG}_bench_mma_bf16x3.json.log | python -c "
import sys,json
j=json.loads(sys.stdin.readlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['achieved'], j['roofline'].get('kernel_ms_per_step'))"
