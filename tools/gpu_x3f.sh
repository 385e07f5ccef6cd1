cd /root/repo; mkdir -p gpurun_out
( time TNR_MMA=bf16x3 timeout 1200 python -m pytest tests -m gpu -q ) > gpurun_out/x3_pytest_gpu.log 2>&1
tail -15 gpurun_out/x3_pytest_gpu.log
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-3000
