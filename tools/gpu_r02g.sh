cd /root/repo
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -m gpu -q ) > gpurun_out/r02g_pytest_gpu.log 2>&1
grep -E "passed|failed" gpurun_out/r02g_pytest_gpu.log
grep -E "^FAILED" gpurun_out/r02g_pytest_gpu.log | head
timeout 300 python tools/bench_i2i.py --model pix2pix 2>/dev/null | tail -1 | cut -c1-200
timeout 300 python tools/bench_i2i.py --model cyclegan --batch 8 2>/dev/null | tail -1 | cut -c1-200
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-200
