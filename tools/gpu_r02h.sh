cd /root/repo
mkdir -p gpurun_out
timeout 120 python tools/microbench_chain.py 2>&1 | grep -v amdgpu.ids | head -3
( time timeout 1200 python -m pytest tests -m gpu -q ) > gpurun_out/r02h_pytest_gpu.log 2>&1
grep -E "passed|failed" gpurun_out/r02h_pytest_gpu.log
grep -E "^FAILED" gpurun_out/r02h_pytest_gpu.log | head
timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-900
