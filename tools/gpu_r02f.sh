cd /root/repo
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -m gpu -q ) > gpurun_out/r02f_pytest_gpu.log 2>&1
tail -4 gpurun_out/r02f_pytest_gpu.log | head -2
timeout 300 python tools/bench_i2i.py --model pix2pix 2>/dev/null | tail -1 | cut -c1-200
timeout 300 python tools/bench_i2i.py --model pix2pix --amp 2>/dev/null | tail -1 | cut -c1-200
timeout 300 python tools/bench_i2i.py --model cyclegan --batch 8 2>/dev/null | tail -1 | cut -c1-200
