# quick GPU check: kernel + net + golden step tests, then a short bench with the per-shape table
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_nets.py -x -q 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_step.py -x -q -k "golden" 2>&1 | tail -3
timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --detail > gpurun_out/bench_q.log 2> gpurun_out/bench_q_detail.log
tail -1 gpurun_out/bench_q.log | cut -c1-1400
