# round-end validation: every GPU test, smoke, the default bench line, rocprofv3 kernel stats of the same command
cd /root/repo
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/full_gpu_tests.log 2>&1
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/full_smoke.log 2>&1
( time timeout 600 python bench.py ) > gpurun_out/full_bench.log 2>&1
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --detail > gpurun_out/full_bench_detail.json 2> gpurun_out/full_bench_detail.log
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_full
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_full -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /root/repo/gpurun_out/prof_full.log 2>&1
cd /root/repo
find gpurun_out/prof_full -name "*kernel_trace.csv" -delete
tail -3 gpurun_out/full_gpu_tests.log; tail -4 gpurun_out/full_smoke.log; tail -5 gpurun_out/full_bench.log | cut -c1-1500
