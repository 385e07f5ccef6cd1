# rocprofv3 kernel stats of the bench command with TNR_MMA=bf16x3
TAG=${1:-r02h}
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_x3_${TAG}
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_x3_${TAG} -- python /root/repo/bench.py --mma bf16x3 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /root/repo/gpurun_out/prof_x3_${TAG}.log 2>&1
cd /root/repo
find gpurun_out/prof_x3_${TAG} -name "*kernel_trace.csv" -delete
cp $(find gpurun_out/prof_x3_${TAG} -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats_bench_steps3_mma_bf16x3.csv
head -12 gpurun_out/${TAG}_kernel_stats_bench_steps3_mma_bf16x3.csv | cut -c1-170
tail -1 gpurun_out/prof_x3_${TAG}.log | cut -c1-200
