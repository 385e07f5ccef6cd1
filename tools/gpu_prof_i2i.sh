cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_i2i
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_i2i -- python /root/repo/tools/bench_i2i.py --model ${1:-pix2pix} --steps 2 --warmup 1 --batch ${2:-16} > /root/repo/gpurun_out/prof_i2i.log 2>&1
cd /root/repo
find gpurun_out/prof_i2i -name "*kernel_trace.csv" -delete
f=$(find gpurun_out/prof_i2i -name "*kernel_stats.csv" | head -1)
head -25 $f | cut -c1-200
tail -2 gpurun_out/prof_i2i.log | cut -c1-300
