cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
for M in pix2pix cyclegan; do
rm -rf /tmp/prof_$M
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$M -- python $R/tools/bench_i2i.py --model $M --steps 6 --warmup 2 > $O/r04q_prof_$M.log 2>&1
cp $(find /tmp/prof_$M -name "*kernel_stats.csv" | head -1) $O/r04q_kernel_stats_$M.csv
head -16 $O/r04q_kernel_stats_$M.csv | cut -c1-170
done
