cd /tmp && export TMPDIR=/tmp
for M in ${MODELS:-cyclegan pix2pix}; do
rm -rf /tmp/prof_$M
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$M -- python /root/repo/tools/bench_i2i.py --model $M --netg resnet --steps 3 --warmup 1 > /root/repo/gpurun_out/${TAG:-r12c}_prof_i2i_$M.log 2>&1
cp $(find /tmp/prof_$M -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/${TAG:-r12c}_kernel_stats_i2i_${M}_resnet.csv
head -30 /root/repo/gpurun_out/${TAG:-r12c}_kernel_stats_i2i_${M}_resnet.csv | cut -c1-170
tail -3 /root/repo/gpurun_out/${TAG:-r12c}_prof_i2i_$M.log | cut -c1-600
done
