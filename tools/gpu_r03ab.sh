# round 3: image-to-image recipes in the default arithmetic (and on the fp32 matrix core), with a rocprofv3 kernel table for each generator
cd /root/repo; mkdir -p gpurun_out
( for m in bf16x3 f32; do
    echo "== TNR_MMA=$m"
    TNR_MMA=$m timeout 200 python tools/bench_i2i.py --model pix2pix 2>/dev/null | tail -1
    TNR_MMA=$m timeout 200 python tools/bench_i2i.py --model pix2pix --netg unet 2>/dev/null | tail -1
    TNR_MMA=$m timeout 300 python tools/bench_i2i.py --model cyclegan --batch 4 2>/dev/null | tail -1
  done
  echo "== use_amp (bf16 operands)"; timeout 200 python tools/bench_i2i.py --model pix2pix --amp 2>/dev/null | tail -1 ) > gpurun_out/r03ab_bench_i2i.txt 2>&1
cut -c1-260 gpurun_out/r03ab_bench_i2i.txt
cd /tmp && export TMPDIR=/tmp
for g in resnet unet; do
  rm -rf /tmp/prof_i2i_$g
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_i2i_$g -- python /root/repo/tools/bench_i2i.py --model pix2pix --netg $g --steps 2 --warmup 1 > /dev/null 2>&1
  f=$(find /tmp/prof_i2i_$g -name "*kernel_stats.csv" | head -1)
  head -22 $f | cut -c1-170 > /root/repo/gpurun_out/r03ab_kernel_stats_pix2pix_$g.csv
done
head -12 /root/repo/gpurun_out/r03ab_kernel_stats_pix2pix_unet.csv
