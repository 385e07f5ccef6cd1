# bench variants of a tree (config 4 = --netd unet --feed resrgan; the input pipeline in the loop; --amp; config 5 = tools/bench_i2i.py)
# usage: bash tools/gpu_variants.sh <tag>
TAG=${1:-rXX}; R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for A in "" "--feed paired" "--netd unet" "--netd unet --feed resrgan" "--amp"; do
  echo "== bench.py $A"
  timeout 600 python bench.py $A --no-cpu-baseline --no-variant 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d.get('roofline') or {}
print(d['value'], d['unit'], d['ms_per_step'], 'ms/step', d['dtype'], 'step_tflops', d.get('step_tflops'), '| dominant:', (r.get('kernel') or '')[:40], r.get('achieved'), r.get('unit'), 'frac', r.get('frac'), 'avg_launch_us', r.get('avg_launch_us'))
print('   kernel_ms_per_step', r.get('kernel_ms_per_step'))"
done
for A in "--model pix2pix" "--model cyclegan" "--model pix2pix --netg unet"; do
  echo "== tools/bench_i2i.py $A"
  timeout 600 python tools/bench_i2i.py $A 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['value'], d['unit'], d['ms_per_step'], 'ms/step', d['dtype'], 'step_tflops', d.get('step_tflops'), 'roofline', d.get('roofline'))"
done
} > $O/${TAG}_bench_variants.txt 2>&1
cat $O/${TAG}_bench_variants.txt | cut -c1-420
