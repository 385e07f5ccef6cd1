cd /root/repo; mkdir -p gpurun_out
( for a in "--netd unet" "--netd unet --feed resrgan" "--feed paired" "--amp"; do
    echo "== bench.py $a"
    timeout 400 python bench.py $a --steps 6 --warmup 2 --no-cpu-baseline --no-variant --no-roofline 2>/dev/null | python -c "
import sys, json
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(j['value'], j['unit'], j['ms_per_step'], 'ms/step', j['dtype'], {k:v for k,v in j['config'].items() if k in ('feed','netd')})"
  done ) > gpurun_out/r03ag_bench_variants.txt 2>&1
cat gpurun_out/r03ag_bench_variants.txt
