# config 5 (tools/bench_i2i.py) with the 7x7 image-side layers as one launch per pass (TNR_K7_FUSED=1, default) vs nine 3x3 blocks (0)
cd /root/repo
for M in "pix2pix resnet" "cyclegan resnet"; do
  set -- $M
  for F in 1 0 1 0; do
    echo "== $1 $2 TNR_K7_FUSED=$F"
    TNR_K7_FUSED=$F timeout 600 python tools/bench_i2i.py --model $1 --netg $2 --steps 8 --warmup 3 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    j = json.loads(l[l.index('{'):])
    fam = j.get('roofline', {}).get('families', {})
    top = sorted(((k, (round(v.get('ms_per_step', 0), 2), v.get('frac'))) for k, v in fam.items()), key=lambda kv: -kv[1][0])[:6]
    print(j.get('value'), j.get('ms_per_step'), j.get('step_tflops'), top)
except Exception as e:
    print('unparsed:', l[:400], e)
"
  done
done
