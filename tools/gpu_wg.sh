cd /root/repo
for v in base "$@"; do
  echo "== $v"
  if [ "$v" != base ]; then export TNR_HIP_LIB=/root/repo/trainner_amd/lib/variants/lib$v.so; else unset TNR_HIP_LIB; fi
  python tools/microbench_conv.py 2>&1 | grep -E "^wgrad"
  timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'].get('wgrad_tile'))"
done
