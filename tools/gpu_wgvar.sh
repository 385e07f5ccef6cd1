cd /root/repo
for v in base "$@"; do
  echo "== $v"
  if [ "$v" != base ]; then export TNR_HIP_LIB=/root/repo/trainner_amd/lib/variants/lib$v.so; else unset TNR_HIP_LIB; fi
  python tools/microbench_conv.py 2>&1 | grep -E "^wgrad"
  python tools/microbench_wgrad.py 2>&1 | grep -E " (16|64) +[0-9.]+ +[0-9.]+$"
done
