# sample power / clocks (rocm-smi, 2 Hz) while bench.py runs on the fp32 matrix core and with TNR_MMA=bf16x3
cd /root/repo; mkdir -p gpurun_out
O=gpurun_out/${1:-r02g}_power_clocks.txt
: > $O
for mode in f32 bf16x3; do
  ( for i in $(seq 1 44); do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|Temperature \(Sensor junction\)" | sed 's/.*GPU\[0\][^:]*: //' | tr '\n' ' ' ; echo; sleep 0.5; done ) > gpurun_out/power_$mode.log 2>&1 &
  SP=$!
  timeout 300 python bench.py --mma $mode --steps 40 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/bench_power_$mode.log 2>/dev/null
  kill $SP 2>/dev/null; wait $SP 2>/dev/null
  echo "== --mma $mode: $(tail -1 gpurun_out/bench_power_$mode.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], 'img/s', j['ms_per_step'], 'ms/step')")" >> $O
  echo "rocm-smi every 0.5 s (idle start of the process, then the steps):" >> $O
  awk 'NR%2==0' gpurun_out/power_$mode.log | cut -c1-220 >> $O
done
cat $O
