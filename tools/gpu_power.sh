# sample power / clocks while bench.py runs
cd /root/repo; mkdir -p gpurun_out
( for i in $(seq 1 60); do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor junction\)|fclk" | tr '\n' ' ' ; echo; sleep 0.5; done ) > gpurun_out/power.log 2>&1 &
SP=$!
timeout 300 python bench.py --steps 30 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/bench_power.log 2>&1
kill $SP 2>/dev/null
tail -1 gpurun_out/bench_power.log | cut -c1-200
sed -n 1,60p gpurun_out/power.log | cut -c1-400 | awk 'NR%3==1'
