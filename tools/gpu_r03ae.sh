cd /root/repo; mkdir -p gpurun_out
( /opt/rocm/bin/rocm-smi --showpower 2>&1 | grep -E "Power" | head -2
  TNR_MMA=bf16x3 timeout 300 python tools/probes/sweep_check.py --time-only 2>&1 | grep -E "sweep  "
  timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-variant 2>/dev/null | python -c "
import sys, json
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_per_step'], 'avg_launch_us', j['roofline']['avg_launch_us'])" ) > gpurun_out/r03ae_box_$1.txt 2>&1
cat gpurun_out/r03ae_box_$1.txt
