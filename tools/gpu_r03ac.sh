# diagnostic: is the four-wave sweep kernel's speed box-dependent?  sweep_check timing + bench for both forms, clocks and CU count
cd /root/repo; mkdir -p gpurun_out
( /opt/rocm/bin/rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|Power" | head -6
  python - <<P
import torch
p=torch.cuda.get_device_properties(0); print("CUs", p.multi_processor_count, p.name, getattr(p,"gcnArchName",""))
P
  for w in 4 8 4; do echo "== TNR_SWEEP_WAVES=$w"; TNR_SWEEP_WAVES=$w TNR_MMA=bf16x3 timeout 300 python tools/probes/sweep_check.py 2>&1 | grep -E "sweep  " ; done
  for w in 4 8; do TNR_SWEEP_WAVES=$w timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-variant 2>/dev/null | python -c "
import sys, json
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('waves $w:', j['value'], j['ms_per_step'], 'chain', j['roofline']['kernel_ms_per_step']['conv_chain'], 'avg_launch_us', j['roofline']['avg_launch_us'])"; done
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power" | head -4 ) > gpurun_out/r03ac_box_diag.txt 2>&1
cat gpurun_out/r03ac_box_diag.txt
