# TNR_MMA_BF16X3: accuracy / speed probe, chain microbench, bench lines
cd /root/repo; mkdir -p gpurun_out
O=gpurun_out/${1:-r02f}_mma_bf16x3.txt
( echo "## tools/probes/mma_x3_check.py"; timeout 300 python tools/probes/mma_x3_check.py 2>&1 | grep -v amdgpu.ids
  echo "## tools/microbench_chain2.py (TNR_MMA=f32 | bf16x3)"
  TNR_MMA=f32 timeout 120 python tools/microbench_chain2.py 2>&1 | grep "n=6"
  TNR_MMA=bf16x3 timeout 120 python tools/microbench_chain2.py 2>&1 | grep "n=6\|error"
  echo "## bench.py --steps 6 --warmup 3: img/s, ms/step, chain TFLOP/s, families"
  for v in f32 bf16x3; do
    TNR_MMA=$v timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.readlines()[-1]); print('TNR_MMA=$v', j['value'], j['ms_per_step'], j['roofline']['achieved'], j['roofline'].get('kernel_ms_per_step'))"
  done ) > $O 2>&1
cat $O
