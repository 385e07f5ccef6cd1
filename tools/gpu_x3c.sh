cd /root/repo; mkdir -p gpurun_out
timeout 300 python tools/probes/mma_x3_check.py 2>&1 | grep -v amdgpu.ids | tail -8
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "bf16x3" 2>&1 | tail -2
TNR_MMA=f32 timeout 120 python tools/microbench_chain2.py 2>&1 | grep "n=6"
TNR_MMA=bf16x3 TNR_CHAIN_X3=1 timeout 120 python tools/microbench_chain2.py 2>&1 | grep "n=6\|error"
for v in 1; do
  TNR_CHAIN_X3=$v timeout 300 python bench.py --mma bf16x3 --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.readlines()[-1]); print('CHAIN_X3=$v', j['value'], j['ms_per_step'], j['roofline']['achieved'], j['roofline'].get('kernel_ms_per_step'))"
done
