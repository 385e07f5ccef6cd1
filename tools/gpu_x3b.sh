cd /root/repo; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py -q -x -s -k "bf16x3" 2>&1 | grep -v amdgpu.ids | tail -12
for v in f32 bf16x3 f32 bf16x3; do
  timeout 300 python bench.py --mma $v --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.readlines()[-1]); print('--mma $v', j['value'], j['ms_per_step'], j['roofline']['achieved'], j['roofline'].get('kernel_ms_per_step'), j['losses'])"
done
