cd /root/repo; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -s -k "bf16x3" 2>&1 | grep -v amdgpu.ids | tail -6
for w in 0 1 0 1; do
  TNR_WGRAD_X3=$w timeout 300 python bench.py --mma bf16x3 --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.readlines()[-1]); print('WGRAD_X3=$w', j['value'], j['ms_per_step'], j['roofline']['achieved'], j['roofline'].get('kernel_ms_per_step'))"
done
