cd /root/repo; mkdir -p gpurun_out
TNR_HIP_LIB=/root/repo/trainner_amd/lib/variants/libx3il.so timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "bf16x3" 2>&1 | tail -1
for lib in "" /root/repo/trainner_amd/lib/variants/libx3il.so "" /root/repo/trainner_amd/lib/variants/libx3il.so; do
  TNR_HIP_LIB=$lib timeout 300 python bench.py --mma bf16x3 --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.readlines()[-1]); print('lib=$lib', j['value'], j['ms_per_step'], j['roofline']['achieved'], j['roofline'].get('kernel_ms_per_step'))"
done
