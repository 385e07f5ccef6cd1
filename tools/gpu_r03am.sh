cd /root/repo; mkdir -p gpurun_out
( TNR_TEST_MMA=bf16x3 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py -m gpu -q -x -k "conv or chain or sweep or golden or bf16x3" 2>&1 | tail -2
  timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-variant 2>/dev/null | python -c "
import sys, json
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=j['roofline']['kernel_ms_per_step']; print(j['value'], j['ms_per_step'], k)" ) > gpurun_out/r03am_split4_vector.txt 2>&1
cat gpurun_out/r03am_split4_vector.txt
