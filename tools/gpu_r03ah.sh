cd /root/repo; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_nets.py tests/test_gpu_step.py tests/test_gpu_i2i.py -m gpu -q -x -k "bn or norm or disc or golden or step or pix2pix or cycle or instnorm" 2>&1 | tail -3
  timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-variant --detail 2> gpurun_out/r03ah_per_shape.txt | python -c "
import sys, json
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_per_step'])" ) > gpurun_out/r03ah_bn.txt 2>&1
cat gpurun_out/r03ah_bn.txt
