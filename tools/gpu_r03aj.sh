cd /root/repo; mkdir -p gpurun_out
( TNR_TEST_MMA=bf16x3 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_nets.py -m gpu -q -x -k "conv or vgg or disc" 2>&1 | tail -2
  for pz in 1 0; do TNR_X3W8_PERSISTENT=$pz timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-variant 2>/dev/null | python -c "
import sys, json
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('persistent $pz:', j['value'], j['ms_per_step'], 'conv_tile_3x3', j['roofline']['kernel_ms_per_step']['conv_tile_3x3'])"; done ) > gpurun_out/r03aj_x3w8_persistent.txt 2>&1
cat gpurun_out/r03aj_x3w8_persistent.txt
