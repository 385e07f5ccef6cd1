cd /root/repo; mkdir -p gpurun_out
( TNR_TEST_MMA=bf16x3 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_nets.py -m gpu -q -x -k "conv or dgrad or s2 or bf16x3 or disc or unet" 2>&1 | tail -2
  for v in "" nowsplit; do
    if [ -n "$v" ]; then export TNR_HIP_LIB=/root/repo/trainner_amd/lib/variants/lib$v.so; else unset TNR_HIP_LIB; fi
    timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-variant 2>/dev/null | python -c "
import sys, json
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); k=j['roofline']['kernel_ms_per_step']; print('${v:-default}:', j['value'], j['ms_per_step'], 'dgrad4x4s2', k['conv_tile_dgrad4x4s2'], '4x4s2', k['conv_tile_4x4s2'])"
  done ) > gpurun_out/r03al_x3w_4tap.txt 2>&1
cat gpurun_out/r03al_x3w_4tap.txt
