# TNR_MMA_BF16X3: 8-wave pre-split kernel against the 4-wave form (bit-identity, speed), kernel test, bench lines
cd /root/repo; mkdir -p gpurun_out
TNR_X3_W8=0 timeout 120 python tools/probes/x3w8_check.py save /tmp/x3ref.pt 2>&1 | tail -1
TNR_X3_W8=1 timeout 120 python tools/probes/x3w8_check.py compare /tmp/x3ref.pt 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "bf16x3" 2>&1 | tail -1
for w in 0 1; do echo "TNR_X3_W8=$w"; TNR_X3_W8=$w timeout 300 python tools/probes/mma_x3_check.py 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-110; done
for w in 0 1 0 1; do
  TNR_X3_W8=$w timeout 300 python bench.py --mma bf16x3 --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.readlines()[-1]); k=j['roofline'].get('kernel_ms_per_step'); print('TNR_X3_W8=$w', j['value'], j['ms_per_step'], 'chain', k['conv_chain'], 'conv3x3', k['conv_tile_3x3'], 'wgrad', k['wgrad_tile'])"
done
