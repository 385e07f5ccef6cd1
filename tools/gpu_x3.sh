# TNR_MMA_BF16X3, 8-wave chain kernel (TNR_CHAIN_X3W8=1): chain parity (bit-identical to per-layer launches), bench lines on / off
cd /root/repo; mkdir -p gpurun_out
TNR_MMA=bf16x3 TNR_CHAIN_X3W8=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "chain or bf16x3" 2>&1 | tail -4
for w in 0 1 0 1; do
  TNR_CHAIN_X3W8=$w timeout 300 python bench.py --mma bf16x3 --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.readlines()[-1]); k=j['roofline'].get('kernel_ms_per_step'); print('TNR_CHAIN_X3W8=$w', j['value'], j['ms_per_step'], 'chain', k['conv_chain'], j['roofline']['achieved'], 'conv3x3', k['conv_tile_3x3'], 'wgrad', k['wgrad_tile'], j['losses']['pix-l1'], j['losses']['l_d_real'])"
done
