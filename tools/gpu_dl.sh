# LDS-DMA staging experiments (csrc/conv_body_dl.h, build: python tools/build_variant.py dl -DTNR_CONV_DL_EXPERIMENT):
# per-shape probe of every form, chain parity + chain microbench + bench line with the chain in form 2
cd /root/repo; mkdir -p gpurun_out
export TNR_HIP_LIB=/root/repo/trainner_amd/lib/variants/libdl.so
O=gpurun_out/${1:-r02e}_conv_dl_experiment.txt
( echo "## tools/probes/conv_dl_check.py"; timeout 300 python tools/probes/conv_dl_check.py 2>&1 | grep -v amdgpu.ids
  echo "## chain parity (tests/test_gpu_kernels.py -k chain) and tools/microbench_chain2.py with TNR_CONV_DL=2"
  TNR_CONV_DL=2 timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "chain" 2>&1 | tail -1
  TNR_CONV_DL=0 timeout 120 python tools/microbench_chain2.py 2>&1 | grep "n=6"
  TNR_CONV_DL=2 timeout 120 python tools/microbench_chain2.py 2>&1 | grep "n=6\|error"
  echo "## bench.py --steps 6 --warmup 3: img/s, ms/step, chain TFLOP/s"
  for v in 0 2 0 2; do
    TNR_CONV_DL=$v timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.readlines()[-1]); print('TNR_CONV_DL=$v', j['value'], j['ms_per_step'], j['roofline']['achieved'])"
  done ) > $O 2>&1
cat $O
