cd /root/repo
mkdir -p gpurun_out
echo "== base"; timeout 120 python tools/microbench_chain.py 2>&1 | grep -v amdgpu.ids | head -3
for v in "$@"; do
  echo "== $v"; TNR_HIP_LIB=/root/repo/trainner_amd/lib/variants/lib$v.so timeout 120 python tools/microbench_chain.py 2>&1 | grep -v amdgpu.ids | sed -n 2,2p\;5,5p
done
last="${@: -1}"
TNR_HIP_LIB=/root/repo/trainner_amd/lib/variants/lib$last.so timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "chain" 2>&1 | tail -2
