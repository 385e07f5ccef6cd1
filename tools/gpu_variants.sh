# A/B of kernel-build variants (tools/build_variant.py): dense-block chain + a few per-layer convolution shapes per variant
cd /root/repo
run() {
  python tools/microbench_chain.py 2>&1 | grep -v amdgpu.ids | sed -n 1,2p
  python tools/microbench_conv.py 2>&1 | grep -E "^conv +(64|256|512) +64 " | head -3
}
echo "== base"; run
for v in "$@"; do echo "== $v"; TNR_HIP_LIB=/root/repo/trainner_amd/lib/variants/lib$v.so run; done
