cd /root/repo; mkdir -p gpurun_out
( TNR_MMA=bf16x3 timeout 300 python tools/probes/sweep_check.py 2>&1 | grep -E "sweep  |BIT|MISMATCH|error flag:"
  echo "== epilogue at the chunk top (S4_EPI_AT_TOP)"; TNR_HIP_LIB=/root/repo/trainner_amd/lib/variants/libsw_epitop.so TNR_MMA=bf16x3 timeout 300 python tools/probes/sweep_check.py --time-only 2>&1 | grep -E "sweep  "
  echo "== default again"; TNR_MMA=bf16x3 timeout 300 python tools/probes/sweep_check.py --time-only 2>&1 | grep -E "sweep  "
  TNR_HIP_LIB=/root/repo/trainner_amd/lib/variants/libsw_tl.so TNR_MMA=bf16x3 timeout 300 python tools/probes/sweep_check.py --time-only 2>&1 | grep -v amdgpu.ids | tail -22 ) > gpurun_out/r03af_sweep_epi_hooks.txt 2>&1
cat gpurun_out/r03af_sweep_epi_hooks.txt
