cd /root/repo; mkdir -p gpurun_out
( for v in "" dma34 dma36 ""; do
    if [ -n "$v" ]; then export TNR_HIP_LIB=/root/repo/trainner_amd/lib/variants/lib$v.so; else unset TNR_HIP_LIB; fi
    echo "== ${v:-default (2 units, every 3rd MFMA)}"
    TNR_MMA=bf16x3 timeout 300 python tools/probes/sweep_check.py 2>&1 | grep -E "sweep  |BIT"
  done ) > gpurun_out/r03an_dma_spread.txt 2>&1
cat gpurun_out/r03an_dma_spread.txt
