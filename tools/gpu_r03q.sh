# round 3: is the sweep kernel's inner loop bound by LDS fragment reads?  Ablation builds that keep the MFMAs and drop the reads.
cd /root/repo; mkdir -p gpurun_out
( for v in "" sw_nofb sw_nofab sw_nomem; do
    echo "== variant ${v:-default}"
    if [ -n "$v" ]; then export TNR_HIP_LIB=/root/repo/trainner_amd/lib/variants/lib$v.so; else unset TNR_HIP_LIB; fi
    TNR_MMA=bf16x3 timeout 300 python tools/probes/sweep_check.py 2>&1 | grep -E "sweep|BIT|MISMATCH" | tail -4
  done ) > gpurun_out/r03q_sweep_lds_ablation.txt 2>&1
cat gpurun_out/r03q_sweep_lds_ablation.txt
