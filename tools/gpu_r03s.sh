# round 3: the four-wave sweep kernel (default) against the eight-wave form (TNR_SWEEP_WAVES=8)
cd /root/repo; mkdir -p gpurun_out
( for w in 4 8 4; do
    echo "== TNR_SWEEP_WAVES=$w"
    TNR_SWEEP_WAVES=$w TNR_MMA=bf16x3 timeout 300 python tools/probes/sweep_check.py 2>&1 | grep -E "sweep|BIT|MISMATCH|error" | tail -6
  done ) > gpurun_out/r03s_sweep4.txt 2>&1
cat gpurun_out/r03s_sweep4.txt
