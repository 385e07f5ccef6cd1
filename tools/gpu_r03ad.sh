cd /root/repo; mkdir -p gpurun_out
( for d in 8 1; do echo "== TNR_SWEEP_DISPENSERS=$d"; TNR_SWEEP_DISPENSERS=$d TNR_MMA=bf16x3 timeout 300 python tools/probes/sweep_check.py --time-only 2>&1 | grep -E "sweep  "
  TNR_SWEEP_DISPENSERS=$d timeout 200 python tools/probes/sweep_hog.py 2>&1 | grep "TNR_SWEEP_WAVES"; done ) > gpurun_out/r03ad_sweep_hog2.txt 2>&1
cat gpurun_out/r03ad_sweep_hog2.txt
