# round 3: sweep kernel with the two-team slot schedule (default) against the one-schedule build and against teams = wave parity
cd /root/repo; mkdir -p gpurun_out
( for v in "" sw_noteams sw_team0 ""; do
    echo "== variant ${v:-default}"
    if [ -n "$v" ]; then export TNR_HIP_LIB=/root/repo/trainner_amd/lib/variants/lib$v.so; else unset TNR_HIP_LIB; fi
    TNR_MMA=bf16x3 timeout 300 python tools/probes/sweep_check.py 2>&1 | grep -v amdgpu.ids | tail -12
  done ) > gpurun_out/r03p_sweep_teams.txt 2>&1
cat gpurun_out/r03p_sweep_teams.txt
