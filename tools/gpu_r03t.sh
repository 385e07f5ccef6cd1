cd /root/repo; mkdir -p gpurun_out
TNR_MMA=bf16x3 timeout 300 python tools/probes/sweep_check.py 2>&1 | grep -E "sweep|BIT|MISMATCH|error" | tail -5 > gpurun_out/r03t_sweep4.txt
TNR_HIP_LIB=/root/repo/trainner_amd/lib/variants/libsw_tl.so TNR_MMA=bf16x3 timeout 300 python tools/probes/sweep_check.py 2>&1 | grep -v amdgpu.ids | tail -21 >> gpurun_out/r03t_sweep4.txt
cat gpurun_out/r03t_sweep4.txt
