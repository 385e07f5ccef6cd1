cd /root/repo; mkdir -p gpurun_out
TNR_HIP_LIB=/root/repo/trainner_amd/lib/variants/libsw_tl.so TNR_MMA=bf16x3 timeout 300 python tools/probes/sweep_check.py --time-only 2>&1 | grep -v amdgpu.ids | tail -26 > gpurun_out/r03ap_sweep4_units.txt
cat gpurun_out/r03ap_sweep4_units.txt
