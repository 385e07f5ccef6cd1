cd /root/repo; mkdir -p gpurun_out
( TNR_MMA=bf16x3 timeout 300 python tools/probes/sweep_check.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r03g_sweep_check.txt 2>&1
cat gpurun_out/r03g_sweep_check.txt | tail -40
( TNR_MMA=bf16x3 timeout 300 python -m pytest tests/test_gpu_step.py -q -x -k "k10" 2>&1 | tail -5 )
( TNR_MMA=bf16x3 TNR_CONV_SWEEP=0 timeout 300 python -m pytest tests/test_gpu_step.py -q -x -k "k10" 2>&1 | tail -5 )
