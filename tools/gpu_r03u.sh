cd /root/repo; mkdir -p gpurun_out
for v in sw_tl_nodma; do echo "== $v"; TNR_HIP_LIB=/root/repo/trainner_amd/lib/variants/lib$v.so TNR_MMA=bf16x3 timeout 300 python tools/probes/sweep_check.py 2>&1 | grep -v amdgpu.ids | tail -21; done > gpurun_out/r03u_dma_ablation.txt
cat gpurun_out/r03u_dma_ablation.txt
