cd /root/repo; mkdir -p gpurun_out
( for v in wg_noitems wg_noreads; do echo "== $v"; TNR_HIP_LIB=/root/repo/trainner_amd/lib/variants/lib$v.so TNR_MMA=bf16x3 timeout 200 python tools/microbench_wgrad.py 2>&1 | grep -E "^  (192|128) .* 16 |Cin"; done ) > gpurun_out/r03y_wgrad_ablation.txt 2>&1
cat gpurun_out/r03y_wgrad_ablation.txt
