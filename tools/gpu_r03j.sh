cd /root/repo; mkdir -p gpurun_out
( for m in bf16x3 f32; do echo "== TNR_MMA=$m"; TNR_MMA=$m timeout 300 python tools/chain_stress.py 8 --rccl 2>&1 | grep "chain_stress\|dense-block\|MISMATCH\|Error\|error"; done ) > gpurun_out/r03j_chain_stress_rccl.txt 2>&1
cat gpurun_out/r03j_chain_stress_rccl.txt
