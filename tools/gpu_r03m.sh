cd /root/repo; mkdir -p gpurun_out
timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --detail > gpurun_out/r03m_bench.json.log 2> gpurun_out/r03m_per_shape_table.txt
python - <<'P'
import json
j=json.loads(open("gpurun_out/r03m_bench.json.log").read().strip().split("\n")[-1])
print(j["value"], j["ms_per_step"], j["dtype"], j["roofline"]["achieved"], j["roofline"]["frac"], j["roofline"]["kernel_ms_per_step"], j.get("variant_f32_mfma"))
P
grep "wgrad_tile" gpurun_out/r03m_per_shape_table.txt | head -12
cd /tmp && export TMPDIR=/tmp
cat > /tmp/one_wgrad.py <<'P'
import sys, os, torch
sys.path.insert(0, "/root/repo")
os.environ["TNR_MMA"]="bf16x3"
from trainner_amd import ops
dev=torch.device("cuda")
N,H,W,Cin,Cout=16,128,128,192,64
x=torch.randn(N,H,W,Cin,device=dev); g=torch.randn(N,H,W,Cout,device=dev); dw=torch.zeros(Cout,Cin,3,3,device=dev); db=torch.zeros(Cout,device=dev)
for _ in range(4): ops.wgrad(ops.View(x),ops.View(g),dw,db)
torch.cuda.synchronize()
P
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU"; do
  N=$(echo $C | cut -d' ' -f1); rm -rf /tmp/pw_$N
  timeout 200 rocprofv3 --pmc $C --output-format csv -d /tmp/pw_$N -- python /tmp/one_wgrad.py > /dev/null 2>&1
  python /root/repo/tools/pmc_summary.py /tmp/pw_$N | grep -E "kernel|wgrad_tile" | head -10
done > /root/repo/gpurun_out/r03m_pmc_wgrad_192_64.txt
cat /root/repo/gpurun_out/r03m_pmc_wgrad_192_64.txt
