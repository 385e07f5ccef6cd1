# PMC passes over one per-layer convolution shape (256 -> 64 and 512 -> 64 at 16 x 128 x 128): where do the matrix-pipe-idle cycles go
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/pmc_conv; mkdir -p $O
cat > /tmp/one_conv.py <<'P'
import sys, torch
sys.path.insert(0, "/root/repo")
from trainner_amd import ops
dev = torch.device("cuda")
N, H, W, Cin, Cout = 16, 128, 128, 512, 64
x = torch.randn(N, H, W, Cin, device=dev); y = torch.empty(N, H, W, Cout, device=dev)
w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05; b = torch.zeros(Cout, device=dev)
p = ops.WeightPacker(dev); i = p.add(w, ops.PACK_FWD); p.run(); wp = p.get(i)
for _ in range(5):
    ops.conv(ops.View(x), wp, ops.View(y), bias=b, act=ops.ACT_LRELU)
torch.cuda.synchronize()
P
for C in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU"; do
  N=$(echo $C | cut -d' ' -f1)
  rm -rf /tmp/pc_$N
  timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/pc_$N -- python /tmp/one_conv.py > $O/run_$N.log 2>&1
  python $R/tools/pmc_summary.py /tmp/pc_$N | grep -E "kernel|conv_tile" | head -6
done
