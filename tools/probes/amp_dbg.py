import os, sys
os.environ.setdefault("TNR_MMA", "bf16x3")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trainner_amd import hip, ops
from tools.probes.sweep_check import block
ops.MMA = hip.MMA_BF16
for shape in [(1, 16, 32), (1, 24, 32)]:
    run = block(*shape, seed=11, grad_shape=False, with_r2=False)
    rb, ro, _ = run("layers")
    gb, go, _ = run("sweep")
    d = (gb - rb).abs()
    print(shape, "per group max", [float(d[..., lo:hi].max()) for lo, hi in ((0, 64), (64, 96), (96, 128), (128, 160), (160, 192))], "out", float((go - ro).abs().max()))
    for rep in range(3):
        g2, o2, _ = run("sweep")
        print("  rerun", rep, "identical to first sweep run:", bool(torch.equal(g2, gb)), " max|d| vs layers", float((g2 - rb).abs().max()))
    for lo, hi in ((96, 128),):
        dd = d[0, :, :, lo:hi].amax(-1)
        print("group x2 per-pixel-row max:", [round(float(v), 6) for v in dd.amax(1)])
        print("group x2 per-pixel-col max:", [round(float(v), 6) for v in dd.amax(0)])
    rel = ((gb - rb).abs() / (rb.abs() + 1e-3))[..., 64:96]
    print("x1 rel max", float(rel.max()), "count > 1e-4:", int((d[..., 64:96] > 1e-4).sum()), "of", d[..., 64:96].numel())
