"""conv_x3w8.h (TNR_MMA_BF16X3, 8-wave workgroup, both operands pre-split in LDS) against conv_tile_body's split-operand form:
bit-identical outputs.  The switch TNR_X3_W8 is read once per process:
    TNR_X3_W8=0 python tools/probes/x3w8_check.py save /tmp/x3ref.pt ; TNR_X3_W8=1 python tools/probes/x3w8_check.py compare /tmp/x3ref.pt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trainner_amd import hip, ops  # noqa: E402


def main(mode, path):
    dev = torch.device("cuda")
    ops.MMA = hip.MMA_BF16X3
    outs = []
    for (N, H, W, Cin, Cout, ct, co, seed) in ((2, 50, 70, 32, 96, 64, 16, 1), (1, 16, 32, 16, 64, 16, 0, 2), (3, 37, 33, 48, 128, 80, 32, 3),
                                               (2, 64, 64, 192, 64, 192, 0, 4), (1, 128, 96, 64, 64, 64, 0, 5)):
        g = torch.Generator().manual_seed(seed)
        xb = torch.randn(N, H, W, ct, generator=g).to(dev)
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05).to(dev)
        b = torch.randn(Cout, generator=g).to(dev)
        r2 = torch.randn(N, H, W, Cout, generator=g).to(dev)
        m = torch.randn(N, H, W, Cout, generator=g).to(dev)
        p = ops.WeightPacker(dev)
        i = p.add(w, ops.PACK_FWD)
        p.run()
        y = torch.full((N, H, W, Cout), 7.0, device=dev)
        ops.conv(ops.View(xb, co, Cin), p.get(i), ops.View(y), bias=b, act=ops.ACT_LRELU)
        y2 = torch.zeros(N, H, W, Cout, device=dev)
        ops.conv(ops.View(xb, co, Cin), p.get(i), ops.View(y2), r2=ops.View(r2), alpha2=0.5, mask=ops.View(m), m_lo=0, m_hi=Cout, m_slope=0.2)
        torch.cuda.synchronize()
        outs += [y.cpu(), y2.cpu()]
    if mode == "save":
        torch.save(outs, path)
        print("saved", len(outs))
        return 0
    ref = torch.load(path)
    bad = sum(0 if torch.equal(a, b) else 1 for a, b in zip(outs, ref))
    print("x3w8 vs conv_tile_body<BF=2>:", "BIT-IDENTICAL" if not bad else "%d of %d outputs differ (max %.3e)" % (
        bad, len(outs), max((a - b).abs().max().item() for a, b in zip(outs, ref))))
    return bad


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))
