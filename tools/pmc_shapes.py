"""A handful of launches of the trunk's main kernel shapes, for rocprofv3 --pmc passes (tools/pmc_summary.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trainner_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda")
    N, H, W = 16, 128, 128
    x = torch.randn(N, H, W, 192, device=dev)
    g = torch.randn(N, H, W, 192, device=dev)
    for Cin, Cout in ((160, 32), (192, 64)):
        y = torch.empty(N, H, W, Cout, device=dev)
        w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
        b = torch.zeros(Cout, device=dev)
        p = ops.WeightPacker(dev)
        i = p.add(w, ops.PACK_FWD)
        p.run()
        for _ in range(6):
            ops.conv(ops.View(x, 0, Cin), p.get(i), ops.View(y), bias=b, act=ops.ACT_LRELU)
    for Cin, Cout in ((64, 32), (96, 32), (192, 64)):
        dw = torch.zeros(Cout, Cin, 3, 3, device=dev)
        db = torch.zeros(Cout, device=dev)
        for _ in range(6):
            ops.wgrad(ops.View(x, 0, Cin), ops.View(g, 0, Cout), dw, db)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
