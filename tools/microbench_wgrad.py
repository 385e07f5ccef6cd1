"""wgrad_tile micro-benchmark: time vs batch (= pixel tiles per workgroup) at fixed channel shape, to
separate the per-launch fixed cost (prologue, partial-slab write, reduce launch) from the per-tile cost.
python tools/microbench_wgrad.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trainner_amd import ops  # noqa: E402
from tools.microbench_conv import timeit  # noqa: E402


def main():
    dev = torch.device("cuda")
    H = W = 128
    print("%5s %5s %4s %9s %9s" % ("Cin", "Cout", "N", "us", "TFLOP/s"))
    for Cout, Cin in ((32, 64), (32, 96), (32, 128), (32, 32), (64, 192), (64, 64)):
        pts = []
        for N in (4, 8, 16, 32, 64):
            x = torch.randn(N, H, W, Cin, device=dev)
            g = torch.randn(N, H, W, Cout, device=dev)
            dw = torch.zeros(Cout, Cin, 3, 3, device=dev)
            db = torch.zeros(Cout, device=dev)
            us = timeit(lambda: ops.wgrad(ops.View(x), ops.View(g), dw, db), reps=10)
            fl = 2.0 * N * H * W * 9 * Cin * Cout
            pts.append((N, us))
            print("%5d %5d %4d %9.1f %9.1f" % (Cin, Cout, N, us, fl / us / 1e6))
        (n0, t0), (n1, t1) = pts[1], pts[-1]
        slope = (t1 - t0) / (n1 - n0)
        fl1 = 2.0 * H * W * 9 * Cin * Cout
        print("   fit: fixed %.1f us + %.2f us/image  (steady state %.1f TFLOP/s)" % (t0 - slope * n0, slope, fl1 / slope / 1e6))


if __name__ == "__main__":
    main()
