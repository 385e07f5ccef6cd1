"""Micro-benchmark of conv_tile / wgrad_tile launches: time vs K (input channels) at fixed output tile
work, to separate the per-launch fixed cost from the per-K-chunk cost.  python tools/microbench_conv.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trainner_amd import ops  # noqa: E402


def timeit(fn, reps=int(os.environ.get("TNR_REPS", "20"))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3   # us


def main():
    dev = torch.device("cuda")
    N, H, W = 16, 128, 128
    print("%-8s %5s %5s %9s %9s %8s" % ("kind", "Cin", "Cout", "us", "TFLOP/s", "chunks"))
    for Cout in (32, 64):
        for Cin in (16, 32, 64, 128, 192, 256, 512):
            x = torch.randn(N, H, W, Cin, device=dev)
            y = torch.empty(N, H, W, Cout, device=dev)
            w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
            b = torch.zeros(Cout, device=dev)
            p = ops.WeightPacker(dev)
            i = p.add(w, ops.PACK_FWD)
            p.run()
            wp = p.get(i)
            us = timeit(lambda: ops.conv(ops.View(x), wp, ops.View(y), bias=b, act=ops.ACT_LRELU))
            fl = 2.0 * N * H * W * 9 * Cin * Cout
            print("%-8s %5d %5d %9.1f %9.1f %8d" % ("conv", Cin, Cout, us, fl / us / 1e6, Cin // 16))
    for Cout, Cin in ((32, 64), (32, 96), (64, 64), (64, 192)):
        x = torch.randn(N, H, W, Cin, device=dev)
        g = torch.randn(N, H, W, Cout, device=dev)
        dw = torch.zeros(Cout, Cin, 3, 3, device=dev)
        db = torch.zeros(Cout, device=dev)
        us = timeit(lambda: ops.wgrad(ops.View(x), ops.View(g), dw, db))
        fl = 2.0 * N * H * W * 9 * Cin * Cout
        print("%-8s %5d %5d %9.1f %9.1f" % ("wgrad", Cin, Cout, us, fl / us / 1e6))
    # HBM-bound reference: axpby over a 200 MB buffer
    a = torch.randn(16, 128, 128, 192, device=dev)
    c = torch.randn(16, 128, 128, 192, device=dev)
    us = timeit(lambda: ops.axpby(ops.View(a), ops.View(c), 1.0, 1.0))
    print("axpby 201MB x3 traffic: %.1f us -> %.2f TB/s" % (us, 3 * a.numel() * 4 / us / 1e6))


if __name__ == "__main__":
    main()
