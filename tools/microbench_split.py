"""Does splitting the batch into two half-batch launch chains on two HIP streams hide the per-launch
prologue / epilogue bursts?  One chain of dense-block-like conv launches at N=16 versus two concurrent
chains at N=8.  python tools/microbench_split.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trainner_amd import ops  # noqa: E402


def chain(x, ys, packs, biases, reps):
    for _ in range(reps):
        for (cin, cout), y, wp, b in zip(SHAPES, ys, packs, biases):
            ops.conv(ops.View(x, 0, cin), wp, ops.View(y), bias=b, act=ops.ACT_LRELU)


SHAPES = [(64, 32), (96, 32), (128, 32), (160, 32), (192, 64)]


def main():
    dev = torch.device("cuda")
    H = W = 128
    p = ops.WeightPacker(dev)
    idx = [p.add(torch.randn(co, ci, 3, 3, device=dev) * 0.05, ops.PACK_FWD) for ci, co in SHAPES]
    p.run()
    packs = [p.get(i) for i in idx]
    biases = [torch.zeros(co, device=dev) for _, co in SHAPES]
    reps = 20

    def bufs(N):
        return torch.randn(N, H, W, 192, device=dev), [torch.empty(N, H, W, co, device=dev) for _, co in SHAPES]

    x16, y16 = bufs(16)
    xa, ya = bufs(8)
    xb, yb = bufs(8)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3 / reps

    def one():
        chain(x16, y16, packs, biases, reps)

    def two():
        ev = torch.cuda.Event()
        ev.record()
        for s, x, y in ((s1, xa, ya), (s2, xb, yb)):
            with torch.cuda.stream(s):
                s.wait_event(ev)
                chain(x, y, packs, biases, reps)
        for s in (s1, s2):
            e = torch.cuda.Event()
            e.record(s)
            torch.cuda.current_stream().wait_event(e)

    def half_serial():
        chain(xa, ya, packs, biases, reps)
        chain(xb, yb, packs, biases, reps)

    fl = sum(2.0 * 16 * H * W * 9 * ci * co for ci, co in SHAPES)
    for name, fn in (("one chain  N=16", one), ("two chains N=8+8 (2 streams)", two), ("two chains N=8,8 (serial)", half_serial)):
        us = timed(fn)
        print("%-32s %8.1f us per dense block  %6.1f TFLOP/s" % (name, us, fl / us / 1e6))


if __name__ == "__main__":
    main()
