"""Which kernels of the step are held back by the power cap?  Each kernel family on random operands and on all-zero operands (same instruction
stream, same traffic), ~0.7 s of back-to-back launches each: a kernel whose time does not move is bound by its structure, one that speeds up
on zeros by the switching power of its operands.    TNR_MMA=bf16x3 python tools/probes/kernel_power.py"""
import os
import sys

os.environ.setdefault("TNR_MMA", "bf16x3")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trainner_amd import hip, ops  # noqa: E402

dev = torch.device("cuda")


def timed(fn, reps):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps


def rnd(shape, zero):
    return torch.zeros(shape, device=dev) if zero else (torch.rand(shape, device=dev) * 2 - 1)


def conv_case(H, Cin, Cout, zero):
    w = rnd((Cout, Cin, 3, 3), zero) * 0.05
    p = ops.WeightPacker(dev)
    i = p.add(w, ops.PACK_FWD)
    p.run()
    x, y = rnd((16, H, H, Cin), zero), torch.empty((16, H, H, Cout), device=dev)
    return (lambda: ops.conv(ops.View(x), p.get(i), ops.View(y), act=ops.ACT_LRELU)), 2.0 * 16 * H * H * 9 * Cin * Cout, (p, w)


def wgrad_case(H, Cin, Cout, zero):
    x, g = rnd((16, H, H, Cin), zero), rnd((16, H, H, Cout), zero)
    dw, db = torch.zeros(Cout, Cin, 3, 3, device=dev), torch.zeros(Cout, device=dev)
    return (lambda: ops.wgrad(ops.View(x), ops.View(g), dw, db, beta=0.0)), 2.0 * 16 * H * H * 9 * Cin * Cout, None


assert ops.MMA == hip.MMA_BF16X3
print("%-44s %10s %10s %8s" % ("kernel (batch 16)", "random us", "zeros us", "ratio"))
for name, make, args in (("conv3x3_d4 256->256 @128", conv_case, (128, 256, 256)), ("conv3x3_d4 64->64 @512", conv_case, (512, 64, 64)),
                         ("wgrad_tile 192->64 @128 (dense conv5)", wgrad_case, (128, 192, 64)), ("wgrad_tile 128->32 @128 (dense conv3/4 piece)", wgrad_case, (128, 128, 32)),
                         ("wgrad_tile 256->256 @128 (VGG/D class)", wgrad_case, (128, 256, 256))):
    res = []
    for zero in (False, True):
        fn, fl, keep = make(*args, zero)
        us = timed(fn, 20)
        reps = max(50, int(0.7e6 / us))
        res.append(timed(fn, reps))
    print("%-44s %10.1f %10.1f %8.3f   (%.1f -> %.1f TFLOP/s fp32-equivalent)" % (name, res[0], res[1], res[1] / res[0], fl / res[0] / 1e6, fl / res[1] / 1e6), flush=True)
