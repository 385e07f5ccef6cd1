"""TNR_MMA_BF16X3 (fp32 operands split into three bf16 values, six partial products on the bf16 matrix core) against the fp32
matrix-core path: error of both against an fp64 reference on the same inputs, and the time of both on benchmark shapes.
python tools/probes/mma_x3_check.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trainner_amd import hip, ops  # noqa: E402
from tools.microbench_conv import timeit  # noqa: E402


def main():
    dev = torch.device("cuda")
    torch.manual_seed(3)
    print("== accuracy: max |y - y64| / max |y64| and rms relative error, fp32 MFMA vs bf16x3")
    for (N, H, W, Cin, Cout) in ((2, 32, 32, 64, 64), (1, 24, 40, 192, 32), (2, 16, 32, 512, 64)):
        x = torch.randn(N, Cin, H, W)
        w = torch.randn(Cout, Cin, 3, 3) * (1.0 / (3 * Cin ** 0.5))
        b = torch.randn(Cout)
        y64 = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        xb = x.permute(0, 2, 3, 1).contiguous().to(dev)
        p = ops.WeightPacker(dev)
        i = p.add(w.to(dev), ops.PACK_FWD)
        p.run()
        wp = p.get(i)
        res = {}
        for name, mma in (("f32", hip.MMA_F32), ("bf16x3", hip.MMA_BF16X3), ("bf16", hip.MMA_BF16)):
            ops.MMA = mma
            y = torch.zeros(N, H, W, Cout, device=dev)
            ops.conv(ops.View(xb), wp, ops.View(y), bias=b.to(dev))
            torch.cuda.synchronize()
            e = (y.cpu().permute(0, 3, 1, 2).double() - y64)
            res[name] = (e.abs().max().item() / y64.abs().max().item(), (e.pow(2).mean().sqrt() / y64.pow(2).mean().sqrt()).item())
        print("%-22s " % ("%d,%d,%d,%d,%d" % (N, H, W, Cin, Cout)) + "  ".join("%s max %.2e rms %.2e" % (k, v[0], v[1]) for k, v in res.items()))
    print("== time (us) and fp32-equivalent TFLOP/s: fp32 MFMA | bf16x3 | bf16")
    for (N, H, W, Cin, Cout, ct) in ((16, 128, 128, 64, 32, 64), (16, 128, 128, 160, 32, 192), (16, 128, 128, 192, 64, 192), (4, 512, 512, 64, 64, 64),
                                     (32, 128, 128, 256, 256, 256), (32, 64, 64, 512, 512, 512)):
        xb = torch.randn(N, H, W, ct, device=dev)
        w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        y = torch.zeros(N, H, W, Cout, device=dev)
        p = ops.WeightPacker(dev)
        i = p.add(w, ops.PACK_FWD)
        p.run()
        wp = p.get(i)
        xv = ops.View(xb, 0, Cin)
        f = lambda: ops.conv(xv, wp, ops.View(y), bias=b, act=ops.ACT_LRELU)   # noqa: E731
        t = {}
        for _ in range(2):
            for name, mma in (("f32", hip.MMA_F32), ("bf16x3", hip.MMA_BF16X3), ("bf16", hip.MMA_BF16)):
                ops.MMA = mma
                t[name] = min(t.get(name, 1e30), timeit(f))
        fl = 2.0 * N * H * W * 9 * Cin * Cout
        byt = 4.0 * N * H * W * (Cin + Cout)
        print("%-24s " % ("%d,%d,%d,%d,%d" % (N, H, W, Cin, Cout)) + "  ".join("%s %7.1f us %6.1f TF %4.2f TB/s" % (k, v, fl / v / 1e6, byt / v / 1e6) for k, v in t.items()))
    ops.MMA = hip.MMA_F32


if __name__ == "__main__":
    main()
