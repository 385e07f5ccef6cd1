"""LDS-DMA staging experiments for the 3x3 convolution (csrc/conv_body_dl.h) against the register-staged kernel: outputs on a set
of shapes (image borders, ragged tiles, channel-offset views, residual / mask epilogues; forms 1 / 4 / 5 must be bit-identical,
form 2 sums 8-channel chunks) and the time of every form per shape, interleaved.  Needs the experiment build:
    python tools/build_variant.py dl -DTNR_CONV_DL_EXPERIMENT
    TNR_HIP_LIB=trainner_amd/lib/variants/libdl.so python tools/probes/conv_dl_check.py
TNR_CONV_DL selects the form per call: 0 register-staged (the product), 1 eight-wave workgroup / 16-channel chunks / two operand
buffers, 2 four-wave workgroups / 8-channel chunks / two buffers (also inside tnr_conv_chain), 4 / 5 form 1 plus two / four
loader waves that issue every DMA piece."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trainner_amd import ops  # noqa: E402
from tools.microbench_conv import timeit  # noqa: E402


def run(dl, fn):
    os.environ["TNR_CONV_DL"] = str(int(dl))
    return fn()


def main():
    dev = torch.device("cuda")
    torch.manual_seed(1)
    bad = 0
    print("%-24s us: form 0 (product)  1  2  4  5  |  TFLOP/s: same order" % "shape N,H,W,Cin,Cout")
    for (N, H, W, Cin, Cout, ct, co) in ((16, 128, 128, 64, 32, 64, 0), (16, 128, 128, 160, 32, 192, 0), (16, 128, 128, 192, 32, 192, 0),
                                         (48, 128, 128, 64, 32, 64, 0), (2, 50, 70, 32, 32, 64, 16), (16, 128, 128, 64, 64, 64, 0), (16, 128, 128, 192, 64, 192, 0), (4, 512, 512, 64, 64, 64, 0),
                                         (48, 128, 128, 64, 64, 64, 0), (48, 64, 64, 128, 128, 128, 0), (48, 32, 32, 256, 256, 256, 0),
                                         (48, 16, 16, 512, 512, 512, 0), (2, 50, 70, 32, 96, 64, 16), (1, 16, 32, 16, 64, 16, 0),
                                         (3, 37, 33, 48, 128, 80, 32)):
        xb = torch.randn(N, H, W, ct, device=dev)
        w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        p = ops.WeightPacker(dev)
        i = p.add(w, ops.PACK_FWD)
        p.run()
        wp = p.get(i)
        xv = ops.View(xb, co, Cin) if (co or ct != Cin) else ops.View(xb)
        VAR = (0, 1, 2, 4, 5)
        ys = {v: torch.full((N, H, W, Cout), 7.0, device=dev) for v in VAR}
        fs = {v: (lambda v=v: ops.conv(xv, wp, ops.View(ys[v]), bias=b, act=ops.ACT_LRELU)) for v in VAR}
        for v in VAR:
            run(v, fs[v])
        torch.cuda.synchronize()
        # forms 1 / 4 keep the summation order of the register-staged kernel (bit-identical); form 2 sums 8-channel chunks
        exact = all(torch.equal(ys[0], ys[v]) for v in (1, 4, 5))
        close = (ys[0] - ys[2]).abs().max().item() <= 1e-4 * max(1.0, ys[0].abs().max().item())
        bad += 0 if (exact and close) else 1
        t = {v: 1e30 for v in VAR}
        for _ in range(3):                       # interleaved, best of 3: the clock ramps during the first launches
            for v in VAR:
                t[v] = min(t[v], run(v, lambda: timeit(fs[v])))
        fl = 2.0 * N * H * W * 9 * Cin * Cout
        print("%-24s " % ("%d,%d,%d,%d,%d" % (N, H, W, Cin, Cout)) + " ".join("%8.1f" % t[v] for v in VAR) + "  |  " +
              " ".join("%6.1f" % (fl / t[v] / 1e6) for v in VAR) + ("  ok" if exact and close else "  MISMATCH"))
    # residual + mask epilogue (the data-gradient form)
    N, H, W, C = 4, 64, 64, 64
    x = torch.randn(N, H, W, C, device=dev)
    r2 = torch.randn(N, H, W, C, device=dev)
    m = torch.randn(N, H, W, C, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    p = ops.WeightPacker(dev)
    i = p.add(w, ops.PACK_DGRAD_3x3)
    p.run()
    wp = p.get(i)
    outs = []
    for dl in (0, 1, 4):
        y = torch.zeros(N, H, W, C, device=dev)
        run(dl, lambda: ops.conv(ops.View(x), wp, ops.View(y), r2=ops.View(r2), alpha2=0.5, mask=ops.View(m), m_lo=0, m_hi=C, m_slope=0.2))
        outs.append(y)
    torch.cuda.synchronize()
    same = torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    bad += 0 if same else 1
    print("residual+mask epilogue:", "ok" if same else "MISMATCH")
    os.environ["TNR_CONV_DL"] = "0"
    print("FAILED" if bad else "ALL EXACT")
    return bad


if __name__ == "__main__":
    sys.exit(main())
