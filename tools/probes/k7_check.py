"""Timing of the 7x7 image-side launches of config 5 (ResnetGenerator, ResNet_arch.py:52-55, :86-88) at the benchmark's shape
(batch 16, 256 x 256, ngf 64): TNR_CONV_7x7_C4 (forward 3 -> 64 and the 64 <- 3 data-gradient on the padded grid), tnr_conv_thin7
(forward 64 -> 3 and the 3 <- 64 data-gradient), tnr_wgrad_thin7 (both sides).    TNR_MMA=bf16x3 python tools/probes/k7_check.py"""
import os
import sys

os.environ.setdefault("TNR_MMA", "bf16x3")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trainner_amd import ops  # noqa: E402
from tools.microbench_conv import timeit  # noqa: E402

dev = torch.device("cuda")
N, H, W, C = 16, 256, 256, 64


def main():
    g = torch.Generator().manual_seed(1)
    w_in = ((torch.rand(C, 3, 7, 7, generator=g) - 0.5) * 0.1).to(dev)
    w_out = ((torch.rand(3, C, 7, 7, generator=g) - 0.5) * 0.1).to(dev)
    x4 = torch.rand(N, H, W, 4, generator=g).to(dev)
    x4[..., 3] = 0
    xp4 = torch.rand(N, H + 6, W + 6, 4, generator=g).to(dev)
    xp4[..., 3] = 0
    f = torch.rand(N, H, W, C, generator=g).to(dev)
    fp = torch.empty(N, H + 6, W + 6, C, device=dev)
    z = torch.empty(N, H, W, C, device=dev)
    o4 = torch.empty(N, H, W, 4, device=dev)
    op4 = torch.empty(N, H + 6, W + 6, 4, device=dev)
    p = ops.WeightPacker(dev)
    i_in, i_out = p.add(w_in, ops.PACK_C4_FWD), p.add(w_out, ops.PACK_C4_DGRAD3)
    p.run()
    dw_in, db_in, dw_out = torch.zeros_like(w_in), torch.zeros(C, device=dev), torch.zeros_like(w_out)
    V = ops.View
    px, pxp = N * H * W, N * (H + 6) * (W + 6)
    rows = [
        ("conv 7x7_c4 fwd 3->64 (reflect)", lambda: ops.conv(V(x4), p.get(i_in), V(z), mode=ops.CONV_7x7_C4, reflect=True), 2.0 * px * 49 * 3 * C, px * C * 4),
        ("conv 7x7_c4 dgrad 3->64 (padded grid)", lambda: ops.conv(V(xp4), p.get(i_out), V(fp), mode=ops.CONV_7x7_C4), 2.0 * pxp * 49 * 3 * C, pxp * C * 4),
        ("conv_thin7 fwd 64->3 (reflect)", lambda: ops.conv_thin7(V(f), w_out, V(o4, 0, 3), pad=3, reflect=True), 2.0 * px * 49 * 3 * C, px * C * 4),
        ("conv_thin7 dgrad 64->3 (pad 6)", lambda: ops.conv_thin7(V(f), w_in, V(op4, 0, 3), pad=6, reflect=False, dgrad=True), 2.0 * pxp * 49 * 3 * C, px * C * 4),
        ("wgrad_thin7 image->64", lambda: ops.wgrad_thin7(V(f), V(xp4), dw_in, db_in, flip=False, beta=0.0), 2.0 * px * 49 * 3 * C, px * C * 4),
        ("wgrad_thin7 64->image (reflected reads)", lambda: ops.wgrad_thin7(V(f), V(x4), dw_out, None, flip=True, rpad=3, off=-6, beta=0.0), 2.0 * pxp * 49 * 3 * C, px * C * 4),
    ]
    for name, fn, fl, by in rows:
        us = min(timeit(fn), timeit(fn))
        print("%-42s %8.1f us   %6.1f TFLOP/s (3 of 4 image channels counted)   wide tensor once: %5.2f TB/s" % (name, us, fl / us / 1e6, by / us / 1e6), flush=True)


if __name__ == "__main__":
    main()
