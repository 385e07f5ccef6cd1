"""Is the dense-block sweep's sustained rate set by the DATA it multiplies?  The same launches (same instruction stream, same memory traffic) on
random operands, on all-zero activations, and on all-zero activations AND weights, timed over a ~1 s run each (the part is at its power cap
in this arithmetic: DESIGN.md 3.1).    TNR_MMA=bf16x3 python tools/probes/sweep_power.py"""
import os
import sys

os.environ.setdefault("TNR_MMA", "bf16x3")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trainner_amd import hip, ops  # noqa: E402

dev = torch.device("cuda")
nf, gc, N, H, W = 64, 32, 16, 128, 128
fl = sum(2.0 * N * H * W * 9 * ci * co for ci, co in [(nf + k * gc, gc) for k in range(4)] + [(nf + 4 * gc, nf)])


def make(zero_x, zero_w):
    g = torch.Generator().manual_seed(3)
    shapes = [(nf + k * gc, gc) for k in range(4)] + [(nf + 4 * gc, nf)]
    p = ops.WeightPacker(dev)
    idx = [p.add(((torch.rand(co, ci, 3, 3, generator=g) * 0.1 - 0.05) * (0.0 if zero_w else 1.0)).to(dev), ops.PACK_FWD) for ci, co in shapes]
    p.run()
    buf = torch.zeros((N, H, W, nf + 4 * gc), device=dev)
    if not zero_x:
        buf[..., :nf] = (torch.rand(N, H, W, nf, generator=g) * 2 - 1).to(dev)
    out = torch.zeros((N, H, W, nf), device=dev)
    st = []
    for k in range(4):
        cin = nf + gc * k
        st.append(dict(x=ops.View(buf, 0, cin), wp=p.get(idx[k]), y=ops.View(buf, cin, gc), act=ops.ACT_LRELU, slope=0.2, fresh_from=(cin - gc if k else None)))
    st.append(dict(x=ops.View(buf), wp=p.get(idx[4]), y=ops.View(out), alpha=0.2, r1=ops.View(buf, 0, nf), fresh_from=nf + 3 * gc))
    return st, p


def timed(st, reps):
    for _ in range(20):
        ops.conv_chain(st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        ops.conv_chain(st)
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps


assert ops.MMA == hip.MMA_BF16X3
for name, zx, zw in (("random activations, random weights", False, False), ("zero activations, random weights", True, False), ("zero activations, zero weights", True, True),
                     ("random activations, random weights (again)", False, False)):
    st, keep = make(zx, zw)
    short, long_ = timed(st, 20), timed(st, 1800)
    print("%-46s 20 launches: %6.1f us   1800 launches (~1 s): %6.1f us = %5.1f TFLOP/s fp32-equivalent" % (name, short, long_, fl / long_ / 1e6), flush=True)
