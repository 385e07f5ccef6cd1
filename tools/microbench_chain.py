"""Dense block forward: five tnr_conv_forward launches vs one tnr_conv_chain launch.  python tools/microbench_chain.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trainner_amd import ops  # noqa: E402
from tools.microbench_conv import timeit  # noqa: E402


def main():
    dev = torch.device("cuda")
    N, H, W, nf, gc = 16, 128, 128, 64, 32
    shapes = [(nf + k * gc, gc) for k in range(4)] + [(nf + 4 * gc, nf)]
    p = ops.WeightPacker(dev)
    idx = [p.add(torch.randn(co, ci, 3, 3, device=dev) * 0.05, ops.PACK_FWD) for ci, co in shapes]
    p.run()
    bs = [torch.zeros(co, device=dev) for _, co in shapes]
    buf = torch.randn(N, H, W, nf + 4 * gc, device=dev)
    out = torch.empty(N, H, W, nf, device=dev)
    st = []
    for k in range(4):
        cin = nf + gc * k
        st.append(dict(x=ops.View(buf, 0, cin), wp=p.get(idx[k]), y=ops.View(buf, cin, gc), bias=bs[k], act=ops.ACT_LRELU,
                       fresh_from=(cin - gc if k else None)))
    st.append(dict(x=ops.View(buf), wp=p.get(idx[4]), y=ops.View(out), bias=bs[4], alpha=0.2, r1=ops.View(buf, 0, nf),
                   fresh_from=nf + 3 * gc))
    fl = sum(2.0 * N * H * W * 9 * ci * co for ci, co in shapes)

    def per_layer():
        for d in st:
            ops.conv(**{k: v for k, v in d.items() if k != "fresh_from"})

    for name, fn in (("5 launches", per_layer), ("1 chain launch", lambda: ops.conv_chain(st)),
                     ("chain, first 4 stages", lambda: ops.conv_chain(st[:4])), ("4 launches", lambda: [ops.conv(**{k: v for k, v in d.items() if k != "fresh_from"}) for d in st[:4]])):
        us = timeit(fn)
        f = fl if "4" not in name else sum(2.0 * N * H * W * 9 * ci * co for ci, co in shapes[:4])
        print("%-24s %8.1f us  %6.1f TFLOP/s" % (name, us, f / us / 1e6))
    print("chain error flag:", ops.chain_error_flag())


if __name__ == "__main__":
    main()
