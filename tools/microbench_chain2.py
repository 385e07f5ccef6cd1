"""Marginal cost of a stage inside tnr_conv_chain: n identical 160->32 stages (a) independent (no waits),
(b) each waiting for its predecessor at channel 128, versus n separate launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trainner_amd import ops  # noqa: E402
from tools.microbench_conv import timeit  # noqa: E402


def main():
    dev = torch.device("cuda")
    N, H, W = 16, 128, 128
    cin, cout = 160, 32
    p = ops.WeightPacker(dev)
    i = p.add(torch.randn(cout, cin, 3, 3, device=dev) * 0.05, ops.PACK_FWD)
    p.run()
    b = torch.zeros(cout, device=dev)
    buf = torch.randn(N, H, W, 192, device=dev)
    outs = [torch.empty(N, H, W, cout, device=dev) for _ in range(6)]
    one = dict(x=ops.View(buf, 0, cin), wp=p.get(i), bias=b, act=ops.ACT_LRELU)
    fl1 = 2.0 * N * H * W * 9 * cin * cout
    t_launch = timeit(lambda: ops.conv(y=ops.View(outs[0]), **one), reps=20)
    print("1 launch: %.1f us (%.1f TFLOP/s)" % (t_launch, fl1 / t_launch / 1e6))
    for n in (1, 2, 4, 6):
        free = [dict(y=ops.View(outs[k]), fresh_from=None, **one) for k in range(n)]
        dep = [dict(y=ops.View(outs[k]), fresh_from=(128 if k else None), **one) for k in range(n)]
        ta = timeit(lambda: ops.conv_chain(free), reps=20)
        tb = timeit(lambda: ops.conv_chain(dep), reps=20)
        print("n=%d  chain(no waits) %7.1f us = %6.1f/stage   chain(waits) %7.1f us = %6.1f/stage   launches %7.1f us"
              % (n, ta, ta / n, tb, tb / n, n * t_launch))
    print("chain error flag:", ops.chain_error_flag())


if __name__ == "__main__":
    main()
