"""Staleness stress for tnr_conv_chain: the SAME dense-block buffers are re-used launch after launch with NEW data
(as the rotating gradient buffers of the backward pass are), interleaved with an L2-thrashing copy and with per-layer
launches that read the same addresses.  Every launch must equal five per-layer launches bit for bit.
python tools/chain_stress.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trainner_amd import ops  # noqa: E402


def main(reps=24):
    dev = torch.device("cuda")
    torch.manual_seed(0)
    bad = 0
    for (N, H, W) in ((16, 128, 128), (20, 128, 128), (3, 72, 40)):
        nf, gc = 64, 32
        shapes = [(nf + k * gc, gc) for k in range(4)] + [(nf + 4 * gc, nf)]
        p = ops.WeightPacker(dev)
        idx = [p.add(torch.randn(co, ci, 3, 3, device=dev) * 0.05, ops.PACK_FWD) for ci, co in shapes]
        p.run()
        bs = [torch.randn(co, device=dev) * 0.1 for _, co in shapes]
        bufs = [torch.zeros(N, H, W, nf + 4 * gc, device=dev) for _ in range(2)]     # [0] chain, [1] per-layer reference
        outs = [torch.zeros(N, H, W, nf, device=dev) for _ in range(2)]
        thrash = torch.empty(64 * 1024 * 1024, device=dev)

        def stages(buf, out):
            st = []
            for k in range(4):
                cin = nf + gc * k
                st.append(dict(x=ops.View(buf, 0, cin), wp=p.get(idx[k]), y=ops.View(buf, cin, gc), bias=bs[k], act=ops.ACT_LRELU,
                               fresh_from=(cin - gc if k else None)))
            st.append(dict(x=ops.View(buf), wp=p.get(idx[4]), y=ops.View(out), bias=bs[4], alpha=0.2, r1=ops.View(buf, 0, nf),
                           fresh_from=nf + 3 * gc))
            return st

        for rep in range(reps):
            x0 = torch.randn(N, H, W, nf, device=dev)
            for b in bufs:
                b[..., :nf] = x0                                   # new data at the same addresses; x1..x4 keep the OLD values
            if rep % 3 == 1:
                thrash.fill_(float(rep))                           # evict / dirty the L2s between launches
            ops.conv_chain(stages(bufs[0], outs[0]))
            for d in stages(bufs[1], outs[1]):
                ops.conv(**{k: v for k, v in d.items() if k != "fresh_from"})
            torch.cuda.synchronize()
            nb = int((bufs[0] != bufs[1]).sum().item()) + int((outs[0] != outs[1]).sum().item())
            if nb:
                bad += 1
                print("MISMATCH shape", (N, H, W), "rep", rep, "elements", nb)
    print("chain_stress:", "OK" if bad == 0 and ops.chain_error_flag() == 0 else "FAILED", "| error flag", ops.chain_error_flag())


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 24)
