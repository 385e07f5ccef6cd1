"""Staleness stress for tnr_conv_chain: the SAME dense-block buffers are re-used launch after launch with NEW data
(as the rotating gradient buffers of the backward pass are), interleaved with an L2-thrashing copy and with per-layer
launches that read the same addresses.  Every launch must equal five per-layer launches bit for bit.
python tools/chain_stress.py [reps] [--rccl]
--rccl: every dense-block launch (tnr_conv_chain, or tnr_conv_sweep in TNR_MMA=bf16x3) runs WHILE an RCCL all-reduce of a 64 MB buffer
is in flight on a side stream (1-rank nccl group: the collective is a device kernel competing for the CUs the launch wants all of).
Measures whether the one-launch forms stay correct and how much they slow down next to a collective -- the question behind
ops.COLLECTIVES_IN_FLIGHT (dp.py), which sends dense blocks through per-layer launches while gradient buckets are on the wire."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trainner_amd import ops  # noqa: E402


def main(reps=24, rccl=False):
    dev = torch.device("cuda")
    torch.manual_seed(0)
    bad = 0
    side, grad = None, None
    if rccl:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29631")
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
        side, grad = torch.cuda.Stream(), torch.randn(16 * 1024 * 1024, device=dev)
        t_alone = t_with = 0.0
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
            if rccl:
                import torch.distributed as dist
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev = torch.cuda.Event()
                ev.record()
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    for _ in range(4):
                        dist.all_reduce(grad, op=dist.ReduceOp.AVG)
                e0.record()
                ops.conv_chain(stages(bufs[0], outs[0]))            # (COLLECTIVES_IN_FLIGHT is NOT set here: the one-launch form runs)
                e1.record()
                torch.cuda.synchronize()
                t_with += e0.elapsed_time(e1)
                b0 = bufs[0].clone()
                bufs[0][..., nf:] = 0
                e0.record()
                ops.conv_chain(stages(bufs[0], outs[0]))
                e1.record()
                torch.cuda.synchronize()
                t_alone += e0.elapsed_time(e1)
                assert torch.equal(b0, bufs[0])
            else:
                ops.conv_chain(stages(bufs[0], outs[0]))
            for d in stages(bufs[1], outs[1]):
                ops.conv(**{k: v for k, v in d.items() if k != "fresh_from"})
            torch.cuda.synchronize()
            nb = int((bufs[0] != bufs[1]).sum().item()) + int((outs[0] != outs[1]).sum().item())
            if nb:
                bad += 1
                print("MISMATCH shape", (N, H, W), "rep", rep, "elements", nb)
    print("chain_stress%s:" % (" --rccl" if rccl else ""), "OK" if bad == 0 and ops.chain_error_flag() == 0 else "FAILED",
          "| error flag", ops.chain_error_flag())
    if rccl:
        print("dense-block launch next to a 4 x 64 MB RCCL all-reduce on a side stream: %.3f ms per launch, alone %.3f ms (all shapes, %d launches)"
              % (t_with / (3 * reps), t_alone / (3 * reps), 3 * reps))
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    a = [x for x in sys.argv[1:] if not x.startswith("--")]
    main(int(a[0]) if a else 24, rccl="--rccl" in sys.argv)
