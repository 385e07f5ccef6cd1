"""Do independent matrix-core launches on a SECOND stream fill the tails of the dense-block sweeps?  An RRDB's three gradient-shaped
sweep launches (stream A) and an RRDB's grouped weight-gradient launch (18 jobs of 64 x 64, stream B), timed alone, back to back on one
stream, and concurrently on two streams -- over ~0.3 s each (the part is power-capped: short runs flatter).
    TNR_MMA=bf16x3 python tools/probes/overlap_probe.py"""
import os
import sys

os.environ.setdefault("TNR_MMA", "bf16x3")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trainner_amd import hip, ops  # noqa: E402

dev = torch.device("cuda")
nf, gc, N, H, W = 64, 32, 16, 128, 128
g = torch.Generator().manual_seed(3)


def make_sweep():
    shapes = [(nf + k * gc, gc) for k in range(4)] + [(nf + 4 * gc, nf)]
    p = ops.WeightPacker(dev)
    idx = [p.add((torch.rand(co, ci, 3, 3, generator=g) * 0.1 - 0.05).to(dev), ops.PACK_FWD) for ci, co in shapes]
    p.run()
    buf = torch.zeros((N, H, W, nf + 4 * gc), device=dev)
    buf[..., :nf] = (torch.rand(N, H, W, nf, generator=g) * 2 - 1).to(dev)
    out = torch.zeros((N, H, W, nf), device=dev)
    st = []
    for k in range(4):
        cin = nf + gc * k
        st.append(dict(x=ops.View(buf, 0, cin), wp=p.get(idx[k]), y=ops.View(buf, cin, gc), act=ops.ACT_LRELU, slope=0.2, fresh_from=(cin - gc if k else None)))
    st.append(dict(x=ops.View(buf), wp=p.get(idx[4]), y=ops.View(out), alpha=0.2, r1=ops.View(buf, 0, nf), fresh_from=nf + 3 * gc))
    return st, p, buf


def make_wgrad(bufs):
    items = []
    keep = []
    for xb in bufs:
        GP = (torch.rand(N, H, W, 192, generator=g) * 2 - 1).to(dev)
        dws = [torch.zeros(gc if k < 4 else nf, nf + gc * k, 3, 3, device=dev) for k in range(5)]
        keep += [GP] + dws
        items.append(dict(x=ops.View(xb), g=ops.View(GP, 0, nf), dw=dws[4]))
        for ka, kb in ((3, 2), (1, 0)):
            cin = nf + gc * kb
            items.append(dict(x=ops.View(xb, 0, cin), g=ops.View(GP, nf + (3 - ka) * gc, 2 * gc), dw=dws[ka], pair=(dws[kb], None, gc)))
    return items, keep


sw = [make_sweep() for _ in range(3)]
items, keep = make_wgrad([s[2] for s in sw])
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()


def run_sweeps(reps):
    for _ in range(reps):
        for st, _, _ in sw:
            ops.conv_chain(st)


def run_wgrads(reps):
    for _ in range(reps):
        ops.wgrad_group(items)


def timed(fn_a, fn_b, reps, concurrent):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    if concurrent:
        sA.wait_stream(torch.cuda.current_stream())
        sB.wait_stream(torch.cuda.current_stream())
        # interleave the submissions so that neither queue runs dry
        for _ in range(reps):
            with torch.cuda.stream(sA):
                fn_a(1)
            with torch.cuda.stream(sB):
                fn_b(1)
        torch.cuda.current_stream().wait_stream(sA)
        torch.cuda.current_stream().wait_stream(sB)
    else:
        for _ in range(reps):
            if fn_a:
                fn_a(1)
            if fn_b:
                fn_b(1)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


assert ops.MMA == hip.MMA_BF16X3
run_sweeps(1)
torch.cuda.synchronize()
ref = [(st[-1]["y"].buf.clone(), b.clone()) for st, _, b in sw]          # single-stream results: block output, [x | x1 .. x4] buffer
for _ in range(2):
    run_sweeps(3); run_wgrads(3)
with torch.cuda.stream(sA):
    run_sweeps(2)
with torch.cuda.stream(sB):
    run_wgrads(2)
torch.cuda.synchronize()
print("latch after warm-up:", ops.chain_error_flag(), flush=True)
R = 100
ta = timed(run_sweeps, None, R, False)
print("latch after sweeps alone:", ops.chain_error_flag(), flush=True)
tb = timed(None, run_wgrads, R, False)
ts = timed(run_sweeps, run_wgrads, R, False)
print("latch after one-stream runs:", ops.chain_error_flag(), flush=True)
tc = timed(run_sweeps, run_wgrads, R, True)
print("latch after the first two-stream run:", ops.chain_error_flag(), flush=True)
ts2 = timed(run_sweeps, run_wgrads, R, False)
tc2 = timed(run_sweeps, run_wgrads, R, True)
print("per RRDB: 3 sweeps alone %.3f ms | wgrad group alone %.3f ms | one stream %.3f / %.3f ms | two streams %.3f / %.3f ms (%.1f %% of the serial time)"
      % (ta, tb, ts, ts2, tc, tc2, 100.0 * (tc + tc2) / (ts + ts2)))
print("fault latch:", ops.chain_error_flag())
torch.cuda.synchronize()
same = all(torch.equal(st[-1]["y"].buf, r[0]) and torch.equal(b, r[1]) for (st, _, b), r in zip(sw, ref))
print("results after the concurrent runs bit-identical to the single-stream ones:", same)
