"""tnr_conv_sweep (csrc/conv_sweep.hip) against per-layer launches and tnr_conv_chain, TNR_MMA=bf16x3: bit equality on forward-shaped
and gradient-shaped dense blocks (bias / LeakyReLU / residual / mask epilogues, ragged tiles, several rounds), then timing.
    TNR_MMA=bf16x3 python tools/probes/sweep_check.py"""
import os
import sys

os.environ.setdefault("TNR_MMA", "bf16x3")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trainner_amd import hip, ops  # noqa: E402
from tools.microbench_conv import timeit  # noqa: E402

dev = torch.device("cuda")
nf, gc = 64, 32


def block(N, H, W, seed, grad_shape, with_r2=True, noise=None):
    """noise: an ops.Noise for the last stage (ESRGAN+ multiplier: after the r1 step in the forward shape, after the r2 step in the gradient shape)."""
    nk = dict(noise=noise.at(2 if grad_shape else 1)) if noise is not None else {}
    g = torch.Generator().manual_seed(seed)
    shapes = [(nf + k * gc, gc) for k in range(4)] + [(nf + 4 * gc, nf)]
    p = ops.WeightPacker(dev)
    idx = [p.add((torch.rand(co, ci, 3, 3, generator=g) * 0.1 - 0.05).to(dev), ops.PACK_FWD) for ci, co in shapes]
    p.run()
    bs = [(torch.rand(co, generator=g) - 0.5).to(dev) for _, co in shapes]
    x0 = (torch.rand(N, H, W, nf, generator=g) * 2 - 1).to(dev)
    skip = (torch.rand(N, H, W, nf, generator=g) * 2 - 1).to(dev)
    maskbuf = (torch.rand(N, H, W, nf + 4 * gc, generator=g) * 2 - 1).to(dev)

    def make(buf, out):
        st = []
        for k in range(4):
            cin = nf + gc * k
            if grad_shape:       # gradient mirror: no bias, LeakyReLU' mask from another buffer
                st.append(dict(x=ops.View(buf, 0, cin), wp=p.get(idx[k]), y=ops.View(buf, cin, gc), fresh_from=(cin - gc if k else None),
                               mask=ops.View(maskbuf, nf + (3 - k) * gc, gc), m_lo=0, m_hi=gc, m_slope=0.2))
            else:
                st.append(dict(x=ops.View(buf, 0, cin), wp=p.get(idx[k]), y=ops.View(buf, cin, gc), bias=bs[k], act=ops.ACT_LRELU, slope=0.2,
                               fresh_from=(cin - gc if k else None)))
        if grad_shape:
            st.append(dict(x=ops.View(buf), wp=p.get(idx[4]), y=ops.View(out), fresh_from=nf + 3 * gc, r1=ops.View(buf, 0, nf), beta1=0.2,
                           **(dict(r2=ops.View(skip), alpha2=1.0) if with_r2 else {}), **nk))
        else:
            st.append(dict(x=ops.View(buf), wp=p.get(idx[4]), y=ops.View(out), bias=bs[4], alpha=0.2, r1=ops.View(buf, 0, nf),
                           fresh_from=nf + 3 * gc, **(dict(r2=ops.View(skip), alpha2=0.2) if with_r2 else {}), **nk))
        return st

    def run(how):
        buf = torch.zeros((N, H, W, nf + 4 * gc), device=dev)
        buf[..., :nf] = x0
        out = torch.zeros((N, H, W, nf), device=dev)
        st = make(buf, out)
        if how == "layers":
            for d in st:
                ops.conv(wino=False, **{k: v for k, v in d.items() if k != "fresh_from"})      # (the direct kernels: what the sweep is bit-identical to)
        else:
            ops.CONV_SWEEP = how == "sweep"
            ops.conv_chain(st)
            ops.CONV_SWEEP = True
        torch.cuda.synchronize()
        return buf, out, st

    return run


def main_amp():
    """--amp: TNR_MMA_BF16 (operands rounded to bf16).  The sweep's bf16-operand form against tnr_conv_chain and per-layer launches: the
    same rounded operands and fp32 accumulation, another order of the 16 products inside an MFMA.  The first stage therefore agrees to
    ~5e-7; downstream, an activation within that distance of a bf16 rounding boundary rounds the other way (one bf16 ulp = 2^-8 of its
    value, seen as a 3 x 3 patch of ~2e-4 in the next stage: tools/probes/amp_dbg.py) -- every form is a valid bf16-operand evaluation,
    they agree to bf16 resolution (4e-3 of the scale), and each is deterministic."""
    ops.MMA = hip.MMA_BF16
    ok = True
    for shape in [] if "--time-only" in sys.argv else [(4, 16, 16), (1, 16, 32), (2, 40, 72), (1, 8, 32), (1, 10, 20), (3, 128, 128), (20, 128, 128), (5, 64, 96)]:
        for grad_shape in (False, True):
            run = block(*shape, seed=11, grad_shape=grad_shape, with_r2=(shape[0] % 2 == 1), noise=(ops.Noise(0.1, ops.noise_key(1, 2, 3)) if shape[0] == 5 else None))
            rb, ro, _ = run("layers")
            for how in ("chain", "sweep"):
                gb, go, _ = run(how)
                eb, eo = float((gb - rb).abs().max()), float((go - ro).abs().max())
                sc = max(float(rb.abs().max()), float(ro.abs().max()), 1.0)
                first = float((gb[..., nf:nf + gc] - rb[..., nf:nf + gc]).abs().max())
                good = eb <= 4e-3 * sc and eo <= 4e-3 * sc and first <= 2e-6 * sc
                ok &= good
                print(shape, "grad" if grad_shape else "fwd", how, "max|d| buf %.2e out %.2e (scale %.2f)%s" % (eb, eo, sc, "" if good else "  <-- MISMATCH"), flush=True)
    print("AMP AGREEMENT", "OK" if ok else "FAILED", "; chain error flag", ops.chain_error_flag())
    N, H, W = 16, 128, 128
    fl = sum(2.0 * N * H * W * 9 * ci * co for ci, co in [(nf + k * gc, gc) for k in range(4)] + [(nf + 4 * gc, nf)])
    for grad_shape in (False, True):
        run = block(N, H, W, seed=5, grad_shape=grad_shape)
        _, _, st = run("layers")
        for how in ("chain", "sweep"):
            ops.CONV_SWEEP = how == "sweep"
            us = timeit(lambda: ops.conv_chain(st))
            print("amp %-5s %-6s %8.1f us  %6.1f TFLOP/s" % ("grad" if grad_shape else "fwd", how, us, fl / us / 1e6), flush=True)
        ops.CONV_SWEEP = True


def main():
    if "--amp" in sys.argv:
        return main_amp()
    ok = True
    assert ops.MMA == hip.MMA_BF16X3
    TIME_ONLY = "--time-only" in sys.argv
    for shape in [] if TIME_ONLY else [(4, 16, 16), (1, 16, 32), (2, 40, 72), (1, 8, 32), (2, 24, 24), (1, 10, 20), (4, 32, 32), (3, 128, 128), (20, 128, 128), (5, 64, 96)]:
        for grad_shape in (False, True):
            run = block(*shape, seed=11, grad_shape=grad_shape, with_r2=(shape[0] % 2 == 1))
            rb, ro, _ = run("layers")
            for rep in range(2):
                for how in ("chain", "sweep"):
                    gb, go, _ = run(how)
                    eb, eo = torch.equal(gb, rb), torch.equal(go, ro)
                    if not (eb and eo):
                        ok = False
                        db = (gb - rb).abs()
                        bad = (db > 0).nonzero()
                        first = bad[0].tolist() if len(bad) else None
                        per_group = [float(db[..., lo:hi].max()) for lo, hi in ((0, 64), (64, 96), (96, 128), (128, 160), (160, 192))]
                        print("MISMATCH", shape, "grad" if grad_shape else "fwd", how, "rep", rep, "buf max|d| per group", per_group,
                              "out max|d| %.3e" % float((go - ro).abs().max()), "first bad", first, "nbad", len(bad))
            print(shape, "grad" if grad_shape else "fwd", "done; chain error flag", ops.chain_error_flag(), flush=True)
    if not TIME_ONLY:
        print("BIT-EQUALITY", "OK" if ok else "FAILED")

    N, H, W = 16, 128, 128
    fl = sum(2.0 * N * H * W * 9 * ci * co for ci, co in [(nf + k * gc, gc) for k in range(4)] + [(nf + 4 * gc, nf)])
    for grad_shape in (False, True):
        for nz in (None, ops.Noise(0.1, ops.noise_key(1, 2, 3))):       # the ESRGAN+ multiplier in the last stage's epilogue: its cost
            run = block(N, H, W, seed=5, grad_shape=grad_shape, noise=nz)
            _, _, st = run("layers")
            for how in (("sweep",) if TIME_ONLY else ("chain", "sweep")):
                ops.CONV_SWEEP = how == "sweep"
                us = timeit(lambda: ops.conv_chain(st))
                print("%-5s %-6s %-6s %8.1f us  %6.1f TFLOP/s fp32-equivalent" % ("grad" if grad_shape else "fwd", how, "noise" if nz else "", us, fl / us / 1e6), flush=True)
            ops.CONV_SWEEP = True
    print("chain error flag:", ops.chain_error_flag())

    lib = hip.load()
    if hasattr(lib, "tnr_debug_sweep_timeline"):       # -DSW_TIMELINE probe build: where wave 0 of a workgroup spends its cycles
        import ctypes as C
        names = ["tile prologue", "neighbour wait", "A load issue", "vmcnt wait", "barrier", "DMA issue", "MFMA bodies (four-wave form: whole chunks, incl. 2 syncs + side work)",
                 "A split + LDS store", "epilogue", "  chunks with 3 N-tiles", "  chunks with 2 N-tiles", "  chunks with 1 N-tile", "", "", "kernel total", "slots"]
        run = block(16, 128, 128, seed=5, grad_shape=False)
        _, _, st = run("layers")
        ops.conv_chain(st)
        torch.cuda.synchronize()
        out = (C.c_ulonglong * 16)()
        lib.tnr_debug_sweep_timeline(out, 1)
        out2 = (C.c_ulonglong * 16)()
        if hasattr(lib, "tnr_debug_sweep_units"):
            lib.tnr_debug_sweep_units(out2, 1)
        reps = 5
        for _ in range(reps):
            ops.conv_chain(st)
        torch.cuda.synchronize()
        lib.tnr_debug_sweep_timeline(out, 0)
        wgs = 256 * reps
        print("sweep timeline: mean cycles per workgroup and launch (wave 0; 4 tiles per workgroup), %d slots" % (out[15] // wgs))
        for i, nm in enumerate(names):
            if nm and nm != "slots":
                print("   %-40s %10.0f  (%5.1f %%)" % (nm, out[i] / wgs, 100.0 * out[i] / max(out[14], 1)))
        if hasattr(lib, "tnr_debug_sweep_units"):
            lib.tnr_debug_sweep_units(out2, 0)
            print("four-wave form: mean cycles per unit (12 MFMAs = 384 cycles of matrix core) by unit class")
        names2 = ["unit with the slot synchronisation", "first two units of a slot (weight DMA)", "units with an input-chunk item", "plain units"]
        for i in range(8):
            if out2[8 + i]:
                print("   %d N-tiles: %-45s %8.0f   (%d units per workgroup and launch)" % (3 if i < 4 else 2, names2[i % 4], out2[i] / out2[8 + i], out2[8 + i] // wgs))


if __name__ == "__main__":
    main()
