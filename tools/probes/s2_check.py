"""conv_s2_d4_kernel (csrc/conv_sweep.hip: the 4x4 stride-2 convolution and its data-gradient on the weight-stream machinery) against
conv_tile_kernel (ops.S2_D4 = False): bit equality over the discriminator's layer shapes with their epilogues, then timing of both.
    TNR_MMA=bf16x3 python tools/probes/s2_check.py [--time-only] [--amp]"""
import os
import sys

os.environ.setdefault("TNR_MMA", "bf16x3")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trainner_amd import hip, ops  # noqa: E402
from tools.microbench_conv import timeit  # noqa: E402

dev = torch.device("cuda")


def layer(N, Ho, Wo, Cin, Cout, seed, dgrad=False, epi="plain"):
    """forward: x [N, 2 Ho, 2 Wo, Cin] -> y [N, Ho, Wo, Cout];  dgrad: g [N, Ho, Wo, Cout] -> gx [N, 2 Ho, 2 Wo, Cin] (Cin / Cout: the FORWARD layer's)."""
    g = torch.Generator().manual_seed(seed)
    w = (torch.rand(Cout, Cin, 4, 4, generator=g) * 0.1 - 0.05).to(dev)
    p = ops.WeightPacker(dev)
    i = p.add(w, ops.PACK_DGRAD_S2 if dgrad else ops.PACK_FWD_S2D)
    p.run()
    if dgrad:
        x = (torch.rand(N, Ho, Wo, Cout + 32, generator=g) * 2 - 1).to(dev)
        xin, yc, yh, yw = ops.View(x, 16, Cout), Cin, 2 * Ho, 2 * Wo
    else:
        x = (torch.rand(N, 2 * Ho, 2 * Wo, Cin + 32, generator=g) * 2 - 1).to(dev)
        xin, yc, yh, yw = ops.View(x, 16, Cin), Cout, Ho, Wo
    m = (torch.rand(N, yh, yw, yc, generator=g) * 2 - 1).to(dev)
    b = (torch.rand(yc, generator=g) - 0.5).to(dev)
    kw = {"plain": dict(bias=b), "mask": dict(mask=ops.View(m), m_slope=0.2), "none": {}}[epi]
    mode = ops.DGRAD_4x4_S2 if dgrad else ops.CONV_4x4_S2

    def run(fast):
        y = torch.full((N, yh, yw, yc + 96), 3.0, device=dev)
        ops.S2_D4 = fast
        ops.conv(xin, p.get(i), ops.View(y, 64, yc), mode=mode, **kw)
        ops.S2_D4 = True
        torch.cuda.synchronize()
        return y

    def call(fast):
        ops.S2_D4 = fast
        ops.conv(xin, p.get(i), ops.View(torch.empty((N, yh, yw, yc), device=dev)), mode=mode, **kw)

    return run, p, call


def main():
    amp = "--amp" in sys.argv
    if amp:
        ops.MMA = hip.MMA_BF16
    ok = True
    if "--time-only" not in sys.argv:
        cases = [(2, 32, 32, 64, 64, False, "plain"), (1, 40, 72, 128, 128, False, "plain"), (3, 17, 33, 64, 128, False, "none"), (2, 32, 32, 512, 512, False, "plain"),
                 (2, 32, 32, 64, 64, True, "mask"), (1, 40, 72, 128, 128, True, "mask"), (3, 17, 33, 128, 64, True, "none"), (2, 32, 32, 512, 512, True, "mask"), (1, 8, 32, 32, 64, True, "plain")]
        for k, (N, Ho, Wo, Cin, Cout, dg, epi) in enumerate(cases):
            run, _, _ = layer(N, Ho, Wo, Cin, Cout, 60 + k, dg, epi)
            ref = run(False)
            for rep in range(2):
                got = run(True)
                same = bool(torch.equal(got, ref)) if not amp else float((got - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
                ok &= same
                if not same:
                    d = (got - ref).abs()
                    print("MISMATCH", (N, Ho, Wo, Cin, Cout, dg, epi), "rep", rep, "max|d| %.3e" % float(d.max()), "nbad", int((d > 0).sum()), "first", (d > 0).nonzero()[0].tolist(), "scale", float(ref.abs().max()))
            print((N, Ho, Wo, Cin, Cout, "dgrad" if dg else "fwd", epi), "done", flush=True)
        print("BIT-EQUALITY" if not amp else "AGREEMENT", "OK" if ok else "FAILED")
    print("%-34s %10s %10s %9s %9s" % ("layer (batch 16)", "old us", "new us", "old TF/s", "new TF/s"))
    for (Ho, Cin, Cout) in ((256, 64, 64), (128, 128, 128), (64, 256, 256), (32, 512, 512)):
        for dg in (False, True):
            _, _, call = layer(16, Ho, Ho, Cin, Cout, 7, dg, "mask" if dg else "plain")
            t0 = min(timeit(lambda: call(False)), timeit(lambda: call(False)))
            t1 = min(timeit(lambda: call(True)), timeit(lambda: call(True)))
            fl = 2.0 * 16 * Ho * Ho * 16 * Cin * Cout
            print("%-34s %10.1f %10.1f %9.1f %9.1f   x%.2f" % ("%s %d->%d out %dx%d" % ("dgrad" if dg else "fwd  ", Cin, Cout, Ho, Ho), t0, t1, fl / t0 / 1e6, fl / t1 / 1e6, t0 / t1), flush=True)
    ops.S2_D4 = True


if __name__ == "__main__":
    main()
