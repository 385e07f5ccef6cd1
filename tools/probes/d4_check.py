"""conv3x3_d4_kernel (csrc/conv_sweep.hip: 64-cout 3x3 layers on the direct-sweep machinery, weights as a pre-split stream) against the
kernels it replaces (conv3x3_x3w8_kernel / conv_tile_kernel, ops.X3_D4 = False): bit equality over the benchmark's layer shapes with every
epilogue form, then timing of both.    TNR_MMA=bf16x3 [TNR_D4_OCC=1|2] python tools/probes/d4_check.py [--time-only]"""
import os
import sys

os.environ.setdefault("TNR_MMA", "bf16x3")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trainner_amd import hip, ops  # noqa: E402
from tools.microbench_conv import timeit  # noqa: E402

dev = torch.device("cuda")


def layer(N, H, W, Cin, Cout, seed, epi="plain", dgrad=False):
    g = torch.Generator().manual_seed(seed)
    w = (torch.rand((Cin, Cout, 3, 3) if dgrad else (Cout, Cin, 3, 3), generator=g) * 0.1 - 0.05).to(dev)
    p = ops.WeightPacker(dev)
    i = p.add(w, ops.PACK_DGRAD_3x3 if dgrad else ops.PACK_FWD)
    p.run()
    x = (torch.rand(N, H, W, Cin + 32, generator=g) * 2 - 1).to(dev)            # a channel window of a wider buffer
    r1, r2, m = [(torch.rand(N, H, W, Cout, generator=g) * 2 - 1).to(dev) for _ in range(3)]
    b = (torch.rand(Cout, generator=g) - 0.5).to(dev)
    kw = {"plain": dict(bias=b), "lrelu": dict(bias=b, act=ops.ACT_LRELU, slope=0.2), "res": dict(bias=b, alpha=0.2, r1=ops.View(r1), r2=ops.View(r2), alpha2=0.2),
          "mask": dict(mask=ops.View(m), m_slope=0.2), "noise": dict(bias=b, alpha=0.2, r1=ops.View(r1), noise=ops.Noise(0.1, ops.noise_key(3, 1, 4))),
          "reflect": dict(bias=b, act=ops.ACT_RELU, reflect=True)}[epi]

    def run(d4):
        """d4: True / False = the direct weight-stream kernel / the staged-weights kernels; "wino" = the Winograd form (csrc/conv_wino.hip)."""
        y = torch.full((N, H, W, Cout + 96), 3.0, device=dev)
        ops.X3_D4 = d4 is not False
        ops.conv(ops.View(x, 16, Cin), p.get(i), ops.View(y, 64, Cout), wino=(d4 == "wino"), **kw)
        ops.X3_D4 = True
        torch.cuda.synchronize()
        return y

    return run, p, lambda d4: (setattr(ops, "X3_D4", d4), ops.conv(ops.View(x, 16, Cin), p.get(i), ops.View(torch.empty((N, H, W, Cout), device=dev)), wino=False, **kw))


def main():
    amp = "--amp" in sys.argv           # TNR_MMA_BF16 (use_amp): the kernel's bf16-operand form against conv_tile_body<BF = 1> -- agreement, not equality
    if amp:
        ops.MMA = hip.MMA_BF16
    assert ops.MMA == (hip.MMA_BF16 if amp else hip.MMA_BF16X3)
    ok = True
    if "--time-only" not in sys.argv:
        cases = [(2, 40, 72, 64, 64, "lrelu", False), (1, 8, 32, 32, 64, "plain", False), (3, 17, 33, 128, 128, "res", False), (2, 64, 64, 256, 256, "mask", True),
                 (1, 32, 32, 512, 512, "plain", False), (16, 128, 128, 64, 64, "noise", False), (2, 96, 160, 64, 128, "lrelu", True), (1, 9, 45, 96, 192, "res", False), (2, 64, 64, 256, 256, "reflect", False), (1, 20, 37, 64, 64, "reflect", False)]
        for k, (N, H, W, Cin, Cout, epi, dg) in enumerate(cases):
            run, _, _ = layer(N, H, W, Cin, Cout, 50 + k, epi, dg)
            ref = run(False)
            for rep in range(2):
                got = run(True)
                same = bool(torch.equal(got, ref)) if not amp else float((got - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
                ok &= same
                if not same:
                    d = (got - ref).abs()
                    print("MISMATCH", (N, H, W, Cin, Cout, epi, dg), "rep", rep, "max|d| %.3e" % float(d.max()), "nbad", int((d > 0).sum()), "first", (d > 0).nonzero()[0].tolist())
            print((N, H, W, Cin, Cout, epi, "dgrad" if dg else "fwd"), "done", flush=True)
        print("BIT-EQUALITY", "OK" if ok else "FAILED")
    print("%-28s %10s %10s %9s %9s" % ("layer (batch 16)", "old us", "d4 us", "old TF/s", "d4 TF/s"))
    for (H, Cin, Cout) in ((512, 64, 64), (256, 128, 128), (128, 256, 256), (64, 512, 512), (32, 512, 512), (128, 64, 64), (256, 64, 128)):
        _, _, call = layer(16, H, H, Cin, Cout, 7, "lrelu")
        t0 = timeit(lambda: call(False))
        t1 = timeit(lambda: call(True))
        fl = 2.0 * 16 * H * H * 9 * Cin * Cout
        print("%-28s %10.1f %10.1f %9.1f %9.1f" % ("%dx%d %d->%d" % (H, H, Cin, Cout), t0, t1, fl / t0 / 1e6, fl / t1 / 1e6), flush=True)
    ops.X3_D4 = True


if __name__ == "__main__":
    main()
