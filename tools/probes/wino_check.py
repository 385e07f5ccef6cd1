"""The Winograd F(2x2, 3x3) form of the 64-cout 3x3 layers (csrc/conv_wino.hip) against fp64 and against the direct weight-stream
kernel (conv3x3_d4_kernel), TNR_MMA=bf16x3: error ratios on small shapes (fp64 reference on the CPU), then timing on the step's shapes.
    TNR_MMA=bf16x3 python tools/probes/wino_check.py [--time-only]
VERDICT r5 item 1 (a): go if >= 1.35 x the direct kernel at conv5's shape (192 -> 64, 128^2, batch 16) and on the 64 -> 64 HR layer, with
error vs fp64 <= 3 x the fp32 matrix-core path's."""
import os
import sys

os.environ.setdefault("TNR_MMA", "bf16x3")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trainner_amd import hip, ops  # noqa: E402
from tools.microbench_conv import timeit  # noqa: E402

dev = torch.device("cuda")


def layer(N, H, W, Cin, Cout, seed, epi="plain", reflect=False):
    g = torch.Generator().manual_seed(seed)
    w = (torch.rand(Cout, Cin, 3, 3, generator=g) - 0.5) * (2.0 / (9 * Cin) ** 0.5) * 3.0
    b = (torch.rand(Cout, generator=g) - 0.5)
    x = torch.rand(N, H, W, Cin, generator=g) * 2 - 1
    r1 = torch.rand(N, H, W, Cout, generator=g) * 2 - 1
    r2 = torch.rand(N, H, W, Cout, generator=g) * 2 - 1
    m = torch.rand(N, H, W, Cout, generator=g) * 2 - 1
    p = ops.WeightPacker(dev)
    i = p.add(w.to(dev), ops.PACK_FWD)
    p.run()
    xd, bd, r1d, r2d, md = x.to(dev), b.to(dev), r1.to(dev), r2.to(dev), m.to(dev)
    kw = {}
    if epi == "plain":
        kw = dict(bias=bd, act=ops.ACT_LRELU, slope=0.2)
    elif epi == "res":
        kw = dict(bias=bd, alpha=0.2, r1=ops.View(r1d), r2=ops.View(r2d), alpha2=0.2)
    elif epi == "mask":
        kw = dict(mask=ops.View(md), m_lo=0, m_hi=Cout, m_slope=0.2, r1=ops.View(r1d), beta1=0.5)
    if reflect:
        kw["reflect"] = True

    def run(mode):
        y = torch.zeros(N, H, W, Cout, device=dev)
        prev = ops.MMA
        ops.MMA = hip.MMA_F32 if mode == "f32" else hip.MMA_BF16X3
        try:
            ops.conv(ops.View(xd), p.get(i), ops.View(y), wino=(mode == "wino"), **kw)
        finally:
            ops.MMA = prev
        torch.cuda.synchronize()
        return y

    def ref64():
        xx = x.double().permute(0, 3, 1, 2)
        if reflect:
            xx = torch.nn.functional.pad(xx, (1, 1, 1, 1), mode="reflect")
            v = torch.nn.functional.conv2d(xx, w.double(), None)
        else:
            v = torch.nn.functional.conv2d(xx, w.double(), None, padding=1)
        v = v.permute(0, 2, 3, 1)
        if epi == "plain":
            v = torch.nn.functional.leaky_relu(v + b.double(), 0.2)
        elif epi == "res":
            v = ((v + b.double()) * 0.2 + r1.double()) * 0.2 + r2.double()
        elif epi == "mask":
            v = (v + 0.5 * r1.double()) * torch.where(m.double() > 0, 1.0, 0.2)
        return v

    return run, ref64


def one(which, reps=10):
    """--one d4|wino: `reps` launches of conv5's shape in one form (for rocprofv3 --pmc passes)."""
    N, H, W, Cin, Cout = 16, 128, 128, 192, 64
    g = torch.Generator().manual_seed(3)
    w = ((torch.rand(Cout, Cin, 3, 3, generator=g) - 0.5) * 0.1).to(dev)
    x = torch.randn(N, H, W, Cin, device=dev)
    y = torch.empty(N, H, W, Cout, device=dev)
    p = ops.WeightPacker(dev)
    i = p.add(w, ops.PACK_FWD)
    p.run()
    lib = hip.load()
    tl = hasattr(lib, "tnr_debug_wino_timeline") and which == "wino"
    if tl:
        import ctypes as C
        ops.conv(ops.View(x), p.get(i), ops.View(y), act=ops.ACT_LRELU, slope=0.2, wino=True)
        torch.cuda.synchronize()
        out = (C.c_ulonglong * 32)()
        lib.tnr_debug_wino_timeline(out, 1)
    for _ in range(reps):
        ops.conv(ops.View(x), p.get(i), ops.View(y), act=ops.ACT_LRELU, slope=0.2, wino=(which == "wino"))
    torch.cuda.synchronize()
    if tl:      # -DWN_TIMELINE probe build: where waves 0 / 4 of a workgroup spend their cycles (per workgroup and launch)
        lib.tnr_debug_wino_timeline(out, 0)
        names = ["transform section", "barrier after transform", "multiply section", "barrier after multiply", "chunk loop (per tile sum)", "kernel total", "", "chunks"]
        for h in range(2):
            n = 256.0 * reps
            o = out[h * 16:(h + 1) * 16]
            print("half %d: chunks per workgroup %d" % (h, o[7] / n))
            for k, nm in enumerate(names[:6]):
                print("   %-28s %10.0f cycles per workgroup and launch   %8.0f per chunk" % (nm, o[k] / n, o[k] / max(o[7], 1)))
            ch = float(max(o[7], 1))      # stamps 8 .. 12 are SUMS of absolute times: differences of sums = summed intervals
            print("   inside the transform, per chunk: store raw + request weights %.0f | raw LDS reads landed %.0f | row step %.0f | first 4 positions split + stored %.0f | last 4 %.0f"
                  % ((o[8] - o[13]) / ch, (o[9] - o[8]) / ch, (o[10] - o[9]) / ch, (o[11] - o[10]) / ch, (o[12] - o[11]) / ch))


def main():
    assert ops.MMA == hip.MMA_BF16X3
    if "--one" in sys.argv:
        return one(sys.argv[sys.argv.index("--one") + 1])
    ok = True
    if "--time-only" not in sys.argv:
        cases = [(2, 16, 16, 32, 64, "plain", False), (1, 40, 72, 64, 64, "plain", False), (2, 24, 40, 192, 64, "res", False),
                 (1, 17, 33, 64, 128, "mask", False), (3, 32, 32, 128, 128, "plain", True), (1, 9, 11, 32, 64, "res", True),
                 (1, 64, 64, 512, 64, "plain", False)]
        for (N, H, W, Cin, Cout, epi, refl) in cases:
            run, ref64 = layer(N, H, W, Cin, Cout, seed=7 + Cin + H, epi=epi, reflect=refl)
            r = ref64()
            scale = float(r.abs().max())
            e = {}
            for mode in ("f32", "d4", "wino"):
                y = run(mode).double().cpu()
                d = (y - r).abs()
                e[mode] = (float(d.max()), float(d.pow(2).mean().sqrt()))
            good = e["wino"][0] <= 3.0 * e["f32"][0] + 2e-7 * scale and e["wino"][1] <= 3.0 * e["f32"][1] + 2e-7 * scale
            ok &= good
            print("%-28s scale %.2f  max|err| f32 %.2e d4 %.2e wino %.2e (x%.2f)   rms f32 %.2e d4 %.2e wino %.2e (x%.2f)%s" % (
                str((N, H, W, Cin, Cout, epi, "refl" if refl else "zero")), scale, e["f32"][0], e["d4"][0], e["wino"][0], e["wino"][0] / max(e["f32"][0], 1e-30),
                e["f32"][1], e["d4"][1], e["wino"][1], e["wino"][1] / max(e["f32"][1], 1e-30), "" if good else "   <-- ABOVE 3 x"), flush=True)
        print("WINOGRAD ERROR BOUND", "OK" if ok else "FAILED")
    # timing on the step's shapes (random operands: the part sits at its power cap in this arithmetic)
    shapes = [("RDB conv5      ", 16, 128, 128, 192, 64), ("HR conv 64->64 ", 16, 512, 512, 64, 64), ("LR conv 64->64 ", 16, 128, 128, 64, 64),
              ("up1 64->64 @256", 16, 256, 256, 64, 64), ("VGG 128->128   ", 16, 256, 256, 128, 128), ("VGG 256->256   ", 16, 128, 128, 256, 256),
              ("VGG 512->512@64", 16, 64, 64, 512, 512), ("VGG 512->512@32", 16, 32, 32, 512, 512), ("D 64->128 @256 ", 16, 256, 256, 64, 128)]
    for name, N, H, W, Cin, Cout in shapes:
        g = torch.Generator().manual_seed(3)
        w = ((torch.rand(Cout, Cin, 3, 3, generator=g) - 0.5) * 0.1).to(dev)
        b = torch.zeros(Cout, device=dev)
        x = torch.randn(N, H, W, Cin, device=dev)
        y = torch.empty(N, H, W, Cout, device=dev)
        p = ops.WeightPacker(dev)
        i = p.add(w, ops.PACK_FWD)
        p.run()
        fl = 2.0 * N * H * W * 9 * Cin * Cout
        t = {}
        for mode in ("d4", "wino", "d4", "wino"):
            us = timeit(lambda: ops.conv(ops.View(x), p.get(i), ops.View(y), bias=b, act=ops.ACT_LRELU, slope=0.2, wino=(mode == "wino")))
            t.setdefault(mode, []).append(us)
        d4, wn = min(t["d4"]), min(t["wino"])
        print("%s N%d %dx%d %d->%d   d4 %8.1f us (%6.1f TF)   wino %8.1f us (%6.1f TF algorithmic, frac %.3f of 419.4)   x%.2f" % (
            name, N, H, W, Cin, Cout, d4, fl / d4 / 1e6, wn, fl / wn / 1e6, fl / wn / 1e6 / 419.43, d4 / wn), flush=True)
    print("chain error flag:", ops.chain_error_flag())


if __name__ == "__main__":
    main()
