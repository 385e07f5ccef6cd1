"""Randomised cross-check of the launch forms added in round 6 against the forms they replace, on shapes and epilogues the fixed test
cases do not enumerate (TNR_MMA=bf16x3, one GPU):
  * Winograd form (csrc/conv_wino.hip) vs the direct kernels: max |d| <= 3e-6 of the scale, nothing outside the channel window touched;
  * four-tap weight-stream kernel (conv_s2_d4_kernel) vs conv_tile_kernel: bit for bit;
  * pixel-shuffle store (tnr_conv_desc.shuffle) vs conv + tnr_depth_to_space: bit for bit;
  * the dense-block sweep (packed operand split in its stagers) vs one launch per layer: bit for bit, ragged tilings, both shapes;
  * config 5's one-launch 7x7 image-side kernels vs torch in fp64 (2e-5 of the scale).
    TNR_MMA=bf16x3 python tools/probes/fuzz_round6.py [cases] [seed]
Prints one line per failing case and a summary; exit status 1 on any failure."""
import os
import random
import sys

os.environ.setdefault("TNR_MMA", "bf16x3")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trainner_amd import hip, ops  # noqa: E402

dev = torch.device("cuda")
EPIS = ("plain", "lrelu", "res", "mask", "noise", "reflect")


def wino_case(rng, k):
    from tools.probes.d4_check import layer
    N = rng.choice((1, 1, 2, 3))
    H, W = rng.randint(8, 70), rng.randint(8, 90)
    Cin = 16 * rng.randint(2, 12)
    Cout = 64 * rng.randint(1, 3)
    dg = rng.random() < 0.4
    epi = rng.choice(EPIS)
    run, _, _ = layer(N, H, W, Cin, Cout, 1000 + k, epi, dg)
    ref = run(True)
    got = run("wino")
    sc = max(1.0, float(ref[..., 64:64 + Cout].abs().max()))
    err = float((got - ref).abs().max()) / sc
    clean = float(got[..., :64].min()) == 3.0 and float(got[..., :64].max()) == 3.0 and float(got[..., 64 + Cout:].min()) == 3.0 and float(got[..., 64 + Cout:].max()) == 3.0
    again = torch.equal(run("wino"), got)
    return ("wino", (N, H, W, Cin, Cout, epi, "dgrad" if dg else "fwd")), (err <= 3e-6 and clean and again), "err/scale %.2e clean %s deterministic %s" % (err, clean, again)


def s2_case(rng, k):
    from tools.probes.s2_check import layer
    N = rng.choice((1, 2, 3))
    Ho, Wo = rng.randint(8, 40), rng.randint(32, 80)
    dg = rng.random() < 0.5
    # (the FORWARD layer's channels; the launch's output side -- Cout forward, Cin for the data-gradient -- comes in 64-channel blocks)
    Cin, Cout = (64 * rng.randint(1, 3), 16 * rng.randint(2, 10)) if dg else (16 * rng.randint(2, 10), 64 * rng.randint(1, 3))
    epi = rng.choice(("plain", "mask", "none"))
    run, _, _ = layer(N, Ho, Wo, Cin, Cout, 2000 + k, dg, epi)
    ref = run(False)
    got = run(True)
    yc = Cin if dg else Cout
    clean = float(got[..., :64].min()) == 3.0 and float(got[..., 64 + yc:].min()) == 3.0
    same = torch.equal(got, ref)
    return ("s2", (N, Ho, Wo, Cin, Cout, epi, "dgrad" if dg else "fwd")), (same and clean), "equal %s clean %s max|d| %.2e" % (same, clean, float((got - ref).abs().max()))


def shuffle_case(rng, k):
    N = rng.choice((1, 2))
    H, W = rng.randint(8, 48), rng.randint(32, 80)
    nf = 64 * rng.randint(1, 2)
    cin = 16 * rng.randint(2, 8)
    g = torch.Generator().manual_seed(3000 + k)
    w = ((torch.rand(4 * nf, cin, 3, 3, generator=g) - 0.5) * 0.2).to(dev)
    b = (torch.rand(4 * nf, generator=g) - 0.5).to(dev)
    x = (torch.rand(N, H, W, cin, generator=g) * 2 - 1).to(dev)
    p = ops.WeightPacker(dev)
    i = p.add(w, ops.PACK_FWD)
    p.run()
    act = rng.choice(("relu", "lrelu", "none"))
    kw = dict(bias=b, **{"relu": dict(act=ops.ACT_RELU), "lrelu": dict(act=ops.ACT_LRELU, slope=0.2), "none": {}}[act])
    t = torch.zeros(N, H, W, 4 * nf, device=dev)
    ref = torch.zeros(N, 2 * H, 2 * W, nf, device=dev)
    ops.conv(ops.View(x), p.get(i), ops.View(t), wino=False, **kw)
    ops.depth_to_space(ops.View(t), ops.View(ref))
    off = 4 * rng.randint(0, 3)
    got = torch.full((N, 2 * H, 2 * W, nf + 16), 7.0, device=dev)
    done = ops.conv_shuffle2(ops.View(x), p.get(i), ops.View(got, off, nf), **kw)
    same = done is True and torch.equal(got[..., off:off + nf], ref)
    clean = (off == 0 or float(got[..., :off].min()) == 7.0) and float(got[..., off + nf:].min()) == 7.0 and float(got[..., off + nf:].max()) == 7.0
    return ("shuffle", (N, H, W, cin, nf, act, off)), (same and clean), "done %s equal %s clean %s" % (done, same, clean)


def sweep_case(rng, k):
    """The dense-block sweep (whose stagers took the packed operand split this round) against per-layer launches: bit for bit."""
    from tools.probes.sweep_check import block, nf
    N = rng.choice((1, 1, 2, 3, 5))
    H, W = rng.randint(8, 96), rng.randint(16, 140)
    grad = rng.random() < 0.5
    r2 = rng.random() < 0.5
    noise = ops.Noise(0.1, ops.noise_key(1, 2, k)) if rng.random() < 0.25 else None
    run = block(N, H, W, seed=4000 + k, grad_shape=grad, with_r2=r2, noise=noise)
    rb, ro, _ = run("layers")
    gb, go, _ = run("sweep")
    same = torch.equal(gb, rb) and torch.equal(go, ro)
    return ("sweep", (N, H, W, "grad" if grad else "fwd", "r2" if r2 else "-", "noise" if noise else "-")), same, "equal %s max|d| buf %.2e out %.2e" % (
        same, float((gb - rb).abs().max()), float((go - ro).abs().max()))


def k7_case(rng, k):
    """Config 5's one-launch 7x7 image-side kernels (TNR_CONV_7x7_C4, tnr_conv_thin7, tnr_wgrad_thin7) against torch in fp64 on the CPU."""
    import torch.nn.functional as F
    N = rng.choice((1, 2, 3))
    H, W = rng.randint(4, 40), rng.randint(4, 70)
    C = rng.choice((16, 32, 64))
    g = torch.Generator().manual_seed(5000 + k)
    img = torch.rand(N, 3, H, W, generator=g, dtype=torch.float64) * 2 - 1
    feat = torch.rand(N, C, H, W, generator=g, dtype=torch.float64) * 2 - 1
    w_in = (torch.rand(C, 3, 7, 7, generator=g, dtype=torch.float64) - 0.5) * 0.2
    w_out = (torch.rand(3, C, 7, 7, generator=g, dtype=torch.float64) - 0.5) * 0.1
    refl = lambda t: F.pad(t, (3, 3, 3, 3), mode="reflect")
    nhwc = lambda t, c=None: (F.pad(t, (0, 0, 0, 0, 0, (c or t.shape[1]) - t.shape[1])).permute(0, 2, 3, 1).contiguous().float().to(dev))
    V = ops.View
    which = rng.choice(("c4_fwd", "c4_dgrad", "thin_fwd", "thin_dgrad", "wg_in", "wg_out"))
    if which == "c4_fwd":
        ref = F.conv2d(refl(img), w_in)
        p = ops.WeightPacker(dev)
        i = p.add(w_in.float().to(dev), ops.PACK_C4_FWD)
        p.run()
        y = torch.zeros(N, H, W, C, device=dev)
        ops.conv(V(nhwc(img, 4)), p.get(i), V(y), mode=ops.CONV_7x7_C4, reflect=True)
        got = y.permute(0, 3, 1, 2)
    elif which == "c4_dgrad":
        xp = torch.zeros(N, C, H + 6, W + 6, dtype=torch.float64, requires_grad=True)
        (ref,) = torch.autograd.grad(F.conv2d(xp, w_out), xp, img)
        p = ops.WeightPacker(dev)
        i = p.add(w_out.float().to(dev), ops.PACK_C4_DGRAD3)
        p.run()
        y = torch.zeros(N, H + 6, W + 6, C, device=dev)
        ops.conv(V(nhwc(F.pad(img, (3, 3, 3, 3)), 4)), p.get(i), V(y), mode=ops.CONV_7x7_C4)
        got = y.permute(0, 3, 1, 2)
    elif which == "thin_fwd":
        ref = F.conv2d(refl(feat), w_out)
        y = torch.zeros(N, H, W, 4, device=dev)
        ops.conv_thin7(V(nhwc(feat)), w_out.float().to(dev), V(y, 0, 3), pad=3, reflect=True)
        got = y.permute(0, 3, 1, 2)[:, :3]
    elif which == "thin_dgrad":
        xp = torch.zeros(N, 3, H + 6, W + 6, dtype=torch.float64, requires_grad=True)
        (ref,) = torch.autograd.grad(F.conv2d(xp, w_in), xp, feat)
        y = torch.zeros(N, H + 6, W + 6, 4, device=dev)
        ops.conv_thin7(V(nhwc(feat)), w_in.float().to(dev), V(y, 0, 3), pad=6, reflect=False, dgrad=True)
        got = y.permute(0, 3, 1, 2)[:, :3]
    elif which == "wg_in":
        w0 = torch.zeros(C, 3, 7, 7, dtype=torch.float64, requires_grad=True)
        (ref,) = torch.autograd.grad(F.conv2d(refl(img), w0), w0, feat)
        dw = torch.zeros(C, 3, 7, 7, device=dev)
        ops.wgrad_thin7(V(nhwc(feat)), V(nhwc(refl(img), 4)), dw, None, flip=False, beta=0.0)
        got = dw
    else:
        w0 = torch.zeros(3, C, 7, 7, dtype=torch.float64, requires_grad=True)
        (ref,) = torch.autograd.grad(F.conv2d(refl(feat), w0), w0, img)
        dw = torch.zeros(3, C, 7, 7, device=dev)
        ops.wgrad_thin7(V(nhwc(feat)), V(nhwc(img, 4)), dw, None, flip=True, rpad=3, off=-6, beta=0.0)
        got = dw
    err = float((got.double().cpu() - ref).abs().max()) / max(1.0, float(ref.abs().max()))
    return ("k7", (which, N, H, W, C)), err <= 2e-5, "err/scale %.2e" % err


def main():
    assert ops.MMA == hip.MMA_BF16X3
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 20261001)
    bad = 0
    counts = {}
    for k in range(n):
        fn = (wino_case, sweep_case, s2_case, k7_case, wino_case, shuffle_case, sweep_case, k7_case)[k % 8]
        try:
            name, ok, note = fn(rng, k)
        except Exception as e:          # (a refused launch is a finding too)
            name, ok, note = (fn.__name__, k), False, "raised %r" % (e,)
        counts[name[0]] = counts.get(name[0], 0) + 1
        if not ok:
            bad += 1
            print("FAIL", name, note, flush=True)
        elif k % 10 == 0:
            print("ok  ", name, note, flush=True)
    print("cases", counts, "failures", bad, "chain error flag", ops.chain_error_flag())
    print("FUZZ", "OK" if bad == 0 else "FAILED")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
