"""Kernel-level parity (-m gpu): every C-ABI op against a plain fp32 PyTorch-CPU statement of the
same op (F.conv2d + autograd, F.batch_norm, ...), on shapes that exercise ragged tiles, channel
windows of wider buffers (the dense-block layout), 3/4-channel images and every conv geometry.
Tolerance: fp32 round-off only -- max |err| <= 2e-5 * (max |ref| + 1) unless stated.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from trainner_amd import ops
    return ops


def rnd(*shape, seed=0, lo=-1.0, hi=1.0):
    from oracle.detrand import uniform
    return uniform(shape, seed, lo, hi)


def nhwc_buf(x_nchw, ctot=None, coff=0, fill=7.0):
    """NCHW cpu tensor -> NHWC cuda buffer with `ctot` channels, data at [coff, coff+C)."""
    N, C, H, W = x_nchw.shape
    ctot = ctot or C
    buf = torch.full((N, H, W, ctot), fill, dtype=torch.float32)
    buf[..., coff:coff + C] = x_nchw.permute(0, 2, 3, 1)
    return buf.to(DEV).contiguous()


def to_nchw(buf, coff, C):
    return buf[..., coff:coff + C].permute(0, 3, 1, 2).contiguous().cpu()


def close(got, ref, tol=2e-5, what=""):
    scale = ref.abs().max().item() + 1.0
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, "%s: max err %.3e (scale %.3e)" % (what, err, scale)


def pack(ops, w, kind):
    p = ops.WeightPacker(torch.device(DEV))
    i = p.add(w, kind)
    p.run()
    return p.get(i), p


CONV_CASES = [
    # N, H, W, Cin, Cout, x_ctot, x_coff, y_ctot, y_coff
    (2, 20, 37, 64, 32, 192, 0, 192, 64),
    (1, 33, 33, 96, 32, 192, 0, 192, 96),
    (2, 16, 16, 192, 64, 192, 0, 64, 0),
    (1, 9, 40, 4, 64, 4, 0, 64, 0),       # image input (3 valid channels + zero pad)
    (1, 40, 9, 64, 3, 64, 0, 4, 0),       # image output
    (2, 8, 8, 128, 96, 128, 0, 96, 0),    # Cout not a multiple of 64, small tile (TW=8)
    (1, 18, 18, 32, 160, 32, 0, 160, 0),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3x3_forward(case):
    ops = _ops()
    N, H, W, Cin, Cout, xct, xco, yct, yco = case
    cin_real = 3 if Cin == 4 else Cin
    x = rnd(N, cin_real, H, W, seed=1)
    w = rnd(Cout, cin_real, 3, 3, seed=2, lo=-0.2, hi=0.2)
    b = rnd(Cout, seed=3)
    ref = F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.2)
    xb = nhwc_buf(F.pad(x, (0, 0, 0, 0, 0, Cin - cin_real)), xct, xco, fill=0.0 if Cin == 4 else 7.0)
    yb = torch.full((N, H, W, yct), -3.0, device=DEV)
    wd, bd = w.to(DEV), b.to(DEV)
    wp, _keep = pack(ops, wd, ops.PACK_FWD)
    ops.conv(ops.View(xb, xco, Cin), wp, ops.View(yb, yco, Cout), bias=bd, act=ops.ACT_LRELU, slope=0.2)
    torch.cuda.synchronize()
    close(to_nchw(yb, yco, Cout), ref, what="conv3x3 fwd")
    # channels outside the output window are untouched
    mask = torch.ones(yct, dtype=torch.bool)
    mask[yco:yco + Cout] = False
    assert (yb[..., mask.to(DEV)] == -3.0).all()


def test_conv3x3_epilogue_residuals_and_mask():
    ops = _ops()
    N, H, W, Cin, Cout = 2, 19, 35, 64, 64
    x, w, b = rnd(N, Cin, H, W, seed=4), rnd(Cout, Cin, 3, 3, seed=5, lo=-0.1, hi=0.1), rnd(Cout, seed=6)
    r1, r2, m = rnd(N, Cout, H, W, seed=7), rnd(N, Cout, H, W, seed=8), rnd(N, Cout, H, W, seed=9)
    conv = F.conv2d(x, w, b, padding=1)
    ref = (conv * 0.2 + 0.5 * r1)
    ref[:, 32:] = (conv * 0.2)[:, 32:]                       # r1 applies to channels < r1_ch only
    ref = ref * 0.3 + r2
    mm = torch.where(m > 0, torch.ones_like(m), torch.full_like(m, 0.2))
    ref[:, 16:48] = ref[:, 16:48] * mm[:, 16:48]
    xb, r1b, r2b, mb = nhwc_buf(x), nhwc_buf(r1), nhwc_buf(r2, 96, 32), nhwc_buf(m)
    yb = torch.zeros((N, H, W, Cout), device=DEV)
    wp, _k = pack(ops, w.to(DEV), ops.PACK_FWD)
    ops.conv(ops.View(xb), wp, ops.View(yb), bias=b.to(DEV), alpha=0.2, r1=ops.View(r1b), r1_ch=32, beta1=0.5,
             r2=ops.View(r2b, 32, Cout), alpha2=0.3, mask=ops.View(mb), m_lo=16, m_hi=48, m_slope=0.2)
    close(to_nchw(yb, 0, Cout), ref, what="epilogue")


def test_conv3x3_inplace_accumulate():
    """dgrad-style `y += conv(x)` where y and x are disjoint channel windows of ONE buffer."""
    ops = _ops()
    N, H, W = 1, 24, 40
    g = rnd(N, 192, H, W, seed=10)
    w = rnd(160, 32, 3, 3, seed=11, lo=-0.1, hi=0.1)
    ref = g[:, :160] + F.conv2d(g[:, 160:192], w, None, padding=1)
    gb = nhwc_buf(g)
    wp, _k = pack(ops, w.to(DEV), ops.PACK_FWD)
    acc = ops.View(gb, 0, 160)
    ops.conv(ops.View(gb, 160, 32), wp, acc, r1=acc, beta1=1.0)
    close(to_nchw(gb, 0, 160), ref, what="in-place accumulate")
    close(to_nchw(gb, 160, 32), g[:, 160:192], tol=0, what="input window untouched")


@pytest.mark.parametrize("shape", [(2, 12, 21, 64, 64), (1, 16, 16, 64, 32)])
def test_conv3x3_up2_forward(shape):
    ops = _ops()
    N, H, W, Cin, Cout = shape
    x, w, b = rnd(N, Cin, H, W, seed=12), rnd(Cout, Cin, 3, 3, seed=13, lo=-0.1, hi=0.1), rnd(Cout, seed=14)
    ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, padding=1)
    xb = nhwc_buf(x)
    yb = torch.zeros((N, 2 * H, 2 * W, Cout), device=DEV)
    wp, _k = pack(ops, w.to(DEV), ops.PACK_FWD)
    ops.conv(ops.View(xb), wp, ops.View(yb), mode=ops.CONV_3x3_UP2, bias=b.to(DEV))
    close(to_nchw(yb, 0, Cout), ref, what="conv3x3 up2")


@pytest.mark.parametrize("shape", [(2, 32, 48, 16, 16), (1, 64, 64, 64, 64), (2, 8, 8, 128, 96), (1, 16, 16, 32, 160)])
def test_conv4x4s2_forward(shape):
    ops = _ops()
    N, H, W, Cin, Cout = shape
    x, w, b = rnd(N, Cin, H, W, seed=15), rnd(Cout, Cin, 4, 4, seed=16, lo=-0.1, hi=0.1), rnd(Cout, seed=17)
    ref = F.conv2d(x, w, b, stride=2, padding=1)
    xb = nhwc_buf(x)
    yb = torch.zeros((N, H // 2, W // 2, Cout), device=DEV)
    wp, _k = pack(ops, w.to(DEV), ops.PACK_FWD_S2D)
    ops.conv(ops.View(xb), wp, ops.View(yb), mode=ops.CONV_4x4_S2, bias=b.to(DEV))
    close(to_nchw(yb, 0, Cout), ref, what="conv4x4s2")


@pytest.mark.parametrize("shape", [(2, 20, 37, 64, 32), (1, 16, 16, 192, 64), (1, 12, 12, 3, 64), (1, 10, 34, 64, 3)])
def test_dgrad3x3(shape):
    ops = _ops()
    N, H, W, Cin, Cout = shape
    x = rnd(N, Cin, H, W, seed=18).requires_grad_(True)
    w = rnd(Cout, Cin, 3, 3, seed=19, lo=-0.1, hi=0.1)
    g = rnd(N, Cout, H, W, seed=20)
    (ref,) = torch.autograd.grad(F.conv2d(x, w, None, padding=1), x, g)
    cg = (Cout + 3) // 4 * 4
    gb = nhwc_buf(F.pad(g, (0, 0, 0, 0, 0, cg - Cout)), fill=0.0)
    co = (Cin + 3) // 4 * 4
    yb = torch.zeros((N, H, W, co), device=DEV)
    wp, _k = pack(ops, w.to(DEV), ops.PACK_DGRAD_3x3)
    ops.conv(ops.View(gb), wp, ops.View(yb, 0, Cin), mode=ops.CONV_3x3)
    close(to_nchw(yb, 0, Cin), ref, what="dgrad3x3")


@pytest.mark.parametrize("shape", [(2, 32, 48, 16, 16), (1, 16, 16, 64, 64), (1, 8, 8, 128, 96)])
def test_dgrad4x4s2(shape):
    ops = _ops()
    N, H, W, Cin, Cout = shape
    x = rnd(N, Cin, H, W, seed=21).requires_grad_(True)
    w = rnd(Cout, Cin, 4, 4, seed=22, lo=-0.1, hi=0.1)
    g = rnd(N, Cout, H // 2, W // 2, seed=23)
    (ref,) = torch.autograd.grad(F.conv2d(x, w, None, stride=2, padding=1), x, g)
    gb = nhwc_buf(g)
    yb = torch.zeros((N, H, W, Cin), device=DEV)
    wp, _k = pack(ops, w.to(DEV), ops.PACK_DGRAD_S2)
    ops.conv(ops.View(gb), wp, ops.View(yb), mode=ops.DGRAD_4x4_S2)
    close(to_nchw(yb, 0, Cin), ref, what="dgrad4x4s2")


WGRAD_CASES = [
    # mode, N, H, W, Cin, Cout
    ("3x3", 2, 20, 37, 64, 32), ("3x3", 1, 24, 24, 96, 32), ("3x3", 1, 16, 16, 128, 32), ("3x3", 2, 16, 16, 192, 64),
    ("3x3", 1, 17, 19, 64, 64), ("3x3", 1, 12, 20, 3, 64), ("3x3", 1, 12, 20, 64, 3), ("3x3", 1, 8, 8, 128, 160),
    ("up2", 1, 10, 12, 64, 64), ("s2", 2, 32, 48, 16, 16), ("s2", 1, 16, 16, 64, 64), ("s2", 1, 8, 8, 128, 96),
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_wgrad(case):
    ops = _ops()
    mode, N, H, W, Cin, Cout = case
    x = rnd(N, Cin, H, W, seed=24)
    if mode == "3x3":
        w = rnd(Cout, Cin, 3, 3, seed=25).requires_grad_(True)
        y = F.conv2d(x, w, None, padding=1)
        m = ops.CONV_3x3
    elif mode == "up2":
        w = rnd(Cout, Cin, 3, 3, seed=25).requires_grad_(True)
        y = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, None, padding=1)
        m = ops.CONV_3x3_UP2
    else:
        w = rnd(Cout, Cin, 4, 4, seed=25).requires_grad_(True)
        y = F.conv2d(x, w, None, stride=2, padding=1)
        m = ops.CONV_4x4_S2
    g = rnd(*y.shape, seed=26)
    (ref_w,) = torch.autograd.grad(y, w, g)
    ref_b = g.sum(dim=(0, 2, 3))
    ci4, co4 = (Cin + 3) // 4 * 4, (Cout + 3) // 4 * 4
    xb = nhwc_buf(F.pad(x, (0, 0, 0, 0, 0, ci4 - Cin)), fill=0.0)
    gb = nhwc_buf(F.pad(g, (0, 0, 0, 0, 0, co4 - Cout)), fill=0.0)
    dw0 = rnd(*w.shape, seed=27)
    db0 = rnd(Cout, seed=28)
    dw, db = dw0.to(DEV), db0.to(DEV)
    ops.wgrad(ops.View(xb, 0, Cin), ops.View(gb, 0, Cout), dw, db, mode=m, alpha=0.5, beta=1.0)
    scale = ref_w.abs().max().item() + 1.0
    assert (dw.cpu() - (dw0 + 0.5 * ref_w)).abs().max().item() <= 5e-5 * scale, "wgrad weights"
    assert (db.cpu() - (db0 + 0.5 * ref_b)).abs().max().item() <= 5e-5 * (ref_b.abs().max().item() + 1), "wgrad bias"


def test_wgrad_cin_split():
    """160 input channels as 96 + 64 (how the dense block's conv4 is launched)."""
    ops = _ops()
    N, H, W, Cin, Cout = 1, 16, 24, 160, 32
    x, g = rnd(N, Cin, H, W, seed=29), rnd(N, Cout, H, W, seed=30)
    w = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(x, w, None, padding=1), w, g)
    xb, gb = nhwc_buf(x, 192, 0), nhwc_buf(g, 192, 160)
    dw = torch.zeros(Cout, Cin, 3, 3, device=DEV)
    gv = ops.View(gb, 160, Cout)
    ops.wgrad(ops.View(xb, 0, 96), gv, dw, None, cin_begin=0)
    ops.wgrad(ops.View(xb, 96, 64), gv, dw, None, cin_begin=96)
    close(dw.cpu(), ref, tol=5e-5, what="wgrad cin split")


@pytest.mark.parametrize("shape", [(2, 40, 72), (1, 16, 32), (3, 128, 128), (20, 128, 128)])
def test_conv_chain_equals_per_layer_launches(shape):
    """A dense block's five convolutions (fused LeakyReLU, residual epilogue) as one tnr_conv_chain launch must
    equal five tnr_conv_forward launches bit for bit -- including across tile borders (neighbour hand-off):
    (2,40,72) ragged tiles, (3,128,128) = 96 tiles, (20,128,128) = 640 tiles > the 512 resident workgroups, so
    workgroups walk several tiles per stage and publish across tile boundaries."""
    ops = _ops()
    N, H, W = shape
    nf, gc = 64, 32
    ws = [rnd(gc, nf + k * gc, 3, 3, seed=70 + k, lo=-0.05, hi=0.05).to(DEV) for k in range(4)]
    ws.append(rnd(nf, nf + 4 * gc, 3, 3, seed=74, lo=-0.05, hi=0.05).to(DEV))
    bs = [rnd(w.shape[0], seed=80 + k).to(DEV) for k, w in enumerate(ws)]
    p = ops.WeightPacker(DEV)
    idx = [p.add(w, ops.PACK_FWD) for w in ws]
    p.run()
    x0 = rnd(N, nf, H, W, seed=90).permute(0, 2, 3, 1).contiguous().to(DEV)
    skip = rnd(N, nf, H, W, seed=91).permute(0, 2, 3, 1).contiguous().to(DEV)

    def run(chain):
        buf = torch.zeros((N, H, W, nf + 4 * gc), device=DEV)
        buf[..., :nf] = x0
        out = torch.zeros((N, H, W, nf), device=DEV)
        st = []
        for k in range(4):
            cin = nf + gc * k
            st.append(dict(x=ops.View(buf, 0, cin), wp=p.get(idx[k]), y=ops.View(buf, cin, gc), bias=bs[k],
                           act=ops.ACT_LRELU, slope=0.2, fresh_from=(cin - gc if k else None)))
        st.append(dict(x=ops.View(buf), wp=p.get(idx[4]), y=ops.View(out), bias=bs[4], alpha=0.2, r1=ops.View(buf, 0, nf),
                       r2=ops.View(skip), alpha2=0.2, fresh_from=nf + 3 * gc))
        if chain:
            ops.conv_chain(st)
        else:
            for d in st:        # the engine's own per-layer form of a dense block: direct kernels (the Winograd form is not bit-identical)
                ops.conv(wino=False, **{k: v for k, v in d.items() if k != "fresh_from"})
        torch.cuda.synchronize()
        return buf.cpu(), out.cpu()

    assert ops.CONV_CHAIN
    ref_buf, ref_out = run(False)
    for rep in range(3):                    # repeated: the progress counters are reused with a growing epoch
        got_buf, got_out = run(True)
        assert torch.equal(got_buf, ref_buf), "dense buffer differs (rep %d)" % rep
        assert torch.equal(got_out, ref_out), "block output differs (rep %d)" % rep
    assert ops.chain_error_flag() == 0
    # and against autograd-free fp32 PyTorch for the first stage (sanity of the reference itself)
    y1 = F.leaky_relu(F.conv2d(x0.cpu().permute(0, 3, 1, 2), ws[0].cpu(), bs[0].cpu(), padding=1), 0.2)
    close(ref_buf[..., nf:nf + gc].permute(0, 3, 1, 2), y1, what="chain stage 0")


@pytest.mark.parametrize("grad_shape", [False, True])
@pytest.mark.parametrize("shape", [(1, 8, 32), (1, 10, 20), (2, 24, 24), (5, 64, 96), (9, 128, 128)])
def test_dense_block_sweep_equals_per_layer_launches(shape, grad_shape, mma_mode):
    """tnr_conv_sweep (TNR_MMA_BF16X3: a dense block, or its gradient mirror with LeakyReLU' masks, in one launch; conv_sweep.hip) against five
    per-layer launches, bit for bit: a single tile, ragged tiles in both directions, several images per round of the grid, and
    (9,128,128) = 576 tiles dispensed to 256 workgroups (more than two rounds, the last one partial).  Forward-shaped: bias + LeakyReLU
    stages, residual + skip epilogue; gradient-shaped: mask epilogues, no biases, beta1 residual.  Repeated launches reuse the progress
    counters and the tile dispenser."""
    ops = _ops()
    from trainner_amd import hip
    if ops.MMA != hip.MMA_BF16X3:
        pytest.skip("the sweep kernel is the dense block of the bf16x3 arithmetic")
    from tools.probes.sweep_check import block
    run = block(*shape, seed=31, grad_shape=grad_shape, with_r2=(shape[0] % 2 == 1))
    rb, ro, _ = run("layers")
    for rep in range(3):
        gb, go, _ = run("sweep")
        assert torch.equal(gb, rb), "dense buffer differs (rep %d)" % rep
        assert torch.equal(go, ro), "block output differs (rep %d)" % rep
    assert ops.chain_error_flag() == 0


def test_sweep_images_rebuilt_in_one_launch(mma_mode, monkeypatch):
    """tnr_conv_sweep_pack_batch (ops.SWEEP_PACK_BATCH: every dense block's sweep image of a packer rebuilt in one launch after the weights
    changed) gives the images the per-block tnr_conv_sweep_pack builds, bit for bit -- two blocks, three weight updates."""
    ops = _ops()
    from trainner_amd import hip
    if ops.MMA != hip.MMA_BF16X3:
        pytest.skip("the sweep kernel is the dense block of the bf16x3 arithmetic")
    from tools.probes.sweep_check import block
    outs = {}
    for batch in (False, True):
        monkeypatch.setattr(ops, "SWEEP_PACK_BATCH", batch)
        runs = [block(2, 24, 40, seed=13 + i, grad_shape=bool(i), with_r2=True) for i in range(2)]
        res = []
        for step in range(3):
            for run in runs:
                gb, go, st = run("sweep")
                res.append((gb.clone(), go.clone()))
                packer = st[0]["wp"].owner
            for run in runs:                       # an "optimiser step": scale the weights, re-pack (the batch rebuilds every image)
                _, _, st = run("layers")
                p = st[0]["wp"].owner
                for j in p.jobs:
                    j[0].mul_(0.9)
                p.run()
        outs[batch] = res
        if batch:
            assert "_sweep_batch" in packer.__dict__ and len(packer.__dict__["_sweep_batch"]["items"]) == 1
    for (a, b), (c, d) in zip(outs[False], outs[True]):
        assert torch.equal(a, c) and torch.equal(b, d)


def test_dense_block_form_is_chosen_per_box(mma_mode, monkeypatch):
    """ops.SWEEP_AUTO: a dense block of the training shape is timed ONCE, explicitly (ops.calibrate_dense_block_form -- the models call
    it from their constructor on scratch buffers, never from inside a forward), as one launch and as five per-layer launches, and the
    per-layer path is taken only where the sweep is more than 10 % slower (boxes whose coherent hand-off path is slow: DESIGN.md 3.2).
    The calibration records both timings, either choice gives the same bits, an uncalibrated process runs the sweep, and a block whose
    output aliases one of its inputs is refused (re-executing it would not be idempotent)."""
    ops = _ops()
    from trainner_amd import hip
    if ops.MMA != hip.MMA_BF16X3:
        pytest.skip("the choice exists in the bf16x3 arithmetic (where the two forms are bit-identical)")
    from tools.probes.sweep_check import block
    monkeypatch.setattr(ops, "SWEEP_AUTO", True)
    monkeypatch.setattr(ops, "SWEEP_AUTO_STATE", {"choice": None, "sweep_us": None, "layers_us": None})
    run = block(8, 128, 128, seed=77, grad_shape=True, with_r2=True)          # 131 072 pixels: a full-size block
    rb, ro, stages = run("layers")
    rb, ro = rb.clone(), ro.clone()
    gb, go, _ = run("sweep")                                                    # NOT calibrated: the sweep, no timing inside the call
    assert ops.SWEEP_AUTO_STATE["choice"] is None and ops.SWEEP_AUTO_STATE["sweep_us"] is None
    assert torch.equal(gb, rb) and torch.equal(go, ro)
    st = ops.calibrate_dense_block_form(stages)
    assert st["choice"] in ("sweep", "layers") and st["sweep_us"] > 0 and st["layers_us"] > 0, st
    assert (st["choice"] == "layers") == (st["sweep_us"] > 1.10 * st["layers_us"]), st
    for forced in ("layers", "sweep"):
        ops.SWEEP_AUTO_STATE["choice"] = forced
        gb, go, _ = run("sweep")
        assert torch.equal(gb, rb) and torch.equal(go, ro), forced
    print("dense-block form on this box:", st)
    bad = [dict(s) for s in stages]
    bad[-1]["y"] = ops.View(bad[0]["x"].buf, 0, bad[-1]["y"].C)                # the block's output over its own input channels
    with pytest.raises(AssertionError):
        ops.calibrate_dense_block_form(bad)
    assert ops.chain_error_flag() == 0


def test_model_constructor_calibrates_the_dense_block_form(mma_mode, monkeypatch, tmp_path):
    """SRModel.__init__ -> calibrate_engine: the choice exists before the first step, measured on the TRAINING shape (batch x crop / scale)."""
    ops = _ops()
    from trainner_amd import hip
    if ops.MMA != hip.MMA_BF16X3:
        pytest.skip("bf16x3 only")
    from oracle import ref_harness
    from trainner_amd.models import create_model
    from trainner_amd.options import options
    monkeypatch.setattr(ops, "SWEEP_AUTO", True)
    monkeypatch.setattr(ops, "SWEEP_AUTO_STATE", {"choice": None, "sweep_us": None, "layers_us": None})
    opt = options.parse(ref_harness.esrgan_yaml(name="calib", out_root=str(tmp_path), gpu_ids="[0]", nb=1, batch=2, crop=128, d_nf=16), is_train=True)
    create_model(opt, verbose=False)
    st = ops.SWEEP_AUTO_STATE
    assert st["choice"] in ("sweep", "layers") and st["sweep_us"] > 0 and st["layers_us"] > 0, st


def test_dense_block_sweeps_next_to_other_queues(mma_mode):
    """One-launch dense blocks while OTHER hardware queues oversubscribe the chip -- what a data-parallel run's RCCL kernels or a
    feeder's kernels do to them: two streams each running dense-block sweeps over their own buffers (2 x 256 workgroups that want a
    CU each) and a third running 64-cout convolutions, enqueued without any synchronisation between them.  Every sweep must still
    be bit-identical to its single-stream result and no hand-off wait may report a timeout.  The waits are bounded by a POLL COUNT
    (conv_handoff.h): the clock-based bound they used to have fired spuriously next to other queues in tools/probes/overlap_probe.py
    (4 runs of 11 against 0 of 11 with the poll count, results correct either way: profiles/r08a_handoff_bound.txt) -- this test
    covers the situation; it did not reproduce that failure in 8 runs of the old bound, the probe is the record of it."""
    ops = _ops()
    from trainner_amd import hip
    if ops.MMA != hip.MMA_BF16X3:
        pytest.skip("the sweep kernel is the dense block of the bf16x3 arithmetic")
    from tools.probes.sweep_check import block
    runs = [block(16, 128, 128, seed=51 + i, grad_shape=bool(i & 1), with_r2=True) for i in range(2)]
    sets = []
    for run in runs:
        rb, ro, st = run("sweep")                        # single stream: the reference (and the stage list, bound to rb / ro)
        sets.append((st, rb.clone(), ro.clone()))
    # a third population: plain 64-cout 3x3 layers (weight-stream kernel, two workgroups per CU)
    x3 = rnd(4, 64, 256, 256, seed=60)
    w3 = rnd(64, 64, 3, 3, seed=61) * 0.1
    xb3, yb3 = nhwc_buf(x3), torch.zeros((4, 256, 256, 64), device=DEV)
    wp3, keep = pack(ops, w3.to(DEV), ops.PACK_FWD)
    streams = [torch.cuda.Stream() for _ in range(3)]
    torch.cuda.synchronize()
    for rep in range(6):
        # (the situation in which the clock-based bound fired in the probe: a deep queue of sweeps on the launch stream, the same blocks
        #  -- same values, so the same results -- enqueued on a second stream meanwhile, other kernels on a third)
        for _ in range(3):
            for st, _, _ in sets:
                ops.conv_chain(st)
        for (st, _, _), s in zip(sets, streams):
            with torch.cuda.stream(s):
                for _ in range(3):
                    ops.conv_chain(st)
        with torch.cuda.stream(streams[2]):
            for _ in range(4):
                ops.conv(ops.View(xb3), wp3, ops.View(yb3))
    torch.cuda.synchronize()
    assert ops.chain_error_flag() == 0, "a hand-off wait reported a timeout next to other queues"
    for i, (st, rb, ro) in enumerate(sets):
        assert torch.equal(st[0]["x"].buf, rb) and torch.equal(st[-1]["y"].buf, ro), "sweep %d differs from its single-stream result" % i


@pytest.mark.parametrize("case", [(2, 40, 72, 64, 64, "lrelu", False), (1, 8, 32, 32, 64, "plain", False), (3, 17, 33, 128, 128, "res", False),
                                  (2, 64, 64, 256, 256, "mask", True), (1, 32, 32, 512, 512, "plain", False), (2, 96, 160, 64, 128, "lrelu", True),
                                  (1, 9, 45, 96, 192, "res", False), (2, 24, 40, 64, 64, "noise", False),
                                  (2, 64, 64, 256, 256, "reflect", False), (1, 20, 37, 64, 64, "reflect", False)])    # ReflectionPad2d(1): ResnetGenerator's blocks
def test_conv3x3_weight_stream_kernel_equals_staged_weights(case):
    """TNR_MMA_BF16X3, 3x3 layers with Cout % 64 == 0: conv3x3_d4_kernel (tnr_conv_desc.wq: weights as a pre-split stream read straight
    into registers; csrc/conv_sweep.hip) against the kernels that split and stage the slab per workgroup (ops.X3_D4 = False), bit for bit:
    ragged tiles, a single tile, several 64-cout blocks, channel windows of wider buffers, forward and data-gradient packings, bias /
    LeakyReLU / residual / mask / noise epilogues; and the staged kernels themselves against fp32 F.conv2d (test_conv3x3_forward...)."""
    ops = _ops()
    from trainner_amd import hip
    if ops.MMA != hip.MMA_BF16X3:
        pytest.skip("the weight stream is the split arithmetic's (three bf16 planes)")
    from tools.probes.d4_check import layer
    N, H, W, Cin, Cout, epi, dg = case
    run, packer, _ = layer(N, H, W, Cin, Cout, 77, epi, dg)
    ref = run(False)
    for rep in range(2):
        assert torch.equal(run(True), ref), rep
    assert len(packer.__dict__.get("_wq_images", {})) <= 1 and float(ref[..., :64].min()) == 3.0 and float(ref[..., 64 + Cout:].min()) == 3.0
    assert len(packer.__dict__.get("_wq_images", {})) == 1 or N * H * W <= 16384         # (small-spatial 512-channel layers run split-K instead)
    # the stream follows the weights: after the packer runs again (optimiser step) the image is rebuilt
    packer.jobs[0][0].mul_(0.5)
    packer.run()
    half = run(True)
    assert torch.equal(half, run(False)) and not torch.equal(half, ref)


@pytest.mark.parametrize("case", [(2, 16, 16, 32, 64, "plain", False), (1, 40, 72, 64, 64, "plain", False), (2, 24, 40, 192, 64, "res", False),
                                  (1, 17, 33, 64, 128, "mask", False), (3, 32, 32, 128, 128, "plain", True), (1, 9, 11, 32, 64, "res", True),
                                  (1, 64, 64, 512, 64, "plain", False)])
def test_conv3x3_winograd_form_error_vs_fp64(case):
    """TNR_MMA_BF16X3, the Winograd F(2x2, 3x3) form of the 64-cout 3x3 layers (csrc/conv_wino.hip: 16 transform-domain products per 2 x 2
    output patch instead of 36; transforms in fp32, products in the split arithmetic) against an fp64 convolution on the CPU, next to
    the fp32 matrix-core path and the direct weight-stream kernel on the same operands.  Bound (VERDICT r5 item 1 a): max and rms error
    <= 3 x the fp32 matrix-core path's + 2e-7 of the output scale -- the direct forms are held to 1.5 x.  Ragged tiles in both
    directions, a single 16 x 16 tile, Cin 32 .. 512, 64 and 128 couts, bias / LeakyReLU / two residuals / mask epilogues, zero and
    reflection padding."""
    ops = _ops()
    from trainner_amd import hip
    if ops.MMA != hip.MMA_BF16X3:
        pytest.skip("the Winograd form exists in the split arithmetic (three bf16 planes)")
    from tools.probes.wino_check import layer
    N, H, W, Cin, Cout, epi, refl = case
    run, ref64 = layer(N, H, W, Cin, Cout, seed=7 + Cin + H, epi=epi, reflect=refl)
    r = ref64()
    scale = float(r.abs().max())
    e = {}
    for mode in ("f32", "d4", "wino"):
        d = (run(mode).double().cpu() - r).abs()
        e[mode] = (float(d.max()), float(d.pow(2).mean().sqrt()))
    assert e["wino"][0] <= 3.0 * e["f32"][0] + 2e-7 * scale and e["wino"][1] <= 3.0 * e["f32"][1] + 2e-7 * scale, e
    assert torch.equal(run("wino"), run("wino"))          # deterministic


@pytest.mark.parametrize("case", [(3, 17, 33, 128, 128, "res", False), (2, 64, 64, 256, 256, "mask", True), (1, 32, 32, 512, 512, "plain", False),
                                  (2, 96, 160, 64, 128, "lrelu", True), (1, 9, 45, 96, 192, "res", False), (2, 24, 40, 64, 64, "noise", False),
                                  (2, 64, 64, 256, 256, "reflect", False), (1, 20, 37, 64, 64, "reflect", False)])
def test_conv3x3_winograd_form_agrees_with_direct_kernel(case):
    """The Winograd form in the places the engine uses it from: channel windows of wider buffers (nothing outside the window is touched),
    forward and data-gradient packings, several 64-cout blocks, the ESRGAN+ noise epilogue, reflection padding -- against the direct
    weight-stream kernel on the same launch: two fp32-class evaluations of one convolution (max |d| <= 3e-6 of the scale), and the
    stream follows the weights after an optimiser step."""
    ops = _ops()
    from trainner_amd import hip
    if ops.MMA != hip.MMA_BF16X3:
        pytest.skip("the Winograd form exists in the split arithmetic (three bf16 planes)")
    from tools.probes.d4_check import layer
    N, H, W, Cin, Cout, epi, dg = case
    run, packer, _ = layer(N, H, W, Cin, Cout, 91, epi, dg)
    ref = run(True)
    got = run("wino")
    sc = max(1.0, float(ref[..., 64:64 + Cout].abs().max()))
    assert float((got - ref).abs().max()) <= 3e-6 * sc, float((got - ref).abs().max()) / sc
    assert float(got[..., :64].min()) == 3.0 and float(got[..., :64].max()) == 3.0 and float(got[..., 64 + Cout:].min()) == 3.0
    packer.jobs[0][0].mul_(0.5)
    packer.run()
    half = run("wino")
    assert float((half - run(True)).abs().max()) <= 3e-6 * sc and not torch.equal(half, got)


@pytest.mark.parametrize("case", [(2, 32, 32, 64, 64, False, "plain"), (1, 40, 72, 128, 128, False, "plain"), (3, 17, 33, 64, 128, False, "none"),
                                  (2, 32, 32, 512, 512, False, "plain"), (2, 32, 32, 64, 64, True, "mask"), (1, 40, 72, 128, 128, True, "mask"),
                                  (3, 17, 33, 128, 64, True, "none"), (2, 32, 32, 512, 512, True, "mask"), (1, 8, 32, 32, 64, True, "plain")])
def test_four_tap_weight_stream_kernel_equals_staged_weights(case, monkeypatch):
    """conv_s2_d4_kernel (csrc/conv_sweep.hip: the discriminators' 4x4 stride-2 convolution, discriminators.py:24-34, as 2 x 2 taps over the
    parity planes, and its data-gradient per output parity class, with the weights as a pre-split stream) against conv_tile_kernel
    (ops.S2_D4 = False), bit for bit in the split arithmetic: ragged tiles, channel windows of wider buffers, bias / mask epilogues,
    64 .. 512 channels; with bf16 operands (`use_amp`) the two agree to 2e-5 of the scale (another order of the products inside an MFMA)."""
    ops = _ops()
    from trainner_amd import hip
    if ops.MMA != hip.MMA_BF16X3:
        pytest.skip("the weight stream is the split arithmetic's (three bf16 planes)")
    from tools.probes.s2_check import layer
    N, Ho, Wo, Cin, Cout, dg, epi = case
    for mma in (hip.MMA_BF16X3, hip.MMA_BF16):
        monkeypatch.setattr(ops, "MMA", mma)
        run, _, _ = layer(N, Ho, Wo, Cin, Cout, 60, dg, epi)
        ref = run(False)
        got = run(True)
        if mma == hip.MMA_BF16X3:
            assert torch.equal(got, ref) and torch.equal(run(True), ref)
        else:
            assert float((got - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
        yc = Cin if dg else Cout
        assert float(got[..., :64].min()) == 3.0 and float(got[..., 64 + yc:].min()) == 3.0


@pytest.mark.parametrize("shape", [(2, 32, 32, 64), (1, 40, 72, 64), (2, 24, 40, 128), (3, 9, 33, 64)])
@pytest.mark.parametrize("act", ["relu", "lrelu", "none"])
def test_pixel_shuffle_folded_into_the_conv_store(shape, act, monkeypatch):
    """block.pixelshuffle_block (block.py:374-387: conv nf -> 4 nf, nn.PixelShuffle(2), act) as ONE launch (tnr_conv_desc.shuffle: the
    weight stream ordered by sub-pixel, every 64-cout block stored as 256-byte rows of the shuffled tensor) against the two-pass form
    -- the same convolution kernel followed by tnr_depth_to_space -- bit for bit; ragged tiles, nf 64 and 128, the three epilogue
    activations, also with bf16 operands (`use_amp`).  The fp32-matrix-core arithmetic has no such kernel: ops.conv_shuffle2 says so."""
    ops = _ops()
    from trainner_amd import hip
    N, H, W, nf = shape
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(5 + H)
    w = ((torch.rand(4 * nf, nf, 3, 3, generator=g) - 0.5) * 0.2).to(dev)
    b = (torch.rand(4 * nf, generator=g) - 0.5).to(dev)
    x = (torch.rand(N, H, W, nf, generator=g) * 2 - 1).to(dev)
    p = ops.WeightPacker(dev)
    i = p.add(w, ops.PACK_FWD)
    p.run()
    kw = dict(bias=b, **{"relu": dict(act=ops.ACT_RELU), "lrelu": dict(act=ops.ACT_LRELU, slope=0.2), "none": {}}[act])
    modes = [ops.MMA] + ([hip.MMA_BF16] if ops.MMA == hip.MMA_BF16X3 else [])
    for mma in modes:
        monkeypatch.setattr(ops, "MMA", mma)
        t = torch.zeros(N, H, W, 4 * nf, device=dev)
        ref = torch.zeros(N, 2 * H, 2 * W, nf, device=dev)
        ops.conv(ops.View(x), p.get(i), ops.View(t), wino=False, **kw)
        ops.depth_to_space(ops.View(t), ops.View(ref))
        want = torch.nn.functional.pixel_shuffle(t.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
        assert torch.equal(ref, want)                                  # (tnr_depth_to_space is nn.PixelShuffle(2))
        got = torch.full((N, 2 * H, 2 * W, nf + 8), 7.0, device=dev)   # a channel window of a wider buffer
        done = ops.conv_shuffle2(ops.View(x), p.get(i), ops.View(got, 4, nf), **kw)
        if mma == hip.MMA_F32:
            assert done is False
            continue
        assert done is True
        assert torch.equal(got[..., 4:4 + nf], ref), float((got[..., 4:4 + nf] - ref).abs().max())
        assert float(got[..., :4].min()) == 7.0 and float(got[..., 4 + nf:].min()) == 7.0


def test_winograd_policy(monkeypatch):
    """ops.conv picks the Winograd form by itself only where it wins (>= ops.WINO_MIN_CIN input channels, enough pixels), never in the
    fp32-matrix-core arithmetic or under `use_amp`, never for the per-layer form of a dense block, and TNR_WINO=0 turns it off."""
    ops = _ops()
    from trainner_amd import hip
    if ops.MMA != hip.MMA_BF16X3:
        pytest.skip("bf16x3 only")
    dev = torch.device("cuda")
    def probe(Cin, pixels_side, **kw):
        w = torch.zeros(64, Cin, 3, 3, device=dev)
        p = ops.WeightPacker(dev)
        i = p.add(w, ops.PACK_FWD)
        p.run()
        x = torch.zeros(1, pixels_side, pixels_side, Cin, device=dev)
        y = torch.zeros(1, pixels_side, pixels_side, 64, device=dev)
        prof = ops.ConvProfile()
        monkeypatch.setattr(ops, "PROFILE", prof)
        ops.conv(ops.View(x), p.get(i), ops.View(y), **kw)
        monkeypatch.setattr(ops, "PROFILE", None)
        return list(prof.summary())[0]

    monkeypatch.setattr(ops, "WINO", True)
    monkeypatch.setattr(ops, "WINO_MIN_CIN", 128)
    assert probe(128, 64) == "conv_wino_3x3" and probe(64, 64) == "conv_tile_3x3" and probe(256, 32) == "conv_tile_3x3"
    assert probe(64, 64, wino=True) == "conv_wino_3x3" and probe(128, 64, wino=False) == "conv_tile_3x3"
    monkeypatch.setattr(ops, "WINO", False)
    assert probe(128, 64) == "conv_tile_3x3"
    monkeypatch.setattr(ops, "WINO", True)
    monkeypatch.setattr(ops, "MMA", hip.MMA_BF16)
    assert probe(128, 64) == "conv_tile_3x3"


@pytest.mark.parametrize("grad_shape", [False, True])
@pytest.mark.parametrize("shape", [(1, 8, 32), (2, 40, 72), (5, 64, 96)])
def test_amp_dense_block_sweep_agrees_with_per_layer(shape, grad_shape, monkeypatch):
    """`use_amp` (TNR_MMA_BF16: operands rounded to bf16, fp32 accumulate): the dense block through the sweep's bf16-operand form
    (conv_sweep4_kernel<true, true>: plane 0 of the weight stream and of the input tile, units of 2 MFMAs, nine-deep fragment ring)
    against five per-layer launches.  Same rounded operands; the 16 products of an MFMA are summed in another order, so the first
    stage agrees to 2e-6 of the scale and the later ones to bf16 resolution (an activation within 5e-7 of a rounding boundary rounds
    the other way: one bf16 ulp of ONE input, a 3 x 3 patch of ~2e-4).  Deterministic: two runs are bit-identical."""
    ops = _ops()
    from trainner_amd import hip
    from tools.probes.sweep_check import block
    monkeypatch.setattr(ops, "MMA", hip.MMA_BF16)
    run = block(*shape, seed=35, grad_shape=grad_shape, with_r2=(shape[0] % 2 == 1))
    rb, ro, _ = run("layers")
    gb, go, _ = run("sweep")
    sc = max(float(rb.abs().max()), float(ro.abs().max()), 1.0)
    assert float((gb[..., 64:96] - rb[..., 64:96]).abs().max()) <= 2e-6 * sc
    assert float((gb - rb).abs().max()) <= 4e-3 * sc and float((go - ro).abs().max()) <= 4e-3 * sc
    assert float((gb - rb).abs().mean()) <= 2e-5 * sc
    g2, o2, _ = run("sweep")
    assert torch.equal(g2, gb) and torch.equal(o2, go) and ops.chain_error_flag() == 0


@pytest.mark.parametrize("case", [(2, 40, 72, 64, 64, "lrelu", False), (3, 17, 33, 128, 128, "res", False), (2, 64, 64, 256, 256, "mask", True),
                                  (1, 9, 45, 96, 192, "res", False)])
def test_amp_conv3x3_weight_stream_kernel(case, monkeypatch):
    """TNR_MMA_BF16: conv3x3_d4_kernel<2, true> (bf16-operand form of the weight-stream kernel) against conv_tile_body<BF = 1>: one layer,
    the same rounded operands -> agreement to fp32 summation order (2e-5 of the scale)."""
    ops = _ops()
    from trainner_amd import hip
    from tools.probes.d4_check import layer
    monkeypatch.setattr(ops, "MMA", hip.MMA_BF16)
    N, H, W, Cin, Cout, epi, dg = case
    run, packer, _ = layer(N, H, W, Cin, Cout, 78, epi, dg)
    ref, got = run(False), run(True)
    assert float((got - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max())) and len(packer.__dict__.get("_wq_images", {})) == 1


# ------------------------------------------------------------------------------------------------------------------------
# ESRGAN+ GaussianNoise (block.py:587-600): the counter-based multiplier field of csrc/gauss_noise.h
# ------------------------------------------------------------------------------------------------------------------------
def _device_field(ops, N, H, W, Cc, nz, src=None):
    dst = torch.empty((N, H, W, Cc), device=DEV)
    ops.gauss_mult(ops.View(dst), None if src is None else ops.View(src), nz)
    torch.cuda.synchronize()
    return dst


def test_gauss_field_is_the_documented_function():
    """tnr_gauss_mult against oracle/gauss_noise.py (the integer hash bit for bit, Box-Muller in float32 with libm): the device's
    v_log / v_sqrt / v_sin / v_cos are a few ulp from libm, so |dm| <= 2e-6 at sigma 0.1; the field of a rank's shard (pix0) is the
    slice of the global field; a channel window of a wider buffer is written in place; src * m is ONE rounding of the product."""
    ops = _ops()
    from oracle import gauss_noise
    N, H, W, Cc = 3, 20, 36, 64
    key = ops.noise_key(4242, 7, 11)
    assert (key & 0xFFFFFFFF, key >> 32) == gauss_noise.noise_key(4242, 7, 11)
    nz = ops.Noise(0.1, key)
    got = _device_field(ops, N, H, W, Cc, nz).cpu()
    ref = torch.from_numpy(gauss_noise.multiplier(N * H * W, Cc, 0.1, nz.key0, nz.key1)).view(N, H, W, Cc)
    assert (got - ref).abs().max().item() <= 2e-6
    shard = ops.Noise(0.1, key, pix0=2 * H * W)
    assert torch.equal(_device_field(ops, 1, H, W, Cc, shard).cpu(), got[2:3])
    wide = torch.full((N, H, W, 192), 7.0, device=DEV)
    src = rnd(N, H, W, Cc, seed=5).to(DEV)
    ops.gauss_mult(ops.View(wide, 64, Cc), ops.View(src), nz)
    assert torch.equal(wide[..., 64:128].cpu(), src.cpu() * got) and float(wide[..., :64].min()) == 7.0 and float(wide[..., 128:].max()) == 7.0


def test_gauss_field_statistics():
    """n = (m - 1) / sigma over 16.8 M elements (one dense block at batch 16, 128 x 128): mean 0, variance 1, kurtosis 3, |n| <= 4.86
    (16-bit radius); no correlation between two blocks' fields, two training forwards' fields, neighbouring channels (the four
    lanes of a quad come from two hashes), neighbouring pixels; relative noise: std of x * m - x is 0.1 |x|."""
    ops = _ops()
    N, H, W, Cc = 16, 128, 128, 64
    f = [((_device_field(ops, N, H, W, Cc, ops.Noise(0.1, ops.noise_key(9, call, blk))) - 1.0) / 0.1).double() for call, blk in ((0, 0), (0, 1), (1, 0))]
    n = f[0]
    cnt = n.numel()
    se = 1.0 / math.sqrt(cnt)
    assert abs(n.mean().item()) < 5 * se and abs(n.var().item() - 1.0) < 5 * math.sqrt(2.0) * se
    assert abs((n ** 4).mean().item() - 3.0) < 5 * math.sqrt(96.0) * se and n.abs().max().item() <= 4.86

    def corr(a, b):
        return ((a * b).mean() / (a.std() * b.std())).item()

    assert abs(corr(f[0], f[1])) < 5 * se and abs(corr(f[0], f[2])) < 5 * se
    q = n.view(-1, 4)
    for i in range(4):
        for j in range(i + 1, 4):
            assert abs(corr(q[:, i], q[:, j])) < 5 * 2 * se, (i, j)
    assert abs(corr(n[:, :, :-1], n[:, :, 1:])) < 5 * se and abs(corr(n[:, :-1], n[:, 1:])) < 5 * se and abs(corr(n[..., :-4], n[..., 4:])) < 5 * se
    x = rnd(N, H, W, Cc, seed=3, lo=0.5, hi=2.0).to(DEV)
    y = _device_field(ops, N, H, W, Cc, ops.Noise(0.1, ops.noise_key(9, 0, 0)), src=x)
    rel = ((y - x) / x).double()
    assert abs(rel.std().item() - 0.1) < 1e-4 and abs(rel.mean().item()) < 1e-4


@pytest.mark.parametrize("pos", [1, 2])
@pytest.mark.parametrize("shape", [(2, 20, 37, 96, 32), (1, 33, 33, 192, 64)])
def test_conv_epilogue_noise_multiplier(shape, pos):
    """tnr_conv_desc.noise_*: the per-layer kernels' epilogue multiplies by the field of tnr_gauss_mult -- after the r1 step
    (pos 1: noise(x5*0.2 + x), then the RRDB residual) or after the r2 step (pos 2) -- bit for bit (same function, one rounding
    per operation): compared with the same convolution without noise and WITHOUT r2, finished in torch."""
    ops = _ops()
    N, H, W, Cin, Cout = shape
    w = rnd(Cout, Cin, 3, 3, seed=41, lo=-0.05, hi=0.05).to(DEV)
    wp, _ = pack(ops, w, ops.PACK_FWD)
    b = rnd(Cout, seed=42).to(DEV)
    xb = rnd(N, H, W, Cin, seed=43).to(DEV)
    r1, r2 = rnd(N, H, W, Cout, seed=44).to(DEV), rnd(N, H, W, Cout, seed=45).to(DEV)
    nz = ops.Noise(0.1, ops.noise_key(1, 2, 3), pos=pos, pix0=1234)
    z, y = torch.empty((N, H, W, Cout), device=DEV), torch.empty((N, H, W, Cout), device=DEV)
    ops.conv(ops.View(xb), wp, ops.View(z), bias=b, alpha=0.2, r1=ops.View(r1))
    ops.conv(ops.View(xb), wp, ops.View(y), bias=b, alpha=0.2, r1=ops.View(r1), r2=ops.View(r2), alpha2=0.2, noise=nz)
    m = _device_field(ops, N, H, W, Cout, nz)
    ref = (z * m) * 0.2 + r2 if pos == 1 else (z * 0.2 + r2) * m
    assert torch.equal(y.cpu(), ref.cpu())
    y2 = torch.empty_like(y)
    ops.conv(ops.View(xb), wp, ops.View(y2), bias=b, alpha=0.2, r1=ops.View(r1), noise=nz)       # without r2 both positions coincide
    assert torch.equal(y2.cpu(), (z * m).cpu())


@pytest.mark.parametrize("grad_shape", [False, True])
@pytest.mark.parametrize("shape", [(1, 10, 20), (5, 64, 96)])
def test_dense_block_one_launch_forms_with_noise(shape, grad_shape):
    """The dense block's one-launch forms (tnr_conv_sweep in TNR_MMA_BF16X3, tnr_conv_chain otherwise) with the noise multiplier on
    the last stage against five per-layer launches, bit for bit; and against the same block without noise times the field."""
    ops = _ops()
    from tools.probes.sweep_check import block
    nz = ops.Noise(0.1, ops.noise_key(5, 0, 17), pix0=77)
    run = block(*shape, seed=33, grad_shape=grad_shape, with_r2=False, noise=nz)
    rb, ro, _ = run("layers")
    for rep in range(2):
        gb, go, _ = run("sweep")
        assert torch.equal(gb, rb) and torch.equal(go, ro), rep
    _, plain, _ = block(*shape, seed=33, grad_shape=grad_shape, with_r2=False)("sweep")
    N, H, W = shape
    assert torch.equal(ro.cpu(), (plain * _device_field(ops, N, H, W, 64, nz)).cpu())
    assert ops.chain_error_flag() == 0


@pytest.mark.parametrize("case", [("3x3", 2, 8, 8, 256, 96), ("3x3", 16, 4, 4, 512, 512), ("s2", 2, 16, 16, 128, 64),
                                  ("s2", 4, 8, 8, 512, 160), ("dgrad3", 2, 8, 8, 96, 256)])
def test_conv_small_im2col_splitk(case):
    """Small-spatial layers as tnr_im2col + TNR_CONV_1x1 (split-K partial sums + reduce launch) against F.conv2d."""
    ops = _ops()
    kind, N, H, W, Cin, Cout = case
    x = rnd(N, Cin, H, W, seed=101)
    b = rnd(Cout, seed=103)
    if kind == "3x3":
        w = rnd(Cout, Cin, 3, 3, seed=102, lo=-0.05, hi=0.05)
        ref = F.conv2d(x, w, b, padding=1)
        wp, _ = pack(ops, w.to(DEV), ops.PACK_COL_FWD)
        y = torch.zeros((N, H, W, Cout), device=DEV)
        ops.conv_small(ops.View(nhwc_buf(x)), wp, ops.View(y), 3, 1, bias=b.to(DEV))
    elif kind == "s2":
        w = rnd(Cout, Cin, 4, 4, seed=102, lo=-0.05, hi=0.05)
        ref = F.conv2d(x, w, b, stride=2, padding=1)
        wp, _ = pack(ops, w.to(DEV), ops.PACK_COL_FWD)
        y = torch.zeros((N, H // 2, W // 2, Cout), device=DEV)
        ops.conv_small(ops.View(nhwc_buf(x)), wp, ops.View(y), 4, 2, bias=b.to(DEV))
    else:   # data-gradient of a 3x3 convolution with Cin -> Cout channels: g has Cout channels, result Cin
        w = rnd(Cout, Cin, 3, 3, seed=102, lo=-0.05, hi=0.05)
        xin = x.clone().requires_grad_(True)
        g = rnd(N, Cout, H, W, seed=104)
        (ref,) = torch.autograd.grad(F.conv2d(xin, w, None, padding=1), xin, g)
        wp, _ = pack(ops, w.to(DEV), ops.PACK_COL_DGRAD3)
        y = torch.zeros((N, H, W, Cin), device=DEV)
        ops.conv_small(ops.View(nhwc_buf(g)), wp, ops.View(y), 3, 1)
    close(to_nchw(y, 0, y.shape[3]), ref, what="conv_small " + kind)


@pytest.mark.parametrize("shape", [(2, 20, 37), (1, 64, 96)])
def test_conv3x3_image_channels_c4(shape):
    """TNR_CONV_3x3_C4: 3 -> 64 forward over an NHWC4 image and the 64 <- 3 data-gradient (taps folded into K)."""
    ops = _ops()
    N, H, W = shape
    x = rnd(N, 3, H, W, seed=121)
    w = rnd(64, 3, 3, 3, seed=122, lo=-0.2, hi=0.2)
    b = rnd(64, seed=123)
    ref = F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.2)
    x4 = nhwc_buf(F.pad(x, (0, 0, 0, 0, 0, 1)), fill=0.0)
    wp, _ = pack(ops, w.to(DEV), ops.PACK_C4_FWD)
    y = torch.zeros((N, H, W, 64), device=DEV)
    ops.conv(ops.View(x4), wp, ops.View(y), mode=ops.CONV_3x3_C4, bias=b.to(DEV), act=ops.ACT_LRELU, slope=0.2)
    close(to_nchw(y, 0, 64), ref, what="c4 forward")
    # data-gradient of a 64 -> 3 convolution: incoming gradient has 3 (+1 zero) channels, result 64
    w2 = rnd(3, 64, 3, 3, seed=124, lo=-0.2, hi=0.2)
    xin = rnd(N, 64, H, W, seed=125).requires_grad_(True)
    g = rnd(N, 3, H, W, seed=126)
    (gref,) = torch.autograd.grad(F.conv2d(xin, w2, None, padding=1), xin, g)
    g4 = nhwc_buf(F.pad(g, (0, 0, 0, 0, 0, 1)), fill=0.0)
    wp2, _ = pack(ops, w2.to(DEV), ops.PACK_C4_DGRAD3)
    gx = torch.zeros((N, H, W, 64), device=DEV)
    ops.conv(ops.View(g4), wp2, ops.View(gx), mode=ops.CONV_3x3_C4)
    close(to_nchw(gx, 0, 64), gref, what="c4 dgrad")


@pytest.mark.parametrize("shape", [(2, 20, 37, 64), (1, 33, 16, 16), (1, 64, 96, 64)])
def test_conv_thin_small_cout(shape):
    """tnr_conv_thin: 64 -> 3 forward (bias, alpha) and the 3 <- 64 data-gradient of a 3 -> 64 layer (vector-ALU kernel)."""
    ops = _ops()
    N, H, W, C = shape
    x = rnd(N, C, H, W, seed=131)
    w = rnd(3, C, 3, 3, seed=132, lo=-0.1, hi=0.1)
    b = rnd(3, seed=133)
    ref = 0.5 * F.conv2d(x, w, b, padding=1)
    y = torch.full((N, H, W, 4), 9.0, device=DEV)
    ops.conv_thin(ops.View(nhwc_buf(x)), w.to(DEV), ops.View(y, 0, 3), bias=b.to(DEV), alpha=0.5)
    close(to_nchw(y, 0, 3), ref, what="thin forward")
    assert (y[..., 3] == 9.0).all()                   # a 3-channel view leaves the 4th channel alone
    # data-gradient of a 3 -> C layer: g has C channels, the result 3 (+ a zero 4th channel)
    w2 = rnd(C, 3, 3, 3, seed=134, lo=-0.1, hi=0.1)
    xin = rnd(N, 3, H, W, seed=135).requires_grad_(True)
    g = rnd(N, C, H, W, seed=136)
    (gref,) = torch.autograd.grad(F.conv2d(xin, w2, None, padding=1), xin, g)
    gx = torch.full((N, H, W, 4), 9.0, device=DEV)
    ops.conv_thin(ops.View(nhwc_buf(g)), w2.to(DEV), ops.View(gx), dgrad=True)
    close(to_nchw(gx, 0, 3), gref, what="thin dgrad")
    assert (gx[..., 3] == 0.0).all()


@pytest.mark.parametrize("shape", [(2, 20, 37, 64), (1, 9, 530, 16), (3, 33, 16, 32)])
def test_wgrad_thin_image_layers(shape):
    """tnr_wgrad_thin against autograd: a 3 -> C layer (flip=0) and a C -> 3 layer (flip=1), with bias, alpha, beta;
    W = 530 exercises the 512-pixel row passes."""
    ops = _ops()
    N, H, W, C = shape
    img = rnd(N, 3, H, W, seed=141)
    img4 = nhwc_buf(F.pad(img, (0, 0, 0, 0, 0, 1)), fill=0.0)
    # flip = 0: y = conv(img; w[C,3,3,3]) with gradient g[C]
    w = torch.zeros(C, 3, 3, 3, requires_grad=True)
    g = rnd(N, C, H, W, seed=142)
    (ref_w,) = torch.autograd.grad(F.conv2d(img, w, None, padding=1), w, g)
    dw0, db0 = rnd(C, 3, 3, 3, seed=143), rnd(C, seed=144)
    dw, db = dw0.to(DEV), db0.to(DEV)
    ops.wgrad_thin(ops.View(nhwc_buf(g)), ops.View(img4), dw, db, flip=False, alpha=0.5, beta=1.0)
    close(dw.cpu(), dw0 + 0.5 * ref_w, tol=5e-5, what="thin wgrad 3->C weights")
    close(db.cpu(), db0 + 0.5 * g.sum(dim=(0, 2, 3)), tol=5e-5, what="thin wgrad 3->C bias")
    # flip = 1: y = conv(x[C]; w[3,C,3,3]) with gradient = the 3-channel image tensor
    x = rnd(N, C, H, W, seed=145)
    w2 = torch.zeros(3, C, 3, 3, requires_grad=True)
    (ref_w2,) = torch.autograd.grad(F.conv2d(x, w2, None, padding=1), w2, img)
    dw2, db2 = torch.zeros(3, C, 3, 3, device=DEV), torch.zeros(3, device=DEV)
    ops.wgrad_thin(ops.View(nhwc_buf(x)), ops.View(img4), dw2, db2, flip=True, alpha=1.0, beta=0.0)
    close(dw2.cpu(), ref_w2, tol=5e-5, what="thin wgrad C->3 weights")
    close(db2.cpu(), img.sum(dim=(0, 2, 3)), tol=5e-5, what="thin wgrad C->3 bias")


@pytest.mark.parametrize("shape", [(2, 20, 37, 64), (1, 64, 96, 32), (3, 9, 16, 16)])
def test_conv7x7_image_channels_c4(shape):
    """TNR_CONV_7x7_C4 (ResnetGenerator's image-side layers, ResNet_arch.py:52-55 / :86-88, 49 taps folded into K = 196 -> 208):
    ReflectionPad2d(3) + conv7x7(3 -> C) with bias, and the C <- 3 data-gradient of a conv7x7(C -> 3) taken with respect to its
    reflection-padded input (zero-embedded gradient, zero borders), against torch."""
    ops = _ops()
    N, H, W, C = shape
    x = rnd(N, 3, H, W, seed=151)
    w = rnd(C, 3, 7, 7, seed=152, lo=-0.1, hi=0.1)
    b = rnd(C, seed=153)
    ref = F.conv2d(F.pad(x, (3, 3, 3, 3), mode="reflect"), w, b)
    x4 = nhwc_buf(F.pad(x, (0, 0, 0, 0, 0, 1)), fill=0.0)
    wp, _ = pack(ops, w.to(DEV), ops.PACK_C4_FWD)
    y = torch.full((N, H, W, C + 8), 5.0, device=DEV)
    ops.conv(ops.View(x4), wp, ops.View(y, 4, C), mode=ops.CONV_7x7_C4, bias=b.to(DEV), reflect=True)
    close(to_nchw(y, 4, C), ref, what="7x7 c4 forward (reflect)")
    assert (y[..., :4] == 5.0).all() and (y[..., 4 + C:] == 5.0).all()
    # data-gradient of a C -> 3 layer with respect to its padded input: g embedded at offset 3 in an (H + 6) x (W + 6) canvas
    w2 = rnd(3, C, 7, 7, seed=154, lo=-0.1, hi=0.1)
    xp = rnd(N, C, H + 6, W + 6, seed=155).requires_grad_(True)
    g = rnd(N, 3, H, W, seed=156)
    (gref,) = torch.autograd.grad(F.conv2d(xp, w2, None), xp, g)
    gc = nhwc_buf(F.pad(F.pad(g, (0, 0, 0, 0, 0, 1)), (3, 3, 3, 3)), fill=0.0)
    wp2, _ = pack(ops, w2.to(DEV), ops.PACK_C4_DGRAD3)
    gx = torch.zeros((N, H + 6, W + 6, C), device=DEV)
    ops.conv(ops.View(gc), wp2, ops.View(gx), mode=ops.CONV_7x7_C4)
    close(to_nchw(gx, 0, C), gref, what="7x7 c4 dgrad (padded domain)")


@pytest.mark.parametrize("shape", [(2, 20, 37, 64), (1, 33, 16, 16), (1, 64, 96, 32)])
def test_conv_thin7_small_cout(shape):
    """tnr_conv_thin7: ReflectionPad2d(3) + conv7x7(C -> 3) (bias, alpha) in one launch, and the 3 <- C data-gradient of a
    conv7x7(3 -> C) with respect to its reflection-padded input (pad 6, zero borders, (H + 6) x (W + 6) output)."""
    ops = _ops()
    N, H, W, C = shape
    x = rnd(N, C, H, W, seed=161)
    w = rnd(3, C, 7, 7, seed=162, lo=-0.05, hi=0.05)
    b = rnd(3, seed=163)
    ref = 0.5 * F.conv2d(F.pad(x, (3, 3, 3, 3), mode="reflect"), w, b)
    y = torch.full((N, H, W, 4), 9.0, device=DEV)
    ops.conv_thin7(ops.View(nhwc_buf(x, ctot=C + 8, coff=4), 4, C), w.to(DEV), ops.View(y, 0, 3), pad=3, reflect=True, bias=b.to(DEV), alpha=0.5)
    close(to_nchw(y, 0, 3), ref, what="thin7 forward")
    assert (y[..., 3] == 9.0).all()
    w2 = rnd(C, 3, 7, 7, seed=164, lo=-0.05, hi=0.05)
    xp = rnd(N, 3, H + 6, W + 6, seed=165).requires_grad_(True)
    g = rnd(N, C, H, W, seed=166)
    (gref,) = torch.autograd.grad(F.conv2d(xp, w2, None), xp, g)
    gx = torch.full((N, H + 6, W + 6, 4), 9.0, device=DEV)
    ops.conv_thin7(ops.View(nhwc_buf(g)), w2.to(DEV), ops.View(gx, 0, 3), pad=6, reflect=False, dgrad=True)
    close(to_nchw(gx, 0, 3), gref, what="thin7 dgrad")
    assert (gx[..., 3] == 9.0).all()
    # a 4-channel output view takes the lanes = pixels kernel: the same values, a zero 4th channel (no weights)
    gx4 = torch.full((N, H + 6, W + 6, 4), 9.0, device=DEV)
    ops.conv_thin7(ops.View(nhwc_buf(g)), w2.to(DEV), ops.View(gx4), pad=6, reflect=False, dgrad=True)
    close(to_nchw(gx4, 0, 3), gref, what="thin7 dgrad (4-channel view)")
    assert (gx4[..., 3] == 0.0).all()
    assert float((gx4[..., :3] - gx[..., :3]).abs().max()) <= 2e-5 * (float(gref.abs().max()) + 1.0)


@pytest.mark.parametrize("shape", [(2, 20, 37, 64), (1, 9, 330, 16), (3, 33, 16, 32)])
def test_wgrad_thin7_image_layers(shape):
    """tnr_wgrad_thin7 against autograd: ReflectionPad2d(3) + conv7x7 with the image on the input side (flip=0: over the padded NHWC4
    image, with bias, alpha, beta) and on the output side (flip=1: the wide input read through the reflection map, no padded copy);
    W = 330 exercises the 320-column passes.  Deterministic."""
    ops = _ops()
    N, H, W, C = shape
    img = rnd(N, 3, H, W, seed=171)
    imgp4 = nhwc_buf(F.pad(F.pad(img, (3, 3, 3, 3), mode="reflect"), (0, 0, 0, 0, 0, 1)), fill=0.0)
    w = torch.zeros(C, 3, 7, 7, requires_grad=True)
    g = rnd(N, C, H, W, seed=172)
    (ref_w,) = torch.autograd.grad(F.conv2d(F.pad(img, (3, 3, 3, 3), mode="reflect"), w, None), w, g)
    dw0, db0 = rnd(C, 3, 7, 7, seed=173), rnd(C, seed=174)
    dw, db = dw0.to(DEV), db0.to(DEV)
    gbuf = nhwc_buf(g, ctot=C + 8, coff=4)
    ops.wgrad_thin7(ops.View(gbuf, 4, C), ops.View(imgp4), dw, db, flip=False, alpha=0.5, beta=1.0)
    close(dw.cpu(), dw0 + 0.5 * ref_w, tol=5e-5, what="thin7 wgrad 3->C weights")
    close(db.cpu(), db0 + 0.5 * g.sum(dim=(0, 2, 3)), tol=5e-5, what="thin7 wgrad 3->C bias")
    again = dw0.to(DEV)
    ops.wgrad_thin7(ops.View(gbuf, 4, C), ops.View(imgp4), again, db0.to(DEV), flip=False, alpha=0.5, beta=1.0)
    assert torch.equal(again, dw)
    # flip = 1: y = conv7(reflect_pad(x[C]); w[3,C,7,7]) with gradient = a 3-channel image
    x = rnd(N, C, H, W, seed=175)
    img4 = nhwc_buf(F.pad(img, (0, 0, 0, 0, 0, 1)), fill=0.0)
    w2 = torch.zeros(3, C, 7, 7, requires_grad=True)
    (ref_w2,) = torch.autograd.grad(F.conv2d(F.pad(x, (3, 3, 3, 3), mode="reflect"), w2, None), w2, img)
    dw2 = torch.zeros(3, C, 7, 7, device=DEV)
    ops.wgrad_thin7(ops.View(nhwc_buf(x)), ops.View(img4), dw2, None, flip=True, rpad=3, off=-6, alpha=1.0, beta=0.0)
    close(dw2.cpu(), ref_w2, tol=5e-5, what="thin7 wgrad C->3 weights")


def test_conv_direct_splitk():
    """The direct 3x3 kernel with a split-K workspace (512 -> 512 channels at 16x16: 16 tiles for 512 slots)."""
    ops = _ops()
    N, H, W, Cin, Cout = 2, 16, 16, 512, 512
    x, w, b = rnd(N, Cin, H, W, seed=111), rnd(Cout, Cin, 3, 3, seed=112, lo=-0.03, hi=0.03), rnd(Cout, seed=113)
    ref = F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.2)
    wp, _ = pack(ops, w.to(DEV), ops.PACK_FWD)
    y = torch.zeros((N, H, W, Cout), device=DEV)
    from trainner_amd.hip import ConvDesc
    d = ConvDesc()
    ops._conv_desc(d, ops.View(nhwc_buf(x)), wp, ops.View(y), ops.CONV_3x3, bias=b.to(DEV), act=ops.ACT_LRELU, slope=0.2)
    import ctypes
    from trainner_amd import hip
    assert hip.load().tnr_conv_workspace_bytes(ctypes.byref(d)) > 0, "launch should be split"
    ops.conv(ops.View(nhwc_buf(x)), wp, ops.View(y), bias=b.to(DEV), act=ops.ACT_LRELU, slope=0.2)
    close(to_nchw(y, 0, Cout), ref, what="direct split-K")


def test_wgrad_group_dense_block():
    """The 64-input pieces of a dense block's conv1 / conv3 (2x) / conv4 in ONE launch, with biases,
    alpha and beta, against autograd; then the error path for layers of different tile classes."""
    ops = _ops()
    N, H, W, nf, gc = 2, 24, 40, 64, 32
    x = rnd(N, 192, H, W, seed=41)
    gs = {k: rnd(N, gc, H, W, seed=42 + k) for k in (0, 2, 3)}
    xb = nhwc_buf(x, 192, 0)
    gbuf = torch.zeros((N, H, W, 192), device=DEV)
    items, checks = [], []
    for k, (lo, n) in ((0, (0, 64)), (2, (0, 64)), (2, (64, 64)), (3, (96, 64))):
        cin = nf + gc * k
        goff = nf + (3 - k) * gc
        gbuf[..., goff:goff + gc] = gs[k].permute(0, 2, 3, 1).to(DEV)
        key = "dw%d" % k
        if key not in {c[0] for c in checks}:
            w = torch.zeros(gc, cin, 3, 3, requires_grad=True)
            (ref_w,) = torch.autograd.grad(F.conv2d(x[:, :cin], w, None, padding=1), w, gs[k])
            dw0, db0 = rnd(gc, cin, 3, 3, seed=50 + k), rnd(gc, seed=60 + k)
            dw, db = dw0.to(DEV), db0.to(DEV)
            checks.append((key, dw, db, dw0 + 0.25 * ref_w, db0 + 0.25 * gs[k].sum(dim=(0, 2, 3)), lo, n, cin))
        else:
            (_, dw, db, *_r) = [c for c in checks if c[0] == key][0]
        items.append(dict(x=ops.View(xb, lo, n), g=ops.View(gbuf, goff, gc), dw=dw, db=db if lo == 0 else None,
                          cin_begin=lo, alpha=0.25, beta=1.0))
    ops.wgrad_group(items)
    for key, dw, db, ref_w, ref_b, lo, n, cin in checks:
        covered = {0: (0, 64), 2: (0, 128), 3: (96, 160)}[int(key[2:])]
        sl = slice(*covered)
        scale = ref_w.abs().max().item() + 1.0
        assert (dw.cpu()[:, sl] - ref_w[:, sl]).abs().max().item() <= 5e-5 * scale, key
        if covered[0] == 0:
            assert (db.cpu() - ref_b).abs().max().item() <= 5e-5 * (ref_b.abs().max().item() + 1), key + " bias"
    with pytest.raises(RuntimeError, match="tile class"):
        ops.wgrad_group([dict(x=ops.View(xb, 0, 64), g=ops.View(gbuf, 64, gc), dw=checks[0][1]),
                         dict(x=ops.View(xb, 0, 96), g=ops.View(gbuf, 96, gc), dw=torch.zeros(gc, 96, 3, 3, device=DEV))])


@pytest.mark.parametrize("shape", [(2, 24, 40), (1, 37, 19), (3, 16, 16)])
def test_wgrad_cout_pairs_of_a_dense_block(shape):
    """tnr_wgrad_desc.cout_split: a dense block's [g4 | g3] over x | x1 | x2 (128 channels) and [g2 | g1] over x (64) as 64-cout PAIR jobs
    next to conv5's 64 x 192 job in ONE launch (what RRDBNet's backward enqueues per RRDB), plus the two 32-channel remainders in a
    second launch -- every layer's weight and bias gradient against autograd, with alpha / beta, and the untouched parts of dw intact."""
    ops = _ops()
    N, H, W = shape
    nf, gc = 64, 32
    x = rnd(N, 192, H, W, seed=71)
    g5 = rnd(N, nf, H, W, seed=72)
    gs = {k: rnd(N, gc, H, W, seed=73 + k) for k in range(4)}           # k = 0 .. 3 <-> conv1 .. conv4
    xb = nhwc_buf(x, 192, 0)
    GP = torch.zeros((N, H, W, 192), device=DEV)
    GP[..., :nf] = g5.permute(0, 2, 3, 1).to(DEV)
    for k in range(4):
        o = nf + (3 - k) * gc
        GP[..., o:o + gc] = gs[k].permute(0, 2, 3, 1).to(DEV)
    ref, dw, db, dw0, db0 = {}, {}, {}, {}, {}
    for k in range(5):
        cin, cout, gk = nf + gc * k, (gc if k < 4 else nf), (gs[k] if k < 4 else g5)
        w = torch.zeros(cout, cin, 3, 3, requires_grad=True)
        (rw,) = torch.autograd.grad(F.conv2d(x[:, :cin], w, None, padding=1), w, gk)
        dw0[k], db0[k] = rnd(cout, cin, 3, 3, seed=80 + k), rnd(cout, seed=90 + k)
        dw[k], db[k] = dw0[k].to(DEV), db0[k].to(DEV)
        ref[k] = (dw0[k] + 0.5 * rw, db0[k] + 0.5 * gk.sum(dim=(0, 2, 3)))
    V = ops.View
    big = [dict(x=V(xb), g=V(GP, 0, nf), dw=dw[4], db=db[4], alpha=0.5, beta=1.0)]
    rest = []
    for ka, kb in ((3, 2), (1, 0)):
        cin = nf + gc * kb
        big.append(dict(x=V(xb, 0, cin), g=V(GP, nf + (3 - ka) * gc, 2 * gc), dw=dw[ka], db=db[ka], alpha=0.5, beta=1.0,
                        pair=(dw[kb], db[kb], gc)))
        rest.append(dict(x=V(xb, cin, gc), g=V(GP, nf + (3 - ka) * gc, gc), dw=dw[ka], cin_begin=cin, alpha=0.5, beta=1.0))
    ops.wgrad_group(big)
    ops.wgrad_group(rest)
    for k in range(5):
        scale = ref[k][0].abs().max().item() + 1.0
        assert (dw[k].cpu() - ref[k][0]).abs().max().item() <= 5e-5 * scale, ("dw", k)
        assert (db[k].cpu() - ref[k][1]).abs().max().item() <= 5e-5 * (ref[k][1].abs().max().item() + 1.0), ("db", k)
    with pytest.raises(RuntimeError, match="cout pair"):
        ops.wgrad_group([dict(x=V(xb, 0, 64), g=V(GP, nf, 2 * gc), dw=dw[3], pair=(dw[2], None, 16))])


def test_layout_roundtrip_and_norm():
    ops = _ops()
    x = rnd(2, 3, 10, 14, seed=31)
    scale, shift = torch.tensor([2.0, 3.0, 4.0]), torch.tensor([0.1, 0.2, 0.3])
    buf = torch.full((2, 10, 14, 4), 9.0, device=DEV)
    ops.nchw_to_nhwc(x.to(DEV), ops.View(buf), Cpad=4, scale=scale.to(DEV), shift=shift.to(DEV))
    ref = x * scale.view(1, 3, 1, 1) + shift.view(1, 3, 1, 1)
    close(to_nchw(buf, 0, 3), ref, tol=1e-6, what="nchw->nhwc")
    assert (buf[..., 3] == 0).all()
    out = torch.full((2, 3, 10, 14), 1.0, device=DEV)
    ops.nhwc_to_nchw(ops.View(buf, 0, 3), out, scale=scale.to(DEV), accumulate=True)
    close(out.cpu(), ref * scale.view(1, 3, 1, 1) + 1.0, tol=1e-6, what="nhwc->nchw")


def test_upsample_pixelshuffle_pool():
    ops = _ops()
    N, H, W, C = 2, 6, 10, 64
    # nearest x2 backward with mask
    gup, m = rnd(N, C, 2 * H, 2 * W, seed=32), rnd(N, C, H, W, seed=33)
    ref = F.avg_pool2d(gup, 2) * 4 * torch.where(m > 0, torch.ones_like(m), torch.full_like(m, 0.2))
    gx = torch.zeros((N, H, W, C), device=DEV)
    ops.upsample2x_bwd(ops.View(nhwc_buf(gup)), ops.View(gx), mask=ops.View(nhwc_buf(m)), mslope=0.2)
    close(to_nchw(gx, 0, C), ref, tol=1e-6, what="upsample2x bwd")
    # pixel shuffle forward / backward
    x = rnd(N, 4 * C, H, W, seed=34)
    ref = F.pixel_shuffle(x, 2)
    y = torch.zeros((N, 2 * H, 2 * W, C), device=DEV)
    ops.depth_to_space(ops.View(nhwc_buf(x)), ops.View(y))
    close(to_nchw(y, 0, C), ref, tol=0, what="depth_to_space")
    gy = rnd(N, C, 2 * H, 2 * W, seed=35)
    mk = torch.where(ref > 0, torch.ones_like(ref), torch.zeros_like(ref))
    refg = F.pixel_unshuffle(gy * mk, 2)
    gxx = torch.zeros((N, H, W, 4 * C), device=DEV)
    ops.space_to_depth_bwd(ops.View(nhwc_buf(gy)), ops.View(gxx), mask=ops.View(y), mslope=0.0)
    close(to_nchw(gxx, 0, 4 * C), refg, tol=0, what="space_to_depth bwd")
    # max pool forward / backward (through ReLU)
    a = F.relu(rnd(N, C, 2 * H, 2 * W, seed=36)).requires_grad_(True)
    pooled = F.max_pool2d(a, 2, 2)
    gp = rnd(N, C, H, W, seed=37)
    pre = rnd(N, C, 2 * H, 2 * W, seed=36).requires_grad_(True)
    (refgx,) = torch.autograd.grad(F.max_pool2d(F.relu(pre), 2, 2), pre, gp)
    ab = nhwc_buf(a.detach())
    pb = torch.zeros((N, H, W, C), device=DEV)
    ops.maxpool2_fwd(ops.View(ab), ops.View(pb))
    close(to_nchw(pb, 0, C), pooled.detach(), tol=0, what="maxpool fwd")
    gxb = torch.zeros((N, 2 * H, 2 * W, C), device=DEV)
    ops.maxpool2_bwd(ops.View(nhwc_buf(gp)), ops.View(ab), ops.View(gxb))
    close(to_nchw(gxb, 0, C), refgx, tol=0, what="maxpool bwd")


@pytest.mark.parametrize("C", [16, 64, 512])
def test_batchnorm_train(C):
    ops = _ops()
    N, H, W = 4, 9, 11
    z = (rnd(N, C, H, W, seed=38) * 3 + 0.7).requires_grad_(True)
    gamma, beta = rnd(C, seed=39, lo=0.5, hi=1.5).requires_grad_(True), rnd(C, seed=40).requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    y = F.leaky_relu(F.batch_norm(z, rm, rv, gamma, beta, True, 0.1, 1e-5), 0.2)
    gy = rnd(N, C, H, W, seed=41)
    gz_ref, gg_ref, gb_ref = torch.autograd.grad(y, (z, gamma, beta), gy)
    zb = nhwc_buf(z.detach())
    yb = torch.zeros_like(zb)
    drm, drv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    nbt = torch.zeros((), dtype=torch.int64, device=DEV)
    sm, si = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    gd, bd = gamma.detach().to(DEV), beta.detach().to(DEV)
    ops.bn_train_fwd(ops.View(zb), ops.View(yb), gd, bd, drm, drv, nbt, sm, si)
    close(to_nchw(yb, 0, C), y.detach(), what="bn fwd")
    close(drm.cpu(), rm, tol=1e-5, what="running mean")
    close(drv.cpu(), rv, tol=1e-5, what="running var")
    assert int(nbt) == 1
    gzb = torch.zeros_like(zb)
    dg, dbt = torch.ones(C, device=DEV), torch.ones(C, device=DEV)
    ops.bn_train_bwd(ops.View(nhwc_buf(gy)), ops.View(yb), ops.View(zb), ops.View(gzb), gd, sm, si, dgamma=dg, dbeta=dbt,
                     acc_beta=1.0)
    close(to_nchw(gzb, 0, C), gz_ref, tol=1e-4, what="bn bwd dx")
    close(dg.cpu() - 1, gg_ref, tol=1e-4, what="bn dgamma")
    close(dbt.cpu() - 1, gb_ref, tol=1e-4, what="bn dbeta")
    # the mask recomputed from z (tnr_bn_train_bwd_z; what Discriminator_VGG's backward calls): BIT-identical to the mask read from y
    gzb2 = torch.zeros_like(zb)
    dg2, dbt2 = torch.ones(C, device=DEV), torch.ones(C, device=DEV)
    assert ops.BN_MASK_FROM_Z
    ops.bn_train_bwd(ops.View(nhwc_buf(gy)), None, ops.View(zb), ops.View(gzb2), gd, sm, si, dgamma=dg2, dbeta=dbt2, acc_beta=1.0, beta=bd)
    assert torch.equal(gzb2, gzb) and torch.equal(dg2, dg) and torch.equal(dbt2, dbt)


def test_linear_and_losses_and_optim():
    ops = _ops()
    from trainner_amd import hip
    N, In, Out = 5, 300, 17
    x, w, b = rnd(N, In, seed=42), rnd(Out, In, seed=43, lo=-0.1, hi=0.1).requires_grad_(True), rnd(Out, seed=44).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    y = F.leaky_relu(F.linear(xr, w, b), 0.2)
    gy = rnd(N, Out, seed=45)
    gx_r, gw_r, gb_r = torch.autograd.grad(y, (xr, w, b), gy)
    xd, wd, bd = x.to(DEV), w.detach().to(DEV), b.detach().to(DEV)
    yd = torch.zeros(N, Out, device=DEV)
    ops.linear_fwd(xd, wd, bd, yd, act=ops.ACT_LRELU, slope=0.2)
    close(yd.cpu(), y.detach(), what="linear fwd")
    gxd, dwd, dbd = torch.zeros(N, In, device=DEV), torch.zeros(Out, In, device=DEV), torch.zeros(Out, device=DEV)
    ops.linear_bwd(xd, wd, gy.to(DEV), yd, gx=gxd, dw=dwd, db=dbd, mslope=0.2)
    close(gxd.cpu(), gx_r, what="linear gx")
    close(dwd.cpu(), gw_r, what="linear dw")
    close(dbd.cpu(), gb_r, what="linear db")
    # L1
    a, bb = rnd(3, 3, 20, 20, seed=46).requires_grad_(True), rnd(3, 3, 20, 20, seed=47)
    l = F.l1_loss(a, bb)
    (ga_r,) = torch.autograd.grad(l * 0.7, a)
    ad, bbd = a.detach().to(DEV), bb.to(DEV)
    out = torch.zeros((), device=DEV)
    ops.l1_mean_fwd(ad, bbd, 1.0, out)
    assert abs(float(out) - float(l)) < 1e-6
    gsc = torch.tensor([0.7], device=DEV)
    gad = torch.zeros_like(ad)
    ops.l1_mean_bwd(ad, bbd, 1.0, gsc, gad)
    close(gad.cpu(), ga_r, tol=1e-6, what="l1 bwd")
    # relativistic BCE, both stages
    from oracle import sr_oracle as O
    lib = hip.load()
    # (6 logits: the one-block kernels; 3 x 64 x 70 per-pixel logits of a U-Net discriminator: the two-stage fixed-order reductions)
    for stage, nlog in ((0, 6), (1, 6), (0, 13440), (1, 13440)):
        pf = rnd(nlog, 1, seed=48).requires_grad_(True)
        pr = rnd(nlog, 1, seed=49).requires_grad_(True)
        if stage == 0:
            loss = 5e-3 * O.ragan_g_loss(pf, pr)
            gf_r, = torch.autograd.grad(loss, pf)
            gr_r = torch.zeros_like(pr)
        else:
            lr_, lf_ = O.ragan_d_loss(pf, pr)
            loss = (lr_ + lf_) * 0.5
            gf_r, gr_r = torch.autograd.grad(loss, (pf, pr))
        wgt = 5e-3 if stage == 0 else 1.0
        pfd, prd = pf.detach().to(DEV).view(-1), pr.detach().to(DEV).view(-1)
        sums, o5 = torch.zeros(8, device=DEV), torch.zeros(5, device=DEV)
        gf, gr = torch.zeros(nlog, device=DEV), torch.zeros(nlog, device=DEV)
        s = hip.stream()
        ws = torch.zeros(lib.tnr_reduce_workspace_bytes() // 8, dtype=torch.float64, device=DEV)
        hip.check(lib.tnr_ragan_phase_a(pfd.data_ptr(), prd.data_ptr(), nlog, sums.data_ptr(), ws.data_ptr(), s))
        hip.check(lib.tnr_ragan_phase_b(pfd.data_ptr(), prd.data_ptr(), nlog, stage, sums.data_ptr(), ws.data_ptr(), s))
        hip.check(lib.tnr_ragan_phase_c(pfd.data_ptr(), prd.data_ptr(), nlog, stage, wgt, sums.data_ptr(), o5.data_ptr(),
                                        gf.data_ptr(), gr.data_ptr(), s))
        assert abs(float(o5[0]) - float(loss)) < 1e-6, (stage, float(o5[0]), float(loss))
        close(gf.cpu().view(nlog, 1), gf_r, tol=1e-6, what="ragan gf")
        close(gr.cpu().view(nlog, 1), gr_r, tol=1e-6, what="ragan gr")
        if nlog > 4096:       # without scratch: the one-block kernels, the same numbers
            sums1, o51 = torch.zeros(8, device=DEV), torch.zeros(5, device=DEV)
            hip.check(lib.tnr_ragan_phase_a(pfd.data_ptr(), prd.data_ptr(), nlog, sums1.data_ptr(), None, s))
            hip.check(lib.tnr_ragan_phase_b(pfd.data_ptr(), prd.data_ptr(), nlog, stage, sums1.data_ptr(), None, s))
            assert (sums1 - sums).abs().max().item() <= 1e-6 * sums.abs().max().item()
    # clip + Adam against torch
    n = 10000
    p0, g0 = rnd(n, seed=50), rnd(n, seed=51) * 0.01
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=1e-4)
    pd, gd_ = p0.to(DEV), g0.to(DEV)
    md, vd = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for t in range(1, 4):
        pt.grad = g0.clone() * t
        torch.nn.utils.clip_grad_norm_([pt], 0.1)
        opt.step()
        gcur = (g0 * t).to(DEV)
        ss = torch.zeros(1, dtype=torch.float64, device=DEV)
        ops.sumsq(gcur, ss)
        ops.clip_by_norm(gcur, ss, 0.1)
        close(gcur.cpu(), pt.grad, tol=1e-6, what="clip")
        ops.adam_step(pd, gcur, md, vd, 1e-4 / (1 - 0.9 ** t), 0.9, 0.999, math.sqrt(1 - 0.999 ** t), 1e-8)
    assert (pd.cpu() - pt.detach()).abs().max().item() < 2e-7, "adam"


def test_fault_latch_blocks_the_optimizer_step(tmp_path):
    """The fault latch (tnr_set_fault_word): once a tile hand-off wait of a one-launch dense block has given up, every Adam launch
    (tnr_adam_step_guarded) leaves weights and moments untouched ON THE DEVICE -- no host sync is needed to keep a faulted step from
    being applied -- and the next host-side check (log read-out / checkpoint) raises.  The latch is set by hand here."""
    import test_gpu_step as TS
    from trainner_amd import hip
    ops = _ops()
    n = 5000
    p0, g0 = rnd(n, seed=60), rnd(n, seed=61) * 0.01
    pd, gd_ = p0.to(DEV), g0.to(DEV)
    md, vd = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    word = ops.fault_word(pd.device)
    assert ops.chain_error_flag() == 0
    opt, model = TS.build_engine_model(dict(nb=1, batch=2, crop=64, d_nf=16), tmp_path)
    LR, HR = __import__("oracle.detrand", fromlist=["x"]).synthetic_pair(2, 64, 5)
    try:
        word.fill_(1)
        ops.adam_step(pd, gd_, md, vd, 1e-4, 0.9, 0.999, 1.0, 1e-8)
        assert torch.equal(pd.cpu(), p0) and float(md.abs().max()) == 0.0 and float(vd.abs().max()) == 0.0
        before = {k: v.detach().clone() for k, v in model.netG.state_dict().items()}
        model.feed_data({"LR": LR, "HR": HR})
        model.optimize_parameters(1)                    # runs, but applies nothing
        for k, v in model.netG.state_dict().items():
            assert torch.equal(v, before[k]), k
        with pytest.raises(hip.HipEngineError):
            model.get_current_log()
        with pytest.raises(hip.HipEngineError):
            model.save(1)
    finally:
        word.zero_()
    ops.adam_step(pd, gd_, md, vd, 1e-4, 0.9, 0.999, 1.0, 1e-8)
    assert not torch.equal(pd.cpu(), p0) and ops.chain_error_flag() == 0


@pytest.mark.parametrize("t", [0, 1, 2, 3, 4])
def test_dense_block_gradient_pack(t):
    """tnr_pack_dense_dgrad + conv_tile == sum over consumers of conv_transpose(g_k, W_k[:, target])."""
    ops = _ops()
    nf, gc, N, H, W = 64, 32, 1, 20, 36
    ws = [rnd(gc if k < 4 else nf, nf + k * gc, 3, 3, seed=60 + k, lo=-0.1, hi=0.1) for k in range(5)]
    gp = rnd(N, nf + 4 * gc, H, W, seed=70)                      # [g5 | g4 | g3 | g2 | g1]
    scale5 = 0.04
    tlo, ntar = (0, nf) if t == 4 else (nf + (3 - t) * gc, gc)
    ref = scale5 * F.conv_transpose2d(gp[:, :nf], ws[4][:, tlo:tlo + ntar], None, padding=1)
    for m in range(t):
        ref = ref + F.conv_transpose2d(gp[:, nf + m * gc: nf + (m + 1) * gc], ws[3 - m][:, tlo:tlo + ntar], None, padding=1)
    dp = ops.DensePacker(torch.device(DEV))
    wd = [w.to(DEV) for w in ws]
    idx = dp.add_block(wd, nf, gc, scale5)
    dp.run()
    gb = nhwc_buf(gp)
    yb = torch.zeros((N, H, W, ntar), device=DEV)
    ops.conv(ops.View(gb, 0, nf + t * gc), dp.get(idx[t]), ops.View(yb))
    close(to_nchw(yb, 0, ntar), ref, what="dense-block gradient step %d" % t)


@pytest.mark.parametrize("shape", [(2, 5, 7, 8), (1, 1, 1, 4), (2, 16, 24, 32), (1, 1, 9, 12)])
def test_bilinear2x_and_skip_kernels(shape):
    """tnr_bilinear2x_fwd / _bwd against F.interpolate(scale_factor=2, mode='bilinear', align_corners=False) and its
    autograd adjoint (UNetDiscriminator's decoder, discriminators.py:745-769), inside channel windows of wider buffers;
    tnr_add2 / tnr_mask_copy (skip sums and the out-of-place LeakyReLU backward) are exact."""
    ops = _ops()
    N, H, W, C = shape
    x = rnd(N, C, H, W, seed=301)
    xb = nhwc_buf(x, C + 8, 4)
    yb = torch.full((N, 2 * H, 2 * W, C + 4), 7.0, device=DEV)
    ops.bilinear2x_fwd(ops.View(xb, 4, C), ops.View(yb, 0, C))
    ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    close(to_nchw(yb, 0, C), ref, tol=2e-6, what="bilinear fwd")
    assert float(yb[..., C:].min()) == 7.0                              # channels outside the view are untouched
    g = rnd(N, C, 2 * H, 2 * W, seed=302)
    xr = x.clone().requires_grad_(True)
    (gref,) = torch.autograd.grad(F.interpolate(xr, scale_factor=2, mode="bilinear", align_corners=False), xr, g)
    gb = nhwc_buf(g, C, 0)
    m = rnd(N, C, H, W, seed=303)
    mb = nhwc_buf(m, C, 0)
    gx = torch.zeros(N, H, W, C, device=DEV)
    gz = torch.zeros(N, H, W, C + 4, device=DEV)
    ops.bilinear2x_bwd(ops.View(gb), gx=ops.View(gx), gz=ops.View(gz, 4, C), mask=ops.View(mb), mslope=0.2)
    close(to_nchw(gx, 0, C), gref, tol=5e-6, what="bilinear bwd")
    gate = torch.where(m > 0, torch.ones_like(m), torch.full_like(m, 0.2))
    close(to_nchw(gz, 4, C), gref * gate, tol=5e-6, what="bilinear bwd gated")
    gz2 = torch.zeros(N, H, W, C, device=DEV)
    ops.bilinear2x_bwd(ops.View(gb), gx=None, gz=ops.View(gz2), mask=ops.View(mb), mslope=0.2)
    assert torch.equal(gz2, gz[..., 4:].contiguous())
    # skip sum and out-of-place gate
    a, b = rnd(N, C, H, W, seed=304), rnd(N, C, H, W, seed=305)
    ab, bb = nhwc_buf(a, C, 0), nhwc_buf(b, C + 4, 4)
    d = torch.zeros(N, H, W, C, device=DEV)
    ops.add2(ops.View(d), ops.View(ab), ops.View(bb, 4, C))
    assert torch.equal(to_nchw(d, 0, C), a + b)
    ops.mask_copy(ops.View(d), ops.View(ab), ops.View(mb), 0.2)
    assert torch.equal(to_nchw(d, 0, C), a * gate)
    assert torch.equal(to_nchw(ab, 0, C), a)                            # the source stays intact


def _bf(t):
    """round to bf16 (nearest even) and back: what TNR_MMA_BF16 feeds the matrix core"""
    return t.to(torch.bfloat16).to(torch.float32)


def test_bf16_operand_mode(monkeypatch):
    """TNR_MMA_BF16 (`use_amp: true`): activations and weights rounded to bf16 in front of the matrix core, fp32 accumulate,
    fp32 epilogue / storage.  Products of bf16 values are exact in fp32, so every launch must equal the fp32 PyTorch op
    on bf16-ROUNDED operands up to summation order (tolerance as for the fp32 kernels) -- forward with fused epilogue,
    strided forward, both data-gradients, the up-sampling stager, the dense-block chain, and every weight-gradient class."""
    from trainner_amd import hip
    ops = _ops()
    monkeypatch.setattr(ops, "MMA", hip.MMA_BF16)
    # 3x3 forward, window of a wider buffer, bias + LeakyReLU + residual
    N, H, W, Cin, Cout = 2, 20, 37, 96, 32
    x, w, b, r = rnd(N, Cin, H, W, seed=1), rnd(Cout, Cin, 3, 3, seed=2, lo=-0.2, hi=0.2), rnd(Cout, seed=3), rnd(N, Cout, H, W, seed=4)
    ref = F.leaky_relu(F.conv2d(_bf(x), _bf(w), b, padding=1), 0.2) * 0.5 + r
    xb, rb = nhwc_buf(x, 192, 0), nhwc_buf(r)
    yb = torch.full((N, H, W, 192), -3.0, device=DEV)
    wp, _k1 = pack(ops, w.to(DEV), ops.PACK_FWD)
    ops.conv(ops.View(xb, 0, Cin), wp, ops.View(yb, 96, Cout), bias=b.to(DEV), act=ops.ACT_LRELU, slope=0.2, alpha=0.5, r1=ops.View(rb))
    close(to_nchw(yb, 96, Cout), ref, what="bf16 conv3x3")
    plain = F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.2) * 0.5 + r
    assert (to_nchw(yb, 96, Cout) - plain).abs().max().item() > 1e-4        # and it really is the rounded-operand result
    # 64-cout tile class, nearest-x2 stager
    w2 = rnd(64, 64, 3, 3, seed=5, lo=-0.2, hi=0.2)
    x2 = rnd(1, 64, 12, 20, seed=6)
    wp2, _k2 = pack(ops, w2.to(DEV), ops.PACK_FWD)
    y2 = torch.zeros(1, 24, 40, 64, device=DEV)
    ops.conv(ops.View(nhwc_buf(x2)), wp2, ops.View(y2), mode=ops.CONV_3x3_UP2)
    close(to_nchw(y2, 0, 64), F.conv2d(F.interpolate(_bf(x2), scale_factor=2.0, mode="nearest"), _bf(w2), None, padding=1), what="bf16 up2")
    # 4x4 s2 forward and its data-gradient, 3x3 data-gradient
    w4 = rnd(64, 32, 4, 4, seed=7, lo=-0.2, hi=0.2)
    x4 = rnd(2, 32, 16, 24, seed=8)
    wp4, _k4 = pack(ops, w4.to(DEV), ops.PACK_FWD_S2D)
    y4 = torch.zeros(2, 8, 12, 64, device=DEV)
    ops.conv(ops.View(nhwc_buf(x4)), wp4, ops.View(y4), mode=ops.CONV_4x4_S2)
    close(to_nchw(y4, 0, 64), F.conv2d(_bf(x4), _bf(w4), None, stride=2, padding=1), what="bf16 4x4s2")
    g4 = rnd(2, 64, 8, 12, seed=9)
    wd4, _k5 = pack(ops, w4.to(DEV), ops.PACK_DGRAD_S2)
    gx4 = torch.zeros(2, 16, 24, 32, device=DEV)
    ops.conv(ops.View(nhwc_buf(g4)), wd4, ops.View(gx4), mode=ops.DGRAD_4x4_S2)
    close(to_nchw(gx4, 0, 32), F.conv_transpose2d(_bf(g4), _bf(w4), None, stride=2, padding=1), what="bf16 dgrad4x4s2")
    g3 = rnd(N, Cout, H, W, seed=10)
    wd3, _k6 = pack(ops, w.to(DEV), ops.PACK_DGRAD_3x3)
    gx3 = torch.zeros(N, H, W, Cin, device=DEV)
    ops.conv(ops.View(nhwc_buf(g3)), wd3, ops.View(gx3))
    close(to_nchw(gx3, 0, Cin), F.conv_transpose2d(_bf(g3), _bf(w), None, padding=1), what="bf16 dgrad3x3")
    # weight gradients: every workgroup tile class (32 x 32/64/96/128 pixel-split and shared-pixel, 64 x 64, strided)
    for (mode, Nn, Hh, Ww, ci, co) in (("3x3", 2, 32, 32, 32, 32), ("3x3", 1, 32, 48, 64, 32), ("3x3", 2, 16, 16, 96, 32),
                                       ("3x3", 1, 24, 32, 128, 32), ("3x3", 2, 16, 32, 192, 64), ("s2", 2, 16, 32, 64, 64)):
        xx = rnd(Nn, ci, Hh, Ww, seed=11)
        k = 3 if mode == "3x3" else 4
        ww = rnd(co, ci, k, k, seed=12).requires_grad_(True)
        yy = F.conv2d(_bf(xx), ww, None, padding=1) if mode == "3x3" else F.conv2d(_bf(xx), ww, None, stride=2, padding=1)
        gg = rnd(*yy.shape, seed=13)
        (rw,) = torch.autograd.grad(yy, ww, _bf(gg))
        dw, db = torch.zeros(co, ci, k, k, device=DEV), torch.zeros(co, device=DEV)
        ops.wgrad(ops.View(nhwc_buf(xx)), ops.View(nhwc_buf(gg)), dw, db, mode=ops.CONV_3x3 if mode == "3x3" else ops.CONV_4x4_S2, beta=0.0)
        close(dw.cpu(), rw, tol=5e-5, what="bf16 wgrad %s %d->%d" % (mode, ci, co))
        close(db.cpu(), gg.sum(dim=(0, 2, 3)), tol=5e-5, what="bf16 wgrad bias (fp32 sum of the un-rounded gradient)")
    # dense-block chain == per-layer launches, bit for bit, in this mode too
    nf, gc = 64, 32
    ws = [rnd(gc, nf + k * gc, 3, 3, seed=70 + k, lo=-0.05, hi=0.05).to(DEV) for k in range(4)] + [rnd(nf, nf + 4 * gc, 3, 3, seed=74, lo=-0.05, hi=0.05).to(DEV)]
    p = ops.WeightPacker(DEV)
    idx = [p.add(wk, ops.PACK_FWD) for wk in ws]
    p.run()
    x0 = rnd(3, nf, 40, 72, seed=90).permute(0, 2, 3, 1).contiguous().to(DEV)

    def run(chain):
        buf = torch.zeros((3, 40, 72, nf + 4 * gc), device=DEV)
        buf[..., :nf] = x0
        out = torch.zeros((3, 40, 72, nf), device=DEV)
        st = [dict(x=ops.View(buf, 0, nf + gc * k), wp=p.get(idx[k]), y=ops.View(buf, nf + gc * k, gc), act=ops.ACT_LRELU, slope=0.2,
                   fresh_from=(nf + gc * (k - 1) if k else None)) for k in range(4)]
        st.append(dict(x=ops.View(buf), wp=p.get(idx[4]), y=ops.View(out), alpha=0.2, r1=ops.View(buf, 0, nf), fresh_from=nf + 3 * gc))
        if chain:
            ops.conv_chain(st)
        else:
            for d in st:
                ops.conv(**{kk: v for kk, v in d.items() if kk != "fresh_from"})
        torch.cuda.synchronize()
        return buf.cpu(), out.cpu()

    monkeypatch.setattr(ops, "X3_D4", False)              # per-layer launches on conv_tile_body<BF = 1> (not the weight-stream kernel's bf16 form)
    rb_, ro_ = run(False)
    monkeypatch.setattr(ops, "AMP_SWEEP", False)          # tnr_conv_chain: the same k-order inside every MFMA as that per-layer kernel
    gb_, go_ = run(True)
    assert torch.equal(gb_, rb_) and torch.equal(go_, ro_) and ops.chain_error_flag() == 0
    monkeypatch.setattr(ops, "X3_D4", True)
    monkeypatch.setattr(ops, "AMP_SWEEP", True)           # the sweep's bf16-operand form (the default): bf16 resolution downstream of stage 1
    gs_, os_ = run(True)                                  # (test_amp_dense_block_sweep_agrees_with_per_layer)
    sc_ = max(1.0, float(rb_.abs().max()))
    assert float((gs_ - rb_).abs().max()) <= 4e-3 * sc_ and float((os_ - ro_).abs().max()) <= 4e-3 * sc_ and ops.chain_error_flag() == 0
    x1 = F.leaky_relu(F.conv2d(_bf(x0.cpu().permute(0, 3, 1, 2)), _bf(ws[0].cpu()), None, padding=1), 0.2)
    close(rb_[..., nf:nf + gc].permute(0, 3, 1, 2), x1, what="bf16 chain stage 0")


def test_bf16x3_split_operand_mode(monkeypatch):
    """TNR_MMA_BF16X3: fp32 arithmetic on the bf16 matrix core -- every operand split EXACTLY into hi + mid + lo (three bf16
    values), the six largest of the nine exact partial products accumulated in fp32.  Not a reduced-precision mode: on every
    geometry (3x3 forward with fused epilogue in a channel window, 64-cout and 32-cout tile classes, nearest-x2 stager, 4x4 s2
    forward, both data-gradients, the taps-in-K image kernel, reflection borders, the im2col / split-K route, the dense-block
    chain instantiation) the result must pass the fp32 kernels' own tolerance AND its error against an fp64 reference must not
    exceed 1.5 x the fp32 matrix-core path's error on the same inputs (+ 2e-7 of the output scale)."""
    from trainner_amd import hip
    ops = _ops()
    worst = []

    def both(name, launch, ref64):
        err = {}
        for mma in (hip.MMA_F32, hip.MMA_BF16X3):
            monkeypatch.setattr(ops, "MMA", mma)
            got = launch().double()
            err[mma] = (got - ref64).abs().max().item()
        scale = ref64.abs().max().item()
        worst.append((name, err[hip.MMA_F32] / scale, err[hip.MMA_BF16X3] / scale))
        assert err[hip.MMA_BF16X3] <= 2e-5 * (scale + 1.0), (name, err)
        assert err[hip.MMA_BF16X3] <= 1.5 * err[hip.MMA_F32] + 2e-7 * scale, (name, err)

    d = torch.double
    # 3x3 forward, window of a wider buffer, bias + LeakyReLU + residual; 32-cout and 64-cout tile classes, large K
    for (N, H, W, Cin, Cout, seed) in ((2, 20, 37, 96, 32, 1), (1, 24, 40, 192, 64, 2), (1, 16, 32, 512, 128, 3)):
        x, w = rnd(N, Cin, H, W, seed=seed), rnd(Cout, Cin, 3, 3, seed=seed + 10, lo=-0.2, hi=0.2)
        b, r = rnd(Cout, seed=seed + 20), rnd(N, Cout, H, W, seed=seed + 30)
        ref = F.leaky_relu(F.conv2d(x.to(d), w.to(d), b.to(d), padding=1), 0.2) * 0.5 + r.to(d)
        xb, rb = nhwc_buf(x, 512, 0), nhwc_buf(r)
        wp, _k = pack(ops, w.to(DEV), ops.PACK_FWD)

        def launch():
            yb = torch.full((N, H, W, 192), -3.0, device=DEV)
            ops.conv(ops.View(xb, 0, Cin), wp, ops.View(yb, 32, Cout), bias=b.to(DEV), act=ops.ACT_LRELU, slope=0.2, alpha=0.5, r1=ops.View(rb))
            return to_nchw(yb, 32, Cout)
        both("conv3x3 %d->%d" % (Cin, Cout), launch, ref)
    # nearest-x2 stager
    w2, x2 = rnd(64, 64, 3, 3, seed=5, lo=-0.2, hi=0.2), rnd(1, 64, 12, 20, seed=6)
    wp2, _k2 = pack(ops, w2.to(DEV), ops.PACK_FWD)

    def up2():
        y2 = torch.zeros(1, 24, 40, 64, device=DEV)
        ops.conv(ops.View(nhwc_buf(x2)), wp2, ops.View(y2), mode=ops.CONV_3x3_UP2)
        return to_nchw(y2, 0, 64)
    both("up2", up2, F.conv2d(F.interpolate(x2.to(d), scale_factor=2.0, mode="nearest"), w2.to(d), None, padding=1))
    # 4x4 s2 forward and its data-gradient, 3x3 data-gradient
    w4, x4, g4 = rnd(64, 32, 4, 4, seed=7, lo=-0.2, hi=0.2), rnd(2, 32, 16, 24, seed=8), rnd(2, 64, 8, 12, seed=9)
    wp4, _k4 = pack(ops, w4.to(DEV), ops.PACK_FWD_S2D)
    wd4, _k5 = pack(ops, w4.to(DEV), ops.PACK_DGRAD_S2)

    def s2():
        y4 = torch.zeros(2, 8, 12, 64, device=DEV)
        ops.conv(ops.View(nhwc_buf(x4)), wp4, ops.View(y4), mode=ops.CONV_4x4_S2)
        return to_nchw(y4, 0, 64)

    def ds2():
        gx4 = torch.zeros(2, 16, 24, 32, device=DEV)
        ops.conv(ops.View(nhwc_buf(g4)), wd4, ops.View(gx4), mode=ops.DGRAD_4x4_S2)
        return to_nchw(gx4, 0, 32)
    both("4x4s2", s2, F.conv2d(x4.to(d), w4.to(d), None, stride=2, padding=1))
    both("dgrad4x4s2", ds2, F.conv_transpose2d(g4.to(d), w4.to(d), None, stride=2, padding=1))
    w3, g3 = rnd(32, 96, 3, 3, seed=12, lo=-0.2, hi=0.2), rnd(2, 32, 20, 37, seed=10)
    wd3, _k6 = pack(ops, w3.to(DEV), ops.PACK_DGRAD_3x3)

    def d3():
        gx3 = torch.zeros(2, 20, 37, 96, device=DEV)
        ops.conv(ops.View(nhwc_buf(g3)), wd3, ops.View(gx3))
        return to_nchw(gx3, 0, 96)
    both("dgrad3x3", d3, F.conv_transpose2d(g3.to(d), w3.to(d), None, padding=1))
    # reflection borders
    wr, xr = rnd(64, 64, 3, 3, seed=14, lo=-0.2, hi=0.2), rnd(2, 64, 20, 37, seed=15)
    wpr, _k7 = pack(ops, wr.to(DEV), ops.PACK_FWD)

    def refl():
        yr = torch.zeros(2, 20, 37, 64, device=DEV)
        ops.conv(ops.View(nhwc_buf(xr)), wpr, ops.View(yr), reflect=True)
        return to_nchw(yr, 0, 64)
    both("reflect", refl, F.conv2d(F.pad(xr.to(d), (1, 1, 1, 1), mode="reflect"), wr.to(d), None))
    # small-spatial route: im2col + 1x1 GEMM with split-K
    ws_, xs = rnd(96, 256, 3, 3, seed=16, lo=-0.05, hi=0.05), rnd(2, 256, 8, 8, seed=17)
    wps, _k8 = pack(ops, ws_.to(DEV), ops.PACK_COL_FWD)

    def small():
        ys = torch.zeros(2, 8, 8, 96, device=DEV)
        ops.conv_small(ops.View(nhwc_buf(xs)), wps, ops.View(ys), 3, 1)
        return to_nchw(ys, 0, 96)
    both("im2col splitk", small, F.conv2d(xs.to(d), ws_.to(d), None, padding=1))
    # weight gradients: every workgroup tile class (32 x 32/64/96/128 pixel-split and shared-pixel, 64 x 64, strided)
    monkeypatch.setattr(ops, "WGRAD_X3", True)
    for (mode, Nn, Hh, Ww, ci, co) in (("3x3", 2, 32, 32, 32, 32), ("3x3", 1, 32, 48, 64, 32), ("3x3", 2, 16, 16, 96, 32),
                                       ("3x3", 1, 24, 32, 128, 32), ("3x3", 2, 16, 32, 192, 64), ("s2", 2, 16, 32, 64, 64)):
        xx = rnd(Nn, ci, Hh, Ww, seed=11)
        k = 3 if mode == "3x3" else 4
        ww = rnd(co, ci, k, k, seed=12).to(d).requires_grad_(True)
        yy = F.conv2d(xx.to(d), ww, None, padding=1) if mode == "3x3" else F.conv2d(xx.to(d), ww, None, stride=2, padding=1)
        gg = rnd(*yy.shape, seed=13)
        (rw,) = torch.autograd.grad(yy, ww, gg.to(d))
        xg, ggb = nhwc_buf(xx), nhwc_buf(gg)

        def wg():
            dw, db = torch.zeros(co, ci, k, k, device=DEV), torch.zeros(co, device=DEV)
            ops.wgrad(ops.View(xg), ops.View(ggb), dw, db, mode=ops.CONV_3x3 if mode == "3x3" else ops.CONV_4x4_S2, beta=0.0)
            return dw.cpu()
        both("wgrad %s %d->%d" % (mode, ci, co), wg, rw.detach())
    # the chain kernel's split-operand instantiation
    monkeypatch.setattr(ops, "CHAIN_X3", True)
    nf, gc = 64, 32
    cw = [rnd(gc, nf + k * gc, 3, 3, seed=70 + k, lo=-0.05, hi=0.05) for k in range(2)]
    p = ops.WeightPacker(DEV)
    idx = [p.add(wk.to(DEV), ops.PACK_FWD) for wk in cw]
    p.run()
    x0 = rnd(2, nf, 40, 72, seed=90)
    x1 = F.leaky_relu(F.conv2d(x0.to(d), cw[0].to(d), None, padding=1), 0.2)
    x2_ = F.leaky_relu(F.conv2d(torch.cat([x0.to(d), x1], 1), cw[1].to(d), None, padding=1), 0.2)

    def chain():
        buf = torch.zeros((2, 40, 72, nf + 2 * gc), device=DEV)
        buf[..., :nf] = x0.permute(0, 2, 3, 1).to(DEV)
        st = [dict(x=ops.View(buf, 0, nf + gc * k), wp=p.get(idx[k]), y=ops.View(buf, nf + gc * k, gc), act=ops.ACT_LRELU, slope=0.2,
                   fresh_from=(nf + gc * (k - 1) if k else None)) for k in range(2)]
        ops.conv_chain(st)
        torch.cuda.synchronize()
        return to_nchw(buf, nf, 2 * gc)
    both("chain", chain, torch.cat([x1, x2_], 1))
    assert ops.chain_error_flag() == 0
    print("\nbf16x3 vs fp32 matrix core, max error / scale against fp64:", ["%s %.1e %.1e" % w_ for w_ in worst])


@pytest.mark.parametrize("case", [(7, 1, 3, True, 3, 64, 2, 20, 28), (7, 1, 3, True, 64, 3, 1, 16, 24), (4, 1, 1, False, 32, 48, 2, 9, 13),
                                  (4, 1, 1, False, 64, 1, 1, 12, 12), (3, 2, 1, False, 8, 12, 2, 10, 14), (3, 1, 1, True, 16, 8, 1, 7, 9)])
def test_generic_conv_family(case):
    """tnr_gconv_fwd / _dgrad / _wgrad (vector ALUs; the 7x7 reflection-padded and 4x4 stride-1 layers of the image-to-image
    networks) against F.conv2d + autograd, zero and reflection padding, inside channel windows of wider buffers."""
    ops = _ops()
    k, stride, pad, reflect, Cin, Cout, N, H, W = case
    x = rnd(N, Cin, H, W, seed=401)
    w = rnd(Cout, Cin, k, k, seed=402, lo=-0.3, hi=0.3).requires_grad_(True)
    b = rnd(Cout, seed=403)
    xr = x.clone().requires_grad_(True)
    xp = F.pad(xr, (pad,) * 4, mode="reflect") if reflect else F.pad(xr, (pad,) * 4)
    ref = F.leaky_relu(F.conv2d(xp, w, b, stride=stride), 0.2)
    Ho, Wo = ref.shape[2:]
    g = rnd(N, Cout, Ho, Wo, seed=404)
    pre = F.conv2d(xp, w, b, stride=stride)
    gx_ref, gw_ref = torch.autograd.grad(pre, (xr, w), g)
    ci4, co4 = (Cin + 3) // 4 * 4, (Cout + 3) // 4 * 4
    xb = nhwc_buf(F.pad(x, (0, 0, 0, 0, 0, ci4 - Cin)), ci4 + 4, 4, fill=0.0)
    yb = torch.full((N, Ho, Wo, co4), 5.0, device=DEV)
    wd, bd = w.detach().to(DEV), b.to(DEV)
    ops.gconv_fwd(ops.View(xb, 4, ci4), wd, ops.View(yb, 0, co4), bias=bd, stride=stride, pad=pad, reflect=reflect, act=ops.ACT_LRELU, slope=0.2)
    close(to_nchw(yb, 0, Cout), ref.detach(), what="gconv fwd")
    gb = nhwc_buf(F.pad(g, (0, 0, 0, 0, 0, co4 - Cout)), fill=0.0)
    gxb = torch.full((N, H, W, ci4), 5.0, device=DEV)
    ops.gconv_dgrad(ops.View(gb), wd, ops.View(gxb), stride=stride, pad=pad, reflect=reflect)
    close(to_nchw(gxb, 0, Cin), gx_ref, what="gconv dgrad")
    dw0, db0 = rnd(Cout, Cin, k, k, seed=405), rnd(Cout, seed=406)
    dw, db = dw0.to(DEV), db0.to(DEV)
    ops.gconv_wgrad(ops.View(xb, 4, ci4), ops.View(gb), dw, db, stride=stride, pad=pad, reflect=reflect, alpha=0.5, beta=1.0)
    close(dw.cpu(), dw0 + 0.5 * gw_ref, tol=5e-5, what="gconv wgrad")
    close(db.cpu(), db0 + 0.5 * g.sum(dim=(0, 2, 3)), tol=5e-5, what="gconv bias grad")


def test_pad_tanh_ganloss_kernels():
    ops = _ops()
    x = rnd(2, 8, 9, 11, seed=411)
    xb = nhwc_buf(x)
    for refl_ in (True, False):
        yb = torch.zeros(2, 9 + 4, 11 + 4, 8, device=DEV)
        ops.pad2d(ops.View(xb), ops.View(yb), 2, refl_)
        want = F.pad(x, (2,) * 4, mode="reflect") if refl_ else F.pad(x, (2,) * 4)
        assert torch.equal(to_nchw(yb, 0, 8), want)
        back = torch.zeros(2, 9, 11, 8, device=DEV)
        ops.unpad2d(ops.View(yb), ops.View(back), 2, False)
        assert torch.equal(to_nchw(back, 0, 8), x)
    gp = rnd(2, 8, 13, 15, seed=412)
    xr = x.clone().requires_grad_(True)
    (fold_ref,) = torch.autograd.grad(F.pad(xr, (2,) * 4, mode="reflect"), xr, gp)
    fb = torch.zeros(2, 9, 11, 8, device=DEV)
    ops.unpad2d(ops.View(nhwc_buf(gp)), ops.View(fb), 2, True)
    close(to_nchw(fb, 0, 8), fold_ref, tol=1e-6, what="reflection fold")
    t = rnd(3, 5, 7, seed=413, lo=-3, hi=3).to(DEV)
    y = torch.empty_like(t)
    ops.tanh_fwd(t, y)
    close(y.cpu(), torch.tanh(t.cpu()), tol=1e-6, what="tanh")
    g = rnd(3, 5, 7, seed=414).to(DEV)
    gx = torch.empty_like(t)
    ops.tanh_bwd(g, y, gx)
    close(gx.cpu(), g.cpu() * (1 - torch.tanh(t.cpu()) ** 2), tol=1e-6, what="tanh bwd")
    p = rnd(2, 1, 30, 30, seed=415, lo=-4, hi=4)
    for kind, target in ((0, 1.0), (0, 0.0), (1, 1.0), (1, 0.0)):
        pr = p.clone().requires_grad_(True)
        tt = torch.full_like(pr, target)
        l = F.binary_cross_entropy_with_logits(pr, tt) if kind == 0 else F.mse_loss(pr, tt)
        (gr,) = torch.autograd.grad(l, pr)
        out, grad = torch.zeros(1, device=DEV), torch.zeros_like(p, device=DEV)
        ops.gan_loss(p.to(DEV), kind, target, out, grad)
        assert abs(float(out) - float(l)) < 1e-6 * max(1.0, abs(float(l)))
        close(grad.cpu(), gr, tol=1e-6, what="gan loss grad")
    out = torch.zeros(1, device=DEV)
    ops.gan_loss(p.to(DEV), 2, 0.0, out, None)            # kind 2: mean of the logits (D_real / D_fake log entries)
    assert abs(float(out) - float(p.mean())) < 1e-6


@pytest.mark.parametrize("case", [(2, 31, 31, 64, 96), (1, 32, 32, 256, 512), (3, 16, 20, 32, 4)])
def test_conv4x4_s1_through_the_patch_matrix(case):
    """The PatchGAN's 4x4 stride-1 pad-1 layers (discriminators.py:543-560) on the matrix cores: forward and data-gradient as
    tnr_im2col + the 1x1 GEMM kernel over the natural image shape (ops.conv_col; pad 2 + flipped taps for the gradient),
    tnr_window2d (zero-embedding at an offset), and the weight gradient as four shifted 3x3 windows (_K4S1.wgrad)."""
    ops = _ops()
    from trainner_amd.models.modules.architectures import block as B
    from trainner_amd.models.modules.architectures.discriminators import _K4S1
    N, H, W, Cin, Cout = case
    conv = B.Conv2dHIP(Cin, 1 if Cout == 4 else Cout, 4, 1).to(DEV)
    co_real = conv.out_channels
    w = rnd(co_real, Cin, 4, 4, seed=402, lo=-0.05, hi=0.05)
    b = rnd(co_real, seed=403)
    with torch.no_grad():
        conv.weight.copy_(w.to(DEV))
        conv.bias.copy_(b.to(DEV))
    conv.weight.grad = torch.zeros_like(conv.weight)
    conv.bias.grad = torch.zeros_like(conv.bias)
    x = rnd(N, Cin, H, W, seed=401)
    g = rnd(N, co_real, H - 1, W - 1, seed=404)
    xr = x.clone().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, br, padding=1)
    ref.backward(g)
    packer = ops.WeightPacker(torch.device(DEV))
    op = _K4S1(conv, packer, Cin)
    op.refresh()
    packer.run()
    xb = ops.View(nhwc_buf(x))
    y = torch.zeros((N, H - 1, W - 1, Cout), device=DEV)
    op.fwd(xb, ops.View(y))
    close(to_nchw(y, 0, co_real), ref.detach(), what="k4s1 fwd")
    gb = ops.View(nhwc_buf(g, ctot=Cout, fill=0.0))
    gx = torch.zeros((N, H, W, Cin), device=DEV)
    op.dgrad(gb, ops.View(gx))
    close(to_nchw(gx, 0, Cin), xr.grad, what="k4s1 dgrad")
    op.wgrad(xb, gb)
    close(conv.weight.grad.cpu(), wr.grad, tol=5e-5, what="k4s1 wgrad")
    close(conv.bias.grad.cpu(), br.grad, tol=5e-5, what="k4s1 bias grad")
    # window2d on its own: offset crop and offset embed
    src = rnd(2, 8, 5, 7, seed=405)
    dst = torch.full((2, 6, 9, 8), 3.0, device=DEV)
    ops.window2d(ops.View(nhwc_buf(src)), ops.View(dst), -1, 2)
    exp = torch.zeros(2, 8, 6, 9)
    exp[:, :, 1:6, 0:5] = src[:, :, 0:5, 2:7]
    assert torch.equal(to_nchw(dst, 0, 8), exp)


@pytest.mark.parametrize("shape", [(3, 17, 23, 64), (2, 64, 64, 256), (16, 8, 8, 16)])
def test_instance_norm_grouped_launch(shape):
    """tnr_instnorm_fwd / bwd (InstanceNorm2d without affine, ResNet_arch.py:40-50) over a batch in one set of launches,
    with the fused ReLU, against F.instance_norm under autograd."""
    ops = _ops()
    N, H, W, C = shape
    x = rnd(N, C, H, W, seed=501) * 2.0 + rnd(1, C, 1, 1, seed=502)
    g = rnd(N, C, H, W, seed=503)
    for relu in (True, False):
        xr = x.clone().requires_grad_(True)
        ref = F.instance_norm(xr, eps=1e-5)
        ref = F.relu(ref) if relu else ref
        ref.backward(g)
        zb = ops.View(nhwc_buf(x))
        y = torch.zeros((N, H, W, C), device=DEV)
        mean, inv = torch.zeros(N * C, device=DEV), torch.zeros(N * C, device=DEV)
        ops.instnorm_fwd(zb, ops.View(y), mean, inv, act=ops.ACT_RELU if relu else ops.ACT_NONE, slope=0.0)
        close(to_nchw(y, 0, C), ref.detach(), tol=1e-5, what="instnorm fwd")
        close(mean.view(N, C).cpu(), x.mean(dim=(2, 3)), tol=1e-6, what="instnorm mean")
        gz = torch.zeros((N, H, W, C), device=DEV)
        ops.instnorm_bwd(ops.View(nhwc_buf(g)), ops.View(y), zb, ops.View(gz), mean, inv, mslope=0.0 if relu else 1.0)
        close(to_nchw(gz, 0, C), xr.grad, tol=2e-5, what="instnorm bwd")


@pytest.mark.parametrize("shape", [(2, 33, 17, 64), (16, 64, 64, 128), (1, 5, 7, 12)])
def test_bias_grad_column_sum(shape):
    """tnr_bias_grad: db = beta db + alpha sum over pixels (two-stage, fixed order), on a channel window of a wider buffer."""
    ops = _ops()
    N, H, W, C = shape
    g = rnd(N, C, H, W, seed=601)
    buf = nhwc_buf(g, ctot=C + 8, coff=4)
    db = torch.full((C,), 2.0, device=DEV)
    ops.bias_grad(ops.View(buf, 4, C), db, alpha=0.5, beta=1.0)
    close(db.cpu(), 2.0 + 0.5 * g.double().sum(dim=(0, 2, 3)).float(), tol=2e-6, what="bias_grad")


@pytest.mark.parametrize("case", [(2, 20, 37, 64, 64), (1, 64, 64, 256, 256), (2, 9, 33, 32, 96)])
def test_conv3x3_reflection_borders(case):
    """tnr_conv_desc.pad_mode / tnr_wgrad_desc.pad_mode = 1: ReflectionPad2d(1) + conv3x3 (the ResnetGenerator's residual blocks,
    ResNet_arch.py:118-146) with the reflection done by the stagers of the forward and weight-gradient MFMA kernels."""
    ops = _ops()
    N, H, W, Cin, Cout = case
    x = rnd(N, Cin, H, W, seed=701)
    w = rnd(Cout, Cin, 3, 3, seed=702, lo=-0.05, hi=0.05)
    b = rnd(Cout, seed=703)
    g = rnd(N, Cout, H, W, seed=704)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), wr, br)
    ref.backward(g)
    wp, _ = pack(ops, w.to(DEV), ops.PACK_FWD)
    xb = ops.View(nhwc_buf(x))
    y = torch.zeros((N, H, W, Cout), device=DEV)
    ops.conv(xb, wp, ops.View(y), bias=b.to(DEV), reflect=True)
    close(to_nchw(y, 0, Cout), ref.detach(), what="reflect conv")
    dw, db = torch.zeros((Cout, Cin, 3, 3), device=DEV), torch.zeros(Cout, device=DEV)
    ops.wgrad(xb, ops.View(nhwc_buf(g)), dw, db, beta=0.0, reflect=True)
    close(dw.cpu(), wr.grad, tol=5e-5, what="reflect wgrad")
    close(db.cpu(), br.grad, tol=5e-5, what="reflect bias grad")
