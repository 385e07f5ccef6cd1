"""Network-level parity (-m gpu): each HIP-engine network (one autograd node) against the CPU
oracle's functional restatement driven by torch autograd, on identical seeded weights/inputs:
outputs, input gradients, every parameter gradient, BatchNorm running statistics.

Gradients that pass through ReLU / LeakyReLU / max-pool gates are arbitrated by fp64 (oracle/gated.py): the
oracle is re-evaluated in double precision with every gate pinned to the branch the HIP engine actually took
(read from the engine's own saved activations).  With the gates pinned the function is smooth, so EVERY gradient
element must agree to fp32 round-off (max norm, 2e-4) -- this replaces the former "relative L2 < 0.05 / 0.1"
bounds, which could have hidden a real kernel bug behind the gate-flip argument.  The ungated comparison against
the free-running fp64 oracle is kept as a secondary statistic with an absolute bound (`dist64`; a flip is a
discrete event, so the CPU fp32 oracle may sit at 2e-6 from fp64 while an equally valid run sits at 4e-4 --
measured on MI355X for features.17.weight -- which is why no ratio between the two is asserted).
"""
import pytest
import torch
import torch.nn.functional as F

from oracle import detrand, gated, sr_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel_err(got, ref):
    return (got.detach().cpu() - ref.detach()).abs().max().item() / (ref.detach().abs().max().item() + 1e-12)


def robust_err(got, ref):
    """(relative L2 error, median |diff| / max |ref|).  Used for gradients that pass through ReLU /
    max-pool gates deep in VGG: an activation within round-off of zero may gate differently in two
    correct implementations, which changes a whole receptive field of the input gradient by O(1)
    while leaving almost every element untouched -- so bound the L2 norm and the median, not the max."""
    d = (got.detach().cpu() - ref.detach()).double()
    r = ref.detach().double()
    return (d.norm() / (r.norm() + 1e-30)).item(), (d.abs().median() / (r.abs().max() + 1e-30)).item()


def to64(sd):
    return {k: (v.detach().double() if v.is_floating_point() else v.detach().clone()) for k, v in sd.items()}


def dist64(got, ref32, ref64):
    """relative L2 distances to the free-running fp64 evaluation: (engine, fp32 CPU oracle)."""
    r = ref64.detach().double()
    n = r.norm().item() + 1e-300
    return (got.detach().cpu().double() - r).norm().item() / n, (ref32.detach().double() - r).norm().item() / n


def seeded(net, seed, **kw):
    sd = net.state_dict()
    detrand.fill_state_dict_(sd, seed, **kw)
    return {k: v.clone() for k, v in sd.items()}


def oracle_params(sd):
    out = {}
    for k, v in sd.items():
        t = v.clone()
        if t.is_floating_point() and not k.startswith("running") and ".running_" not in k:
            t.requires_grad_(True)
        out[k] = t
    return out


@pytest.mark.parametrize("mode", ["upconv", "pixelshuffle"])
def test_rrdbnet_forward_backward(mode):
    from trainner_amd.models.modules.architectures.RRDBNet_arch import RRDBNet
    net = RRDBNet(3, 3, 64, 2, upsample_mode=mode)
    sd = seeded(net, 5)
    net = net.to(DEV)
    lr = detrand.uniform((2, 3, 24, 20), 3, 0.0, 1.0)
    gout = detrand.uniform((2, 3, 96, 80), 4, -1.0, 1.0)
    out = net(lr.to(DEV))
    out.backward(gout.to(DEV))
    osd = oracle_params(sd)
    ref = O.rrdbnet_forward(lr, osd, 2, 4, mode)
    ref.backward(gout)
    assert rel_err(out, ref) < 2e-5
    for k, p in net.named_parameters():
        assert rel_err(p.grad, osd[k].grad) < 2e-4, k
    # no-grad inference path (buffer ring) gives the same image
    with torch.no_grad():
        out2 = net(lr.to(DEV))
    assert rel_err(out2, ref) < 2e-5
    # gradients accumulate across backward calls (virtual batch semantics)
    net(lr.to(DEV)).backward(gout.to(DEV))
    for k, p in net.named_parameters():
        assert rel_err(p.grad, 2 * osd[k].grad) < 2e-4, k


def _engine_noise_fields(N, h, w, seed, call, nblocks, sigma=0.1, nf=64):
    """The multiplier fields m = 1 + sigma n the engine draws in training forward `call` (tnr_gauss_mult), as NCHW CPU tensors."""
    from trainner_amd import ops
    out = []
    for i in range(nblocks):
        buf = torch.empty((N, h, w, nf), device=DEV)
        ops.gauss_mult(ops.View(buf), None, ops.Noise(sigma, ops.noise_key(seed, call, i)))
        out.append(buf.permute(0, 3, 1, 2).contiguous().cpu())
    return out


def test_rrdbnet_gaussian_noise():
    """gaussian_noise=True (ESRGAN+, the reference's default: defaults.py:59; ResidualDenseBlock_5C.forward RRDBNet_arch.py:160-163,
    block.py:587-600): forward and every parameter gradient against the oracle evaluated ON THE ENGINE'S OWN DRAW (the fields are
    read back with tnr_gauss_mult), i.e. the multiplier sits after `x5*0.2 + x`, before the RRDB residual, and the gradient flows
    through both terms.  Then: the same (seed, forward count) reproduces the step bit for bit, the next forward draws a different
    field, eval() and sigma = 0 are bit-identical to a network built with gaussian_noise=False, and a no-grad training-mode
    forward (buffer ring) draws the same field as the graph-building one."""
    from trainner_amd.models.modules.architectures.RRDBNet_arch import RRDBNet
    nb = 2
    net = RRDBNet(3, 3, 64, nb, gaussian_noise=True)
    plain = RRDBNet(3, 3, 64, nb, gaussian_noise=False)
    sd = seeded(net, 5)
    plain.load_state_dict(sd)
    net, plain = net.to(DEV), plain.to(DEV)
    assert net.noise_sigma == 0.1 and plain.noise_sigma == 0.0 and net.training
    net.noise_seed = 777
    lr = detrand.uniform((2, 3, 24, 20), 3, 0.0, 1.0)
    gout = detrand.uniform((2, 3, 96, 80), 4, -1.0, 1.0)
    taken = []
    inner = net.engine_forward

    def tapped(x, save):
        o, saved = inner(x, save)
        taken.append(saved)
        return o, saved

    net.engine_forward = tapped
    out = net(lr.to(DEV))
    out.backward(gout.to(DEV))
    net.engine_forward = inner
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    ms = _engine_noise_fields(2, 24, 20, 777, 0, 3 * nb)
    # fp64 with the engine's own LeakyReLU branches AND the engine's own draw: smooth, so every gradient element is held to round-off
    gates = gated.gates_of_rrdbnet(taken[0])
    gates["noise"] = [m.double() for m in ms]
    p64 = {k: v.detach().double().requires_grad_(True) for k, v in sd.items()}
    ref = gated.rrdbnet_forward_gated(lr.double(), p64, nb, gates)
    ref.backward(gout.double())
    assert rel_err(out, ref) < 2e-5
    for k, p in net.named_parameters():
        assert rel_err(p.grad, p64[k].grad) < 2e-4, k
    # and the free-running fp32 oracle on the same draw (forward only: its LeakyReLU inputs within round-off of zero gate either way)
    assert rel_err(out, O.rrdbnet_forward(lr, sd, nb, 4, "upconv", noise=ms)) < 2e-5
    # the noise matters at this tolerance: without it the oracle is far away
    assert rel_err(out, O.rrdbnet_forward(lr, sd, nb, 4, "upconv")) > 1e-3
    # same seed, same forward count: bit-identical forward and backward
    net._noise_calls = 0
    net.flat_params().grad.zero_()            # (every p.grad is a view of the flat gradient buffer)
    out_b = net(lr.to(DEV))
    out_b.backward(gout.to(DEV))
    assert torch.equal(out_b, out)
    for k, p in net.named_parameters():
        assert torch.equal(p.grad, grads[k]), k
    # a no-grad training-mode forward of the same count draws the same field; the next count a different one
    net._noise_calls = 0
    with torch.no_grad():
        assert torch.equal(net(lr.to(DEV)), out)
        assert net._noise_calls == 1 and not torch.equal(net(lr.to(DEV)), out)
    # eval() is the identity (block.py:595: `if self.training and self.sigma != 0`), and so is sigma = 0: bit-identical to gaussian_noise=False
    base = plain(lr.to(DEV))
    net.eval()
    with torch.no_grad():
        assert torch.equal(net(lr.to(DEV)), base)
    net.train()
    net.noise_sigma = 0.0
    assert torch.equal(net(lr.to(DEV)), base)


def test_srresnet_forward_backward():
    from trainner_amd.models.modules.architectures.SRResNet_arch import SRResNet
    net = SRResNet(3, 3, 64, 3)
    sd = seeded(net, 6)
    net = net.to(DEV)
    lr = detrand.uniform((2, 3, 16, 24), 7, 0.0, 1.0)
    gout = detrand.uniform((2, 3, 64, 96), 8, -1.0, 1.0)
    out = net(lr.to(DEV))
    out.backward(gout.to(DEV))
    osd = oracle_params(sd)
    ref = O.srresnet_forward(lr, osd, 3, 4)
    ref.backward(gout)
    assert rel_err(out, ref) < 2e-5
    for k, p in net.named_parameters():
        assert rel_err(p.grad, osd[k].grad) < 2e-4, k


@pytest.mark.parametrize("size,nf", [(64, 16), (128, 64)])
def test_discriminator_vgg(size, nf):
    from trainner_amd.models.modules.architectures.discriminators import Discriminator_VGG
    net = Discriminator_VGG(size, 3, nf)
    sd = seeded(net, 9)
    net = net.to(DEV).train()
    x = detrand.uniform((3, 3, size, size), 10, 0.0, 1.0)
    gout = detrand.uniform((3, 1), 11, -1.0, 1.0)
    xd = x.clone().to(DEV).requires_grad_(True)
    out = net(xd)
    gates, hid_gate = gated.gates_of_discriminator(out.grad_fn.saved)      # the branches the engine took
    out.backward(gout.to(DEV))
    osd = oracle_params(sd)
    xr = x.detach().clone().requires_grad_(True)
    ref = O.disc_vgg_forward(xr, osd, size, nf, training=True)
    ref.backward(gout)
    assert rel_err(out, ref) < 5e-5
    # fp64 arbitration (see three_way): LeakyReLU gates within round-off of zero and batch-3 BatchNorm statistics make
    # two fp32 evaluations differ by far more than 1 ulp; neither may be further from fp64 than the other by > 4x
    osd64 = oracle_params(to64(sd))
    x64 = x.detach().double().requires_grad_(True)
    ref64 = O.disc_vgg_forward(x64, osd64, size, nf, training=True)
    ref64.backward(gout.double())
    e_hip, e_cpu = dist64(xd.grad, xr.grad, x64.grad)
    assert e_hip < 2e-2, ("input grad", e_hip, e_cpu)   # (MI355X: 5.9e-3; the CPU fp32 oracle itself: 7.3e-3)
    from oracle.fixtures import bn_shadowed_biases
    shadow = bn_shadowed_biases([(k, None) for k in sd])
    for k, p in net.named_parameters():
        if k in shadow:
            continue                                   # true gradient is exactly zero: noise on both sides
        e_hip, e_cpu = dist64(p.grad, osd[k].grad, osd64[k].grad)
        assert e_hip < 2e-2, (k, e_hip, e_cpu)          # ungated: flipped gates (MI355X 5.5e-3, CPU fp32 oracle 6.8e-3; was: L2 < 0.05)
    # the strong check: fp64 with the engine's own gates -> every element of every gradient, max norm
    gsd64 = oracle_params(to64(sd))
    xg64 = x.detach().double().requires_grad_(True)
    out64 = gated.disc_vgg_forward_gated(xg64, gsd64, size, nf, gates, hid_gate)
    out64.backward(gout.double())
    assert rel_err(out, out64) < 2e-5
    assert rel_err(xd.grad, xg64.grad) < 2e-4, rel_err(xd.grad, xg64.grad)
    for k, p in net.named_parameters():
        if k not in shadow:
            assert rel_err(p.grad, gsd64[k].grad) < 2e-4, (k, rel_err(p.grad, gsd64[k].grad))
    new = net.state_dict()
    for k in sd:
        if "running_" in k:
            assert rel_err(new[k], osd[k]) < 1e-4, k
        if k.endswith("num_batches_tracked"):
            assert int(new[k]) == 1
    # G-step usage: parameters frozen -> data gradient only, statistics still updated
    for p in net.parameters():
        p.requires_grad_(False)
    before = [p.grad.clone() for p in net.parameters()]
    xd2 = x.detach().clone().to(DEV).requires_grad_(True)
    net(xd2).backward(gout.to(DEV))
    for b, p in zip(before, net.parameters()):
        assert torch.equal(b, p.grad)
    assert int(net.state_dict()["features.3.num_batches_tracked"]) == 2


@pytest.mark.parametrize("nf,skip", [(16, True), (64, True), (16, False)])
def test_unet_discriminator(nf, skip):
    """UNetDiscriminator (Real-ESRGAN's D, discriminators.py:686-779): per-pixel logits, input gradient and every
    parameter gradient against the oracle; LeakyReLU gates arbitrated by the gate-pinned fp64 run (oracle/gated.py)."""
    from trainner_amd.models.modules.architectures.discriminators import UNetDiscriminator
    net = UNetDiscriminator(3, nf, skip_connection=skip)
    sd = seeded(net, 21)
    assert list(sd) == ["conv0.weight", "conv0.bias"] + ["conv%d.weight" % i for i in range(1, 10)] + ["conv9.bias"]
    net = net.to(DEV).train()
    x = detrand.uniform((2, 3, 48, 64), 22, 0.0, 1.0)
    gout = detrand.uniform((2, 1, 48, 64), 23, -1.0, 1.0)
    xd = x.clone().to(DEV).requires_grad_(True)
    out = net(xd)
    gates = gated.gates_of_unet(out.grad_fn.saved)
    out.backward(gout.to(DEV))
    osd = oracle_params(sd)
    xr = x.detach().clone().requires_grad_(True)
    ref = O.unet_disc_forward(xr, osd, skip)
    ref.backward(gout)
    assert out.shape == ref.shape == (2, 1, 48, 64)
    assert rel_err(out, ref) < 5e-5
    e_hip, e_cpu = dist64(xd.grad, xr.grad, xr.grad)         # ungated fp32 vs fp32: bounded loosely (flips), see below
    assert e_hip < 2e-2, e_hip
    gsd64 = oracle_params(to64(sd))
    xg64 = x.detach().double().requires_grad_(True)
    out64 = gated.unet_disc_forward_gated(xg64, gsd64, gates, skip)
    out64.backward(gout.double())
    assert rel_err(out, out64) < 2e-5
    assert rel_err(xd.grad, xg64.grad) < 2e-4, rel_err(xd.grad, xg64.grad)
    for k, p in net.named_parameters():
        assert rel_err(p.grad, gsd64[k].grad) < 2e-4, (k, rel_err(p.grad, gsd64[k].grad))
    # gradients accumulate; frozen parameters (G step) give the data gradient only
    for p in net.parameters():
        p.requires_grad_(False)
    before = [p.grad.clone() for p in net.parameters()]
    xd2 = x.detach().clone().to(DEV).requires_grad_(True)
    net(xd2).backward(gout.to(DEV))
    for b, p in zip(before, net.parameters()):
        assert torch.equal(b, p.grad)
    assert rel_err(xd2.grad, xg64.grad) < 2e-4


@pytest.mark.parametrize("norm,nb", [("instance", 2), ("batch", 1)])
def test_resnet_generator(norm, nb):
    """ResnetGenerator (CycleGAN / Pix2Pix G, ResNet_arch.py:11-149): output, input gradient and every parameter gradient
    against the oracle (reflection padding, stride-2 and transposed convolutions, Instance / BatchNorm, tanh)."""
    from trainner_amd.models.modules.architectures.ResNet_arch import ResnetGenerator
    net = ResnetGenerator(3, 3, ngf=16, norm_type=norm, n_blocks=nb)
    sd = seeded(net, 31, gain=0.5)
    net = net.to(DEV).train()
    x = detrand.uniform((2, 3, 32, 40), 32, -1.0, 1.0)
    gout = detrand.uniform((2, 3, 32, 40), 33, -1.0, 1.0)
    xd = x.clone().to(DEV).requires_grad_(True)
    out = net(xd)
    out.backward(gout.to(DEV))
    osd = oracle_params(sd)
    xr = x.detach().clone().requires_grad_(True)
    ref = O.resnet_generator_forward(xr, osd, nb, norm)
    ref.backward(gout)
    assert out.shape == ref.shape and rel_err(out, ref) < 5e-5
    # ReLU gates: bounded like the other ungated comparisons (robust_err), tight on the median
    l2, med = robust_err(xd.grad, xr.grad)
    assert l2 < 2e-2 and med < 2e-5, (l2, med)
    last_bias = "model.%d.bias" % (17 + nb)
    for k, p in net.named_parameters():
        if k.endswith(".bias") and k != last_bias and norm == "instance":
            continue          # a conv bias in front of a normalisation has an exactly-zero true gradient: noise on both sides
        l2, med = robust_err(p.grad, osd[k].grad)
        assert l2 < 2e-2 and med < 5e-5, (k, l2, med)
    if norm == "batch":
        new = net.state_dict()
        for k in sd:
            if "running_" in k:
                assert rel_err(new[k], osd[k]) < 1e-4, k


@pytest.mark.parametrize("norm,downs,size", [("batch", 5, (32, 64)), ("instance", 6, (64, 64)), ("batch", 7, (128, 128))])
def test_unet_generator(norm, downs, size):
    """UnetGenerator (the shipped Pix2Pix recipe's generator, UNet_arch.py:11-162): output, INPUT gradient (CycleGAN chains two
    generators) and every parameter gradient against the oracle -- 4x4 stride-2 convolutions down to 1 x 1 (or 1 x 2), transposed
    convolutions as the data-gradient kernel, in-place activation semantics of the skip connections, Batch / InstanceNorm, tanh."""
    from trainner_amd.models.modules.architectures.UNet_arch import UnetGenerator
    net = UnetGenerator(3, 3, downs, ngf=16, norm_type=norm)
    sd = seeded(net, 51, gain=0.5)
    net = net.to(DEV).train()
    x = detrand.uniform((2, 3) + size, 52, -1.0, 1.0)
    gout = detrand.uniform((2, 3) + size, 53, -1.0, 1.0)
    xd = x.clone().to(DEV).requires_grad_(True)
    out = net(xd)
    out.backward(gout.to(DEV))
    osd = oracle_params(sd)
    xr = x.detach().clone().requires_grad_(True)
    ref = O.unet_generator_forward(xr, osd, downs, norm)
    ref.backward(gout)
    assert out.shape == ref.shape and rel_err(out, ref) < 5e-5
    l2, med = robust_err(xd.grad, xr.grad)          # LeakyReLU / ReLU gates: bounded like the other ungated comparisons
    assert l2 < 2e-2 and med < 2e-5, (l2, med)
    last_bias = "model.model.3.bias"
    for k, p in net.named_parameters():
        if k.endswith(".bias") and k != last_bias and norm == "instance" and "model.model.0." not in k:
            # a conv bias in front of an InstanceNorm has an exactly-zero true gradient (the outermost down-convolution and the
            # last transposed convolution have no norm behind them)
            inner_down = k.count("model.") == downs + 1 and k.endswith(".1.bias")       # innermost down-convolution: no norm either
            if not inner_down:
                continue
        l2, med = robust_err(p.grad, osd[k].grad)
        assert l2 < 2e-2 and med < 5e-5, (k, l2, med)
    if norm == "batch":
        new = net.state_dict()
        for k in sd:
            if "running_" in k:
                assert rel_err(new[k], osd[k]) < 1e-4, k


@pytest.mark.parametrize("in_nc,size", [(6, 64), (3, 48)])
def test_patchgan_discriminator(in_nc, size):
    """NLayerDiscriminator (PatchGAN, discriminators.py:472-579) with 6 input channels (conditional pix2pix pair) and 3."""
    from trainner_amd.models.modules.architectures.discriminators import NLayerDiscriminator
    net = NLayerDiscriminator(in_nc, ndf=16, n_layers=3)
    sd = seeded(net, 41, gain=0.5)
    net = net.to(DEV).train()
    x = detrand.uniform((2, in_nc, size, size), 42, -1.0, 1.0)
    xd = x.clone().to(DEV).requires_grad_(True)
    out = net(xd)
    osd = oracle_params(sd)
    xr = x.detach().clone().requires_grad_(True)
    ref = O.patchgan_forward(xr, osd, 3)
    assert out.shape == ref.shape == (2, 1, size // 8 - 2, size // 8 - 2)
    gout = detrand.uniform(tuple(ref.shape), 43, -1.0, 1.0)
    out.backward(gout.to(DEV))
    ref.backward(gout)
    assert rel_err(out, ref) < 5e-5
    l2, med = robust_err(xd.grad, xr.grad)
    assert l2 < 2e-2 and med < 2e-5, (l2, med)
    for k, p in net.named_parameters():
        l2, med = robust_err(p.grad, osd[k].grad)
        assert l2 < 2e-2 and med < 1e-4, (k, l2, med)


def test_vgg19_features():
    from trainner_amd.models.modules.architectures.perceptual import FeatureExtractor
    from oracle import fixtures as FX
    net = FeatureExtractor(["conv5_4"], allow_random_init=True)
    vsd = FX.vgg_state(77)
    net.load_state_dict({**{k: v for k, v in net.state_dict().items() if k in ("mean", "std")}, **vsd})
    net = net.to(DEV)
    x = detrand.uniform((2, 3, 64, 96), 12, 0.0, 1.0)
    xd = x.clone().to(DEV).requires_grad_(True)
    feat = net(xd)["conv5_4"]
    relu_gates, pool_idx = gated.gates_of_vgg(feat.grad_fn.saved)
    xr = x.detach().clone().requires_grad_(True)
    ref = O.vgg19_conv54(xr, vsd)
    assert feat.shape == ref.shape
    assert rel_err(feat, ref) < 5e-5
    g = detrand.uniform(tuple(ref.shape), 13, -1.0, 1.0)
    feat.backward(g.to(DEV).contiguous(memory_format=torch.channels_last))
    ref.backward(g)
    vsd64 = to64(vsd)
    x64 = x.detach().double().requires_grad_(True)
    O.vgg19_conv54(x64, vsd64).backward(g.double())
    e_hip, e_cpu = dist64(xd.grad, xr.grad, x64.grad)     # ungated: a few flipped ReLU / max-pool gates (was: L2 < 0.1)
    assert e_hip < 2e-2, (e_hip, e_cpu)
    l2, med = robust_err(xd.grad, xr.grad)
    assert med < 2e-5, (l2, med)
    # the strong check: fp64 with the engine's own ReLU gates and max-pool winners, max norm over every pixel
    xg64 = x.detach().double().requires_grad_(True)
    f64 = gated.vgg19_conv54_gated(xg64, vsd64, relu_gates, pool_idx)
    f64.backward(g.double())
    assert rel_err(feat, f64) < 2e-5
    assert rel_err(xd.grad, xg64.grad) < 2e-4, rel_err(xd.grad, xg64.grad)
    # the perceptual criterion end to end
    from trainner_amd.models.losses import L1Loss
    y = detrand.uniform((2, 3, 64, 96), 14, 0.0, 1.0)
    xd2 = x.detach().clone().to(DEV).requires_grad_(True)
    with torch.no_grad():
        fy = net(y.to(DEV))["conv5_4"]
    loss = L1Loss()(net(xd2)["conv5_4"], fy)
    loss.backward()
    xr2 = x.detach().clone().requires_grad_(True)
    lref = F.l1_loss(O.vgg19_conv54(xr2, vsd), O.vgg19_conv54(y, vsd))
    lref.backward()
    assert abs(float(loss) - float(lref)) < 2e-5 * max(1.0, abs(float(lref)))
    # d|f - fy|/df = sign(f - fy): a feature element whose difference is at rounding level takes the other sign
    # under a different (equally valid) fp32 summation order -- the split-K GEMM of the 4x6 / 8x12 layers here --
    # and conv5_4's receptive field spans this whole image, so a handful of flips moves every input-gradient
    # element a little: bound the median at 5e-4 (measured 8e-5 with flips, 2e-7 without) and the L2 as before
    x64 = x.detach().double().requires_grad_(True)
    F.l1_loss(O.vgg19_conv54(x64, vsd64), O.vgg19_conv54(y.double(), vsd64)).backward()
    e_hip, e_cpu = dist64(xd2.grad, xr2.grad, x64.grad)
    assert e_hip < 5e-2, (e_hip, e_cpu)               # + sign(f - fy) flips of the L1 criterion (see below)
    l2, med = robust_err(xd2.grad, xr2.grad)
    assert med < 5e-4, (l2, med)
