"""Step-control options of `optimize_parameters` against the LIVE reference (-m "not gpu", build container only; skipped where
/root/reference is absent).  The goldens pin the plain recipe (one optimizer step per call, D every step); these pin the options
around it that train.py users set in the same files (sr_model.py:195-267, base_model.py:246-300,805-850):

  * `virtual_batch_size` > `batch_size`: gradients accumulate over `accumulations` calls, losses are divided by it, both optimizers
    step on the last call only;
  * `D_update_ratio`, `D_init_iters` (the WGAN-style gates): the GENERATOR trains every n-th step only, and not during the
    discriminator's initial iterations; the discriminator trains on every step;
  * learning-rate policies as train.py drives them (`update_learning_rate(step, warmup_iter)` after every step): MultiStepLR with
    the shipped recipe's relative milestones + warm-up, and the i2i recipes' Linear policy.

The engine side runs over tests/emul_backend.py (the C ABI's contract in torch-CPU); what is under test is host logic."""
import pytest
import torch

import emul_backend
import test_gpu_i2i as TI
import test_gpu_step as TS
from oracle import detrand, fixtures as FX, ref_harness as R

pytestmark = pytest.mark.skipif(not R.reference_available(), reason="needs the reference checkout (build container)")
KW = dict(nb=1, batch=2, crop=64, d_nf=16)
LR_ = 1e-4


@pytest.fixture(autouse=True)
def emulated(monkeypatch):
    emul_backend.install(monkeypatch)
    monkeypatch.setattr(TS, "DEV", "cpu")
    monkeypatch.setattr(TI, "DEV", "cpu")
    torch.set_num_threads(8)


def _edit(yml, repl):
    txt = open(yml).read()
    for a, b in repl:
        assert a in txt, a
        txt = txt.replace(a, b, 1)
    with open(yml, "w") as fh:
        fh.write(txt)
    return yml


def _both(tmp_path, repl, **kw):
    """The same configuration on both sides, the same seeded initial state."""
    from trainner_amd.models import create_model
    from trainner_amd.options import options
    k = dict(KW, **kw)
    ryml = _edit(R.esrgan_yaml(name="opts", out_root=str(tmp_path / "ref"), **k), repl)
    eyml = _edit(R.esrgan_yaml(name="opts", out_root=str(tmp_path / "eng"), gpu_ids="[0]", **k), repl)
    ropt, ref = R.build_reference_model(ryml, seed=0)
    eopt = options.parse(eyml, is_train=True)
    eng = create_model(eopt, verbose=False)
    g = detrand.fill_state_dict_({n: v.detach().cpu().clone() for n, v in eng.netG.state_dict().items()}, 101)
    d = detrand.fill_state_dict_({n: v.detach().cpu().clone() for n, v in eng.netD.state_dict().items()}, 202)
    f = FX.vgg_state(77)
    TS.load_initial(eng, g, d, f)
    ref.netG.load_state_dict(g)
    ref.netD.load_state_dict(d)
    nf = R.reference_netF(ref)
    sd = nf.state_dict()
    sd.update(f)
    nf.load_state_dict(sd)
    return (ropt, ref), (eopt, eng)


def _step(ref, eng, s, seed=6000):
    LR, HR = detrand.synthetic_pair(KW["batch"], KW["crop"], seed + s)
    log_ref = R.reference_step(ref, LR, HR, s)
    eng.feed_data({"LR": LR, "HR": HR})
    eng.optimize_parameters(s)
    return dict(eng.get_current_log()), log_ref


def _logs_close(a, b, tol=2e-4):
    assert list(a) == list(b), (list(a), list(b))
    for k, v in b.items():
        t = max(tol, 2e-3) if k.startswith("D_") else tol       # raw mean logits behind batch-2 BatchNorms (cf. test_gpu_i2i.check_logs)
        assert abs(a[k] - v) <= t * max(1.0, abs(v)), (k, a[k], v)


def _weights(eng_net, ref_net):
    """(mean, worst) |difference| over the trainable tensors in units of lr (BatchNorm-shadowed conv biases excluded)."""
    sd_r = ref_net.state_dict()
    skip = FX.bn_shadowed_biases([(k, tuple(v.shape)) for k, v in sd_r.items()])
    tot = n = worst = 0
    for k, v in sd_r.items():
        if v.dtype.is_floating_point and k not in skip and "running" not in k:
            d = (eng_net.state_dict()[k].detach().cpu() - v).abs()
            tot, n, worst = tot + d.sum().item(), n + d.numel(), max(worst, d.max().item())
    return tot / n / LR_, worst / LR_


def _moved(net, before):
    return any(not torch.equal(v.detach().cpu(), before[k]) for k, v in net.state_dict().items() if k.endswith("weight"))


def _snapshot(net):
    return {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}


def test_virtual_batch_accumulates_like_the_reference(tmp_path):
    (ropt, ref), (eopt, eng) = _both(tmp_path, [("virtual_batch_size: 2", "virtual_batch_size: 6")])
    assert eng.accumulations == 3 == ref.accumulations
    g0, d0 = _snapshot(eng.netG), _snapshot(eng.netD)
    for s in range(1, 7):
        le, lr = _step(ref, eng, s)
        _logs_close(le, lr)
        if s in (1, 2):                 # gradients only accumulate: nothing has been applied yet, on either side
            assert not _moved(eng.netG, g0) and not _moved(eng.netD, d0)
            assert not _moved(ref.netG, g0) and not _moved(ref.netD, d0)
        if s == 3:
            assert _moved(eng.netG, g0) and _moved(eng.netD, d0)
    for a, b in ((eng.netG, ref.netG), (eng.netD, ref.netD)):
        mean, worst = _weights(a, b)
        # two optimizer steps in: round-off through Adam on average; an element whose gradient is rounding noise may have moved
        # +-lr the other way in each step (the goldens' bound: 2.05 lr per step)
        assert mean <= 0.02 and worst <= 2.05 * 2, (mean, worst)


def test_d_update_ratio_and_init_iters_like_the_reference(tmp_path):
    (ropt, ref), (eopt, eng) = _both(tmp_path, [("  gan_weight: 5e-3", "  gan_weight: 5e-3\n  D_update_ratio: 2\n  D_init_iters: 1")])
    trained_g = []
    for s in range(1, 6):
        dg_e, dg_r, dd_e, dd_r = _snapshot(eng.netG), _snapshot(ref.netG), _snapshot(eng.netD), _snapshot(ref.netD)
        le, lr = _step(ref, eng, s)
        _logs_close(le, lr, 2e-4 if s == 1 else 3e-3)           # (later steps: trajectory drift of two fp32 implementations, as in the goldens)
        me, mr = _moved(eng.netG, dg_e), _moved(ref.netG, dg_r)
        assert me == mr and _moved(eng.netD, dd_e) and _moved(ref.netD, dd_r), (s, me, mr)
        trained_g.append(me)
    # the WGAN-style gates act on the GENERATOR (sr_model.py:246-247: eff_step % D_update_ratio == 0 and eff_step > D_init_iters);
    # the discriminator trains on every step
    assert trained_g == [False, True, False, True, False]
    for a, b in ((eng.netG, ref.netG), (eng.netD, ref.netD)):
        mean, worst = _weights(a, b)
        assert mean <= 0.03 and worst <= 2.05 * 5, (mean, worst)


@pytest.mark.parametrize("mix", ["pixel_feature_no_gan", "pixel_gan_no_feature", "no_clip"])
def test_loss_mixes_like_the_reference(tmp_path, mix):
    """The recipe with parts switched off: no discriminator at all (PSNR-style training with the perceptual term), no perceptual
    network, no gradient clipping -- optional pieces of `optimize_parameters` (sr_model.py:162-267) must drop out the same way."""
    from trainner_amd.models import create_model
    from trainner_amd.options import options
    kw = dict(KW, **{"pixel_feature_no_gan": dict(gan=False), "pixel_gan_no_feature": dict(feature=False),
                     "no_clip": dict(grad_clip=False)}[mix])
    ropt, ref = R.build_reference_model(R.esrgan_yaml(name="mix", out_root=str(tmp_path / "ref"), **kw), seed=0)
    eng = create_model(options.parse(R.esrgan_yaml(name="mix", out_root=str(tmp_path / "eng"), gpu_ids="[0]", **kw), is_train=True),
                       verbose=False)
    has_d, has_f = kw.get("gan", True), kw.get("feature", True)
    g = detrand.fill_state_dict_({n: v.detach().cpu().clone() for n, v in eng.netG.state_dict().items()}, 101)
    d = detrand.fill_state_dict_({n: v.detach().cpu().clone() for n, v in eng.netD.state_dict().items()}, 202) if has_d else None
    f = FX.vgg_state(77) if has_f else None
    TS.load_initial(eng, g, d, f)
    ref.netG.load_state_dict(g)
    if has_d:
        ref.netD.load_state_dict(d)
    if has_f:
        nf = R.reference_netF(ref)
        sd = nf.state_dict()
        sd.update(f)
        nf.load_state_dict(sd)
    assert (R.reference_netF(ref) is not None) == has_f and bool(getattr(ref, "cri_gan", False)) == has_d
    for s in range(1, 4):
        le, lr = _step(ref, eng, s)
        _logs_close(le, lr, 2e-4 if s == 1 else 3e-3)
        # after the first step: round-off only (0.0003 lr measured).  Later the two fp32 trajectories separate -- without the
        # perceptual term G's gradient is small and passes through a batch-2 BatchNorm discriminator: measured 0.008 lr after
        # step 2, 0.047 after step 3, 0.2 % of the elements more than half an lr apart -- while every logged loss still agrees
        for a, b in ((eng.netG, ref.netG),) + (((eng.netD, ref.netD),) if has_d else ()):
            mean, worst = _weights(a, b)
            assert mean <= (0.002 if s == 1 else 0.03 * s) and worst <= 2.05 * s, (mix, s, mean, worst)


def test_auto_gradient_clip_like_the_reference(tmp_path):
    """`grad_clip_value: auto` (the alternative the shipped recipe names, train_sr.yml:190; base_model.py:896-922): the clip norm
    is the 10th percentile of the running history of gradient norms."""
    (ropt, ref), (eopt, eng) = _both(tmp_path, [("  grad_clip_value: 0.1", "  grad_clip_value: auto")])
    for s in range(1, 5):
        le, lr = _step(ref, eng, s)
        _logs_close(le, lr, 2e-4 if s == 1 else 3e-3)
        assert len(eng.grad_history) == len(ref.grad_history)
        for i, (a, b) in enumerate(zip(eng.grad_history, ref.grad_history)):
            # (first entry: round-off; later entries carry the trajectory drift -- these norms are far above the recipe's 0.1)
            assert abs(a - b) <= (1e-4 if i == 0 else 1e-2) * abs(b) + 1e-9, (s, eng.grad_history, ref.grad_history)
    for a, b in ((eng.netG, ref.netG), (eng.netD, ref.netD)):
        mean, worst = _weights(a, b)
        assert mean <= 0.03 and worst <= 2.05 * 4, (mean, worst)


@pytest.mark.parametrize("policy", ["multistep_rel_warmup", "multistep_restarts"])
def test_learning_rate_policy_like_the_reference(tmp_path, policy):
    """train.py:306-309: `model.update_learning_rate(current_step, warmup_iter=opt['train']['warmup_iter'])` after every step; the
    logger prints `get_current_learning_rate()`.  Only the schedulers run here (no optimizer steps needed for the lr values)."""
    if policy == "multistep_rel_warmup":        # the shipped recipe's keys (train_sr.yml: lr_steps_rel, niter) + warm-up
        repl = [("  lr_steps: [50000, 100000]", "  lr_steps_rel: [0.1, 0.2, 0.4, 0.6]\n  warmup_iter: 5"), ("  niter: 500000", "  niter: 50")]
    else:
        repl = [("  lr_steps: [50000, 100000]", "  lr_steps: [10, 20, 35, 45]\n  restarts: [25]\n  restart_weights: [0.5]"), ("  niter: 500000", "  niter: 60")]
    (ropt, ref), (eopt, eng) = _both(tmp_path, repl)
    assert list(eopt["train"]["lr_steps"]) == list(ropt["train"]["lr_steps"])
    warm = ropt["train"].get("warmup_iter") or -1
    assert (eopt["train"].get("warmup_iter") or -1) == warm
    import warnings
    traj_r, traj_e = [], []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # scheduler.step() before optimizer.step()
        for s in range(1, 56):
            with R.reference_env():
                ref.update_learning_rate(s, warmup_iter=warm)
                traj_r.append((ref.get_current_learning_rate(), [g["lr"] for o in ref.optimizers for g in o.param_groups]))
            eng.update_learning_rate(s, warmup_iter=warm)
            traj_e.append((eng.get_current_learning_rate(), [g["lr"] for o in eng.optimizers for g in o.param_groups]))
    for s, (a, b) in enumerate(zip(traj_e, traj_r), 1):
        assert abs(a[0] - b[0]) <= 1e-12 and all(abs(x - y) <= 1e-12 for x, y in zip(a[1], b[1])), (s, a, b)
    assert len({round(b[1][0], 12) for b in traj_r}) >= 4             # the policy did something over these steps


def test_linear_policy_of_the_i2i_recipes_like_the_reference(tmp_path):
    """options/i2i/train_{pix2pix,cyclegan}.yml: lr_scheme Linear with fixed_niter / niter_decay."""
    from trainner_amd.models import create_model
    from trainner_amd.options import options
    kw = dict(model="pix2pix", batch=2, crop=64, n_blocks=1, ngf=16, ndf=16, lr_scheme="Linear")
    repl = [("  fixed_niter: 25000", "  fixed_niter: 10"), ("  niter_decay: 25000", "  niter_decay: 20")]
    ryml = _edit(R.i2i_yaml(name="lin", out_root=str(tmp_path / "ref"), **kw), repl)
    eyml = _edit(R.i2i_yaml(name="lin", out_root=str(tmp_path / "eng"), gpu_ids="[0]", **kw), repl)
    ropt, ref = R.build_reference_model(ryml, seed=0)
    eng = create_model(options.parse(eyml, is_train=True), verbose=False)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for s in range(1, 36):
            with R.reference_env():
                ref.update_learning_rate(s, warmup_iter=-1)
                lr_r = [g["lr"] for o in ref.optimizers for g in o.param_groups]
            eng.update_learning_rate(s, warmup_iter=-1)
            lr_e = [g["lr"] for o in eng.optimizers for g in o.param_groups]
            assert all(abs(x - y) <= 1e-12 for x, y in zip(lr_e, lr_r)), (s, lr_e, lr_r)
    assert lr_r[0] < 2e-4 * 0.2               # decayed
