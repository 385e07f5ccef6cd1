"""Checkpoint / resume interchange with the REAL reference, both directions (-m "not gpu", build container only: needs
/root/reference; skipped elsewhere).  INTEGRATION.md says `<iter>_G.pth`, `<iter>_D.pth` (base_model.py:353-375) and `<iter>.state`
(optimizers + schedulers, base_model.py:454-477) written by either side continue training on the other.  Here that is executed:

  reference -> engine : the reference trains 2 steps, `save(2)` + `save_training_state(0, 2)`; the engine is constructed the way
                        train.py:81-104,176-189 resumes (path.resume_state -> options.check_resume -> create_model loads
                        2_G.pth / 2_D.pth -> resume_training -> update_schedulers), takes step 3; the reference takes step 3 too.
  engine -> reference : the same with the roles swapped.

The engine side runs over tests/emul_backend.py (the C ABI's contract in torch-CPU; the product has no CPU path) -- what is
under test is the file formats and the host logic (flat-buffer Adam <-> torch.optim.Adam state layout, scheduler state, key
layout), which are the same on the GPU.  Bounds: step-3 logs 2e-4 relative (round-off only: both sides continue from identical
weights AND identical Adam moments; were the moments or step counts lost, Adam's first step would be +-lr on every element and
the post-step weights below would differ by ~lr = 1e-4 on average, against the 0.02 lr mean / 0.6 lr worst-element bounds)."""
import os

import pytest
import torch

import emul_backend
import test_gpu_step as TS
from oracle import detrand, fixtures as FX, ref_harness as R

pytestmark = pytest.mark.skipif(not R.reference_available(), reason="needs the reference checkout (build container)")
KW = dict(nb=1, batch=2, crop=64, d_nf=16)
SEED = 4100
LR_ = 1e-4           # lr_G = lr_D of the configuration (defaults of the recipe, train_sr.yml)


@pytest.fixture(autouse=True)
def emulated(monkeypatch):
    emul_backend.install(monkeypatch)
    monkeypatch.setattr(TS, "DEV", "cpu")
    torch.set_num_threads(8)


def _fill(model, netF):
    g = detrand.fill_state_dict_({k: v.detach().cpu().clone() for k, v in model.netG.state_dict().items()}, 101)
    d = detrand.fill_state_dict_({k: v.detach().cpu().clone() for k, v in model.netD.state_dict().items()}, 202)
    f = FX.vgg_state(77)
    model.netG.load_state_dict(g)
    model.netD.load_state_dict(d)
    sd = netF.state_dict()
    sd.update(f)
    netF.load_state_dict(sd)
    return f


def _engine_netF(model):
    return [l["function"].network for l in model.generatorlosses.loss_list if "fea" in l["name"]][0]


def _resume_yaml(yml, state_path):
    txt = open(yml).read().replace("path:\n", "path:\n  resume_state: %s\n" % state_path, 1)
    out = yml.replace(".yml", "_resume.yml")
    with open(out, "w") as fh:
        fh.write(txt)
    return out


def _pair(s):
    return detrand.synthetic_pair(KW["batch"], KW["crop"], SEED + s)


def _check(log_a, log_b, sd_a, sd_b):
    assert list(log_a) == list(log_b)
    for k, v in log_b.items():
        assert abs(log_a[k] - v) <= 2e-4 * max(1.0, abs(v)), (k, log_a[k], v)
    # (conv biases in front of a BatchNorm have an exactly-zero true gradient: rounding noise that Adam turns into +-lr either way)
    skip = FX.bn_shadowed_biases([(k, tuple(v.shape)) for k, v in sd_b.items()])
    tot = n = 0
    for k, v in sd_b.items():
        if v.dtype.is_floating_point and k not in skip:
            d = (sd_a[k].detach().cpu().float() - v.detach().cpu().float()).abs()
            if k.endswith(("running_mean", "running_var")):
                assert d.max().item() <= 1e-5 * max(1.0, v.abs().max().item()), (k, d.max().item())
                continue
            # one Adam step apart at most: an element whose gradient is rounding noise moves by a fraction of lr either way
            assert d.max().item() <= 0.6 * LR_, (k, d.max().item())
            tot, n = tot + d.sum().item(), n + d.numel()
    if n:
        assert tot / n <= 0.02 * LR_, tot / n


def test_engine_resumes_a_reference_checkpoint(tmp_path):
    from trainner_amd.models import create_model
    from trainner_amd.options import options
    root = str(tmp_path)
    yml = R.esrgan_yaml(name="xresume", out_root=root, **KW)
    opt, ref = R.build_reference_model(yml, seed=0)
    f = _fill(ref, R.reference_netF(ref))
    for s in (1, 2):
        R.reference_step(ref, *_pair(s), s)
    for key in ("models", "training_state"):              # train.py:36-40 creates the experiment folders
        os.makedirs(opt["path"][key], exist_ok=True)
    with R.reference_env():
        ref.save(2)
        ref.save_training_state(0, 2)
    state_path = os.path.join(opt["path"]["training_state"], "2.state")
    assert os.path.isfile(state_path) and os.path.isfile(os.path.join(opt["path"]["models"], "2_G.pth"))
    log_ref = R.reference_step(ref, *_pair(3), 3)

    # engine: train.py:81-104 (get_resume_state + check_resume), create_model, train.py:176-189 (resume_training)
    eyml = _resume_yaml(R.esrgan_yaml(name="xresume", out_root=root, gpu_ids="[0]", **KW), state_path)
    eopt = options.parse(eyml, is_train=True)
    resume_state = torch.load(eopt["path"]["resume_state"], weights_only=False)
    options.check_resume(eopt)
    assert eopt["path"]["pretrain_model_G"].endswith("2_G.pth") and eopt["path"]["pretrain_model_D"].endswith("2_D.pth")
    eng = create_model(eopt, verbose=False)
    sd = _engine_netF(eng).state_dict()
    sd.update(f)
    _engine_netF(eng).load_state_dict(sd)
    eng.resume_training(resume_state)
    eng.update_schedulers(eopt["train"])
    assert resume_state["iter"] == 2 and len(resume_state["optimizers"]) == 2
    LR, HR = _pair(3)
    eng.feed_data({"LR": LR, "HR": HR})
    eng.optimize_parameters(3)
    _check(dict(eng.get_current_log()), log_ref, eng.netG.state_dict(), ref.netG.state_dict())
    _check({}, {}, eng.netD.state_dict(), ref.netD.state_dict())
    # the Adam step counters continued (3, not 1) on both optimizers
    for o in eng.optimizers:
        steps = {int(st["step"]) for st in o.state_dict()["state"].values()}
        assert steps == {3}, steps


def test_reference_resumes_an_engine_checkpoint(tmp_path):
    import sys
    root = str(tmp_path)
    opt, eng = TS.build_engine_model(dict(KW), tmp_path)
    f = _fill(eng, _engine_netF(eng))
    for s in (1, 2):
        LR, HR = _pair(s)
        eng.feed_data({"LR": LR, "HR": HR})
        eng.optimize_parameters(s)
    eng.save(2)
    eng.save_training_state(0, 2)
    state_path = os.path.join(opt["path"]["training_state"], "2.state")
    assert os.path.isfile(state_path)
    LR, HR = _pair(3)
    eng.feed_data({"LR": LR, "HR": HR})
    eng.optimize_parameters(3)
    log_eng = dict(eng.get_current_log())

    # reference: same experiment name and root => same models/ and training_state/ folders; its own resume flow
    ryml = _resume_yaml(R.esrgan_yaml(name="engine_case", out_root=root, **KW), state_path)
    with R.reference_env():
        for m in [k for k in sys.modules if k.split(".")[0] in ("models", "options", "utils", "dataops", "data", "cv2", "torchvision")]:
            del sys.modules[m]
        import options.options as O
        from models import create_model as ref_create
        ropt = O.parse(ryml, is_train=True)
        resume_state = torch.load(ropt["path"]["resume_state"], weights_only=False)
        O.check_resume(ropt)
        assert ropt["path"]["pretrain_model_G"].endswith("2_G.pth")
        ref = ref_create(ropt, verbose=False)
        netF = R.reference_netF(ref)
        sd = netF.state_dict()
        sd.update(f)
        netF.load_state_dict(sd)
        ref.resume_training(resume_state)
        ref.update_schedulers(ropt["train"])
    log_ref = R.reference_step(ref, LR, HR, 3)
    _check(log_eng, log_ref, eng.netG.state_dict(), ref.netG.state_dict())
    _check({}, {}, eng.netD.state_dict(), ref.netD.state_dict())
    for o in ref.optimizers:
        steps = {int(st["step"]) for st in o.state_dict()["state"].values()}
        assert steps == {3}, steps


# ----------------------------------------------------------------------------------------------------------------- CycleGAN
# Two networks per optimizer (G_A + G_B, D_A + D_B: cyclegan_model.py:120-133), four checkpoint files, check_resume's _A / _B keys
# (options.py:676-678,699-714).
I2I = dict(model="cyclegan", batch=1, crop=64, n_blocks=1, ngf=16, ndf=16, pixel_weight=10.0, lr_scheme="Linear")
LR_I2I = 2e-4
NETS = ("G_A", "G_B", "D_A", "D_B")


def _ab(s):
    return (detrand.uniform((1, 3, 64, 64), 7000 + s, -1.0, 1.0), detrand.uniform((1, 3, 64, 64), 12000 + s, -1.0, 1.0))


def _i2i_fill(model):
    for i, n in enumerate(NETS):
        net = getattr(model, "net" + n)
        net.load_state_dict(detrand.fill_state_dict_({k: v.detach().cpu().clone() for k, v in net.state_dict().items()}, 310 + i))


def _i2i_ref_step(ref, s):
    A, B = _ab(s)
    with R.reference_env():
        ref.feed_data({"A": A, "B": B, "A_path": ["a"], "B_path": ["b"]})
        ref.optimize_parameters(s)
    return dict(ref.get_current_log())


def _i2i_eng_step(eng, s):
    A, B = _ab(s)
    eng.feed_data({"A": A, "B": B, "A_path": ["a"]})
    eng.optimize_parameters(s)
    return dict(eng.get_current_log())


def _i2i_check(eng, ref, log_e, log_r):
    # (the D entries of CycleGAN's log lag one step -- they are folded in inside backward_G, cyclegan_model.py:296-307 -- so the
    #  side that continues still shows the previous step's, the freshly constructed side has none yet: compare what both have)
    common = [k for k in log_r if k in log_e]
    assert len(common) >= 4 and (set(log_e) <= set(log_r) or set(log_r) <= set(log_e)), (list(log_e), list(log_r))
    for k in common:
        if k.startswith(("l_d_", "D_")) and len(log_e) != len(log_r):
            continue
        assert abs(log_e[k] - log_r[k]) <= 3e-3 * max(1.0, abs(log_r[k])) + 5e-6, (k, log_e[k], log_r[k])
    for n in NETS:
        sd_e, sd_r = getattr(eng, "net" + n).state_dict(), getattr(ref, "net" + n).state_dict()
        skip = FX.norm_shadowed_biases([(k, tuple(v.shape)) for k, v in sd_r.items()], "instance") if n.startswith("G") else ()
        tot = cnt = 0
        for k, v in sd_r.items():
            if v.dtype.is_floating_point and k not in skip and "running" not in k:
                d = (sd_e[k].detach().cpu() - v.detach().cpu()).abs()
                tot, cnt = tot + d.sum().item(), cnt + d.numel()
        # lost moments or step counts would put every element ~lr apart (Adam's first step is +-lr); ReLU-gate flips between the
        # chained generators move a few receptive fields the other way (tests/test_gpu_i2i.py header), hence 0.15 and not 0.02
        assert tot / cnt <= 0.15 * LR_I2I, (n, tot / cnt / LR_I2I)


def _steps_of(optimizers):
    return [{int(st["step"]) for st in o.state_dict()["state"].values()} for o in optimizers]


def test_cyclegan_checkpoints_interchange_with_the_reference(tmp_path):
    import sys
    from trainner_amd.models import create_model
    from trainner_amd.options import options
    # ---- reference -> engine
    root = str(tmp_path / "r2e")
    ropt, ref = R.build_reference_model(R.i2i_yaml(name="xcyc", out_root=root, **I2I), seed=0)
    _i2i_fill(ref)
    for s in (1, 2):
        _i2i_ref_step(ref, s)
    for key in ("models", "training_state"):
        os.makedirs(ropt["path"][key], exist_ok=True)
    with R.reference_env():
        ref.save(2)
        ref.save_training_state(0, 2)
    state_path = os.path.join(ropt["path"]["training_state"], "2.state")
    assert sorted(os.listdir(ropt["path"]["models"])) == ["2_D_A.pth", "2_D_B.pth", "2_G_A.pth", "2_G_B.pth"]
    log_r = _i2i_ref_step(ref, 3)
    eopt = options.parse(_resume_yaml(R.i2i_yaml(name="xcyc", out_root=root, gpu_ids="[0]", **I2I), state_path), is_train=True)
    resume_state = torch.load(eopt["path"]["resume_state"], weights_only=False)
    options.check_resume(eopt)
    assert all(eopt["path"]["pretrain_model_" + n].endswith("2_%s.pth" % n) for n in NETS)
    eng = create_model(eopt, verbose=False)
    eng.resume_training(resume_state)
    eng.update_schedulers(eopt["train"])
    log_e = _i2i_eng_step(eng, 3)
    _i2i_check(eng, ref, log_e, log_r)
    assert _steps_of(eng.optimizers) == [{3}, {3}]
    # ---- engine -> reference (the engine above goes on: saves at 3, both take step 4)
    for key in ("models", "training_state"):
        os.makedirs(eopt["path"][key], exist_ok=True)
    eng.save(3)
    eng.save_training_state(0, 3)
    state3 = os.path.join(eopt["path"]["training_state"], "3.state")
    log_e = _i2i_eng_step(eng, 4)
    ryml = _resume_yaml(R.i2i_yaml(name="xcyc", out_root=root, **I2I), state3)
    with R.reference_env():
        for m in [k for k in sys.modules if k.split(".")[0] in ("models", "options", "utils", "dataops", "data", "cv2", "torchvision")]:
            del sys.modules[m]
        import options.options as O
        from models import create_model as ref_create
        ropt2 = O.parse(ryml, is_train=True)
        rs = torch.load(ropt2["path"]["resume_state"], weights_only=False)
        O.check_resume(ropt2)
        ref2 = ref_create(ropt2, verbose=False)
        ref2.resume_training(rs)
        ref2.update_schedulers(ropt2["train"])
    log_r = _i2i_ref_step(ref2, 4)
    _i2i_check(eng, ref2, log_e, log_r)
    assert _steps_of(ref2.optimizers) == [{4}, {4}]
