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
