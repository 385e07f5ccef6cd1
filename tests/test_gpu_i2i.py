"""Step-level parity (-m gpu) of the image-to-image models (BASELINE.json configs[4]): trainner_amd's
Pix2PixModel / CycleGANModel.optimize_parameters against the golden fixtures produced by the REAL reference
(tests/golden/{pix2pix,cyclegan}_*.pt, oracle/make_golden_i2i.py), plus a live oracle comparison at 256 x 256 with the
ResNet-9 generator.  Tolerances (fp32 MFMA == fmaf chains; only summation order differs from the CPU reference):
step 1 is round-off only: log entries 2e-4 relative.  From step 2 on two correct fp32 implementations follow slightly
different trajectories: Adam's first steps are sign-like (+-lr per element), and an element whose gradient differs moves the
other way.  In these models that happens through ReLU gates: the generators are chained (rec_B = G_A(G_B(B))), the second
net sees an input that differs by 1e-6 from the reference's, a pre-activation within that distance of zero takes the other
branch and changes one receptive field of the encoder gradients (measured with the torch-CPU stand-in of the C ABI: 12 of
2 352 / 17 of 4 608 / 14 of 18 432 elements of G_B's three encoder weights flipped in step 1 of cyclegan_rn2_crop64, while
the same network on the SAME input tensor matches the oracle to 2e-6 -- test_resnet_generator and the fp64-gated net tests
are the tight checks).  Hence: log entries 3e-3 relative from step 2 on; generated images of step 1 mean |d| <= 2e-5, max <= 5e-4, of the last
step (1-4 sign-like Adam updates later: measured 5.5e-3 mean on rec_A of the 5-step case while every loss over those images
still agrees to 7e-4) mean <= 1e-2, max <= 1e-1 (tanh output range 2); post-step weights mean |dp| <= 15 % of lr*steps, worst 2 lr*steps... the K = 10 bounds of
test_gpu_step.CASE_TOL.
"""
import os
import random

import pytest
import torch

from oracle import detrand, fixtures as FX, ref_harness

pytestmark = pytest.mark.gpu
DEV = "cuda"

I2I_CASES = ["pix2pix_rn2_crop64", "pix2pix_rn1_bn_lsgan", "cyclegan_rn2_crop64", "cyclegan_rn1_noidt",
             "pix2pix_unet128",                  # the shipped Pix2Pix recipe's U-Net generator (which_model_G: unet_net)
             "cyclegan_rn1_relativistic"]        # the shipped CycleGAN recipe's GAN form (no gan_opt => relativistic), pools


def build_i2i_model(yaml_kw, tmp_path):
    from trainner_amd.models import create_model
    from trainner_amd.options import options
    yml = ref_harness.i2i_yaml(name="engine_i2i", out_root=str(tmp_path), gpu_ids="[0]", **yaml_kw)
    opt = options.parse(yml, is_train=True)
    return opt, create_model(opt, verbose=False)


def check_logs(log, ref_log, tol):
    assert list(log.keys()) == list(ref_log.keys()), (list(log.keys()), list(ref_log.keys()))
    for k, v in ref_log.items():
        t = max(tol, 2e-3) if k.startswith("D_") else tol        # raw mean logits (values near 0): absolute-ish bound
        assert abs(log[k] - v) <= t * max(1.0, abs(v)) + 5e-6, (k, log[k], v)


@pytest.mark.parametrize("case", I2I_CASES)
def test_i2i_step_matches_reference_golden(case, tmp_path):
    fx = FX.load(case)
    opt, model = build_i2i_model(fx["spec"]["yaml"], tmp_path)
    assert dict(opt["network_G"]) == fx["network_G"] and dict(opt["network_D"]) == fx["network_D"]
    assert list(model.model_names) == fx["model_names"]
    for n, sd in FX.i2i_initial_states(fx).items():
        net = getattr(model, "net" + n)
        assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == fx["keys"][n]
        net.load_state_dict(sd)
    random.seed(fx["seeds"]["pool"])
    for (s, (A, B)), ref_log in zip(FX.i2i_batches(fx), fx["logs"]):
        model.feed_data({"A": A, "B": B, "A_path": ["a"] * A.shape[0]})
        model.optimize_parameters(s)
        check_logs(model.get_current_log(), ref_log, 2e-4 if s < 2 else 3e-3)
        if s == 1:      # images of the first step: no weight update has reached them -- forward parity, round-off only
            for k, ref in fx["images_step1"].items():
                diff = (getattr(model, k).detach().cpu() - ref).abs()
                assert diff.mean().item() <= 2e-5 and diff.max().item() <= 5e-4, (k, diff.mean().item(), diff.max().item())
    for k, ref in fx["images"].items():
        diff = (getattr(model, k).detach().cpu() - ref).abs()
        assert diff.mean().item() <= 1e-2 and diff.max().item() <= 1e-1, (k, diff.mean().item(), diff.max().item())
    lr_steps = 2e-4 * fx["spec"]["steps"]
    for n in fx["model_names"]:
        sd = {k: v.detach().cpu() for k, v in getattr(model, "net" + n).state_dict().items()}
        skip = FX.norm_shadowed_biases(fx["keys"][n], fx["network_G"]["norm_type"]) if n.startswith("G") else ()
        worst, mean, k = FX.state_error(sd, fx["states"][n], skip, lr_steps=lr_steps)
        assert mean < 0.15 and worst < 2.05, (n, k, worst, mean)
        e, k = FX.buffers_error(sd, fx["states"][n])
        assert e < (2e-3 if fx["spec"]["steps"] <= 2 else 1e-2), (n, "running stats", k, e)     # (carry the trajectory drift)


def test_i2i_step_golden_with_the_nine_block_form_of_the_7x7_layers(tmp_path, monkeypatch):
    """TNR_K7_FUSED=0 (ResNet_arch._K7Image: the 7x7 image-side layers as nine 3x3 blocks of taps instead of one launch per pass) stays a
    working A/B switch: the same reference golden, the same bounds."""
    from trainner_amd.models.modules.architectures import ResNet_arch
    monkeypatch.setattr(ResNet_arch, "K7_FUSED", False)
    test_i2i_step_matches_reference_golden("pix2pix_rn2_crop64", tmp_path)


def _shipped_i2i_recipe(tmp_path, rel, edit=None):
    """The reference's options/<rel> (tests/golden/shipped_recipes.json: every key and value, locations re-rooted,
    oracle/make_golden_options.py) + the pretrained generators the recipe names, here seeded state_dicts of the recipe's nets."""
    from trainner_amd.models import create_model
    from trainner_amd.options import options
    yml = FX.write_recipe(rel, str(tmp_path), edit)
    opt = options.parse(yml, is_train=True)
    pre = {k: v for k, v in opt["path"].items() if k.startswith("pretrain_model_") and v}
    bare = options.parse(yml, is_train=True)
    for k in pre:
        bare["path"][k] = None
    torch.manual_seed(0)
    shapes = create_model(bare, verbose=False)
    saved = {}
    for i, (k, path) in enumerate(sorted(pre.items())):
        net = getattr(shapes, "net" + k[len("pretrain_model_"):])
        sd = detrand.fill_state_dict_({kk: vv.detach().cpu().clone() for kk, vv in net.state_dict().items()}, 700 + i)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        torch.save(sd, path)
        saved[k[len("pretrain_model_"):]] = sd
    del shapes
    torch.manual_seed(opt["train"]["manual_seed"])
    return opt, create_model(opt, verbose=False), saved


def _i2i_steps(opt, model, steps, seed):
    ds = opt["datasets"]["train"]
    logs = []
    for s in range(1, steps + 1):
        A = detrand.uniform((ds["batch_size"], 3, ds["crop_size"], ds["crop_size"]), seed + s, -1.0, 1.0)
        B = detrand.uniform((ds["batch_size"], 3, ds["crop_size"], ds["crop_size"]), seed + 5000 + s, -1.0, 1.0)
        model.feed_data({"A": A, "B": B, "A_path": ["a"] * ds["batch_size"]})
        model.optimize_parameters(s)
        logs.append(dict(model.get_current_log()))
    return logs


@pytest.mark.timeout(600)
def test_reference_shipped_cyclegan_recipe_runs_unmodified(tmp_path):
    """options/i2i/train_cyclegan.yml as shipped: resnet_net (9 blocks) x 2 + PatchGAN x 2 at 256 x 256, batch 1, use_amp, pool
    50, Linear LR policy, two pretrained generators, and NO `gan_opt` => the relativistic form (golden cyclegan_rn1_relativistic
    pins that form against the real reference).  Constructs, loads both generators, steps; finite logs with the reference's keys."""
    random.seed(7)
    opt, model, saved = _shipped_i2i_recipe(tmp_path, "i2i/train_cyclegan.yml")
    assert opt["network_G"]["type"] == "resnet_net" and opt["network_D"]["type"] == "patchgan" and opt["use_amp"] is True
    assert opt["datasets"]["train"]["batch_size"] == 1 and opt["datasets"]["train"]["crop_size"] == 256 and opt["pool_size"] == 50
    assert model.adversarial.form == "relativistic" and list(model.model_names) == ["G_A", "G_B", "D_A", "D_B"]
    for n, sd in saved.items():
        k = next(kk for kk in sd if kk.endswith(".weight"))
        assert torch.equal(getattr(model, "net" + n).state_dict()[k].detach().cpu(), sd[k]), (n, k)      # pretrained, not the init
    logs = _i2i_steps(opt, model, 3, 900)
    assert list(logs[0]) == ["l_g_gan_A", "pix-l1_idt_A", "pix-l1_A", "l_g_gan_B", "pix-l1_idt_B", "pix-l1_B"] or \
        list(logs[0]) == ["l_g_gan_A", "pix-l1_A", "l_g_gan_B", "pix-l1_B"], list(logs[0])
    assert {"l_d_real_A", "l_d_fake_A", "D_real_A", "D_fake_A", "l_d_real_B", "l_d_fake_B", "D_real_B", "D_fake_B"} <= set(logs[-1])
    for log in logs:
        assert all(v == v and abs(v) < 1e3 for v in log.values()), log


@pytest.mark.timeout(600)
def test_reference_shipped_pix2pix_recipe(tmp_path):
    """options/i2i/train_pix2pix.yml as shipped: unet_net (unet_256: 8 down-samplings) + conditional PatchGAN (in_nc 6), batch 2,
    256 x 256, use_amp, a pretrained generator.  It has no `gan_opt` either, and Pix2Pix calls the generator stage without the
    real image (pix2pix_model.py:152-154): the REFERENCE raises on its own recipe (TypeError: conv2d on None at
    losses.py:401-403, run in the build container) and so does the engine, at the same place, saying what to set.  With the one
    line the recipe needs -- train.gan_opt.form: standard, like options/i2i/train_wbc.yml:137-138 -- it steps."""
    opt, model, _ = _shipped_i2i_recipe(tmp_path, "i2i/train_pix2pix.yml")
    assert opt["network_G"]["type"] == "unet_net" and opt["network_D"]["input_nc"] == 6 and opt["use_amp"] is True
    with pytest.raises(TypeError, match="gan_opt.form: standard"):
        _i2i_steps(opt, model, 1, 910)
    del model
    tmp2 = tmp_path / "standard"
    tmp2.mkdir()
    def standard_form(tree):
        tree["train"]["gan_opt"] = {"form": "standard"}

    opt, model, saved = _shipped_i2i_recipe(tmp2, "i2i/train_pix2pix.yml", edit=standard_form)
    assert model.adversarial.form == "standard" and model.adversarial.conditional
    k = next(kk for kk in saved["G"] if kk.endswith(".weight"))
    assert torch.equal(model.netG.state_dict()[k].detach().cpu(), saved["G"][k])
    logs = _i2i_steps(opt, model, 2, 920)
    assert list(logs[-1]) == ["l_d_real", "l_d_fake", "D_real", "D_fake", "l_g_gan", "pix-l1"], list(logs[-1])
    for log in logs:
        assert all(v == v and abs(v) < 1e3 for v in log.values()), log


@pytest.mark.timeout(600)
@pytest.mark.parametrize("kind", ["pix2pix", "cyclegan"])
def test_i2i_step_matches_oracle_at_256(kind, tmp_path):
    """BASELINE.json configs[4]: 256 x 256, ResNet-9 generator (ngf 64), PatchGAN (ndf 64), batch 1: one live step against
    the CPU oracle (oracle/i2i_oracle.py, pinned to the reference by the goldens above)."""
    from oracle import i2i_oracle
    kw = dict(model=kind, batch=1, crop=256, n_blocks=9, ngf=64, ndf=64, pixel_weight=100.0 if kind == "pix2pix" else 10.0,
              lambda_identity=0.5 if kind == "cyclegan" else None)
    opt, model = build_i2i_model(kw, tmp_path)
    states = {}
    for i, n in enumerate(model.model_names):
        net = getattr(model, "net" + n)
        states[n] = detrand.fill_state_dict_({k: v.detach().cpu().clone() for k, v in net.state_dict().items()}, 400 + i)
        net.load_state_dict(states[n])
    common = dict(n_blocks=9, norm="instance", gan_type="vanilla", pixel_weight=kw["pixel_weight"])
    if kind == "pix2pix":
        orc = i2i_oracle.OraclePix2PixStep(states["G"], states["D"], **common)
    else:
        orc = i2i_oracle.OracleCycleGANStep(states["G_A"], states["G_B"], states["D_A"], states["D_B"], lambda_identity=0.5, **common)
    A, B = detrand.uniform((1, 3, 256, 256), 91, -1.0, 1.0), detrand.uniform((1, 3, 256, 256), 92, -1.0, 1.0)
    torch.set_num_threads(16)
    ref_log = orc.step(A, B)
    model.feed_data({"A": A, "B": B, "A_path": ["a"]})
    model.optimize_parameters(1)
    check_logs(model.get_current_log(), ref_log, 2e-4)
    diff = (model.fake_B.detach().cpu() - orc.fake_B.detach()).abs()
    assert diff.mean().item() <= 5e-5 and diff.max().item() <= 2e-3, (diff.mean().item(), diff.max().item())


@pytest.mark.parametrize("kind", ["pix2pix", "cyclegan"])
def test_i2i_amp_bf16_step_tracks_the_fp32_oracle(kind, tmp_path):
    """`use_amp: true` as shipped by options/i2i/train_{pix2pix,cyclegan}.yml: bf16 matrix-core operands (every MFMA launch of
    the generators and PatchGANs: 3x3 / 4x4-s2 tiles, the im2col GEMMs of the 4x4-s1 layers, the taps-in-K image layers and
    all weight gradients), fp32 accumulation, InstanceNorm / BatchNorm / losses / Adam in fp32.  Two steps against the fp32
    CPU oracle: the loss scalars within 3 % (bf16 resolution 2^-8 per operand), the generated image within 6e-2 (range 2)
    and NOT bit-equal to an fp32 run."""
    from oracle import i2i_oracle
    from trainner_amd import hip, ops
    kw = dict(model=kind, batch=2, crop=64, n_blocks=2, ngf=16, ndf=16, pixel_weight=100.0 if kind == "pix2pix" else 10.0,
              lambda_identity=0.5 if kind == "cyclegan" else None)
    try:
        opt, model = build_i2i_model(dict(kw, amp=True), tmp_path)
        assert model.amp and ops.MMA == ops.FP32_MMA          # the bf16 region covers the training step only
        states = {}
        for i, n in enumerate(model.model_names):
            net = getattr(model, "net" + n)
            states[n] = detrand.fill_state_dict_({k: v.detach().cpu().clone() for k, v in net.state_dict().items()}, 500 + i)
            net.load_state_dict(states[n])
        common = dict(n_blocks=2, norm="instance", gan_type="vanilla", pixel_weight=kw["pixel_weight"])
        if kind == "pix2pix":
            orc = i2i_oracle.OraclePix2PixStep(states["G"], states["D"], **common)
        else:
            orc = i2i_oracle.OracleCycleGANStep(states["G_A"], states["G_B"], states["D_A"], states["D_B"], lambda_identity=0.5, **common)
        worst = 0.0
        for s in (1, 2):
            A, B = detrand.uniform((2, 3, 64, 64), 95 + s, -1.0, 1.0), detrand.uniform((2, 3, 64, 64), 85 + s, -1.0, 1.0)
            ref_log = orc.step(A, B)
            model.feed_data({"A": A, "B": B, "A_path": ["a", "b"]})
            model.optimize_parameters(s)
            log = model.get_current_log()
            assert list(log.keys()) == list(ref_log.keys())
            for k, v in ref_log.items():
                if k.startswith("D_"):
                    assert abs(log[k] - v) <= 2e-2, (s, k, log[k], v)        # raw mean logits around 0
                else:
                    assert abs(log[k] - v) <= 0.03 * abs(v) + 1e-4, (s, k, log[k], v)
            worst = max(worst, (model.fake_B.detach().cpu() - orc.fake_B.detach()).abs().max().item())
        assert 1e-5 < worst < 6e-2, worst          # (two InstanceNorm-normalised ResNet blocks + tanh: range 2)
    finally:
        ops.MMA = ops.FP32_MMA
