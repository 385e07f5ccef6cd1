"""Host-logic tests (-m "not gpu"): the engine's hand-written forward/backward SCHEDULES, autograd
bridge, flat-parameter optimiser plumbing and SRModel step, executed with the torch-CPU stand-in for
the C ABI (tests/emul_backend.py) and checked against the oracle and the reference-generated goldens.
The very same test bodies run on the GPU box through libtrainner_hip.so (test_gpu_nets / test_gpu_step).
"""
import pytest
import torch

import emul_backend
import test_gpu_i2i as TI
import test_gpu_nets as TN
import test_gpu_step as TS


@pytest.fixture(autouse=True)
def emulated(monkeypatch):
    emul_backend.install(monkeypatch)
    monkeypatch.setattr(TN, "DEV", "cpu")
    monkeypatch.setattr(TS, "DEV", "cpu")
    monkeypatch.setattr(TI, "DEV", "cpu")
    torch.set_num_threads(8)


@pytest.mark.parametrize("mode", ["upconv", "pixelshuffle"])
def test_rrdbnet_schedule(mode):
    TN.test_rrdbnet_forward_backward(mode)


def test_rrdbnet_gaussian_noise_schedule():
    TN.test_rrdbnet_gaussian_noise()


def test_srresnet_schedule():
    TN.test_srresnet_forward_backward()


def test_discriminator_schedule():
    TN.test_discriminator_vgg(64, 16)


@pytest.mark.parametrize("skip", [True, False])
def test_unet_discriminator_schedule(skip):
    TN.test_unet_discriminator(16, skip)


@pytest.mark.parametrize("norm,nb", [("instance", 2), ("batch", 1)])
def test_resnet_generator_schedule(norm, nb):
    TN.test_resnet_generator(norm, nb)


def test_patchgan_schedule():
    TN.test_patchgan_discriminator(6, 64)


def test_vgg_schedule():
    TN.test_vgg19_features()


@pytest.mark.parametrize("case", ["cfg1_srresnet", "esrgan_nb1_crop64", "esrgan_nb1_pixelshuffle", "esrgan_nb2_crop64_k10",
                                  "esrgan_nb1_unet", "esrgan_nb2_crop64_gauss", "esrgan_nb2_crop128_b16"])
def test_step_vs_reference_golden(case, tmp_path):
    TS.test_step_matches_reference_golden(case, tmp_path)


def test_step_vs_reference_golden_split_operand_mode(tmp_path, monkeypatch):
    """TNR_MMA=bf16x3 is an fp32 mode: the host logic (setup_amp keeps ops.FP32_MMA, the chain / weight-gradient policies in
    ops.py, descriptors carrying mma = 2) runs the same step; the stand-in backend computes fp32 for it, as the contract says."""
    TS.test_step_matches_reference_golden_bf16x3("esrgan_nb1_crop64", tmp_path, monkeypatch)


@pytest.mark.parametrize("case", TI.I2I_CASES)
def test_i2i_step_vs_reference_golden(case, tmp_path):
    TI.test_i2i_step_matches_reference_golden(case, tmp_path)


def test_i2i_step_nine_block_form_of_the_7x7_layers(tmp_path, monkeypatch):
    TI.test_i2i_step_golden_with_the_nine_block_form_of_the_7x7_layers(tmp_path, monkeypatch)


@pytest.mark.parametrize("d_type", ["discriminator_vgg", "unet"])
def test_discriminator_forward_memoization(tmp_path, monkeypatch, d_type):
    TS.test_discriminator_forward_memoization_is_exact(tmp_path, monkeypatch, d_type)


def test_i2i_amp_policy_step(tmp_path):
    TI.test_i2i_amp_bf16_step_tracks_the_fp32_oracle("pix2pix", tmp_path)


def test_amp_policy_step(tmp_path):
    TS.test_amp_bf16_step_tracks_the_fp32_oracle(tmp_path, "discriminator_vgg")


def test_validation_forward_and_self_ensemble(tmp_path):
    TS.test_validation_forward_and_self_ensemble(tmp_path)


def test_freezeD_layers_are_not_trained(tmp_path):
    """train.freeze_loc (FreezeD, base_model.py:641-655 / sr_model.py:249-253): the first layers of the discriminator
    have requires_grad False during the D step, so the reference's Adam never touches them; the engine's flat Adam
    launch must leave them bit-identical too."""
    from oracle import detrand, ref_harness
    from trainner_amd.models import create_model
    from trainner_amd.options import options
    yml = ref_harness.esrgan_yaml(name="freezed", out_root=str(tmp_path), gpu_ids="[0]", nb=1, batch=2, crop=64, d_nf=16)
    txt = open(yml).read().replace("  gan_type: vanilla", "  gan_type: vanilla\n  freeze_loc: 2")
    open(yml, "w").write(txt)
    opt = options.parse(yml, is_train=True)
    model = create_model(opt, verbose=False)
    assert model.feature_loc == 4                       # (loc * 3) - 2: features.0 .. features.3 are frozen
    before = {k: v.detach().clone() for k, v in model.netD.state_dict().items()}
    for s in (1, 2):
        LR, HR = detrand.synthetic_pair(2, 64, 90 + s)
        model.feed_data({"LR": LR, "HR": HR})
        model.optimize_parameters(s)
    after = model.netD.state_dict()
    frozen = [k for k in before if k.split(".")[0] == "features" and int(k.split(".")[1]) < 4
              and (k.endswith("weight") or k.endswith("bias"))]
    assert len(frozen) == 6                             # conv0 w/b, conv1 (features.2) w/b, its BatchNorm (features.3) w/b
    for k in frozen:
        assert torch.equal(before[k], after[k]), k
    moved = [k for k in before if k.endswith("weight") and k not in frozen and not torch.equal(before[k], after[k])]
    assert "features.5.weight" in moved and "classifier.2.weight" in moved


def test_reference_shipped_recipe_runs_unmodified(tmp_path, monkeypatch):
    """options/sr/train_sr.yml as the reference ships it (gaussian noise, AMP, pretrained G, RRDBNet-23) over the emulated C ABI
    (one step, no repeat run: RRDBNet-23 at batch 8 is slow on the CPU stand-in; the GPU test takes three steps twice)."""
    TS.test_reference_shipped_recipe_runs_unmodified(tmp_path, monkeypatch, nsteps=1, repeat=False)


@pytest.mark.parametrize("gaussian", [False, True])
def test_checkpoint_and_state_resume_continues_bit_for_bit(tmp_path, gaussian):
    """save + save_training_state, the train.py resume flow on a fresh model, bit-identical continuation (emulated C ABI)."""
    TS.test_checkpoint_and_state_resume_continues_bit_for_bit(tmp_path, gaussian)


def test_reference_shipped_test_recipe(tmp_path):
    """options/sr/test_sr.yml (inference: is_train False, pretrained RRDB_ESRGAN_x4) over the emulated C ABI."""
    TS.test_reference_shipped_test_recipe(tmp_path)


def test_reference_shipped_json_recipe(tmp_path, monkeypatch):
    """options/sr/train_sr.json (the same recipe in the JSON dialect) over the emulated C ABI."""
    TS.test_reference_shipped_json_recipe(tmp_path, monkeypatch)


def test_reference_shipped_cyclegan_recipe(tmp_path):
    """options/i2i/train_cyclegan.yml unmodified (relativistic form by default, pools, use_amp, two pretrained generators)."""
    TI.test_reference_shipped_cyclegan_recipe_runs_unmodified(tmp_path)


def test_reference_shipped_pix2pix_recipe(tmp_path):
    """options/i2i/train_pix2pix.yml: raises where the reference raises (no gan_opt, no real image in the generator stage);
    steps with gan_opt.form: standard (unet_256 + conditional PatchGAN, use_amp) over the emulated C ABI."""
    TI.test_reference_shipped_pix2pix_recipe(tmp_path)


@pytest.mark.parametrize("gaussian", [False, True])
def test_step_gate_pinned_fp64_trajectory(tmp_path, gaussian):
    """The gate-pinned float64 arbitration of three consecutive steps (tests/test_gpu_step.py) over the emulated C ABI."""
    TS.test_step_gate_pinned_fp64_trajectory(tmp_path, gaussian)
