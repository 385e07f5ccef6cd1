"""Host-logic tests (-m "not gpu"): the engine's hand-written forward/backward SCHEDULES, autograd
bridge, flat-parameter optimiser plumbing and SRModel step, executed with the torch-CPU stand-in for
the C ABI (tests/emul_backend.py) and checked against the oracle and the reference-generated goldens.
The very same test bodies run on the GPU box through libtrainner_hip.so (test_gpu_nets / test_gpu_step).
"""
import pytest
import torch

import emul_backend
import test_gpu_nets as TN
import test_gpu_step as TS


@pytest.fixture(autouse=True)
def emulated(monkeypatch):
    emul_backend.install(monkeypatch)
    monkeypatch.setattr(TN, "DEV", "cpu")
    monkeypatch.setattr(TS, "DEV", "cpu")
    torch.set_num_threads(8)


@pytest.mark.parametrize("mode", ["upconv", "pixelshuffle"])
def test_rrdbnet_schedule(mode):
    TN.test_rrdbnet_forward_backward(mode)


def test_srresnet_schedule():
    TN.test_srresnet_forward_backward()


def test_discriminator_schedule():
    TN.test_discriminator_vgg(64, 16)


def test_vgg_schedule():
    TN.test_vgg19_features()


@pytest.mark.parametrize("case", ["cfg1_srresnet", "esrgan_nb1_crop64", "esrgan_nb1_pixelshuffle"])
def test_step_vs_reference_golden(case, tmp_path):
    TS.test_step_matches_reference_golden(case, tmp_path)


def test_validation_forward_and_self_ensemble(tmp_path):
    TS.test_validation_forward_and_self_ensemble(tmp_path)
