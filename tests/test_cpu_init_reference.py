"""Same seed => the reference's initial weights, bit for bit (-m "not gpu", build container only: needs /root/reference).

train.py:107 seeds python / numpy / torch (`util.set_random_seed(manual_seed)`) before `create_model`; the networks are then
built module by module -- every nn.Conv2d / nn.ConvTranspose2d / nn.Linear draws its default init from torch's generator -- and
`networks.init_weights` (kaiming_normal_ x init_scale, networks.py:41-54,71-100) draws again in `Module.apply` order.  The
engine's parameter holders (block.Conv2dHIP, LinearHIP, ResNet_arch.ConvTranspose2dHIP) make the same draws in the same order, so
a run started from scratch with the reference's seed starts from the reference's weights.  Checked for every network kind built:
RRDBNet, SRResNet, Discriminator_VGG, UNetDiscriminator, ResnetGenerator, UnetGenerator, PatchGAN."""
import random

import numpy as np
import pytest
import torch

import emul_backend
from oracle import ref_harness as R

pytestmark = pytest.mark.skipif(not R.reference_available(), reason="needs the reference checkout (build container)")

CASES = {
    "esrgan": (R.esrgan_yaml, dict(nb=2, batch=2, crop=64, d_nf=16), ("G", "D")),
    "esrgan_unet_d": (R.esrgan_yaml, dict(nb=1, batch=2, crop=64, d_nf=16, d_type="unet"), ("G", "D")),
    "srresnet_psnr": (R.esrgan_yaml, dict(nb=2, batch=2, crop=64, model_G="sr_resnet", gan=False, feature=False), ("G",)),
    "cyclegan": (R.i2i_yaml, dict(model="cyclegan", batch=1, crop=64, n_blocks=2, ngf=16, ndf=16), ("G_A", "G_B", "D_A", "D_B")),
    "pix2pix_unet": (R.i2i_yaml, dict(model="pix2pix", batch=2, crop=128, ngf=16, ndf=16, norm_G="batch", which_G="unet_128"), ("G", "D")),
}


@pytest.mark.parametrize("case", list(CASES))
def test_same_seed_gives_the_reference_initial_weights(case, tmp_path, monkeypatch):
    from trainner_amd.models import create_model
    from trainner_amd.options import options
    emul_backend.install(monkeypatch)
    make, kw, nets = CASES[case]
    opt, ref = R.build_reference_model(make(name="init", out_root=str(tmp_path / "ref"), **kw), seed=7)   # set_random_seed(7)
    eopt = options.parse(make(name="init", out_root=str(tmp_path / "eng"), gpu_ids="[0]", **kw), is_train=True)
    random.seed(7)
    np.random.seed(7)
    torch.manual_seed(7)
    eng = create_model(eopt, verbose=False)
    for n in nets:
        a, b = getattr(eng, "net" + n).state_dict(), getattr(ref, "net" + n).state_dict()
        assert list(a) == list(b)
        for k, v in b.items():
            assert torch.equal(a[k].detach().cpu(), v.detach().cpu()), (n, k)
        w = next(v for k, v in b.items() if k.endswith("weight") and v.dim() == 4)
        assert w.std().item() > 0                                  # (a drawn tensor, not a constant fill)
