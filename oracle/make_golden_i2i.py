"""Generate tests/golden/{pix2pix,cyclegan}_*.pt from the REAL reference (run in the build container only).

TEST INFRASTRUCTURE ONLY.  Usage:  python -m oracle.make_golden_i2i [case ...]

The reference's own `options.parse` + `create_model('pix2pix' | 'cyclegan')` build the model on CPU (oracle/ref_harness.py,
config shaped like codes/options/i2i/train_{pix2pix,cyclegan}.yml with the ResNet generator + PatchGAN of BASELINE.json
configs[4]); every network is overwritten with the bit-reproducible fill of oracle/detrand.py, the A / B batches come from
detrand.uniform in [-1, 1) (znorm images), `random` is seeded right before the first step (image pool draws), and
`feed_data` + `optimize_parameters` run K steps.  Recorded: per-step `log_dict`, the generated images of the first and of the last step, and
probes of every tensor of every post-step state_dict.
"""
import os
import random
import sys

import torch

from . import detrand
from . import ref_harness as R
from .make_golden import OUT, probe_state

CASES = {
    # Pix2Pix: conditional PatchGAN (6 input channels), vanilla GAN (standard form) + 100 x L1, InstanceNorm generator
    "pix2pix_rn2_crop64": dict(yaml=dict(model="pix2pix", batch=2, crop=64, n_blocks=2, ngf=16, ndf=16, pixel_weight=100.0),
                               steps=3, seed=81),
    # the same with a BatchNorm generator and lsgan
    "pix2pix_rn1_bn_lsgan": dict(yaml=dict(model="pix2pix", batch=2, crop=64, n_blocks=1, ngf=16, ndf=16, norm_G="batch",
                                           gan_type="lsgan", pixel_weight=100.0), steps=2, seed=82),
    # the SHIPPED Pix2Pix recipe's generator (options/i2i/train_pix2pix.yml:65: which_model_G unet_net): UnetGenerator with 7
    # down-samplings (unet_128: 128 x 128 down to 1 x 1), BatchNorm, deconv up-sampling, conditional PatchGAN
    "pix2pix_unet128": dict(yaml=dict(model="pix2pix", batch=2, crop=128, ngf=16, ndf=16, norm_G="batch", pixel_weight=100.0,
                                      which_G="unet_128"), steps=3, seed=85),
    # CycleGAN: identity terms, image pools that fill during step 1-2 and draw from step 3 on, lsgan
    "cyclegan_rn2_crop64": dict(yaml=dict(model="cyclegan", batch=2, crop=64, n_blocks=2, ngf=16, ndf=16, gan_type="lsgan",
                                          pixel_weight=10.0, lambda_identity=0.5, pool_size=4), steps=5, seed=83),
    # CycleGAN recipe as shipped (vanilla GAN), no identity term, no pool, batch 1, Linear LR policy
    "cyclegan_rn1_noidt": dict(yaml=dict(model="cyclegan", batch=1, crop=64, n_blocks=1, ngf=16, ndf=16, pixel_weight=10.0,
                                         lr_scheme="Linear"), steps=2, seed=84),
    # the SHIPPED CycleGAN recipe's GAN: options/i2i/train_cyclegan.yml has no `gan_opt`, so Adversarial runs its default form,
    # 'relativistic' (losses.py:366-369): D_A compares fake_B with real_A in the generator stage (cyclegan_model.py:241-243) and
    # the pooled fake with real_B in its own; per-pixel PatchGAN logits, image pools, Linear LR policy
    "cyclegan_rn1_relativistic": dict(yaml=dict(model="cyclegan", batch=1, crop=64, n_blocks=1, ngf=16, ndf=16, pixel_weight=10.0,
                                                lr_scheme="Linear", pool_size=4, gan_form=None), steps=4, seed=86),
}
SEEDS = {"G": 301, "D": 302, "G_A": 303, "G_B": 304, "D_A": 305, "D_B": 306}
POOL_SEED = 4242


def ab_pair(n, size, seed):
    return detrand.uniform((n, 3, size, size), seed, -1.0, 1.0), detrand.uniform((n, 3, size, size), seed + 5000, -1.0, 1.0)


def run_case(name, spec):
    yml = R.i2i_yaml(name="golden_" + name, **spec["yaml"])
    opt, model = R.build_reference_model(yml, seed=0)
    names = list(model.model_names)
    for n in names:
        detrand.fill_state_dict_(getattr(model, "net" + n).state_dict(), SEEDS[n])
    batch, crop = spec["yaml"]["batch"], spec["yaml"]["crop"]
    random.seed(POOL_SEED)
    logs = []
    with R.reference_env():
        for s in range(1, spec["steps"] + 1):
            A, B = ab_pair(batch, crop, spec["seed"] + s)
            model.feed_data({"A": A, "B": B, "A_path": ["a"] * batch})
            model.optimize_parameters(s)
            logs.append(dict(model.get_current_log()))
            if s == 1:      # before any weight update reached these images: forward parity, round-off only
                imgs1 = {k: getattr(model, k).detach().clone() for k in ("fake_B", "fake_A", "rec_A", "rec_B") if hasattr(model, k)}
    imgs = {k: getattr(model, k).detach().clone() for k in ("fake_B", "fake_A", "rec_A", "rec_B") if hasattr(model, k)}
    fx = {"name": name, "spec": spec, "network_G": dict(opt["network_G"]), "network_D": dict(opt["network_D"]),
          "seeds": dict(SEEDS, data=spec["seed"], pool=POOL_SEED), "model_names": names, "logs": logs, "images": imgs, "images_step1": imgs1,
          "states": {n: probe_state(getattr(model, "net" + n).state_dict()) for n in names},
          "keys": {n: [(k, tuple(v.shape)) for k, v in getattr(model, "net" + n).state_dict().items()] for n in names},
          "torch": torch.__version__}
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".pt")
    torch.save(fx, path)
    print(name, "->", path, "%.1f KB" % (os.path.getsize(path) / 1024))
    for l in logs:
        print("   ", {k: round(v, 6) for k, v in l.items()})


def main(argv):
    torch.set_num_threads(os.cpu_count() or 1)
    for n in (argv or list(CASES)):
        run_case(n, CASES[n])


if __name__ == "__main__":
    main(sys.argv[1:])
