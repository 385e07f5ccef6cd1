"""Drive the REAL reference (victorca25/traiNNer, read-only at /root/reference) on CPU.

TEST INFRASTRUCTURE ONLY -- never imported by the product path (trainner_amd/).
Only usable in the build container (the GPU box has no /root/reference); it is what
`oracle/make_golden.py` uses to produce the committed fixtures in tests/golden/ and what
pins `oracle/sr_oracle.py` (the travelling CPU restatement).

Recipe = SURVEY.md 8(c) / Appendix A:
  * stubs for cv2 / torchvision first on sys.path, then /root/reference/codes
  * cwd = /root/reference/codes (preset lookup is relative: codes/options/options.py:177)
  * gpu_ids: [] (CPU, no DataParallel: codes/models/base_model.py:76-81), use_amp: false
  * network_G gaussian: false (codes/models/modules/architectures/block.py:592 is CUDA-only)
"""
import os
import sys
import tempfile
import contextlib

REF_ROOT = "/root/reference"
REF_CODES = os.path.join(REF_ROOT, "codes")
STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stubs")


def reference_available():
    return os.path.isdir(REF_CODES)


@contextlib.contextmanager
def reference_env():
    """sys.path / cwd set up so `import models, options` resolve to the reference."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_CODES)
    sys.dont_write_bytecode = True
    old_path, old_cwd = list(sys.path), os.getcwd()
    sys.path.insert(0, REF_CODES)
    sys.path.insert(0, STUBS)
    os.chdir(REF_CODES)
    try:
        yield
    finally:
        os.chdir(old_cwd)
        sys.path[:] = old_path


def esrgan_yaml(name="oracle_esrgan", batch=2, crop=128, nb=2, nf=64, d_nf=64, model_G="esrgan",
                gan=True, feature=True, pixel_weight=1e-2, grad_clip=True, upsample_mode=None,
                out_root=None, gpu_ids="[]", d_type="discriminator_vgg", amp=False, gaussian=False):
    """A train_sr.yml-shaped config (codes/options/sr/train_sr.yml:1-195) for CPU."""
    out_root = out_root or tempfile.mkdtemp(prefix="tnr_oracle_")
    os.makedirs(out_root, exist_ok=True)
    if model_G == "esrgan":
        netg = "network_G:\n  type: esrgan\n  gaussian: %s\n  nb: %d\n  nf: %d\n" % ("true" if gaussian else "false", nb, nf)
        if upsample_mode:
            netg += "  upsample_mode: %s\n" % upsample_mode
    else:
        netg = "network_G:\n  type: sr_resnet\n  nb: %d\n  nf: %d\n" % (nb, nf)
    netd = "network_D:\n  type: %s\n  nf: %d\n" % (d_type, d_nf) if gan else ""
    train = [
        "  optim_G: adam", "  optim_D: adam", "  lr_scheme: MultiStepLR",
        "  lr_steps: [50000, 100000]", "  lr_gamma: 0.5",
        "  pixel_criterion: l1", "  pixel_weight: %g" % pixel_weight,
    ]
    if feature:
        train += ["  feature_criterion: l1", "  feature_weight: 1"]
        if gpu_ids != "[]":     # engine configs: the parity tests load their own seeded VGG weights (no ImageNet file here)
            train += ["  perceptual_allow_random_init: true"]
    if gan:
        train += ["  gan_type: vanilla", "  gan_weight: 5e-3"]
    train += ["  manual_seed: 0", "  niter: 500000", "  val_freq: 5000"]
    if grad_clip:
        train += ["  grad_clip: norm", "  grad_clip_value: 0.1"]
    txt = "\n".join([
        "name: %s" % name, "use_tb_logger: false", "model: sr", "scale: 4", "gpu_ids: %s" % gpu_ids,
        "use_amp: %s" % ("true" if amp else "false"), "use_swa: false", "use_cem: false", "use_atg: false",
        "datasets:", "  train:", "    name: synth", "    mode: aligned",
        "    dataroot_HR: /tmp/none_hr", "    dataroot_LR: /tmp/none_lr", "    znorm: false",
        "    n_workers: 0", "    batch_size: %d" % batch, "    virtual_batch_size: %d" % batch,
        "    preprocess: crop", "    crop_size: %d" % crop, "    image_channels: 3",
        "path:", "  root: %s" % out_root,
        netg.rstrip("\n"), netd.rstrip("\n"),
        "train:", *train,
        "logger:", "  print_freq: 1", "  save_checkpoint_freq: 1000000", ""])
    path = os.path.join(out_root, name + ".yml")
    with open(path, "w") as f:
        f.write(txt)
    return path


class _EngineDraw:
    """Stands where GaussianNoise keeps its 0-dim `noise` tensor (block.py:592): `.repeat(*size).normal_()` (:597) -- the reference's
    draw from torch's global generator -- returns the field the ENGINE draws for this block instead (oracle/gauss_noise.py restates
    csrc/gauss_noise.h; key = (noise_seed, this block's call count, block index)).  The reference's forward (:594-599: training only,
    scale = sigma * x, x + n * scale) runs unmodified around it."""

    def __init__(self, seed, block):
        self.seed, self.block, self.calls, self.size = seed, block, 0, None

    def repeat(self, *size):
        self.size = size
        return self

    def normal_(self):
        from . import gauss_noise
        N, C, H, W = self.size
        call, self.calls = self.calls, self.calls + 1
        return gauss_noise.normals_nchw(N, C, H, W, self.seed, call, self.block)


def _substitute_gaussian_draw(noise_seed):
    """gaussian: true on CPU.  Two substitutions, neither in the module's arithmetic: the constructor's unconditional
    `.to(torch.device('cuda'))` (block.py:592) and the source of the N(0, 1) draw (_EngineDraw).  Blocks are numbered in
    construction order = RRDBNet's forward order (RRDBNet_arch.py:27-29,72-79)."""
    import torch
    from models.modules.architectures import block as B
    count = [0]

    def init(self, sigma=0.1, is_relative_detach=False):
        torch.nn.Module.__init__(self)
        self.sigma, self.is_relative_detach = sigma, is_relative_detach
        self.noise = _EngineDraw(noise_seed, count[0])
        count[0] += 1

    B.GaussianNoise.__init__ = init


def build_reference_model(yaml_path, seed=0, noise_seed=None):
    """-> (opt, model) built by the reference's own options.parse / create_model.  noise_seed: see _substitute_gaussian_draw
    (required for `gaussian: true`, which the reference cannot construct on CPU)."""
    with reference_env():
        for m in [k for k in sys.modules if k.split(".")[0] in
                  ("models", "options", "utils", "dataops", "data", "cv2", "torchvision")]:
            del sys.modules[m]
        import options.options as O
        from models import create_model
        from utils import util
        if noise_seed is not None:
            _substitute_gaussian_draw(noise_seed)
        with open(os.devnull, "w") as dn, contextlib.redirect_stdout(dn):
            opt = O.parse(yaml_path, is_train=True)
        util.set_random_seed(seed)
        model = create_model(opt, verbose=False)
    return opt, model


def reference_step(model, LR, HR, step):
    """feed_data + optimize_parameters (codes/models/sr_model.py:115-128,195-267)."""
    with reference_env():
        model.feed_data({"LR": LR, "HR": HR})
        model.optimize_parameters(step)
    return dict(model.get_current_log())


def reference_netF(model):
    """The VGG FeatureExtractor inside GeneratorLoss (codes/models/losses.py:117-128)."""
    for l in model.generatorlosses.loss_list:
        if "fea" in l["name"]:
            return l["function"].network
    return None


def i2i_yaml(name="oracle_i2i", model="pix2pix", batch=2, crop=64, n_blocks=2, ngf=16, ndf=16, norm_G="instance",
             gan_type="vanilla", pixel_weight=100.0, lambda_identity=None, pool_size=0, out_root=None, gpu_ids="[]",
             lr_scheme="MultiStepLR", amp=False, which_G="resnet_net", gan_form="standard"):
    """A train_pix2pix.yml / train_cyclegan.yml-shaped config (codes/options/i2i/train_pix2pix.yml:1-143,
    train_cyclegan.yml:1-140) with the ResNet generator + PatchGAN of BASELINE.json configs[4], for CPU."""
    out_root = out_root or tempfile.mkdtemp(prefix="tnr_oracle_")
    os.makedirs(out_root, exist_ok=True)
    d_in = 6 if model == "pix2pix" else 3            # conditional D sees (A, B) pairs (pix2pix_model.py:63-66)
    train = ["  optim_G: adam", "  lr_G: 2e-4", "  beta1_G: 0.5", "  optim_D: adam", "  lr_D: 2e-4", "  beta1_D: 0.5"]
    if lr_scheme == "Linear":
        train += ["  lr_scheme: Linear", "  fixed_niter: 25000", "  niter_decay: 25000"]
    else:
        train += ["  lr_scheme: MultiStepLR", "  lr_steps: [50000, 100000]", "  lr_gamma: 0.5"]
    train += ["  pixel_criterion: l1", "  pixel_weight: %g" % pixel_weight, "  gan_type: %s" % gan_type, "  gan_weight: 1"]
    if gan_form is not None:        # None: no `gan_opt` at all, like the shipped recipes => 'relativistic' (losses.py:366-369)
        train += ["  gan_opt:", "    form: %s" % gan_form]
    if lambda_identity is not None:
        train += ["  lambda_identity: %g" % lambda_identity]
    train += ["  manual_seed: 0", "  niter: 50000", "  val_freq: 5000"]
    txt = "\n".join([
        "name: %s" % name, "use_tb_logger: false", "model: %s" % model, "scale: 1", "gpu_ids: %s" % gpu_ids,
        "use_amp: %s" % ("true" if amp else "false"), "use_swa: false", "use_cem: false", "use_atg: false",
        "pool_size: %d" % pool_size,
        "datasets:", "  train:", "    name: synth", "    mode: aligned", "    outputs: AB",
        "    dataroot_B: /tmp/none_b", "    dataroot_A: /tmp/none_a", "    znorm: true",
        "    n_workers: 0", "    batch_size: %d" % batch, "    virtual_batch_size: %d" % batch,
        "    preprocess: crop", "    crop_size: %d" % crop, "    image_channels: 3", "    input_nc: 3", "    output_nc: 3",
        "path:", "  root: %s" % out_root,
        *(["network_G:", "  which_model_G: resnet_net", "  n_blocks: %d" % n_blocks, "  ngf: %d" % ngf, "  norm_type: %s" % norm_G]
          if which_G == "resnet_net" else
          # the shipped Pix2Pix recipe's generator (options/i2i/train_pix2pix.yml:65): unet_128 = 7 down-samplings, crop 128
          ["network_G:", "  which_model_G: %s" % which_G, "  ngf: %d" % ngf, "  norm_type: %s" % norm_G]),
        "network_D:", "  which_model_D: patchgan", "  in_nc: %d" % d_in, "  nf: %d" % ndf,
        "train:", *train,
        "logger:", "  print_freq: 1", "  save_checkpoint_freq: 1000000", ""])
    path = os.path.join(out_root, name + ".yml")
    with open(path, "w") as f:
        f.write(txt)
    return path
