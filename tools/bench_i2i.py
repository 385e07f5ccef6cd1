"""Variant bench (NOT the headline): images/sec of the full G+D step of the image-to-image models of BASELINE.json configs[4]
-- Pix2PixModel / CycleGANModel.optimize_parameters, 256 x 256, ResNet-9 generator (ngf 64, InstanceNorm), PatchGAN
(ndf 64), fp32 on the matrix cores, synthetic znorm images resident in HBM, random-init weights.

    python tools/bench_i2i.py --model pix2pix|cyclegan [--batch 16] [--steps 8] [--warmup 2]

`step_tflops` uses the convolution FLOP of the step counted from the layer table below (2*MAC, real channels; transposed
and stride-2 layers by their true tap counts): forward G 99.1 GFLOP/img, forward D 6.3 (3 ch) / 6.4 (6 ch) GFLOP/img;
a pass that needs weight + data gradients costs 3x its forward, data-gradient only 2x.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


YAML = """
name: bench_{model}
use_tb_logger: false
model: {model}
scale: 1
gpu_ids: [0]
use_amp: {amp}
pool_size: {pool}
datasets:
  train:
    name: synthetic
    mode: aligned
    outputs: AB
    dataroot_A: /tmp/none_a
    dataroot_B: /tmp/none_b
    znorm: true
    n_workers: 0
    batch_size: {batch}
    virtual_batch_size: {batch}
    preprocess: crop
    crop_size: {crop}
    input_nc: 3
    output_nc: 3
path:
  root: {root}
network_G:
{netg}network_D:
  which_model_D: patchgan
  in_nc: {d_in}
  nf: 64
train:
  optim_G: adam
  lr_G: 2e-4
  beta1_G: 0.5
  optim_D: adam
  lr_D: 2e-4
  beta1_D: 0.5
  lr_scheme: Linear
  fixed_niter: 25000
  niter_decay: 25000
  pixel_criterion: l1
  pixel_weight: {pixel_weight}
  gan_type: vanilla
  gan_weight: 1
  gan_opt:
    form: standard
{idt}  manual_seed: 0
  niter: 50000
  val_freq: 5000
logger:
  print_freq: 200
  save_checkpoint_freq: 5e3
"""


def g_fwd_flop(size=256, ngf=64, nb=9, nc=3):
    s = size
    f = 2 * s * s * 49 * nc * ngf                                   # c7s1-64
    f += 2 * (s // 2) ** 2 * 9 * ngf * 2 * ngf                      # d128
    f += 2 * (s // 4) ** 2 * 9 * 2 * ngf * 4 * ngf                  # d256
    f += nb * 2 * 2 * (s // 4) ** 2 * 9 * (4 * ngf) ** 2            # residual blocks
    f += 2 * (s // 4) ** 2 * 9 * 4 * ngf * 2 * ngf                  # u128 (each input pixel meets all 9 taps)
    f += 2 * (s // 2) ** 2 * 9 * 2 * ngf * ngf                      # u64
    f += 2 * s * s * 49 * ngf * nc                                  # c7s1-3
    return f


def d_fwd_flop(size=256, ndf=64, in_nc=3):
    f, s, c = 0, size, in_nc
    for co in (ndf, 2 * ndf, 4 * ndf):
        s //= 2
        f += 2 * s * s * 16 * c * co
        c = co
    s -= 1
    f += 2 * s * s * 16 * c * 8 * ndf
    f += 2 * (s - 1) ** 2 * 16 * 8 * ndf
    return f


def step_flop(model, size):
    g, d3, d6 = g_fwd_flop(size), d_fwd_flop(size, in_nc=3), d_fwd_flop(size, in_nc=6)
    if model == "pix2pix":
        # G fwd + G bwd (w + d grads); D step: 2 passes fwd + full bwd (the fake pass needs no input gradient: ~3x);
        # G step through D: fwd + data-gradient
        return 3 * g + 2 * 3 * d6 + 2 * d6
    # cyclegan: 4 G passes in forward() + 2 identity passes, all with both gradients; per D: G-side fwd + dgrad, D-side 2 x full
    return 6 * 3 * g + 2 * (2 * d3 + 2 * 3 * d3)


def build(model_name, netg="resnet", batch=16, size=256, amp=False, device=None):
    """(model, batch dict): the reference's options/i2i/train_{pix2pix,cyclegan}.yml shape with synthetic images (also used by bench.py's
    `variant_config5` leg)."""
    from trainner_amd.models import create_model
    from trainner_amd.options import options
    dev = device or torch.device("cuda", 0)
    root = tempfile.mkdtemp(prefix="tnr_bench_i2i_")
    cyc = model_name == "cyclegan"
    yml = os.path.join(root, "bench.yml")
    with open(yml, "w") as f:          # the reference's options/i2i/train_{pix2pix,cyclegan}.yml with the ResNet generator
        g_txt = ("  which_model_G: resnet_net\n  n_blocks: 9\n  ngf: 64\n  norm_type: instance\n" if netg == "resnet" else
                 "  which_model_G: unet_net\n  ngf: 64\n  norm_type: batch\n")
        f.write(YAML.format(netg=g_txt, model=model_name, amp="true" if amp else "false", pool=50 if cyc else 0, batch=batch, crop=size,
                            root=root, d_in=3 if cyc else 6, pixel_weight=10 if cyc else 100,
                            idt="  lambda_identity: 0.5\n" if cyc else ""))
    torch.manual_seed(1234)
    model = create_model(options.parse(yml, is_train=True), verbose=False)
    g = torch.Generator().manual_seed(7)
    data = {"A": (torch.rand(batch, 3, size, size, generator=g) * 2 - 1).to(dev),
            "B": (torch.rand(batch, 3, size, size, generator=g) * 2 - 1).to(dev), "A_path": ["a"] * batch}
    return model, data


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", choices=["pix2pix", "cyclegan"], default="pix2pix")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--amp", action="store_true")
    ap.add_argument("--netg", choices=["resnet", "unet"], default="resnet",
                    help="resnet: ResnetGenerator-9 (BASELINE configs[4]); unet: the reference's Pix2Pix default unet_net (8 downs at 256, BatchNorm)")
    args = ap.parse_args()
    from trainner_amd import hip, ops
    hip.require_device()
    model, data = build(args.model, args.netg, args.batch, args.size, args.amp)
    step = 0
    for _ in range(args.warmup):
        step += 1
        model.feed_data(data)
        model.optimize_parameters(step)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step += 1
        model.feed_data(data)
        model.optimize_parameters(step)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    log = model.get_current_log()
    imgs = args.batch * args.steps
    fl = step_flop(args.model, args.size)
    # instrumented pass (HIP events around every convolution-family launch on the launch stream), as in bench.py
    sys.path.insert(0, ROOT)
    import bench as B
    prof = ops.ConvProfile()
    ops.PROFILE = prof
    for _ in range(2):
        step += 1
        model.feed_data(data)
        model.optimize_parameters(step)
    ops.PROFILE = None
    summ = prof.summary()
    mode = "bf16x3amp" if args.amp else ("bf16x3" if ops.MMA == hip.MMA_BF16X3 else "f32")
    peak = 2516.6 if args.amp else (2516.6 / 6.0 if ops.MMA == hip.MMA_BF16X3 else 157.3)
    fam_tab = B.family_table(summ, 2, peak, "i2i_%s_%s_%s" % (args.model, args.netg, mode), 307.8 if mode == "bf16x3" else None)
    dom_name = max((f for f in summ if f not in ("conv_thin", "wgrad_thin", "gconv")), key=lambda f: summ[f]["ms"])
    dom = fam_tab["families"][dom_name]
    print(json.dumps({
        "metric": "images/sec (G+D step), %s %dx%d" % (args.model, args.size, args.size), "value": round(imgs / dt, 2), "unit": "img/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 2),
        "higher_is_better": True, "dtype": "bf16" if args.amp else ("f32 (bf16x3)" if ops.MMA == hip.MMA_BF16X3 else "f32"), "data": "synthetic", "variant": True,
        "config": {"workload": "%s: %s + PatchGAN (ndf 64), batch %d, %dx%d, vanilla GAN (standard form) + L1%s (BASELINE configs[4])"
                               % (args.model, "ResnetGenerator-9 (ngf 64, InstanceNorm)" if args.netg == "resnet" else "UnetGenerator (8 downs, ngf 64, BatchNorm)",
                                  args.batch, args.size, args.size, ", identity 0.5, pool 50" if args.model == "cyclegan" else ""),
                   "mma": os.environ.get("TNR_MMA", "bf16x3")},
        "conv_gflop_per_img": round(fl / 1e9, 1) if args.netg == "resnet" else None,
        "step_tflops": round(fl * imgs / dt / 1e12, 2) if args.netg == "resnet" else None,
        # the same convention as bench.py's roofline: fp32-equivalent ceiling of the split arithmetic = bf16 dense peak / 6
        # the dominant MATRIX-CORE kernel family of this step (bench.py's convention: algorithmic FLOP of its launches / their HIP-event time,
        # against the mode's dense peak; fp32-equivalent ceiling of the split arithmetic = bf16 dense peak / 6), every family beside it,
        # and the whole step's wall-clock view where the step's FLOP are counted (ResNet generator)
        "roofline": dict({"bound": "mfma", "kernel": dom_name, "achieved": dom["tflops"], "unit": "TFLOP/s", "peak": round(peak, 1), "frac": dom["frac"],
                          "frac_of_sustained_mfma": dom.get("frac_of_sustained_mfma"), "mfma_busy": dom["mfma_busy"], "traffic": dom["traffic"],
                          "launches_per_step": dom["launches_per_step"],
                          "whole_step": ({"achieved": round(fl * imgs / dt / 1e12, 2), "frac": round(fl * imgs / dt / 1e12 / peak, 4),
                                          "scope": "every convolution of G and D, forward and both gradients; wall clock"} if args.netg == "resnet" else None)},
                         **fam_tab),
        "losses": {k: round(v, 5) for k, v in log.items()}}))


if __name__ == "__main__":
    main()
