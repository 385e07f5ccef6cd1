"""bench.py -- HR images/sec of the full G+D step (SRModel.optimize_parameters), ESRGAN x4 128->512.

    python bench.py --gpus N --steps K --warmup W
N > 1: one rank per GPU, RCCL gradient all-reduce.  Under `torch.distributed.run` (RANK / WORLD_SIZE set) each
process is one rank; a plain `python bench.py --gpus N` re-launches itself under torch.distributed.run on 127.0.0.1.
`--dry-run-cpu` exercises the same launch + data-parallel host logic on CPU (gloo, the test suite's torch stand-in
for the C ABI, tiny shapes): a plumbing check whose JSON line is marked invalid -- never a measurement.

Workload = BASELINE.json configs[1]: RRDBNet-23 + Discriminator_VGG(512) + VGG19->conv5_4, batch 16
per GPU (weak scaling), L1 + perceptual + relativistic GAN, clip + Adam -- fp32 arithmetic on the matrix cores: by default the
split-operand form (`--mma bf16x3`: every fp32 operand split exactly into three bf16 values, six of the nine exact partial products on
the bf16 matrix core, fp32 accumulate; error against fp64 comparable to the fp32 matrix-core instruction's: test bound 1.5 x + 2e-7 scale), with the same step on
v_mfma_f32_32x32x2_f32 measured in the same process as `variant_f32_mfma`.
Synthetic HR in [0,1), LR = avg_pool(HR, 4), resident in HBM before the timed region; random-init
(kaiming x0.1) G/D and seeded VGG weights (no network access).  One JSON line on rank 0, with
  roofline     : the dominant kernel family (3x3 implicit-GEMM, forward + data-gradient launches),
                 algorithmic FLOP of every launch / HIP-event time of that launch, both summed over a
                 separate instrumented pass of the same steps (events on the launch stream);
  cpu_baseline : the CPU oracle (a port of the reference step) on this host's cores, by BASELINE.md section 2's protocol (batch 2,
                 1 warm-up + 3 timed steps, same shapes; N=1 runs only);
  cpu_reference: the reference's OWN SRModel timed by the same protocol in the build container (profiles/*_cpu_reference.json,
                 oracle/time_reference.py) -- the GPU box has no /root/reference.
The GPU result is also written to stderr as `BENCH_GPU_LINE {...}` before the CPU leg starts, so a timeout in the
CPU leg cannot lose it; stdout carries exactly one JSON line.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E ~8 TB/s
PEAK_BF16_MFMA_TFLOPS = 2516.6     # v_mfma_f32_32x32x16_bf16, dense: 256 CU x 4 SIMD x 32768 FLOP / 32 cycles x 2.4 GHz
# fp32 arithmetic on the bf16 matrix core (bf16x3): six v_mfma_f32_32x32x16_bf16 (6 x 32 cycles) do the work of eight
# v_mfma_f32_32x32x2_f32 (8 x 64 cycles) per 32 x 32 x 16 block -> fp32-equivalent ceiling = bf16 peak / 6 (2.67 x the fp32 pipe)
PEAK_BF16X3_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0
FLOP_PER_IMG = 3.0327e12           # full optimize_parameters, fp32, 128->512 (the reference's schedule)
D_FWD_FLOP_PER_IMG = 7.34e10       # one Discriminator_VGG(512) forward (the D rows / 4 calls)
# --netd unet (BASELINE configs[3]'s discriminator, discriminators.py:686-779, nf 64 at 512 x 512): forward MACs per image, layer by layer
# conv0 3->64 3x3 @512: 0.453 G; conv1..3 4x4 s2 (64->128 @256, 128->256 @128, 256->512 @64): 3 x 8.59 G; conv4..6 3x3 after bilinear x2
# (512->256 @128, 256->128 @256, 128->64 @512): 3 x 19.33 G; conv7, conv8 64->64 @512: 2 x 9.66 G; conv9 64->1 @512: 0.151 G  = 103.7 GMAC
UNET_D_FWD_FLOP_PER_IMG = 2.074e11
# the step with it: everything but the discriminator (3.0327e12 - 9 x 7.34e10: four forwards, the G stage's data-gradient, two full backward
# passes of the VGG-style D) + the same nine forward-equivalents of the U-Net D
FLOP_PER_IMG_UNET_D = FLOP_PER_IMG - 9 * D_FWD_FLOP_PER_IMG + 9 * UNET_D_FWD_FLOP_PER_IMG
BATCH_PER_GPU = 16
CROP = 512
MMA_TEXT = {
    "bf16x3": "fp32 arithmetic on the bf16 matrix core: every fp32 operand split exactly into 3 bf16 values (hi + mid + lo = x), 6 of the 9 "
              "exact partial products in v_mfma_f32_32x32x16_bf16, fp32 accumulate (convolutions, dense-block sweeps, data- and "
              "weight-gradients); the three dropped products cost ~2^-23 per product: error vs fp64 comparable to the fp32 matrix-core instruction's "
              "(held to <= 1.5 x that + 2e-7 x scale in tests/test_gpu_kernels.py; every reference golden is met with the fp32 bounds)",
    "f32": "fp32 matrix core (v_mfma_f32_32x32x2_f32)",
}

YAML = """
name: bench_esrgan
use_tb_logger: false
model: sr
scale: 4
gpu_ids: {gpu_ids}
use_amp: {amp}
datasets:
  train:
    name: synthetic
    mode: aligned
    dataroot_HR: /tmp/none_hr
    dataroot_LR: /tmp/none_lr
    znorm: false
    n_workers: 0
    batch_size: {batch}
    virtual_batch_size: {batch}
    preprocess: crop
    crop_size: {crop}
path:
  root: {root}
network_G:
  type: esrgan
  gaussian: false
network_D: {netd}
train:
  optim_G: adam
  optim_D: adam
  lr_scheme: MultiStepLR
  lr_steps_rel: [0.1, 0.2, 0.4, 0.6]
  lr_gamma: 0.5
  pixel_criterion: l1
  pixel_weight: 1e-2
  feature_criterion: l1
  feature_weight: 1
  perceptual_allow_random_init: true   # seeded VGG weights are loaded below (no network access for ImageNet weights)
  gan_type: vanilla
  gan_weight: 5e-3
  manual_seed: 0
  niter: 5e5
  val_freq: 5e3
  grad_clip: norm
  grad_clip_value: 0.1
logger:
  print_freq: 200
  save_checkpoint_freq: 5e3
"""


def make_model(batch, crop, rank, world=1, netd="discriminator_vgg", amp=False):
    """batch = per-GPU batch; the YAML carries the reference's GLOBAL batch_size (options/README.md:31)."""
    from trainner_amd.models import create_model
    from trainner_amd.options import options
    root = tempfile.mkdtemp(prefix="tnr_bench_r%d_" % rank)
    path = os.path.join(root, "bench.yml")
    with open(path, "w") as f:
        f.write(YAML.format(batch=batch * world, crop=crop, root=root, gpu_ids=list(range(world)), netd=netd,
                            amp="true" if amp else "false"))
    torch.manual_seed(1234 + rank)                # replicas are made identical by SRModel.sync_replicas (rank 0 wins)
    opt = options.parse(path, is_train=True)
    model = create_model(opt, verbose=False)
    # VGG19: seeded He-normal weights (ImageNet weights cannot be downloaded here)
    netF = [l["function"].network for l in model.generatorlosses.loss_list if "fea" in l["name"]][0]
    g = torch.Generator().manual_seed(1234)
    sd = netF.state_dict()
    for k, v in sd.items():
        if k.endswith("weight") and v.dim() == 4:
            fan_in = v.shape[1] * 9
            sd[k] = torch.randn(v.shape, generator=g) * (2.0 / fan_in) ** 0.5
        elif k.endswith("bias"):
            sd[k] = torch.zeros_like(v)
    netF.load_state_dict(sd)
    return model


class HostBatches:
    """--feed: an endless supply of HOST batches in the engine's wire format (uint8 HWC BGR crop windows + paired-transform
    flags), cycling over a small pre-generated pool -- what a DataLoader over AlignedWindowDataset yields."""

    def __init__(self, n_batches, batch, crop, seed, paired):
        import numpy as np
        rs = np.random.RandomState(seed)
        self.pool = []
        for _ in range(4):
            hr = rs.randint(0, 256, (batch, crop, crop, 3), dtype=np.uint8)
            b = {"HR": torch.from_numpy(hr).pin_memory(), "flags": torch.from_numpy(rs.randint(0, 2, batch).astype("int32") * 1)}
            if paired:
                b["LR"] = torch.from_numpy(np.ascontiguousarray(hr[:, ::4, ::4])).pin_memory()
            self.pool.append(b)
        self.n = n_batches

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            yield self.pool[i % len(self.pool)]


def synthetic(batch, crop, seed, device):
    g = torch.Generator().manual_seed(seed)
    hr = torch.rand(batch, 3, crop, crop, generator=g)
    lr = torch.nn.functional.avg_pool2d(hr, 4)
    return lr.to(device), hr.to(device)


def cpu_reference_record():
    """The reference's own SRModel timed on CPU (N = 2, 1 warm-up + >= 3 steps) in the build container: newest
    committed profiles/*_cpu_reference.json, or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_cpu_reference.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as fh:
            rec = json.load(fh)
        rec["source"] = os.path.relpath(files[-1], ROOT)
        rec.pop("last_log", None)
        rec["note"] = ("recorded on the BUILD CONTAINER's cores (the reference tree does not travel to the GPU box): a different host from "
                       "`cpu_baseline` (the oracle port on this box's cores) -- two reported baselines, not operands of a ratio")
        return rec
    except (OSError, ValueError):
        return None


def cpu_baseline(crop, steps=3, batch=2):
    """The reference step as ported in oracle/sr_oracle.py, timed on this host's cores by BASELINE.md section 2's protocol:
    batch 2 (reported per image), 1 warm-up + >= 3 timed steps."""
    from oracle import detrand, sr_oracle as O
    from trainner_amd.models.modules.architectures import RRDBNet_arch, discriminators
    cores = min(os.cpu_count() or 1, 64)       # more threads than this only slow torch's CPU convs down
    torch.set_num_threads(cores)
    g = {k: v.detach().clone() for k, v in RRDBNet_arch.RRDBNet(3, 3, 64, 23).state_dict().items()}
    d = {k: v.detach().clone() for k, v in discriminators.Discriminator_VGG(crop, 3, 64).state_dict().items()}
    detrand.fill_state_dict_(g, 101, gain=0.1)
    detrand.fill_state_dict_(d, 202, gain=0.1)
    orc = O.OracleSRStep(g, d, O.vgg19_seeded_state(), arch="rrdb_net", nb=23, d_size=crop, d_nf=64)
    LR, HR = detrand.synthetic_pair(batch, crop, 7)
    orc.step(LR, HR)                                # warm-up
    times = []
    for _ in range(steps):
        t0 = time.time()
        orc.step(LR, HR)
        times.append(time.time() - t0)
    dt = sum(times) / steps
    return {"value": round(batch / dt, 4), "unit": "HR img/s", "cores": cores, "kind": "port",
            "s_per_step": round(dt, 3), "step_times_s": [round(t, 3) for t in times],
            "sample": "oracle/sr_oracle.py OracleSRStep (port of SRModel.optimize_parameters), ESRGAN RRDBNet-23 + "
                      "D_VGG(%d) + VGG19, batch %d, %d->%d, fp32, 1 warm-up + %d timed steps (BASELINE.md section 2)"
                      % (crop, batch, crop // 4, crop, steps)}


def pmc_traffic(family="conv_tile_3x3", mode=None):
    """HBM bytes per launch of the dominant kernel family (average over its launches) from the newest committed
    rocprofv3 --pmc summary (profiles/*_pmc_traffic.json, made by tools/pmc_traffic.py from separate
    FETCH_SIZE / WRITE_SIZE passes of this same command).  Counters cannot be read from inside the timed
    process, so this is the recorded figure for the same kernels and shapes; None if no summary exists."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic_%s.json" % mode))) if mode else []
    if not files and mode in (None, "f32"):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))      # (round-2 naming: fp32 matrix core)
    if not files:
        return None, None
    try:
        with open(files[-1]) as fh:
            rec = json.load(fh)
        if not _pmc_record_current(rec):
            return None, "%s was recorded on other kernel sources (stamp %s, this tree %s): not quoted" % (
                os.path.relpath(files[-1], ROOT), rec.get("kernel_source_hash"), _source_hash())
        e = rec[family]
        src = "%s (rocprofv3 --pmc passes of this command on these kernel sources, stamp %s)" % (os.path.relpath(files[-1], ROOT), rec["kernel_source_hash"])
        return (round(e["hbm_bytes_per_launch"]) if e.get("dispatches") else None), src
    except (OSError, KeyError, ValueError):
        return None, None


def pmc_family_records(mode):
    """(mfma_busy record, traffic record) of this matrix-core mode -- the newest committed profiles/*_pmc_{mfma_busy,traffic}_<mode>.json
    whose kernel-source stamp is this tree's (else None): per-family entries keyed like ops.ConvProfile's families (tools/pmc_families.py)."""
    import glob
    out = []
    for kind in ("mfma_busy", "traffic"):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_%s_%s.json" % (kind, mode))))
        rec = None
        if files:
            try:
                with open(files[-1]) as fh:
                    rec = json.load(fh)
                if not _pmc_record_current(rec):
                    rec = None
                else:
                    rec["_file"] = os.path.relpath(files[-1], ROOT)
            except (OSError, ValueError):
                rec = None
        out.append(rec)
    return out


def family_table(summ, nprof, peak, mode, sustained=None):
    """roofline.families: EVERY matrix-core kernel family of the step, so that the whole MFMA share of the step can be re-derived from
    this line alone: HIP-event time and algorithmic FLOP of its launches (ops.ConvProfile, the instrumented pass), TFLOP/s, fraction of
    the mode's dense peak (and of the sustained ceiling where one is measured), and -- from the committed PMC passes of this command on
    these kernel sources -- MFMA-busy and HBM bytes per launch (None when no current record exists)."""
    busy, traffic = pmc_family_records(mode)
    fams = {}
    for name, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
        if v["ms"] <= 0 or v["flops"] <= 0:
            continue
        tf = v["flops"] / (v["ms"] * 1e-3) / 1e12
        vector_alu = name in ("conv_thin", "wgrad_thin")          # 3-channel sides: no matrix instruction in these kernels
        e = {"ms_per_step": round(v["ms"] / nprof, 3), "launches_per_step": v["launches"] // nprof,
             "flop_per_step": v["flops"] / nprof, "tflops": round(tf, 2),
             "frac": None if vector_alu else round(tf / peak, 4),
             "mfma_busy": None if (busy is None or name not in busy) else busy[name]["mfma_busy"],
             "traffic": None if (traffic is None or name not in traffic) else round(traffic[name]["hbm_bytes_per_launch"])}
        if sustained and not vector_alu:
            e["frac_of_sustained_mfma"] = round(tf / sustained, 4)
        if vector_alu:
            e["note"] = "vector-ALU kernel (<= 4 channels on one side): HBM-bound, no MFMA roofline"
        fams[name] = e
    return {"families": fams, "families_mfma_ms_per_step": round(sum(e["ms_per_step"] for e in fams.values()), 2),
            "families_counters_source": {"mfma_busy": busy and busy["_file"], "traffic": traffic and traffic["_file"]}}


def _source_hash():
    from trainner_amd.build import source_hash
    return source_hash()


def _pmc_record_current(rec):
    """A recorded counter is quoted only if it was measured on THIS tree's kernels (tools/pmc_stamp.py writes the hash of
    trainner_amd/csrc + include/trainner_hip.h into the record; records without a stamp predate it)."""
    return rec.get("kernel_source_hash") == _source_hash()


def pmc_mfma_busy(family, mode):
    """SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCD x 1024 SIMD) of the dominant kernel family from the newest committed
    rocprofv3 --pmc summary of this command in this matrix-core mode (profiles/*_pmc_mfma_busy_<mode>.json, tools/pmc_busy.py), or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_mfma_busy_%s.json" % mode)))
    if not files:
        return None
    try:
        with open(files[-1]) as fh:
            rec = json.load(fh)
        if not _pmc_record_current(rec):
            return None
        e = rec.get(family)
        return None if e is None else {"value": e["mfma_busy"], "source": os.path.relpath(files[-1], ROOT), "kernel_source_hash": rec["kernel_source_hash"]}
    except (OSError, KeyError, ValueError):
        return None


def driver_visible_variants(args, model, data, device, rank, barrier, step):
    """variant_amp / variant_feed_paired / variant_config4 / variant_config5 of the default line: {value, ms_per_step, steps, warmup, dominant | whole_step}."""
    from trainner_amd import ops
    from trainner_amd.data.feeder import DeviceFeeder
    n_feed = 2 + args.steps + 1

    def measure(mdl, next_batch, amp):
        nonlocal step
        for _ in range(2):
            step += 1
            mdl.feed_data(next_batch())
            mdl.optimize_parameters(step)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step += 1
            mdl.feed_data(next_batch())
            mdl.optimize_parameters(step)
        barrier()
        dtv = time.perf_counter() - t0
        prof = ops.ConvProfile()
        ops.PROFILE = prof
        step += 1
        mdl.feed_data(next_batch())
        mdl.optimize_parameters(step)
        ops.PROFILE = None
        summ = {k: v for k, v in prof.summary().items() if k not in ("conv_thin", "wgrad_thin") and v["ms"] > 0 and v["flops"] > 0}
        if ops.chain_error_flag():
            raise SystemExit("bench: a conv_chain dependency wait timed out -- results are invalid")
        dom = None
        if summ:
            fam = max(summ, key=lambda f: summ[f]["ms"])
            tf = summ[fam]["flops"] / (summ[fam]["ms"] * 1e-3) / 1e12
            peak = PEAK_BF16_MFMA_TFLOPS if amp else PEAK_BF16X3_TFLOPS
            dom = {"family": fam, "ms_per_step": round(summ[fam]["ms"], 3), "launches_per_step": summ[fam]["launches"],
                   "tflops": round(tf, 2), "peak": round(peak, 1), "frac": round(tf / peak, 4)}
        return {"value": round(args.batch * args.steps / dtv, 3), "unit": "HR img/s", "ms_per_step": round(1e3 * dtv / args.steps, 2),
                "steps": args.steps, "warmup": 2, "dominant": dom}

    out = {}
    # (1) use_amp: the same model with the AMP region on (bf16 matrix-core operands, fp32 accumulate and storage)
    model.opt["use_amp"] = True
    model.setup_amp()
    v = measure(model, lambda: data, True)
    v.update(dtype="bf16", what="`use_amp: true` (options/sr/train_sr.yml:6): same model, same batch, AMP region on")
    out["variant_amp"] = v
    model.opt["use_amp"] = False
    model.setup_amp()
    # (2) the input pipeline in the loop: uint8 host batches -> pinned H2D -> device np2tensor / flip / rot (DeviceFeeder)
    feeder = DeviceFeeder(HostBatches(n_feed, args.batch, args.crop, 1000 + rank, True), device=device)
    it = feeder.iterate()
    v = measure(model, lambda: next(it), False)
    v.update(dtype="f32 (bf16x3)", what="--feed paired: uint8 LR + HR host windows through the double-buffered DeviceFeeder",
             h2d_bytes_per_step=feeder.bytes_uploaded // n_feed)
    out["variant_feed_paired"] = v
    feeder.close()
    # (3) BASELINE configs[3] on one GPU: RRDBNet-23 + UNetDiscriminator, LR synthesised by the GPU degradation pipeline
    from trainner_amd.dataops.degradations import RealESRGANDegradation
    del model
    torch.cuda.empty_cache()
    m4 = make_model(args.batch, args.crop, rank, 1, "unet", False)
    feeder = DeviceFeeder(HostBatches(n_feed, args.batch, args.crop, 1000 + rank, False), device=device,
                          degrade=RealESRGANDegradation(scale=4, seed=rank))
    it = feeder.iterate()
    v = measure(m4, lambda: next(it), False)
    v.update(dtype="f32 (bf16x3)", what="--netd unet --feed resrgan (BASELINE configs[3] at 1 GPU): U-Net discriminator, HR windows only, "
                                        "Real-ESRGAN degradations on the GPU")
    out["variant_config4"] = v
    feeder.close()
    # (4) BASELINE configs[4] (image-to-image, 256 x 256): the workloads of tools/bench_i2i.py -- Pix2Pix and CycleGAN with the 9-block
    # ResnetGenerator + PatchGAN, batch 16; value in images / s, the whole step's convolution FLOP against the split arithmetic's ceiling
    del m4
    torch.cuda.empty_cache()
    sys.path.insert(0, ROOT)
    from tools import bench_i2i
    out["variant_config5"] = {}
    for name in ("pix2pix", "cyclegan"):
        mdl, batch_i = bench_i2i.build(name, "resnet", 16, 256, False, device)
        s_i = 0
        for _ in range(2):
            s_i += 1
            mdl.feed_data(batch_i)
            mdl.optimize_parameters(s_i)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            s_i += 1
            mdl.feed_data(batch_i)
            mdl.optimize_parameters(s_i)
        barrier()
        dtv = time.perf_counter() - t0
        if ops.chain_error_flag():
            raise SystemExit("bench: a conv_chain dependency wait timed out -- results are invalid")
        tf = bench_i2i.step_flop(name, 256) * 16 * args.steps / dtv / 1e12
        out["variant_config5"][name] = {"value": round(16 * args.steps / dtv, 2), "unit": "img/s", "ms_per_step": round(1e3 * dtv / args.steps, 2),
                                        "steps": args.steps, "warmup": 2, "dtype": "f32 (bf16x3)",
                                        "whole_step": {"tflops": round(tf, 2), "peak": round(PEAK_BF16X3_TFLOPS, 1), "frac": round(tf / PEAK_BF16X3_TFLOPS, 4),
                                                       "scope": "every convolution of G and D, forward and both gradients; wall clock"},
                                        "what": "%s, ResnetGenerator-9 (ngf 64, InstanceNorm) + PatchGAN, batch 16, 256 x 256 (tools/bench_i2i.py)" % name}
        del mdl, batch_i
        torch.cuda.empty_cache()
    return out


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="per-GPU batch (BASELINE configs[1]: 16)")
    ap.add_argument("--crop", type=int, default=CROP)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--detail", action="store_true", help="per-shape kernel table on stderr")
    ap.add_argument("--feed", choices=["none", "paired", "resrgan"], default="none",
                    help="variant with the input pipeline in the loop: uint8 host batches through the double-buffered DeviceFeeder "
                         "(paired: LR + HR windows; resrgan: HR windows only, LR synthesised by the GPU degradation pipeline)")
    ap.add_argument("--netd", choices=["discriminator_vgg", "unet"], default="discriminator_vgg")
    ap.add_argument("--amp", action="store_true",
                    help="variant: `use_amp: true` = bf16 matrix-core operands, fp32 accumulate (reported as dtype bf16 with its own "
                         "roofline; the headline run is fp32)")
    ap.add_argument("--mma", choices=["f32", "bf16x3"], default=os.environ.get("TNR_MMA", "bf16x3"),
                    help="fp32 arithmetic of the matrix-core kernels: bf16x3 (default, headline) = operands split exactly into three bf16 "
                         "values, six partial products on the bf16 matrix core, fp32 accumulate; f32 = v_mfma_f32_32x32x2_f32")
    ap.add_argument("--no-variant", action="store_true", help="skip the second measurement (the other fp32 arithmetic)")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="plumbing check on CPU (gloo + tests/emul_backend.py, tiny shapes); the JSON line is marked invalid")
    args = ap.parse_args()
    # dmabuf IPC (RCCL across processes on this driver).  Set HERE, not only in self_launch(): the driver's scaling run starts the ranks
    # itself (`python -m torch.distributed.run ... bench.py --gpus N`) and never passes through self_launch
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("TNR_DP_DEVICE", os.environ.get("LOCAL_RANK", "0")))      # TNR_DP_DEVICE (+ TNR_DP_PG=gloo): the ranks share one device -- a plumbing run
    shared_gpu = world > 1 and "TNR_DP_DEVICE" in os.environ
    if args.gpus != world:
        raise SystemExit("bench: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    os.environ["TNR_MMA"] = args.mma          # read by trainner_amd.ops at import
    from trainner_amd import hip, ops
    dry = args.dry_run_cpu
    if dry:
        # TEST INFRASTRUCTURE: the C ABI replaced by its torch-CPU contract (no GPU in the build container)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import emul_backend
        emul_backend.install()
        os.environ["LOCAL_RANK"] = "0"
        args.crop, args.batch, args.no_roofline, args.no_cpu_baseline = 64, 2, True, True
        device = torch.device("cpu")
        torch.set_num_threads(max(1, min(4, (os.cpu_count() or 1) // world)))
    else:
        hip.require_device()
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)

    model = make_model(args.batch, args.crop, rank, world, args.netd, args.amp)
    assert model.dp.world_size == world and (world == 1 or model.dp.active)
    world_observed = model.dp.observed_world_size()      # a collective when world > 1: EVERY rank calls it (rank 0 prints it)
    LR, HR = synthetic(args.batch, args.crop, 1000 + rank, device)      # this rank's shard of the global batch
    data = {"LR": LR, "HR": HR}

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        if not dry:
            torch.cuda.synchronize()

    step = 0
    feeder = None
    if args.feed != "none":
        from trainner_amd.data.feeder import DeviceFeeder
        degrade = None
        if args.feed == "resrgan":
            from trainner_amd.dataops.degradations import RealESRGANDegradation
            degrade = RealESRGANDegradation(scale=4, seed=rank)
        feeder = DeviceFeeder(HostBatches(args.warmup + args.steps, args.batch, args.crop, 1000 + rank, args.feed == "paired"),
                              device=device, degrade=degrade)
        batches = feeder.iterate()

    last_batch = [None]

    def next_batch():
        if feeder is None:
            return data
        last_batch[0] = next(batches)
        return last_batch[0]

    for _ in range(args.warmup):
        step += 1
        model.feed_data(next_batch())
        model.optimize_parameters(step)
    barrier()
    if model.dp.active:
        model.comm_events = []          # BaseModel._sync_gradients: event pairs around the gradient exchange's tail (no host sync)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step += 1
        model.feed_data(next_batch())
        model.optimize_parameters(step)
    barrier()
    dt = time.perf_counter() - t0
    comm = None
    if model.dp.active and not dry:
        ev, model.comm_events = model.comm_events, None
        per = {}
        for flag, e0, e1 in ev:
            per[flag] = per.get(flag, 0.0) + e0.elapsed_time(e1)
        comm = {"exposed_ms_per_step": {k: round(v / args.steps, 3) for k, v in per.items()},
                "what": "compute-stream time between the last backward kernel and the first optimiser kernel of a network (bucket flush + wait for "
                        "the all-reduces still in flight on the side stream), rank 0",
                "g_overlapped_with_backward": bool(ops.g_buckets_leave_in_backward()),
                "dense_blocks_one_launch_next_to_collectives": ops.COUNTERS["one_launch_next_to_collectives"],
                "dense_blocks_per_layer_next_to_collectives": ops.COUNTERS["per_layer_next_to_collectives"]}
    per_rank_ms = [round(1e3 * dt / args.steps, 2)]
    if world > 1:
        # every rank's own wall time of the timed region (a straggler shows here; `value` uses the MAX, as the contract says)
        mine = torch.tensor([dt], dtype=torch.float64, device=device)
        allt = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allt, mine)
        per_rank_ms = [round(1e3 * float(x.item()) / args.steps, 2) for x in allt]
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if feeder is not None:
        feeder.close()
    log = model.get_current_log()
    if ops.chain_error_flag():
        raise SystemExit("bench: a conv_chain dependency wait timed out -- results are invalid")

    roof = None
    if not args.no_roofline:
        # separate instrumented pass: HIP events around every implicit-GEMM launch, on the launch stream
        # (with the input pipeline in the loop: on the last batch it delivered -- the kernels and shapes are the same)
        if feeder is not None:
            data = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in last_batch[0].items()}
        prof = ops.ConvProfile()
        ops.PROFILE = prof
        nprof = min(2, args.steps)
        for _ in range(nprof):
            step += 1
            model.feed_data(data)
            model.optimize_parameters(step)
        ops.PROFILE = None
        summ = prof.summary()
        if args.detail and rank == 0:
            rows = sorted(prof.summary(by_shape=True).items(), key=lambda kv: -kv[1]["ms"])
            print("%-22s %5s %5s %5s %4s %7s %9s %8s" % ("family", "Cin", "Cout", "H", "k", "launch", "ms/step", "TFLOP/s"), file=sys.stderr)
            for key, v in rows:
                print("%-22s %5d %5d %5d %4d %7d %9.3f %8.1f" % (key[0], key[1], key[2], key[3], key[4], v["launches"] // nprof,
                                                             v["ms"] / nprof, v["flops"] / (v["ms"] * 1e-3) / 1e12), file=sys.stderr)
        # dominant kernel: the dense-block chain (RRDB trunk, forward + data-gradient) when it is in use,
        # else the per-layer 3x3 kernel family
        fam = max((f for f in ("conv_chain", "conv_tile_3x3") if f in summ), key=lambda f: summ[f]["ms"], default=None)
        dom = summ.get(fam)
        if dom:
            tf = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
            per_step_ms = {k: round(v["ms"] / nprof, 3) for k, v in summ.items()}
            kname = {"conv_chain": "conv_chain_kernel (5 dense-block 3x3 convolutions per launch, forward and data-gradient)",
                     "conv_tile_3x3": "conv_tile_kernel<3x3> (forward + data-gradient launches)"}[fam]
            traffic, traffic_src = pmc_traffic(fam, "bf16x3amp" if args.amp else args.mma)      # (tools/gpu.sh prof <tag> bf16x3 --amp)
            peak = PEAK_BF16_MFMA_TFLOPS if args.amp else (PEAK_BF16X3_TFLOPS if args.mma == "bf16x3" else PEAK_F32_MFMA_TFLOPS)
            if fam == "conv_tile_3x3" and (args.amp or args.mma == "bf16x3") and ops.X3_D4:
                kname = "conv3x3_d4_kernel (64-cout 3x3 layers, weights as a pre-split stream; forward + data-gradient launches) + conv_tile_kernel<3x3> for the rest"
            if fam == "conv_chain" and args.mma == "bf16x3" and not args.amp and ops.CONV_SWEEP:
                kname = "conv_sweep4_kernel (a dense block's 5 convolutions per launch, forward and data-gradient; bf16x3)"
            roof = {"bound": "mfma", "kernel": kname,
                    "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(tf / peak, 4), "peak_note": (
                        "fp32-equivalent: bf16 dense peak 2516.6 / 6 MFMAs per fp32-equivalent block" if peak == PEAK_BF16X3_TFLOPS else
                        "v_mfma_f32_32x32x16_bf16 dense" if args.amp else "v_mfma_f32_32x32x2_f32 dense"),
                    "power_note": ("in this arithmetic the part sits at its power cap (1.33 kW, 1.98-2.03 GHz instead of 2.4: profiles/r04j_power_bench_smi.txt); the same "
                                   "launches on all-zero operands run 16 % (dense-block sweep) to 26 % (3x3 kernel) faster (profiles/r04t_sweep_power.txt, r04w_kernel_power.txt): "
                                   "a register-only loop of v_mfma_f32_32x32x16_bf16 on random bf16 operands sustains 1847 TFLOP/s = 0.73 of the dense peak, i.e. 307.8 fp32-equivalent "
                                   "(profiles/r04y_mfma_power.txt); `peak` is the nominal-clock figure") if (args.mma == "bf16x3" and not args.amp) else None,
                    "frac_of_sustained_mfma": (round(tf / 307.8, 4) if (args.mma == "bf16x3" and not args.amp) else None),
                    "mfma_busy": pmc_mfma_busy(fam, args.mma if not args.amp else "bf16x3amp"),
                    "traffic": traffic, "traffic_source": traffic_src,
                    "launches_per_step": dom["launches"] // nprof,
                    "avg_launch_us": round(1e3 * dom["ms"] / dom["launches"], 2),
                    "flop_per_launch_avg": dom["flops"] / dom["launches"],
                    "kernel_ms_per_step": per_step_ms}
            roof.update(family_table(summ, nprof, peak, "bf16x3amp" if args.amp else args.mma,
                                     307.8 if (args.mma == "bf16x3" and not args.amp) else None))
            # Two of the ~25 boxes met in round 5 ran ONLY the dense-block sweep 1.6 x slower (935-941 vs 587-606 us per launch at this
            # configuration; same MFMA-busy cycles, every other kernel at its usual rate: DESIGN.md 3.2, profiles/r07a_slowbox_*,
            # r08b_slowbox_*).  Say so in the line when this run is one of them, so that the number is read for what it is.
            if fam == "conv_chain" and args.mma == "bf16x3" and not args.amp and args.batch == BATCH_PER_GPU and args.crop == CROP \
                    and roof["avg_launch_us"] > 760.0:
                roof["box_note"] = ("the dense-block sweep ran %.0f us per launch on this box; 587-606 us on 23 of the 25 boxes measured in round 5 "
                                    "(69 img/s there), 935-941 us on the other two, where the kernel's system-coherent tile hand-off path was slow "
                                    "while every other kernel ran at its usual rate (DESIGN.md 3.2)" % roof["avg_launch_us"])
            if args.amp:
                kname = "conv_sweep4_kernel<true, true> (a dense block's 5 convolutions per launch, bf16 operands)" if (fam == "conv_chain" and ops.CONV_SWEEP and ops.AMP_SWEEP) else kname
                roof["kernel"] = kname
                if traffic:
                    # bf16 operands leave the bytes untouched (activations stay fp32 in HBM and LDS).  Both views of the same launch, from
                    # a PMC pass of THIS mode: the one with the larger fraction names the bound (round 3's chain kernel moved 1.92 GB per
                    # launch at 5.9 TB/s -- HBM-bound; the sweep's bf16-operand form moves a third of that and is bound by neither roof:
                    # load latency and epilogues)
                    avg_s = 1e-3 * dom["ms"] / dom["launches"]
                    gbs = traffic / avg_s / 1e9
                    hbm = {"achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4)}
                    if hbm["frac"] > roof["frac"]:
                        roof.update({"bound": "hbm", "mfma_view": {"achieved": roof["achieved"], "peak": peak, "unit": "TFLOP/s", "frac": roof["frac"]}})
                        roof.update(hbm)
                    else:
                        roof["hbm_view"] = hbm

    # second measurement in the same process: the same step in the OTHER fp32 arithmetic (headline bf16x3 -> v_mfma_f32_32x32x2_f32 and
    # vice versa).  (1 GPU only: the N > 1 runs are the scaling measurement and carry nothing extra.)
    variant, variant_key = None, None
    if world == 1 and not args.amp and not args.no_variant and not args.no_roofline and feeder is None and not dry:
        other = "f32" if args.mma == "bf16x3" else "bf16x3"
        code = {"f32": hip.MMA_F32, "bf16x3": hip.MMA_BF16X3}
        ops.MMA = ops.FP32_MMA = code[other]
        for _ in range(2):
            step += 1
            model.feed_data(data)
            model.optimize_parameters(step)
        barrier()
        tv = time.perf_counter()
        for _ in range(args.steps):
            step += 1
            model.feed_data(data)
            model.optimize_parameters(step)
        barrier()
        dtv = time.perf_counter() - tv
        ops.MMA = ops.FP32_MMA = code[args.mma]
        if ops.chain_error_flag():
            raise SystemExit("bench: a conv_chain dependency wait timed out -- results are invalid")
        variant_key = "variant_f32_mfma" if other == "f32" else "variant_bf16x3"
        variant = {"mma": MMA_TEXT[other] + "; same step, same process",
                   "value": round(args.batch * world * args.steps / dtv, 3), "unit": "HR img/s",
                   "ms_per_step": round(1e3 * dtv / args.steps, 2), "steps": args.steps, "warmup": 2,
                   "dtype": "f32" if other == "f32" else "f32 (bf16x3)"}

    # further variants the default 1-GPU line carries, each measured in this process AFTER the headline region
    # (2 warm-up + `steps` timed steps + 1 instrumented step for its dominant kernel family): `use_amp`, the input pipeline in the loop,
    # and BASELINE configs[3] on one GPU (U-Net discriminator + LR synthesised by the GPU degradation pipeline)
    extra = {}
    if world == 1 and not args.amp and not args.no_variant and not args.no_roofline and feeder is None and not dry \
            and args.netd == "discriminator_vgg" and args.mma == "bf16x3":
        extra = driver_visible_variants(args, model, data, device, rank, barrier, step)

    if rank == 0:
        imgs = args.batch * world * args.steps
        memo_any = bool(getattr(getattr(model, "netD", None), "memoize", False))
        memo = memo_any and args.netd == "discriminator_vgg"
        out = {
            "metric": "HR images/sec (G+D step), ESRGAN x4 128->512",
            "value": round(imgs / dt, 3), "unit": "HR img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if args.amp else ("f32 (bf16x3)" if args.mma == "bf16x3" else "f32"),
            "data": "synthetic" if not dry else "DRY-RUN on CPU (emulated C ABI, tiny shapes): NOT a measurement",
            "config": {"workload": "ESRGAN RRDBNet-23 x4 + Discriminator_VGG(%d) + VGG19-conv5_4, batch %d/GPU, %d->%d, "
                                   "L1+perceptual+RaGAN, clip+Adam (BASELINE configs[1])" % (args.crop, args.batch, args.crop // 4, args.crop),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                       "world_size_observed": world_observed,
                       "mma": "bf16 operands (use_amp)" if args.amp else MMA_TEXT[args.mma],
                       "hsa_enable_ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                       "collectives": ("gloo through the host, ranks sharing device %d" % local if shared_gpu else ("rccl" if not dry else "gloo")) if world > 1 else "none",
                       # per-box choice between the one-launch dense block and five per-layer launches (bit-identical; ops.SWEEP_AUTO)
                       # chosen ONCE at model set-up on a scratch block of this shape and agreed on by all ranks (all-reduce MAX); `per_rank`
                       # (N > 1) = every rank's own measurement, so a split vote or a slow box is visible in this one line
                       "dense_block_form": dict(ops.SWEEP_AUTO_STATE),
                       "per_rank_ms_per_step": per_rank_ms,
                       "gradient_exchange": comm},
            # executed work: with the discriminator's repeated forwards memoized (engine.HipNet.memoize: the D-stage forwards over
            # the real / generated batch reuse the generator stage's -- same inputs, same weights, bit-identical results) two of
            # the reference schedule's four D forwards are not recomputed and are NOT counted
            "step_tflops": round(((FLOP_PER_IMG - (2 * D_FWD_FLOP_PER_IMG if memo else 0.0)) if args.netd == "discriminator_vgg" else
                                  (FLOP_PER_IMG_UNET_D - (2 * UNET_D_FWD_FLOP_PER_IMG if memo_any else 0.0))) * (args.crop / 512.0) ** 2 * imgs / dt / 1e12, 2),
            "d_forward_memoized": memo_any,
            "roofline": roof,
            (variant_key or "variant_f32_mfma"): variant,
            "variant_amp": extra.get("variant_amp"),
            "variant_feed_paired": extra.get("variant_feed_paired"),
            "variant_config4": extra.get("variant_config4"),
            "variant_config5": extra.get("variant_config5"),
            "losses": {k: round(v, 6) for k, v in log.items()},
        }
        if feeder is not None:
            out["config"]["workload"] += "; INPUT PIPELINE IN THE LOOP: uint8 HWC host batches -> pinned H2D -> device np2tensor/flip/rot" + (
                " + Real-ESRGAN degradations (LR synthesised on the GPU)" if args.feed == "resrgan" else "") + " (DeviceFeeder, double-buffered)"
            out["feed"] = {"mode": args.feed, "h2d_bytes_per_step": feeder.bytes_uploaded // max(args.warmup + args.steps, 1),
                           "note": "variant run: the headline `value` is measured with inputs resident in HBM (default run)"}
        if args.netd != "discriminator_vgg":
            out["config"]["workload"] = out["config"]["workload"].replace("Discriminator_VGG(%d)" % args.crop, "UNetDiscriminator")
        if dry:
            out["value"] = None
            out["invalid"] = "dry run"
        if shared_gpu:
            out["invalid"] = "ranks share one GPU (TNR_DP_DEVICE): the driver's N > 1 command on real kernels, not a scaling measurement"
        if world == 1 and not args.no_cpu_baseline:
            print("BENCH_GPU_LINE " + json.dumps(out), file=sys.stderr, flush=True)    # safe before the CPU leg
            out["cpu_baseline"] = cpu_baseline(args.crop)
            out["cpu_reference"] = cpu_reference_record()
        print(json.dumps(out), flush=True)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
