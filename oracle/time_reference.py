"""Time the REFERENCE's own SRModel.feed_data + optimize_parameters on this host's CPU cores:
ESRGAN RRDBNet-23 + Discriminator_VGG(512) + VGG19, 128 -> 512, N = 2, fp32, 1 warm-up + >= 3 timed steps.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (build container: needs /root/reference).  Usage:
    python -m oracle.time_reference [--batch 2] [--steps 3] [--out profiles/r02_cpu_reference.json]
bench.py carries the committed result as cpu_baseline.reference beside the on-box `port` figure (the GPU box has
no /root/reference, so the reference itself cannot be timed there).
"""
import argparse
import json
import os
import platform
import time

import torch

from . import detrand
from . import ref_harness as R


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--crop", type=int, default=512)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    yml = R.esrgan_yaml(name="time_reference", nb=23, batch=a.batch, crop=a.crop, d_nf=64)
    opt, model = R.build_reference_model(yml, seed=0)
    LR, HR = detrand.synthetic_pair(a.batch, a.crop, 7)
    R.reference_step(model, LR, HR, 1)                   # warm-up
    times = []
    for s in range(2, 2 + a.steps):
        t0 = time.time()
        log = R.reference_step(model, LR, HR, s)
        times.append(time.time() - t0)
    dt = sum(times) / len(times)
    out = {"value": round(a.batch / dt, 4), "unit": "HR img/s", "cores": cores, "kind": "reference",
           "s_per_step": round(dt, 3), "step_times_s": [round(t, 3) for t in times],
           "sample": "the reference's own SRModel.feed_data + optimize_parameters (codes/models/sr_model.py:115-128,195-267) "
                     "on CPU, ESRGAN RRDBNet-23 + Discriminator_VGG(%d) + VGG19, batch %d, %d->%d, fp32, 1 warm-up + %d timed steps"
                     % (a.crop, a.batch, a.crop // 4, a.crop, a.steps),
           "host": "build container (%s, %d cores), torch %s" % (platform.processor() or platform.machine(), cores, torch.__version__),
           "last_log": {k: round(float(v), 6) for k, v in log.items()}}
    print(json.dumps(out))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
