"""Soak run of the bench workload (tools/gpu.sh run <tag> soak python tools/probes/soak.py [--steps N] [--amp] [--gauss]):
N consecutive G+D steps of BASELINE configs[1] on fresh synthetic batches, with the train loop's per-step calls around them
(update_learning_rate, a periodic log read, a checkpoint + state save half way), reporting per-window step time, device memory
(allocated / reserved / the driver's free figure) and the loss ranges.  What it is for: a leak (memory creeping per step), a
slowdown over time (clock / fragmentation), a NaN that only shows after the first tens of Adam steps."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--window", type=int, default=25)
    ap.add_argument("--amp", action="store_true")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--crop", type=int, default=512)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    model = bench.make_model(a.batch, a.crop, 0, amp=a.amp)
    pool = [bench.synthetic(a.batch, a.crop, 100 + i, dev) for i in range(4)]
    logs, rows = [], []
    torch.cuda.synchronize()
    t0 = time.time()
    for s in range(1, a.steps + 1):
        lr, hr = pool[s % len(pool)]
        model.feed_data({"LR": lr, "HR": hr})
        model.optimize_parameters(s)
        model.update_learning_rate(s, warmup_iter=-1)
        if s % 10 == 0:
            logs.append(dict(model.get_current_log()))
        if s == a.steps // 2:
            model.save(s)
            model.save_training_state(0, s)
        if s % a.window == 0:
            torch.cuda.synchronize()
            t1 = time.time()
            free, total = torch.cuda.mem_get_info()
            rows.append((s, (t1 - t0) / a.window * 1e3, torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30,
                         (total - free) / 2**30))
            t0 = time.time()
    print("steps  ms/step  allocated GiB  reserved GiB  device-used GiB")
    for r in rows:
        print("%5d  %7.2f  %13.3f  %12.3f  %15.3f" % r)
    keys = list(logs[0])
    for k in keys:
        v = [l[k] for l in logs]
        assert all(x == x and abs(x) < 1e6 for x in v), (k, v)
        print("%-14s first %.6g  last %.6g  min %.6g  max %.6g" % (k, v[0], v[-1], min(v), max(v)))
    growth = rows[-1][4] - rows[1][4] if len(rows) > 2 else 0.0
    slow = rows[-1][1] / min(r[1] for r in rows[1:]) if len(rows) > 2 else 1.0
    print("device memory growth after the first window: %.3f GiB; last window / fastest window: %.3f" % (growth, slow))
    assert growth < 0.25, "device memory keeps growing"
    print("soak ok")


if __name__ == "__main__":
    main()
