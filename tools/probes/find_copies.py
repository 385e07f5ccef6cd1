"""Does a step issue small device copies?  rocprofv3's kernel table of a short bench run lists ~1 300 `__amd_rocclr_copyBuffer` launches:
this probe puts ONE warmed-up G+D step of BASELINE configs[1] under torch.profiler and counts memcpy / memset / tiny aten events.
Finding (profiles/r13c_copies.txt): 2 hipMemcpyAsync per step (the two weight packers' item tables) -- the 1 300 belong to model
construction (799 parameter tensors moved to the device once), not to the step.    python tools/probes/find_copies.py"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    model = bench.make_model(16, 512, 0)
    pool = [bench.synthetic(16, 512, 100 + i, dev) for i in range(2)]
    for s in range(1, 4):
        lr, hr = pool[s % 2]
        model.feed_data({"LR": lr, "HR": hr})
        model.optimize_parameters(s)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        lr, hr = pool[0]
        model.feed_data({"LR": lr, "HR": hr})
        model.optimize_parameters(4)
        torch.cuda.synchronize()
    by = collections.Counter()
    names = collections.Counter()
    for ev in prof.events():
        nm = ev.name
        if not any(k in nm.lower() for k in ("memcpy", "copy_", "memset", "aten::to", "aten::fill_", "aten::zero_", "aten::add", "aten::mul", "aten::clone", "aten::contiguous", "aten::item", "aten::_local_scalar_dense")):
            continue
        if not nm.startswith("aten::") and "mem" not in nm.lower():
            continue
        names[nm] += 1
        frame = "?"
        for fr in (ev.stack or []):
            if "trainner_amd" in fr or "bench.py" in fr:
                frame = fr.strip()
                break
        by[(nm, frame)] += 1
    print("event names:", names.most_common(20))
    for (nm, frame), c in by.most_common(60):
        print("%5d  %-28s %s" % (c, nm, frame[-150:]))


if __name__ == "__main__":
    main()
