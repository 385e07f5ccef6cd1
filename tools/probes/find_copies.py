"""Where do the device-to-device copies of a training step come from?  One profiled step (torch.profiler with stacks), copy ops grouped by
their Python call site.    python tools/probes/find_copies.py"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402

model = B.make_model(16, 512, 0)
LR, HR = B.synthetic(16, 512, 1000, torch.device("cuda", 0))
data = {"LR": LR, "HR": HR}
for s in range(1, 3):
    model.feed_data(data)
    model.optimize_parameters(s)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    model.feed_data(data)
    model.optimize_parameters(3)
    torch.cuda.synchronize()
sites = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::add_", "aten::mul", "aten::add", "aten::sum", "aten::div", "aten::mul_"):
        st = [f for f in (ev.stack or []) if "trainner_amd" in f]
        sites[(ev.name, st[0] if st else "?")] += 1
par = collections.Counter()
for ev in prof.events():
    if "Memcpy" in ev.name or "memcpy" in ev.name:
        q, chain = ev.cpu_parent, []
        while q is not None and len(chain) < 4:
            chain.append(q.name[:40])
            q = q.cpu_parent
        par[(ev.name, " < ".join(chain))] += 1
for k, n in par.most_common(25):
    print("%5d  %s" % (n, k))
names = collections.Counter(ev.name for ev in prof.events())
for name, n in names.most_common(60):
    print("%5d  %s" % (n, name[:100]))
for ev in prof.events():
    if "copyBuffer" in ev.name or "Memcpy" in ev.name or "memcpy" in ev.name:
        print("COPY", ev.name[:60], ev.device_time, [f for f in (ev.stack or [])][:3])
        break
for (name, site), n in sites.most_common(40):
    print("%4d  %-18s %s" % (n, name, site))
