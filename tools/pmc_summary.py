"""Aggregate a rocprofv3 --pmc counter_collection CSV by kernel: mean counter value per dispatch.
usage: python tools/pmc_summary.py <dir with *counter_collection.csv> > summary.csv"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(d):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                name = r.get("Kernel_Name") or r.get("Kernel Name") or "?"
                name = name.replace("(anonymous namespace)::", "").replace("void ", "")[:70]
                c = r.get("Counter_Name") or r.get("Counter Name")
                v = float(r.get("Counter_Value") or r.get("Counter Value") or 0)
                a = agg[name][c]
                a[0] += v
                a[1] += 1
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch", "total"])
    for k in sorted(agg, key=lambda k: -max(x[0] for x in agg[k].values())):
        for c, (tot, n) in sorted(agg[k].items()):
            w.writerow([k, c, n, "%.4g" % (tot / max(n, 1)), "%.6g" % tot])


if __name__ == "__main__":
    main(sys.argv[1])
