"""Stamp PMC summaries (tools/pmc_traffic.py / pmc_busy.py output) with the hash of the kernel sources they were measured on
(trainner_amd.build.source_hash): bench.py quotes a recorded counter only when the stamp matches the tree it runs.
usage: python tools/pmc_stamp.py FILE.json [...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trainner_amd.build import source_hash  # noqa: E402

for p in sys.argv[1:]:
    d = json.load(open(p))
    d["kernel_source_hash"] = source_hash()
    json.dump(d, open(p, "w"), indent=1)
