"""Same-box A/B of environment switches: runs `bench.py --no-roofline --no-cpu-baseline --no-variant` once per setting, interleaved and
repeated, and prints ms/step per run (boxes differ by +-3 %, so only runs of ONE gpurun call compare).
    python tools/ab_env.py REPS "NAME=VAL,NAME2=VAL2" "NAME=VAL" ...      ("" = the defaults)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
reps, settings = int(sys.argv[1]), sys.argv[2:]
extra = os.environ.get("AB_BENCH_ARGS", "").split()
res = {s: [] for s in settings}
for r in range(reps):
    for s in settings:
        env = dict(os.environ)
        for kv in [x for x in s.split(",") if x]:
            k, v = kv.split("=", 1)
            env[k] = v
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-roofline", "--no-cpu-baseline", "--no-variant", "--steps", "10",
                              "--warmup", "3"] + extra, env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print("FAILED", s, out.stderr[-800:])
            continue
        res[s].append(json.loads(line[-1])["ms_per_step"])
        print("%-40s %8.2f ms/step" % (s or "(defaults)", res[s][-1]), flush=True)
for s in settings:
    if res[s]:
        print("MEAN %-40s %8.2f ms/step over %d runs" % (s or "(defaults)", sum(res[s]) / len(res[s]), len(res[s])))
