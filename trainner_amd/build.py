"""Build libtrainner_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m trainner_amd.build           # rebuild if any source is newer than the .so
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libtrainner_hip.so")
SOURCES = ["conv_tile.hip", "conv_chain.hip", "conv_sweep.hip", "conv_wino.hip", "conv_thin.hip", "wgrad_tile.hip", "wgrad_thin.hip", "pack_api.hip", "elementwise.hip", "norm_loss_optim.hip", "metrics.hip", "feed.hip", "degrade.hip", "dp_api.hip", "gconv.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    import glob
    deps = [os.path.join(CSRC, s) for s in SOURCES] + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + \
        [os.path.join(HERE, "..", "include", "trainner_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def source_hash():
    """sha256 over the kernel sources (csrc/*, include/trainner_hip.h) -- stamped into the PMC records under profiles/ so that
    bench.py can tell whether a recorded counter value belongs to THIS tree's kernels."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(CSRC, "*"))) + [os.path.join(HERE, "..", "include", "trainner_hip.h")]:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    cc = _hipcc()

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [cc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    r = subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
    if verbose:
        print("built", LIB, "(%d KB)" % (os.path.getsize(LIB) // 1024))
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
