"""Build an experimental variant of libtrainner_hip.so with extra -D flags:
    python tools/build_variant.py NAME -DTNR_STAGGER_SLEEPS=0 ...   ->  trainner_amd/lib/variants/libNAME.so
Select it at run time with TNR_HIP_LIB=<path>."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trainner_amd import build as B  # noqa: E402


def main(name, *defs):
    # --only a.hip,b.hip : recompile just these sources with the flags and link them with the SHIPPED build's other objects
    only = None
    defs = list(defs)
    if "--only" in defs:
        k = defs.index("--only")
        only = defs[k + 1].split(",")
        del defs[k:k + 2]
    out_dir = os.path.join(B.LIBDIR, "variants")
    obj_dir = os.path.join(out_dir, "obj_" + name)
    os.makedirs(obj_dir, exist_ok=True)
    cc = B._hipcc()

    def one(src):
        if only is not None and src not in only:
            return os.path.join(B.LIBDIR, "obj", src.replace(".hip", ".o"))
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        r = subprocess.run([cc, *B.FLAGS, *defs, "-c", os.path.join(B.CSRC, src), "-o", obj], capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(r.stderr[-3000:])
        return obj

    with ThreadPoolExecutor(max_workers=len(B.SOURCES)) as ex:
        objs = list(ex.map(one, B.SOURCES))
    lib = os.path.join(out_dir, "lib%s.so" % name)
    subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs], check=True)
    print(lib)


if __name__ == "__main__":
    main(*sys.argv[1:])
