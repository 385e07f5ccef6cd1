"""tests/golden/train_sr_reference.yml: the reference's shipped ESRGAN recipe (codes/options/sr/train_sr.yml) as a fixture.

TEST INFRASTRUCTURE ONLY (build container: reads /root/reference).  Usage:  python -m oracle.make_golden_options

The file is copied TEXTUALLY -- comments included -- with only its filesystem locations replaced by the
placeholder @ROOT@ (dataset folders, `path.root`, `pretrain_model_G`), which the test substitutes with a temporary
directory it populates (a seeded RRDB_PSNR_x4.pth in the reference's checkpoint format).  Every other key -- network_G: esrgan
(gaussian noise on by default), use_amp: true, metrics: 'psnr,ssim,lpips', the *_rel schedules -- is what the reference ships:
the drop-in contract says the engine must accept train_sr.yml-shaped files.
"""
import os
import re

from .ref_harness import REF_CODES

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "train_sr_reference.yml")


def main():
    src = os.path.join(REF_CODES, "options", "sr", "train_sr.yml")
    txt = open(src).read()
    out, n = re.subn(r"'\.\./datasets/", "'@ROOT@/datasets/", txt)
    out, n2 = re.subn(r"root: '\.\./'", "root: '@ROOT@/'", out)
    out, n3 = re.subn(r"'\.\./experiments/", "'@ROOT@/experiments/", out)
    assert n >= 7 and n2 == 1 and n3 >= 1, (n, n2, n3)
    changed = sum(a != b for a, b in zip(txt.splitlines(), out.splitlines()))
    with open(OUT, "w") as f:
        f.write(out)
    print(OUT, "%d lines, %d with a location replaced" % (len(out.splitlines()), changed))


if __name__ == "__main__":
    main()
