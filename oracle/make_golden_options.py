"""tests/golden/train_sr_reference.yml: the reference's shipped ESRGAN recipe (codes/options/sr/train_sr.yml) as a fixture; likewise
its train_sr.json, test_sr.yml, i2i/train_pix2pix.yml and i2i/train_cyclegan.yml (tests/golden/*_reference.*).

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

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
OUT = os.path.join(GOLDEN, "train_sr_reference.yml")
# the other recipes the reference ships for the paths in scope (same treatment: text as is, locations re-rooted)
OTHERS = (("sr/train_sr.json", "train_sr_reference.json"), ("sr/test_sr.yml", "test_sr_reference.yml"),
          ("i2i/train_pix2pix.yml", "train_pix2pix_reference.yml"), ("i2i/train_cyclegan.yml", "train_cyclegan_reference.yml"))


def reroot(txt):
    """every quoted location that starts with ../ (datasets, path.root, pretrained models) -> @ROOT@/"""
    out, n = re.subn(r"(['\"])\.\./", r"\1@ROOT@/", txt)
    return out, n


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
    for rel, name in OTHERS:
        txt = open(os.path.join(REF_CODES, "options", rel)).read()
        out, n = reroot(txt)
        assert n >= 2, (rel, n)
        with open(os.path.join(GOLDEN, name), "w") as f:
            f.write(out)
        print(name, "%d lines, %d locations replaced" % (len(out.splitlines()), n))


if __name__ == "__main__":
    main()
