"""Fixtures for trainner_amd/dataops/common.py from the REAL reference (codes/dataops/common.py:575-767), run in the
build container only:  python -m oracle.make_golden_patches  ->  tests/golden/patches.pt
TEST INFRASTRUCTURE ONLY."""
import os

import torch

from . import detrand, ref_harness

CASES = [
    # name, B, C, H, W, patch, step, scale
    ("exact_grid", 1, 3, 24, 36, 12, 1.0, 4),
    ("ragged", 1, 3, 25, 38, 12, 1.0, 2),
    ("overlap_075", 1, 3, 30, 41, 16, 0.75, 4),
    ("overlap_05", 2, 1, 20, 20, 8, 0.5, 1),
]


def main():
    out = {}
    with ref_harness.reference_env():
        from dataops.common import extract_patches_2d, recompose_tensor
        for name, B, C, H, W, p, step, scale in CASES:
            img = detrand.uniform((B, C, H, W), 500 + len(out), 0.0, 1.0)
            pat = extract_patches_2d(img=img, patch_shape=(p, p), step=[step, step], batch_first=True)
            # stand-in for the network: nearest upscaling by `scale` plus a per-patch offset, so misplaced patches show
            flat = pat.reshape(-1, C, p, p)
            sr = torch.nn.functional.interpolate(flat, scale_factor=scale, mode="nearest") + 0.01 * torch.arange(flat.size(0)).view(-1, 1, 1, 1)
            rec = recompose_tensor(sr[:flat.size(0) // B] if B > 1 else sr, H, W, step=step, scale=scale)
            out[name] = dict(spec=(B, C, H, W, p, step, scale), img=img, patches=pat.clone(), sr=sr.clone(), recomposed=rec.clone())
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "patches.pt")
    torch.save(out, dst)
    print("wrote", dst, {k: (tuple(v["patches"].shape), tuple(v["recomposed"].shape)) for k, v in out.items()})


if __name__ == "__main__":
    main()
