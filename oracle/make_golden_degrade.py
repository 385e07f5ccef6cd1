"""tests/golden/degrade_kernels.pt: blur kernels from the REFERENCE's own generators -- get_gaussian_kernel
(augmennt/extra_functional.py:460-515, angle 0: pure numpy; rotated kernels need cv2.warpAffine and cannot run here) and
get_sinc_kernel (augmennt/spadd.py:16-37).  Build container only:  python -m oracle.make_golden_degrade"""
import os

import numpy as np
import torch

from . import ref_harness as R

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "degrade_kernels.pt")


def main():
    fx = {"gauss": [], "sinc": []}
    with R.reference_env():
        from dataops.augmennt.augmennt import extra_functional as EF
        from dataops.augmennt.augmennt import spadd as SP
        for ks, sig in ((7, (0.2, 0.2)), (9, (1.3, 1.3)), (21, (3.0, 3.0)), (13, (0.7, 2.4)), (21, (2.9, 0.25))):
            k = EF.get_gaussian_kernel(kernel_size=ks, sigma=sig, angle=0)
            fx["gauss"].append(dict(ks=ks, sigma=sig, kernel=torch.from_numpy(np.asarray(k, dtype=np.float64).copy())))
        for ks, cutoff in ((7, np.pi / 3), (11, 2.0), (13, np.pi / 5), (21, np.pi), (19, 1.1)):
            k = SP.get_sinc_kernel(cutoff=cutoff, kernel_size=ks)
            fx["sinc"].append(dict(ks=ks, cutoff=float(cutoff), kernel=torch.from_numpy(np.asarray(k, dtype=np.float64).copy())))
    torch.save(fx, OUT)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KB")


if __name__ == "__main__":
    main()
