"""tests/golden/feed.pt from the REFERENCE's own crop / flip / rotate90 (dataops/augmentations.py:776-830), np2tensor
(dataops/common.py:470-499) and get_params (augmentations.py:457-511) -- build container only:
    python -m oracle.make_golden_feed"""
import os
import random

import numpy as np
import torch

from . import detrand
from . import ref_harness as R

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "feed.pt")


def u8_image(h, w, c, seed):
    return (detrand.uniform((h, w, c), seed, 0.0, 256.0).floor().clamp(0, 255)).to(torch.uint8).numpy()


def main():
    cases, params = [], []
    with R.reference_env():
        from dataops import augmentations as A
        from dataops.common import np2tensor
        k = 0
        for (h, w, c, size, pos) in ((40, 52, 3, 24, (5, 9)), (33, 33, 3, 16, (17, 0)), (20, 28, 4, 12, (3, 2)), (18, 18, 1, 18, (0, 0))):
            img = u8_image(h, w, c, 700 + k)
            for flip in (False, True):
                for rot in (False, True):
                    for vflip in (False, True):
                        for norm in (False, True):
                            x = A.crop(img, pos, size=size, img_type="cv2")
                            x = A.flip(x, flip, img_type="cv2")
                            x = A.rotate90(x, rot, vflip, img_type="cv2")
                            t = np2tensor(np.ascontiguousarray(x), normalize=norm, add_batch=False)
                            cases.append(dict(img=torch.from_numpy(img.copy()), pos=pos, size=size, flip=flip, rot=rot,
                                              vflip=vflip, normalize=norm, out=t.clone()))
            k += 1
        # get_params draws (crop positions in LR coordinates + flags) for fixed seeds
        for seed in (0, 1, 2, 3, 4):
            random.seed(seed)
            np.random.seed(seed)
            p = A.get_params({"crop_size": 32, "preprocess": "crop"}, (100, 80))
            params.append(dict(seed=seed, size_wh=(100, 80), crop=32, crop_pos=tuple(int(v) for v in p["crop_pos"]),
                               flip=bool(p["flip"]), rot=bool(p["rot"]), vflip=bool(p["vflip"]), hrrot=bool(p["hrrot"]),
                               angle=int(p["angle"])))
    torch.save(dict(cases=cases, params=params), OUT)
    print("wrote", OUT, len(cases), "cases", os.path.getsize(OUT) // 1024, "KB;", params[0])


if __name__ == "__main__":
    main()
