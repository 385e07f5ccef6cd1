"""Paired LR / HR dataset that hands out uint8 crop WINDOWS (the engine's wire format) instead of fp32 tensors.

Mirrors what codes/data/aligned_dataset.py does for `mode: aligned` with `preprocess: crop`: read the image pair,
draw the paired transform parameters exactly like get_params (dataops/augmentations.py:457-511: crop position in LR
coordinates, then flip, rot, vflip, hrrot, angle -- the same `random` draws in the same order), crop both images
(crop(), :776-790; HR position = LR position x scale, scale_params :1025-1036) and return the batch-dict entries of
aligned_dataset.py:166-175.  Differences, all deliberate: the crop is a zero-copy slice of the decoded uint8 array; the
flip / rot90 and np2tensor are NOT done here but on the GPU (data/feeder.py), so a sample also carries its `flags`.
Files: .npy arrays (uint8 HWC, BGR like cv2.imread) or anything PIL opens (converted RGB -> BGR); `data_type: lmdb` (a dataroot ending
in .lmdb: dataops/common.py:47-105) reads the encoded images out of one lmdb database (the `lmdb` module is imported on first use).
"""
import os
import random

import numpy as np
import torch.utils.data as data

IMG_EXT = (".npy", ".png", ".jpg", ".jpeg", ".bmp", ".webp", ".tif", ".tiff")


def _list_images(root):
    roots = root if isinstance(root, (list, tuple)) else [root]
    out = []
    for r in roots:
        for d, _, files in sorted(os.walk(r)):
            out += [os.path.join(d, f) for f in sorted(files) if f.lower().endswith(IMG_EXT)]
    if not out:
        raise ValueError("no images found under %s" % (roots,))
    return out


class LmdbSource:
    """Images stored in one lmdb database (dataops/common.py:47-105: `dataroot` ends in `.lmdb`, the keys are the lines of its
    meta_info.txt up to the first dot, the values are encoded image files; _init_lmdb opens it read-only without lock, read-ahead or
    meminit).  The environment is opened lazily, once per process (DataLoader workers each open their own)."""

    def __init__(self, dataroot):
        if not isinstance(dataroot, str) or not dataroot.endswith(".lmdb"):
            raise ValueError(f"Folder {dataroot} should in lmdb format.")
        self.root = dataroot
        with open(os.path.join(dataroot, "meta_info.txt")) as fin:
            self.keys = [line.split(".")[0] for line in fin if line.strip()]
        if not self.keys:
            raise ValueError("%s: empty meta_info.txt" % dataroot)
        self._env = None

    def __len__(self):
        return len(self.keys)

    def path(self, index):
        return self.keys[index]

    def env(self):
        if self._env is None:
            try:
                import lmdb
            except ImportError:
                raise ImportError("Please install lmdb to enable use.")
            self._env = lmdb.open(self.root, readonly=True, lock=False, readahead=False, meminit=False)
        return self._env

    def read(self, index):
        with self.env().begin(write=False) as txn:
            buf = txn.get(self.keys[index].encode("ascii"))
        if buf is None:
            raise KeyError("%s: key %r not found" % (self.root, self.keys[index]))
        return decode_image_bgr(bytes(buf), self.keys[index])


class FolderSource:
    def __init__(self, root):
        self.paths = _list_images(root)

    def __len__(self):
        return len(self.paths)

    def path(self, index):
        return self.paths[index]

    def read(self, index):
        return read_image_bgr(self.paths[index])


def image_source(root, data_type="img"):
    """data_type as options.parse sets it (`lmdb` when the dataroot ends in .lmdb, else `img`; dataops/common.py:76-88)."""
    if data_type == "lmdb" or (isinstance(root, str) and root.lower().endswith(".lmdb")):
        return LmdbSource(root)
    return FolderSource(root)


def decode_image_bgr(content, name="<bytes>"):
    """imfrombytes (dataops/common.py:108-128, flag 'color'): an encoded image file -> uint8 HWC BGR."""
    import io
    from PIL import Image
    with Image.open(io.BytesIO(content)) as im:
        img = np.asarray(im.convert("RGB"))[:, :, ::-1]
    if img.dtype != np.uint8 or img.ndim != 3:
        raise ValueError("%s: expected a uint8 HWC image" % name)
    return img


def read_image_bgr(path):
    """uint8 HWC, BGR channel order (what cv2.imread returns to the reference's datasets)."""
    if path.lower().endswith(".npy"):
        img = np.load(path)
    else:
        from PIL import Image
        with Image.open(path) as im:
            img = np.asarray(im.convert("RGB"))[:, :, ::-1]
    if img.dtype != np.uint8 or img.ndim != 3:
        raise ValueError("%s: expected a uint8 HWC image" % path)
    return img


def paired_params(lr_size_wh, crop_lr):
    """get_params (dataops/augmentations.py:457-511) for preprocess 'crop': same draws, same order."""
    w, h = lr_size_wh
    x = random.randint(0, max(0, w - crop_lr))
    y = random.randint(0, max(0, h - crop_lr))
    flip = random.random() > 0.5
    rot = random.random() > 0.5
    vflip = random.random() > 0.5
    hrrot = random.random() > 0.5
    angle = int(random.uniform(-90, 90))
    return {"crop_pos": (x, y), "flip": flip, "rot": rot, "vflip": vflip, "hrrot": hrrot, "angle": angle}


def window(img, pos, size):
    """crop() of dataops/augmentations.py:776-790: a slice when the image is larger than the window, else the image."""
    x1, y1 = pos
    oh, ow = img.shape[:2]
    if ow > size or oh > size:
        return img[y1:y1 + size, x1:x1 + size, ...]
    return img


class AlignedWindowDataset(data.Dataset):
    def __init__(self, opt):
        self.opt = opt
        self.scale = int(opt.get("scale", 1) or 1)
        self.crop = int(opt["crop_size"])
        self.use_flip, self.use_rot = bool(opt.get("use_flip")), bool(opt.get("use_rot"))
        if opt.get("use_hrrot"):
            raise NotImplementedError("use_hrrot (free-angle rotation) is not implemented by the HIP engine feeder")
        dt = opt.get("data_type", "img") or "img"
        self.hr_src = image_source(opt["dataroot_HR"] if opt.get("dataroot_HR") else opt["dataroot_B"], dt)
        lr_root = opt.get("dataroot_LR") or opt.get("dataroot_A")
        self.lr_src = None
        if lr_root:
            self.lr_src = image_source(lr_root, dt)
            if len(self.lr_src) != len(self.hr_src):
                raise ValueError("LR / HR datasets have different lengths: %d vs %d" % (len(self.lr_src), len(self.hr_src)))
        elif not (opt.get("augs_strategy") or opt.get("degradation")):
            raise NotImplementedError("no dataroot_LR: on-the-fly LR needs `augs_strategy: <preset>` (GPU degradation pipeline)")

    def __len__(self):
        return len(self.hr_src)

    def __getitem__(self, index):
        hr = self.hr_src.read(index)
        if self.lr_src is None:             # HR window only: the LR image is synthesised on the GPU (data/feeder.py)
            p = paired_params((hr.shape[1], hr.shape[0]), self.crop)
            flags = (1 if (self.use_flip and p["flip"]) else 0) | (2 if (self.use_rot and p["rot"]) else 0)
            if flags & 2 and p["vflip"]:
                flags |= 4
            return {"HR": np.ascontiguousarray(window(hr, p["crop_pos"], self.crop)), "flags": flags, "HR_path": self.hr_src.path(index)}
        lr = self.lr_src.read(index)
        crop_lr = self.crop // self.scale
        p = paired_params((lr.shape[1], lr.shape[0]), crop_lr)
        x, y = p["crop_pos"]
        lr_w = window(lr, (x, y), crop_lr)
        hr_w = window(hr, (int(x * self.scale), int(y * self.scale)), self.crop)
        flags = (1 if (self.use_flip and p["flip"]) else 0) | (2 if (self.use_rot and p["rot"]) else 0)
        if flags & 2 and p["vflip"]:
            flags |= 4
        return {"LR": np.ascontiguousarray(lr_w), "HR": np.ascontiguousarray(hr_w), "flags": flags,
                "LR_path": self.lr_src.path(index), "HR_path": self.hr_src.path(index)}


class UnalignedWindowDataset(data.Dataset):
    """Unpaired A / B image folders for the image-to-image models (codes/data/unaligned_dataset.py:8-145, `mode: unaligned`,
    `preprocess: crop`): image A by index (wrapped into range), image B at a random index unless `serial_batches`
    (read_single_dataset, base_dataset.py:343-359 -- the `random.randint` is drawn before A's transform parameters, like
    the reference reads B before get_params), then TWO independent parameter draws (get_params for A, then for B:
    unaligned_dataset.py:93-98), a crop window per image and per-image flags (`flags_A`, `flags_B`: the feeder applies each
    image's own flip / rot90).  Returns the reference's batch-dict entries {'A', 'B', 'A_path', 'B_path'} (:137-139)."""

    def __init__(self, opt):
        self.opt = opt
        self.crop = int(opt["crop_size"])
        self.use_flip, self.use_rot = bool(opt.get("use_flip")), bool(opt.get("use_rot"))
        pre = str(opt.get("preprocess", "crop") or "crop")
        if pre not in ("crop", "none"):
            raise NotImplementedError("preprocess [%s]: the HIP engine feeder ships crop windows (`preprocess: crop`)" % pre)
        self.a_paths = _list_images(opt["dataroot_A"])
        self.b_paths = _list_images(opt["dataroot_B"])
        self.serial = bool(opt.get("serial_batches"))

    def __len__(self):
        return max(len(self.a_paths), len(self.b_paths))

    def _flags(self, p):
        flags = (1 if (self.use_flip and p["flip"]) else 0) | (2 if (self.use_rot and p["rot"]) else 0)
        if flags & 2 and p["vflip"]:
            flags |= 4
        return flags

    def __getitem__(self, index):
        a_path = self.a_paths[index % len(self.a_paths)]
        b_path = self.b_paths[index % len(self.b_paths) if self.serial else random.randint(0, len(self.b_paths) - 1)]
        img_a, img_b = read_image_bgr(a_path), read_image_bgr(b_path)
        pa = paired_params((img_a.shape[1], img_a.shape[0]), self.crop)
        pb = paired_params((img_b.shape[1], img_b.shape[0]), self.crop)
        return {"A": np.ascontiguousarray(window(img_a, pa["crop_pos"], self.crop)),
                "B": np.ascontiguousarray(window(img_b, pb["crop_pos"], self.crop)),
                "flags_A": self._flags(pa), "flags_B": self._flags(pb), "A_path": a_path, "B_path": b_path}


class AlignedABWindowDataset(AlignedWindowDataset):
    """Paired A / B folders with `outputs: AB` (codes/data/aligned_dataset.py:166-175 returns {'A','B','A_path','B_path'} instead
    of LR / HR): the paired windows of AlignedWindowDataset under the image-to-image names (scale 1: one crop position)."""

    def __getitem__(self, index):
        d = super().__getitem__(index)
        return {"A": d["LR"], "B": d["HR"], "flags": d["flags"], "A_path": d["LR_path"], "B_path": d["HR_path"]}
