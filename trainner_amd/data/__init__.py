"""create_dataset / create_dataloader of the engine (reference: codes/data/__init__.py:8-97).

`mode: aligned` builds data.aligned_dataset.AlignedWindowDataset (uint8 crop windows; `outputs: AB` the paired A / B variant),
`mode: unaligned` the unpaired A / B dataset of the image-to-image models; create_dataloader wraps the
torch DataLoader (same parameters as the reference: batch_size, use_shuffle, n_workers, drop_last, pin_memory) in a
DeviceFeeder, so iterating it yields the reference's batch dicts with device fp32 tensors.  `batch_size` keeps the
reference's meaning -- the GLOBAL batch (options/README.md:31): with one process per GPU each rank loads
batch_size / world_size samples of a rank-strided index shard.
"""
import torch.utils.data as tud

from .feeder import DeviceFeeder


def create_dataset(dataset_opt):
    mode = str(dataset_opt["mode"]).lower()
    if mode in ("aligned", "lrhr", "lrhrotf", "lrhrc"):
        from .aligned_dataset import AlignedABWindowDataset, AlignedWindowDataset
        if str(dataset_opt.get("outputs", "LRHR")).upper() == "AB" and dataset_opt.get("dataroot_A") and dataset_opt.get("dataroot_B"):
            return AlignedABWindowDataset(dataset_opt)          # paired image-to-image data (Pix2Pix)
        return AlignedWindowDataset(dataset_opt)
    if mode == "unaligned":                                     # unpaired image-to-image data (CycleGAN)
        from .aligned_dataset import UnalignedWindowDataset
        return UnalignedWindowDataset(dataset_opt)
    raise NotImplementedError("Dataset [{:s}] is outside the SR hot path of the HIP engine".format(mode))


class _RankShard(tud.Sampler):
    """rank-strided subset of a (shuffled) index permutation; same permutation on every rank (seeded per epoch).  Every pass
    over the sampler is a new epoch (the reference's DataLoader(shuffle=True) reshuffles per epoch, data/__init__.py:27-46):
    the epoch counter advances by itself, identically on all ranks; set_epoch() still pins it (resume)."""

    def __init__(self, n, rank, world, shuffle, seed=0):
        self.n, self.rank, self.world, self.shuffle, self.seed, self.epoch = n, rank, world, shuffle, seed, 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __iter__(self):
        import torch
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            idx = torch.randperm(self.n, generator=g).tolist()
            self.epoch += 1
        else:
            idx = list(range(self.n))
        idx = idx[:len(idx) - len(idx) % self.world]
        return iter(idx[self.rank::self.world])

    def __len__(self):
        return self.n // self.world


def create_dataloader(dataset, dataset_opt, gpu_ids=None, device=None, rank=0, world_size=1):
    train = dataset_opt.get("phase", "test") == "train"
    if train:
        gb = int(dataset_opt["batch_size"])
        if gb % world_size:
            raise ValueError("batch_size %d is not divisible by the %d data-parallel ranks" % (gb, world_size))
        sampler = _RankShard(len(dataset), rank, world_size, bool(dataset_opt.get("use_shuffle")))
        loader = tud.DataLoader(dataset, batch_size=gb // world_size, sampler=sampler, drop_last=True, pin_memory=True,
                                num_workers=int(dataset_opt.get("n_workers", 0) or 0))
    else:
        loader = tud.DataLoader(dataset, batch_size=1, shuffle=False, num_workers=1, drop_last=False, pin_memory=True)
    degrade = None
    if train and not dataset_opt.get("dataroot_LR") and (dataset_opt.get("augs_strategy") or dataset_opt.get("degradation")):
        # on-the-fly LR: the presets named by `augs_strategy` / `add_*_preset` (options/presets/README.md:25-33), merged by options.parse
        from ..dataops.degradations import RealESRGANDegradation, degradation_config
        conf = dataset_opt.get("degradation") or degradation_config(dataset_opt, dataset_opt.get("presets_root"))
        degrade = RealESRGANDegradation(scale=int(dataset_opt.get("scale", 4) or 4), preset=conf, seed=int(dataset_opt.get("seed", 0) or 0) + rank)
    return DeviceFeeder(loader, device=device, znorm=bool(dataset_opt.get("znorm")), degrade=degrade)
