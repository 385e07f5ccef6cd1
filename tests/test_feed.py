"""Input feed: wire format = uint8 HWC BGR crop windows; the paired flip / rot90 and np2tensor run
on the device.  Fixtures come from the REFERENCE's own crop / flip / rotate90 / np2tensor / get_params
(tests/golden/feed.pt, oracle/make_golden_feed.py).  Integer / layout work: bit-exact."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import feed_oracle as FO

FX = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "feed.pt"), weights_only=False)


def test_feed_oracle_and_param_draws_match_reference():
    for c in FX["cases"]:
        x = FO.flip_rot(FO.crop(c["img"].numpy(), c["pos"], c["size"]), c["flip"], c["rot"], c["vflip"])
        t = FO.np2tensor(np.ascontiguousarray(x), normalize=c["normalize"])
        assert t.shape == tuple(c["out"].shape) and np.array_equal(t, c["out"].numpy())
    from trainner_amd.data.aligned_dataset import paired_params
    for p in FX["params"]:
        random.seed(p["seed"])
        np.random.seed(p["seed"])
        got = paired_params(p["size_wh"], p["crop"])
        assert got["crop_pos"] == p["crop_pos"] and (got["flip"], got["rot"], got["vflip"], got["hrrot"], got["angle"]) == (
            p["flip"], p["rot"], p["vflip"], p["hrrot"], p["angle"]), (got, p)


def test_aligned_window_dataset_windows_and_flags(tmp_path):
    """dataset on .npy files: windows are slices of the decoded arrays at the reference's paired positions."""
    from trainner_amd.data.aligned_dataset import AlignedWindowDataset
    hr_dir, lr_dir = tmp_path / "hr", tmp_path / "lr"
    hr_dir.mkdir()
    lr_dir.mkdir()
    rng = np.random.RandomState(0)
    for i in range(3):
        np.save(str(hr_dir / ("%d.npy" % i)), rng.randint(0, 256, (96, 128, 3), dtype=np.uint8))
        np.save(str(lr_dir / ("%d.npy" % i)), rng.randint(0, 256, (24, 32, 3), dtype=np.uint8))
    ds = AlignedWindowDataset({"dataroot_HR": str(hr_dir), "dataroot_LR": str(lr_dir), "crop_size": 32, "scale": 4,
                               "use_flip": True, "use_rot": True, "mode": "aligned"})
    assert len(ds) == 3
    random.seed(5)
    s = ds[1]
    random.seed(5)
    from trainner_amd.data.aligned_dataset import paired_params, read_image_bgr
    p = paired_params((32, 24), 8)
    x, y = p["crop_pos"]
    assert s["LR"].shape == (8, 8, 3) and s["HR"].shape == (32, 32, 3) and s["LR"].dtype == np.uint8
    assert np.array_equal(s["LR"], read_image_bgr(s["LR_path"])[y:y + 8, x:x + 8])
    assert np.array_equal(s["HR"], read_image_bgr(s["HR_path"])[4 * y:4 * y + 32, 4 * x:4 * x + 32])
    assert s["flags"] == FO.flags_of(p["flip"], p["rot"], p["vflip"])


@pytest.mark.gpu
def test_device_feed_kernel_matches_reference_fixtures():
    from trainner_amd import hip
    from trainner_amd.dataops.common import np2tensor
    lib = hip.load()
    for c in FX["cases"]:
        win = np.ascontiguousarray(FO.crop(c["img"].numpy(), c["pos"], c["size"]))
        src = torch.from_numpy(win)[None].cuda()
        flags = torch.tensor([FO.flags_of(c["flip"], c["rot"], c["vflip"])], dtype=torch.int32, device="cuda")
        N, H, W, C = src.shape
        out = torch.empty((1, C, H, W), device="cuda")
        hip.check(lib.tnr_feed_u8_to_tensor(src.data_ptr(), 1, H, W, C, flags.data_ptr(), int(c["rot"]), out.data_ptr(), 1, 1.0,
                                            int(c["normalize"]), hip.stream()), "feed")
        assert torch.equal(out[0].cpu(), c["out"]), (c["pos"], c["flip"], c["rot"], c["vflip"], c["normalize"])
    c = FX["cases"][0]
    t = np2tensor(c["img"].numpy())
    assert tuple(t.shape) == (1, 3, 40, 52) and torch.equal(t[0].cpu(), torch.from_numpy(FO.np2tensor(c["img"].numpy())))


@pytest.mark.gpu
def test_device_feeder_double_buffering():
    """Many uint8 batches through DeviceFeeder while the compute stream is kept busy: every delivered batch must be
    the conversion of ITS host batch (no slot is overwritten before its consumer is done), in order, paths intact."""
    from trainner_amd.data.feeder import DeviceFeeder
    rng = np.random.RandomState(1)
    batches = []
    for k in range(9):
        n = 4
        flags = rng.randint(0, 8, n)
        flags = np.where(flags & 2, flags, flags & 1)            # vflip only together with rot
        batches.append({"LR": rng.randint(0, 256, (n, 16, 16, 3), dtype=np.uint8), "HR": rng.randint(0, 256, (n, 64, 64, 3), dtype=np.uint8),
                        "flags": flags, "LR_path": ["lr%d_%d" % (k, i) for i in range(n)], "HR_path": ["hr%d_%d" % (k, i) for i in range(n)]})
    feeder = DeviceFeeder(batches)
    busy = torch.randn(4096, 4096, device="cuda")
    seen = []
    for k, d in enumerate(feeder):
        lr, hr = d["LR"], d["HR"]                                   # consumer keeps them through its "step"
        for _ in range(3):
            busy = busy @ busy * 1e-4                               # compute-stream work after taking the batch
        seen.append((lr.clone(), hr.clone(), d["LR_path"]))
    torch.cuda.synchronize()
    assert len(seen) == len(batches)
    for k, (lr, hr, paths) in enumerate(seen):
        b = batches[k]
        for key, got in (("LR", lr), ("HR", hr)):
            for i in range(got.shape[0]):
                f = int(b["flags"][i])
                want = FO.np2tensor(np.ascontiguousarray(FO.flip_rot(b[key][i], f & 1, f & 2, f & 4)))
                assert torch.equal(got[i].cpu(), torch.from_numpy(want)), (k, key, i, f)
        assert paths == b["LR_path"]
    assert feeder.bytes_uploaded == sum(b["LR"].nbytes + b["HR"].nbytes for b in batches)
    # a consumer that leaves the loop early: the slot it still reads and the pre-uploaded ones get a fresh `free` event,
    # so the next epoch's uploads queue up behind the work that reads them -- the batch taken in the aborted epoch stays intact
    # while that work runs, and the new epoch delivers the right data again
    gen = iter(feeder)
    d0 = next(gen)
    lr0 = d0["LR"]
    for _ in range(6):
        busy = busy @ busy * 1e-4                                   # long compute-stream work that "reads" lr0 afterwards
    keep = lr0.clone()                                              # enqueued behind it on the compute stream
    gen.close()                                                     # GeneratorExit at the yield
    for k, d in enumerate(feeder):
        if k == 0:
            first = d["LR"].clone()
    torch.cuda.synchronize()
    want0 = torch.stack([torch.from_numpy(FO.np2tensor(np.ascontiguousarray(FO.flip_rot(batches[0]["LR"][i], int(batches[0]["flags"][i]) & 1,
                         int(batches[0]["flags"][i]) & 2, int(batches[0]["flags"][i]) & 4)))) for i in range(4)])
    assert torch.equal(keep.cpu(), want0) and torch.equal(first.cpu(), want0) and k == len(batches) - 1


@pytest.mark.gpu
def test_training_step_through_the_feeder(tmp_path):
    """create_dataset + create_dataloader (DeviceFeeder over a torch DataLoader) feeding SRModel.optimize_parameters."""
    import test_gpu_step as TS
    from trainner_amd.data import create_dataloader, create_dataset
    hr_dir, lr_dir = tmp_path / "hr", tmp_path / "lr"
    hr_dir.mkdir()
    lr_dir.mkdir()
    rng = np.random.RandomState(3)
    for i in range(4):
        hr = rng.randint(0, 256, (80, 96, 3), dtype=np.uint8)
        np.save(str(hr_dir / ("%d.npy" % i)), hr)
        np.save(str(lr_dir / ("%d.npy" % i)), hr[::4, ::4].copy())
    opt, model = TS.build_engine_model(dict(nb=1, batch=2, crop=64, d_nf=16), tmp_path)
    ds_opt = dict(opt["datasets"]["train"])
    ds_opt.update(dataroot_HR=str(hr_dir), dataroot_LR=str(lr_dir), use_flip=True, use_rot=True, use_shuffle=True, phase="train",
                  scale=4)
    loader = create_dataloader(create_dataset(ds_opt), ds_opt)
    assert len(loader) == 2
    step = 0
    for data in loader:
        assert data["LR"].is_cuda and tuple(data["LR"].shape) == (2, 3, 16, 16) and tuple(data["HR"].shape) == (2, 3, 64, 64)
        step += 1
        model.feed_data(data)
        model.optimize_parameters(step)
    log = model.get_current_log()
    assert step == 2 and all(np.isfinite(v) for v in log.values())


@pytest.mark.gpu
def test_resrgan_strategy_synthesises_lr_on_device(tmp_path):
    """augs_strategy: resrgan without dataroot_LR: the dataset ships HR windows only, the feeder converts them and runs
    the degradation pipeline on the copy stream; network_D: unet consumes the pair (BASELINE configs[3] on one GPU)."""
    import test_gpu_step as TS
    from trainner_amd.data import create_dataloader, create_dataset
    hr_dir = tmp_path / "hr"
    hr_dir.mkdir()
    rng = np.random.RandomState(4)
    for i in range(4):
        np.save(str(hr_dir / ("%d.npy" % i)), rng.randint(0, 256, (80, 96, 3), dtype=np.uint8))
    opt, model = TS.build_engine_model(dict(nb=1, batch=2, crop=64, d_nf=16, d_type="unet"), tmp_path)
    ds_opt = dict(opt["datasets"]["train"])
    ds_opt.pop("dataroot_LR", None)
    ds_opt.update(dataroot_HR=str(hr_dir), dataroot_LR=None, augs_strategy="resrgan", use_flip=True, use_rot=True, use_shuffle=False,
                  phase="train", scale=4)
    loader = create_dataloader(create_dataset(ds_opt), ds_opt)
    step = 0
    for data in loader:
        assert tuple(data["LR"].shape) == (2, 3, 16, 16) and tuple(data["HR"].shape) == (2, 3, 64, 64)
        assert float(data["LR"].min()) >= 0 and float(data["LR"].max()) <= 1 and data["LR_path"] == data["HR_path"]
        step += 1
        model.feed_data(data)
        model.optimize_parameters(step)
    assert step == 2 and all(np.isfinite(v) for v in model.get_current_log().values())


def test_unaligned_window_dataset_draw_order_and_flags(tmp_path):
    """mode: unaligned (CycleGAN data, codes/data/unaligned_dataset.py:70-139): B's index is drawn first (random.randint),
    then get_params for A, then for B -- the same `random` draws in the same order as the reference; per-image windows, flags."""
    from trainner_amd.data import create_dataset
    from trainner_amd.data.aligned_dataset import paired_params, read_image_bgr
    a_dir, b_dir = tmp_path / "a", tmp_path / "b"
    a_dir.mkdir()
    b_dir.mkdir()
    rng = np.random.RandomState(1)
    for i in range(3):
        np.save(str(a_dir / ("%d.npy" % i)), rng.randint(0, 256, (40, 56, 3), dtype=np.uint8))
    for i in range(5):
        np.save(str(b_dir / ("%d.npy" % i)), rng.randint(0, 256, (48, 44, 3), dtype=np.uint8))
    ds = create_dataset({"mode": "unaligned", "dataroot_A": str(a_dir), "dataroot_B": str(b_dir), "crop_size": 32, "scale": 1,
                         "use_flip": True, "use_rot": True, "preprocess": "crop", "outputs": "AB"})
    assert len(ds) == 5
    random.seed(9)
    s = ds[4]                                   # A index wraps: 4 % 3 = 1
    random.seed(9)
    bi = random.randint(0, 4)
    pa, pb = paired_params((56, 40), 32), paired_params((44, 48), 32)
    assert s["A_path"].endswith("1.npy") and s["B_path"].endswith("%d.npy" % bi)
    (xa, ya), (xb, yb) = pa["crop_pos"], pb["crop_pos"]
    assert np.array_equal(s["A"], read_image_bgr(s["A_path"])[ya:ya + 32, xa:xa + 32])
    assert np.array_equal(s["B"], read_image_bgr(s["B_path"])[yb:yb + 32, xb:xb + 32])
    assert s["flags_A"] == FO.flags_of(pa["flip"], pa["rot"], pa["vflip"]) and s["flags_B"] == FO.flags_of(pb["flip"], pb["rot"], pb["vflip"])
    # paired A / B folders (Pix2Pix): one draw, one crop position, one flag word
    pd = create_dataset({"mode": "aligned", "outputs": "AB", "dataroot_A": str(a_dir), "dataroot_B": str(a_dir), "crop_size": 32,
                         "scale": 1, "use_flip": True, "use_rot": False})
    random.seed(3)
    q = pd[2]
    assert set(q) == {"A", "B", "flags", "A_path", "B_path"} and np.array_equal(q["A"], q["B"]) and q["A"].shape == (32, 32, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["pix2pix", "cyclegan"])
def test_i2i_training_step_through_the_feeder(kind, tmp_path):
    """create_dataset + create_dataloader for the image-to-image models: uint8 A / B windows (unaligned: each image with its own
    flip / rot90 flags) -> DeviceFeeder (znorm: [-1, 1]) -> Pix2PixModel / CycleGANModel.optimize_parameters."""
    import test_gpu_i2i as TI
    from trainner_amd.data import create_dataloader, create_dataset
    a_dir, b_dir = tmp_path / "a", tmp_path / "b"
    a_dir.mkdir()
    b_dir.mkdir()
    rng = np.random.RandomState(5)
    for i in range(4):
        np.save(str(a_dir / ("%d.npy" % i)), rng.randint(0, 256, (80, 96, 3), dtype=np.uint8))
        np.save(str(b_dir / ("%d.npy" % i)), rng.randint(0, 256, (80, 96, 3), dtype=np.uint8))
    opt, model = TI.build_i2i_model(dict(model=kind, batch=2, crop=64, n_blocks=1, ngf=16, ndf=16,
                                         pixel_weight=100.0 if kind == "pix2pix" else 10.0,
                                         lambda_identity=0.5 if kind == "cyclegan" else None), tmp_path)
    ds_opt = dict(opt["datasets"]["train"])
    ds_opt.update(mode="aligned" if kind == "pix2pix" else "unaligned", dataroot_A=str(a_dir), dataroot_B=str(b_dir), use_flip=True,
                  use_rot=True, use_shuffle=True, phase="train", scale=1, preprocess="crop")
    loader = create_dataloader(create_dataset(ds_opt), ds_opt)
    step = 0
    for data in loader:
        assert data["A"].is_cuda and tuple(data["A"].shape) == (2, 3, 64, 64) and tuple(data["B"].shape) == (2, 3, 64, 64)
        assert -1.0 <= float(data["A"].min()) and float(data["A"].max()) <= 1.0 and float(data["A"].min()) < -0.9     # znorm
        step += 1
        model.feed_data(data)
        model.optimize_parameters(step)
    log = model.get_current_log()
    assert step == 2 and len(log) >= 4 and all(np.isfinite(v) for v in log.values())


@pytest.mark.gpu
def test_memoized_discriminator_with_gradient_accumulation_through_the_feeder(tmp_path, monkeypatch):
    """DeviceFeeder refills its two persistent slot buffers with a raw kernel, so batches k and k + 2 share the same
    tensor address AND version counter; with virtual_batch_size = 3 x batch_size the discriminator's parameters do not move
    for three micro-steps.  A forward memo keyed on (storage, version, parameter version) alone would hand batch k's logits
    and saved activations to batch k + 2.  The memo is emptied at the start of every optimize_parameters: six micro-steps
    (two optimizer steps) with and without memoization must agree bit for bit."""
    import test_gpu_step as TS
    from oracle import detrand
    from trainner_amd.data.feeder import DeviceFeeder
    rng = np.random.RandomState(5)
    batches = [{"HR": rng.randint(0, 256, (2, 64, 64, 3), dtype=np.uint8), "flags": np.zeros(2, np.int64)} for _ in range(6)]
    for b in batches:
        b["LR"] = np.ascontiguousarray(b["HR"][:, ::4, ::4])

    def run(memo, sub):
        monkeypatch.setenv("TNR_D_MEMO", "1" if memo else "0")
        (tmp_path / sub).mkdir()
        yml_kw = dict(nb=1, batch=2, crop=64, d_nf=16)
        from oracle import ref_harness
        from trainner_amd.models import create_model
        from trainner_amd.options import options
        yml = ref_harness.esrgan_yaml(name="accum", out_root=str(tmp_path / sub), gpu_ids="[0]", **yml_kw)
        txt = open(yml).read().replace("virtual_batch_size: 2", "virtual_batch_size: 6")
        open(yml, "w").write(txt)
        model = create_model(options.parse(yml, is_train=True), verbose=False)
        assert model.accumulations == 3 and model.netD.memoize == memo
        g = detrand.fill_state_dict_({k: v.detach().cpu().clone() for k, v in model.netG.state_dict().items()}, 101)
        d = detrand.fill_state_dict_({k: v.detach().cpu().clone() for k, v in model.netD.state_dict().items()}, 202)
        TS.load_initial(model, g, d, TS.FX.vgg_state(77))
        logs = []
        ptrs = set()
        for s, data in enumerate(DeviceFeeder(batches), 1):
            ptrs.add(data["HR"].data_ptr())
            model.feed_data(data)
            model.optimize_parameters(s)
            logs.append(model.get_current_log())
        assert len(ptrs) == 2                                  # two slots: batches k and k + 2 really share a buffer
        return logs, {k: v.detach().cpu() for k, v in model.netD.state_dict().items()}, {k: v.detach().cpu() for k, v in model.netG.state_dict().items()}

    la, da, ga = run(True, "memo")
    lb, db, gb = run(False, "plain")
    assert la == lb
    for k in da:
        assert torch.equal(da[k], db[k]), k
    for k in ga:
        assert torch.equal(ga[k], gb[k]), k


def test_lmdb_dataset_reader(tmp_path, monkeypatch):
    """`data_type: lmdb` (a dataroot ending in .lmdb, dataops/common.py:47-105): keys from meta_info.txt, values = encoded image files,
    the environment opened read-only without lock / read-ahead / meminit (_init_lmdb).  Runs against the real `lmdb` module when it is
    installed; otherwise an in-memory module with the same three calls (open, begin(write=False), get) stands in, so the reader's own
    logic -- key list, lazy open, decode to BGR, windows, paired parameters -- is exercised either way."""
    import io
    import sys
    import types
    from PIL import Image
    from trainner_amd.data.aligned_dataset import AlignedWindowDataset, LmdbSource
    rs = np.random.RandomState(4)
    imgs = {"%03d" % i: rs.randint(0, 256, (48, 64, 3), dtype=np.uint8) for i in range(3)}      # RGB as stored in the PNG files

    def png(a):
        b = io.BytesIO()
        Image.fromarray(a, "RGB").save(b, format="PNG")
        return b.getvalue()

    try:
        import lmdb as real
    except ImportError:
        real = None
    stores = {}
    for name, scale in (("hr.lmdb", 1), ("lr.lmdb", 4)):
        root = tmp_path / name
        root.mkdir()
        vals = {k: png(v[::scale, ::scale].copy()) for k, v in imgs.items()}
        (root / "meta_info.txt").write_text("".join("%s.png (%d,%d,3) 1\n" % (k, 48 // scale, 64 // scale) for k in imgs))
        if real is not None:
            env = real.open(str(root), map_size=1 << 24)
            with env.begin(write=True) as txn:
                for k, v in vals.items():
                    txn.put(k.encode("ascii"), v)
            env.close()
        stores[str(root)] = vals
    if real is None:
        opened = []

        class _Txn:
            def __init__(self, vals):
                self.vals = vals

            def __enter__(self):
                return self

            def __exit__(self, *a):
                return False

            def get(self, key):
                return self.vals.get(key.decode("ascii"))

        class _Env:
            def __init__(self, root):
                self.vals = stores[root]

            def begin(self, write=False):
                assert write is False
                return _Txn(self.vals)

        fake = types.ModuleType("lmdb")

        def _open(root, readonly=False, lock=True, readahead=True, meminit=True, **kw):
            assert readonly and not lock and not readahead and not meminit            # _init_lmdb's flags
            opened.append(root)
            return _Env(root)

        fake.open = _open
        monkeypatch.setitem(sys.modules, "lmdb", fake)
    src = LmdbSource(str(tmp_path / "hr.lmdb"))
    assert len(src) == 3 and src.path(1) == "001"
    assert np.array_equal(src.read(2), imgs["002"][:, :, ::-1])                         # decoded to BGR like cv2.imdecode
    opt = dict(scale=4, crop_size=32, dataroot_HR=str(tmp_path / "hr.lmdb"), dataroot_LR=str(tmp_path / "lr.lmdb"), data_type="lmdb",
               use_flip=True, use_rot=True)
    ds = AlignedWindowDataset(opt)
    random.seed(3)
    d = ds[1]
    random.seed(3)
    from trainner_amd.data.aligned_dataset import paired_params
    p = paired_params((16, 12), 8)
    x, y = p["crop_pos"]
    assert d["HR"].shape == (32, 32, 3) and d["LR"].shape == (8, 8, 3) and d["HR_path"] == "001"
    assert np.array_equal(d["HR"], imgs["001"][:, :, ::-1][4 * y:4 * y + 32, 4 * x:4 * x + 32])
    assert np.array_equal(d["LR"], imgs["001"][::4, ::4][:, :, ::-1][y:y + 8, x:x + 8])
    with pytest.raises(ValueError):
        LmdbSource(str(tmp_path))                                                        # not a .lmdb folder
