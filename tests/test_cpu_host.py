"""CPU-only checks (-m "not gpu"): the C-ABI library loads and exports every symbol the header
declares, the options loader reproduces the reference's parsed network dicts, the engine's networks
expose the reference's state_dict keys/shapes, checkpoints round-trip, and the host-side logic
(flat parameters, fused-Adam state layout) behaves -- no kernel is launched here.
"""
import os
import re

import pytest
import torch

from oracle import fixtures as FX, ref_harness

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_match_header():
    from trainner_amd import hip
    lib = hip.load()
    hdr = open(os.path.join(ROOT, "include", "trainner_hip.h")).read()
    declared = set(re.findall(r"\b(tnr_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(hip.EXPORTS), declared ^ set(hip.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.tnr_version() == hip.ABI_VERSION == 3          # (header TNR_ABI_VERSION; a mismatch raises in hip.load)


def test_no_device_fails_loudly():
    from trainner_amd import hip
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    with pytest.raises(hip.HipEngineError):
        hip.require_device()
    from trainner_amd.models.modules.architectures.RRDBNet_arch import RRDBNet
    net = RRDBNet(3, 3, 64, 1)
    with pytest.raises(hip.HipEngineError):
        net(torch.zeros(1, 3, 8, 8))          # no CPU / eager fallback exists


@pytest.mark.parametrize("case", ["cfg1_srresnet", "esrgan_nb1_crop64", "esrgan_nb1_pixelshuffle", "esrgan_nb23_crop128",
                                  "esrgan_nb2_crop64_k10", "esrgan_nb23_crop512_b2", "esrgan_nb23_crop512_b4", "esrgan_nb1_unet",
                                  "esrgan_nb2_crop128_b16", "esrgan_nb23_crop128_b2_k10"])
def test_options_and_state_dict_contract(case, tmp_path):
    """options.parse expands to the dicts the REAL reference produced (stored in the fixtures) and the
    engine's networks carry exactly the reference's state_dict keys and shapes."""
    from trainner_amd.options import options
    from trainner_amd.models import networks
    fx = FX.load(case)
    yml = ref_harness.esrgan_yaml(name="contract", out_root=str(tmp_path), gpu_ids="[0]", **fx["spec"]["yaml"])
    opt = options.parse(yml, is_train=True)
    assert dict(opt["network_G"]) == fx["network_G"]
    assert opt["train"]["no_such_key"] is None                       # NoneDict semantics
    assert opt["path"]["models"].endswith(os.path.join("experiments", "contract", "models"))
    netG = networks.define_G(opt)
    assert [(k, tuple(v.shape)) for k, v in netG.state_dict().items()] == fx["g_keys"]
    if fx["network_D"]:
        assert dict(opt["network_D"]) == fx["network_D"]
        netD = networks.define_D(opt)
        assert [(k, tuple(v.shape)) for k, v in netD.state_dict().items()] == fx["d_keys"]


def test_yaml_scientific_notation_and_json(tmp_path):
    from trainner_amd.options import options
    y = tmp_path / "a.yml"
    y.write_text("a: 1e-4\nb: 5e5\nc: [0.1, 2]\nd: {e: null}\n")
    o = options.read_yaml(str(y))
    assert o["a"] == 1e-4 and isinstance(o["b"], float) and o["d"]["e"] is None
    j = tmp_path / "a.json"
    j.write_text('{ "a": 1, // comment\n "b": {"c": [1,2]} }')
    assert options.read_json(str(j))["b"]["c"] == [1, 2]
    nd = options.dict_to_nonedict({"x": {"y": 1}, "l": [{"z": 2}]})
    assert nd["x"]["q"] is None and nd["l"][0]["w"] is None
    assert options.opt_get(nd, ["x", "y"]) == 1 and options.opt_get(nd, ["x", "nope"], 7) == 7


def test_shipped_recipes_parse(tmp_path):
    """Every recipe the reference ships for the paths in scope (tests/golden/shipped_recipes.json: options/sr/train_sr.{yml,json},
    sr/test_sr.yml, i2i/train_{pix2pix,cyclegan}.yml, locations re-rooted) goes through `options.parse`; the JSON file is
    the YAML recipe (one spelling differs, lr_downscale_types 'cubic' / 'bicubic', which parse maps to one interpolation), so
    both give the same option tree."""
    from trainner_amd.options import options

    def parse(rel, is_train=True):
        return options.parse(FX.write_recipe(rel, str(tmp_path)), is_train=is_train)

    y, j = parse("sr/train_sr.yml"), parse("sr/train_sr.json")
    for k in ("network_G", "network_D", "train", "scale", "use_amp", "model", "gpu_ids", "logger"):
        a, b = y[k], j[k]
        assert (dict(a) == dict(b)) if isinstance(a, dict) else (a == b), (k, a, b)
    assert dict(y["datasets"]["train"]) == dict(j["datasets"]["train"])
    vy, vj = dict(y["datasets"]["val"]), dict(j["datasets"]["val"])
    assert vy.pop("lr_downscale_types") == ["linear", "bicubic"] and vj.pop("lr_downscale_types") == ["linear", "cubic"] and vy == vj
    t = parse("sr/test_sr.yml", is_train=False)
    assert t["is_train"] is False and t["network_G"]["type"] == "rrdb_net" and t["path"]["pretrain_model_G"].endswith("RRDB_ESRGAN_x4.pth")
    p2p, cyc = parse("i2i/train_pix2pix.yml"), parse("i2i/train_cyclegan.yml")
    # (what the reference's own parse gives for these files, run in the build container)
    assert dict(p2p["network_D"]) == {"strict": True, "type": "patchgan", "input_nc": 6, "ndf": 64, "n_layers": 3, "get_feats": False,
                                      "patch": True, "use_spectral_norm": False}
    assert dict(p2p["network_G"]) == {"strict": False, "type": "unet_net", "input_nc": 3, "output_nc": 3, "num_downs": 8, "ngf": 64,
                                      "norm_type": "batch", "use_dropout": False, "upsample_mode": "deconv"}
    assert cyc["network_G"]["type"] == "resnet_net" and cyc["pool_size"] == 50 and cyc["train"]["lr_scheme"] == "Linear"
    assert "gan_opt" not in p2p["train"] and "gan_opt" not in cyc["train"]


@pytest.mark.skipif(not ref_harness.reference_available(), reason="needs the reference checkout (build container)")
def test_shipped_recipe_fixture_is_the_reference_files(tmp_path):
    """tests/golden/shipped_recipes.json against the files themselves: (1) the fixture equals the trees the reference's files
    denote (oracle/make_golden_options.reference_trees); (2) `options.parse` of the REAL file (comments, layout, `5e-3` scalars,
    JSON with // comments) and of the file the tests write from the fixture give the same option tree, modulo the locations."""
    import json
    from collections import OrderedDict
    from oracle import make_golden_options as M
    from trainner_amd.options import options
    with open(os.path.join(FX.GOLDEN_DIR, "shipped_recipes.json")) as f:
        stored = json.load(f, object_pairs_hook=OrderedDict)
    assert json.loads(json.dumps(M.reference_trees())) == json.loads(json.dumps(stored))

    def flat(d, pre=""):
        out = {}
        if isinstance(d, dict):
            for k, v in d.items():
                out.update(flat(v, pre + str(k) + "."))
        else:
            out[pre[:-1]] = d
        return out

    wroot = str(tmp_path / "w")

    def loc(v):
        """a location with its root removed: '../x' (the real file, relative to codes/) and '<wroot>/x' (the written file) -> 'x'"""
        if isinstance(v, list):
            return [loc(x) for x in v]
        if isinstance(v, str):
            if v == ".." or v.startswith("../"):
                return os.path.normpath(v[3:] or ".")
            if v == wroot or v.startswith(wroot + "/"):
                return os.path.normpath(os.path.relpath(v, wroot))
        return v

    for rel in M.RECIPES:
        train = "train" in os.path.basename(rel)
        real = flat(options.parse(os.path.join(ref_harness.REF_CODES, "options", rel), is_train=train))
        mine = flat(options.parse(FX.write_recipe(rel, wroot), is_train=train))
        for k in sorted(set(real) | set(mine)):
            if k != "_options_dir":
                assert loc(real.get(k)) == loc(mine.get(k)), (rel, k, real.get(k), mine.get(k))


def test_defaults_reject_off_path_kinds():
    from trainner_amd.options import defaults
    with pytest.raises(NotImplementedError):
        defaults.get_network_G_config("pan", 4, 128)
    with pytest.raises(NotImplementedError):
        defaults.get_network_D_config("multiscale", 4, 128, "rrdb_net")
    g = defaults.get_network_G_config("esrgan", 4, 128)
    assert g["type"] == "rrdb_net" and g["nb"] == 23 and g["upsample_mode"] == "upconv" and g["gaussian_noise"] is True
    d = defaults.get_network_D_config("discriminator_vgg", 4, 128, "rrdb_net")
    assert d["size"] == 128 and d["base_nf"] == 64 and d["arch"] == "ESRGAN"


def test_flat_params_and_checkpoint_roundtrip(tmp_path):
    from trainner_amd.models.modules.architectures.RRDBNet_arch import RRDBNet
    net = RRDBNet(3, 3, 64, 1)
    before = {k: v.clone() for k, v in net.state_dict().items()}
    fp = net.flat_params()
    assert fp.total >= sum(p.numel() for p in net.parameters())
    for p in net.parameters():
        assert p.grad is not None and p.grad.shape == p.shape
        assert fp.flat.data_ptr() <= p.data_ptr() < fp.flat.data_ptr() + 4 * fp.total
    for k, v in net.state_dict().items():
        assert torch.equal(v, before[k])
    # legacy-format checkpoint round trip through the reference-shaped keys
    path = tmp_path / "g.pth"
    torch.save({k: v.cpu() for k, v in net.state_dict().items()}, str(path), _use_new_zipfile_serialization=False)
    net2 = RRDBNet(3, 3, 64, 1)
    net2.load_state_dict(torch.load(str(path), weights_only=False))
    for (k1, v1), (k2, v2) in zip(net.state_dict().items(), net2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    # in-place load keeps the flat views attached
    fp2 = net.flat_params()
    net.load_state_dict(before)
    assert net.flat_params() is fp2 and fp2._consistent()


def test_fused_adam_state_layout_matches_torch_adam():
    from trainner_amd.models.modules.architectures.SRResNet_arch import SRResNet
    from trainner_amd.models.optimizers import FusedAdam
    net = SRResNet(3, 3, 32, 1)
    net.flat_params()
    opt = FusedAdam(list(net.parameters()), lr=1e-4)
    for g in opt.param_groups:
        opt._ensure_state(g)
    sd = opt.state_dict()
    ref = torch.optim.Adam(list(net.parameters()), lr=1e-4)
    assert set(sd["param_groups"][0]) >= {"lr", "betas", "eps", "weight_decay", "params"}
    assert sd["param_groups"][0]["params"] == ref.state_dict()["param_groups"][0]["params"]
    some = sd["state"][0]
    assert set(some) == {"step", "exp_avg", "exp_avg_sq"}
    opt.load_state_dict(sd)                                            # moments are re-homed into flat buffers
    p0 = opt.param_groups[0]["params"][0]
    assert opt.state[p0]["exp_avg"].shape == p0.shape


def test_fused_adam_resumes_a_checkpoint_with_lagging_step_counts(monkeypatch):
    """A torch.optim.Adam checkpoint may carry different `step` values inside one group (a parameter that was frozen for a while
    lags): FusedAdam resumes it with the most advanced count and a warning; TNR_STRICT_OPTIM_STATE=1 makes it an error."""
    from trainner_amd.models.modules.architectures.SRResNet_arch import SRResNet
    from trainner_amd.models.optimizers import FusedAdam
    net = SRResNet(3, 3, 32, 1)
    net.flat_params()
    opt = FusedAdam(list(net.parameters()), lr=1e-4)
    for g in opt.param_groups:
        opt._ensure_state(g)
    sd = opt.state_dict()
    for i, st in sd["state"].items():
        st["step"] = torch.tensor(7.0 if i else 3.0)                  # parameter 0 lags
    opt.load_state_dict(sd)
    assert opt._t[0] == 7
    monkeypatch.setenv("TNR_STRICT_OPTIM_STATE", "1")
    with pytest.raises(ValueError):
        opt.load_state_dict(sd)


def test_psnr_definition():
    from oracle import sr_oracle as O
    a = torch.zeros(1, 3, 16, 16)
    b = torch.full((1, 3, 16, 16), 10.0 / 255.0)
    import math
    assert abs(O.psnr_reference(a, b) - 20 * math.log10(255.0 / 10.0)) < 1e-9


def test_esrgan_checkpoint_key_converters():
    """old-arch <-> new-arch ESRGAN key renaming (networks.py:400-481 of the reference) for any trunk length:
    round trip restores keys, order and tensors; the new-arch names are the community ones; model_val applies
    the conversion only to rrdb_net / esrgan generators."""
    import torch
    from collections import OrderedDict
    from trainner_amd.models import networks as NW
    old = OrderedDict()
    old["model.0.weight"], old["model.0.bias"] = torch.zeros(4, 3, 3, 3), torch.zeros(4)
    for b in range(2):
        for r in (1, 2, 3):
            for c in range(1, 6):
                pre = "model.1.sub.%d.RDB%d.conv%d.0" % (b, r, c)
                old[pre + ".weight"], old[pre + ".bias"] = torch.full((1,), float(100 * b + 10 * r + c)), torch.zeros(1)
    old["model.1.sub.2.weight"], old["model.1.sub.2.bias"] = torch.ones(1), torch.ones(1)
    for i in (3, 6, 8, 10):
        old["model.%d.weight" % i], old["model.%d.bias" % i] = torch.full((1,), float(i)), torch.zeros(1)
    new = NW.normal2mod(old)
    assert list(new)[:2] == ["conv_first.weight", "conv_first.bias"]
    assert "RRDB_trunk.1.RDB3.conv5.weight" in new and "trunk_conv.bias" in new and "conv_last.weight" in new
    assert float(new["RRDB_trunk.1.RDB2.conv4.weight"]) == 124.0 and float(new["HRconv.weight"]) == 8.0
    back = NW.mod2normal(new)
    assert list(back) == list(old) and all(torch.equal(back[k], old[k]) for k in old)
    assert NW.mod2normal(old) is old and NW.normal2mod(new) is new            # already in the target layout
    opt = {"network_G": {"type": "esrgan"}}
    assert list(NW.model_val(opt, new, "G")) == list(old)
    assert NW.model_val({"network_G": {"type": "sr_resnet"}}, new, "G") is new
    assert NW.model_val(opt, new, "D") is new


def test_patch_extraction_and_recomposition_match_reference():
    """trainner_amd.dataops.common against fixtures produced by the reference's own extract_patches_2d /
    recompose_tensor (oracle/make_golden_patches.py): exact-grid, ragged, 25 % and 50 % overlap."""
    import os
    import torch
    from trainner_amd.dataops.common import extract_patches_2d, recompose_tensor
    fx = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "patches.pt"), weights_only=False)
    assert set(fx) == {"exact_grid", "ragged", "overlap_075", "overlap_05"}
    for name, c in fx.items():
        B, C, H, W, p, step, scale = c["spec"]
        pat = extract_patches_2d(c["img"], (p, p), step=[step, step], batch_first=True)
        assert pat.shape == c["patches"].shape and torch.equal(pat, c["patches"]), name
        other = extract_patches_2d(c["img"], (p, p), step=[step, step])
        assert torch.equal(other.permute(1, 0, 2, 3, 4), pat), name
        sr = c["sr"][:c["sr"].size(0) // B] if B > 1 else c["sr"]
        rec = recompose_tensor(sr, H, W, step=step, scale=scale)
        assert rec.shape == c["recomposed"].shape, name
        assert (rec - c["recomposed"]).abs().max().item() <= 1e-6, name


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No eager / CPU fallback: without the built C-ABI library every op raises HipEngineError naming the build step."""
    import pytest
    from trainner_amd import hip
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setenv("TNR_HIP_LIB", str(tmp_path / "nowhere" / "libtrainner_hip.so"))
    with pytest.raises(hip.HipEngineError, match="trainner_amd.build"):
        hip.load()
    monkeypatch.delenv("TNR_HIP_LIB")
    monkeypatch.setattr(hip, "_lib", None)
    assert hip.load() is not None           # the in-tree build is found again


def test_perceptual_net_refuses_silent_random_init(tmp_path):
    """without perceptual_opt.pretrained_path and without torchvision's ImageNet weights the reference
    would download them (perceptual.py:139-144); the engine must raise instead of training against random features,
    unless the config opts in (benchmarks / parity tests load their own weights)."""
    from trainner_amd import hip
    from trainner_amd.models import networks
    from trainner_amd.options import options
    try:
        import torchvision  # noqa: F401
        pytest.skip("torchvision present: pretrained weights may load")
    except ImportError:
        pass
    yml = ref_harness.esrgan_yaml(name="noinit", out_root=str(tmp_path), gpu_ids="[0]", nb=1, batch=2, crop=64, d_nf=16)
    opt = options.parse(yml, is_train=True)
    assert opt["train"]["perceptual_allow_random_init"] is True
    net = networks.define_F(opt)
    assert net.weights_source == "random-init"
    opt["train"]["perceptual_allow_random_init"] = None
    with pytest.raises(hip.HipEngineError, match="pretrained_path"):
        networks.define_F(opt)
    # a torchvision-layout state_dict on disk is honoured
    sd = {}
    for i, n in enumerate(net.names):
        if n.startswith("conv"):
            sd["features.%d.weight" % i] = torch.full_like(net.feature_net[n].weight, 0.5)
            sd["features.%d.bias" % i] = torch.zeros_like(net.feature_net[n].bias)
    p = tmp_path / "vgg19.pth"
    torch.save(sd, str(p))
    opt["train"]["perceptual_opt"] = {"pretrained_path": str(p)}
    net2 = networks.define_F(opt)
    assert net2.weights_source == str(p) and float(net2.feature_net["conv5_4"].weight.mean()) == 0.5


def test_optimizer_options_keep_falsy_values():
    """beta1_G: 0 / lr_D: 0 are legitimate: only a missing key takes the default (optimizers.py:74-134)."""
    from trainner_amd.models.modules.architectures.SRResNet_arch import SRResNet
    from trainner_amd.models.optimizers import config_optimizer
    from trainner_amd.options.options import dict_to_nonedict
    net = SRResNet(3, 3, 32, 1)
    o = config_optimizer(dict_to_nonedict({"beta1_G": 0, "lr_G": 0.0, "optim_G": "adam"}), "G", [net])
    assert o.param_groups[0]["betas"] == (0, 0.999) and o.param_groups[0]["lr"] == 0.0
    o = config_optimizer(dict_to_nonedict({}), "D", [net])
    assert o.param_groups[0]["betas"] == (0.9, 0.999) and o.param_groups[0]["lr"] == 1e-4


def test_noise_key_derivation_matches_its_restatement():
    """ops.noise_key (splitmix64 chain over seed / training forward / block) == oracle/gauss_noise.noise_key: the reference golden
    with the ESRGAN+ noise on (oracle/ref_harness._substitute_gaussian_draw) and the engine derive the SAME 64-bit key."""
    from oracle import gauss_noise
    from trainner_amd import ops
    for seed, call, block in ((0, 0, 0), (4242, 7, 11), (2 ** 63 + 5, 123456, 68), (99, 1, 3)):
        k = ops.noise_key(seed, call, block)
        assert (k & 0xFFFFFFFF, k >> 32) == gauss_noise.noise_key(seed, call, block)
    nz = ops.Noise(0.1, ops.noise_key(1, 2, 3), pos=1, pix0=2 ** 32 + 7)
    assert nz.pix0 == 7 and nz.at(2).pos == 2 and (nz.at(2).key0, nz.at(2).key1) == (nz.key0, nz.key1)
    # distinct blocks / forwards / seeds -> distinct keys
    keys = {ops.noise_key(s, c, b) for s in (0, 1) for c in range(4) for b in range(69)}
    assert len(keys) == 2 * 4 * 69


def test_bench_family_table_and_pmc_family_map():
    """bench.py's `roofline.families` (every matrix-core family of the step, auditable from the driver's line) and the kernel-name map the
    PMC tools use for the same families: every family ops.ConvProfile can emit has a PMC entry, the table's arithmetic is what it says,
    and counters are quoted only from records stamped with THIS tree's kernel-source hash."""
    import importlib
    import re
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    fams = importlib.import_module("pmc_families").FAMILIES
    src = open(os.path.join(root, "trainner_amd", "ops.py")).read()
    emitted = set(re.findall(r'"(conv_tile_[0-9a-z_]+|conv_chain|wgrad_tile)"', src))
    assert emitted and emitted <= set(fams), emitted - set(fams)
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    summ = {"conv_chain": {"launches": 276, "flops": 2 * 17.3e12, "ms": 2 * 81.0}, "wgrad_tile": {"launches": 152, "flops": 2 * 11.7e12, "ms": 2 * 55.0},
            "conv_thin": {"launches": 6, "flops": 1e11, "ms": 2.2}, "idle": {"launches": 1, "flops": 0.0, "ms": 1.0}}
    tab = bench.family_table(summ, 2, 419.4, "no_such_mode", 307.8)
    f = tab["families"]
    assert list(f) == ["conv_chain", "wgrad_tile", "conv_thin"]                      # largest first, zero-FLOP rows dropped
    assert f["conv_chain"]["launches_per_step"] == 138 and abs(f["conv_chain"]["tflops"] - 17.3e12 / 81.0e-3 / 1e12) < 0.01
    assert abs(f["conv_chain"]["frac"] - f["conv_chain"]["tflops"] / 419.4) < 1e-3 and abs(f["conv_chain"]["frac_of_sustained_mfma"] - f["conv_chain"]["tflops"] / 307.8) < 1e-3
    assert f["conv_thin"]["frac"] is None and "vector-ALU" in f["conv_thin"]["note"]
    assert f["wgrad_tile"]["mfma_busy"] is None and f["wgrad_tile"]["traffic"] is None      # no PMC record of that mode: never guessed
    assert abs(tab["families_mfma_ms_per_step"] - (81.0 + 55.0 + 1.1)) < 0.01
    # the committed records of the default mode belong to this tree's kernels (tools/pmc_stamp.py) and cover the families of the headline step
    busy, traffic = bench.pmc_family_records("bf16x3")
    if busy is not None:
        for name in ("conv_chain", "conv_tile_3x3", "wgrad_tile", "conv_tile_4x4s2", "conv_tile_dgrad4x4s2"):
            assert 0.0 < busy[name]["mfma_busy"] <= 1.0 and traffic[name]["hbm_bytes_per_launch"] > 0, name
