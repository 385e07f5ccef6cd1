"""N-rank SRModel step == 1-rank step (-m "not gpu"): two gloo processes, each running the engine's real host logic
(SRModel.optimize_parameters, the network kernel schedules, flat clip + Adam, dp.py's bucketed gradient averaging,
the relativistic global-mean exchange with its xworld backward coupling, the rank-0 start-state broadcast and the
global-batch sharding of feed_data) over the torch-CPU stand-in for the C ABI (tests/emul_backend.py), against ONE
process stepping the whole batch.  BatchNorm statistics are per replica in the reference (nn.DataParallel, no
SyncBN), so the single process takes them per half-batch (emul_backend.chunked_bn) -- everything else must agree
to fp32 round-off: logs, this rank's slice of fake_H, and the post-step G / D weights.
Reference: codes/models/losses.py:428-433,503-512 (global-batch relativistic means), sr_model.py:195-267.
"""
import os

import pytest
import torch
import torch.multiprocessing as mp

import emul_backend
from oracle import detrand, fixtures as FX, ref_harness

KW = dict(nb=1, batch=int(os.environ.get("TNR_TEST_DP_BATCH", "4")), crop=64, d_nf=16)     # (the env reaches the spawned ranks)
STEPS = 2
BN_BIAS = None


I2I_KW = dict(model="pix2pix", batch=4, crop=64, n_blocks=1, ngf=16, ndf=16, pixel_weight=100.0)


def _build_i2i(tmp, seed_g, seed_d):
    from trainner_amd.models import create_model
    from trainner_amd.options import options
    yml = ref_harness.i2i_yaml(name="dp_i2i", out_root=tmp, gpu_ids="[0]", **I2I_KW)
    model = create_model(options.parse(yml, is_train=True), verbose=False)
    model.netG.load_state_dict(detrand.fill_state_dict_({k: v.clone() for k, v in model.netG.state_dict().items()}, seed_g))
    model.netD.load_state_dict(detrand.fill_state_dict_({k: v.clone() for k, v in model.netD.state_dict().items()}, seed_d))
    return model


def _run_i2i(model):
    logs = []
    for s in range(1, STEPS + 1):
        A = detrand.uniform((I2I_KW["batch"], 3, 64, 64), 170 + s, -1.0, 1.0)     # every rank is fed the GLOBAL batch
        B = detrand.uniform((I2I_KW["batch"], 3, 64, 64), 180 + s, -1.0, 1.0)
        model.feed_data({"A": A, "B": B, "A_path": ["a"] * 4})
        model.optimize_parameters(s)
        logs.append(model.get_current_log())
    return dict(logs=logs, fake=model.fake_B.detach().clone(),
                g={k: v.detach().clone() for k, v in model.netG.state_dict().items()},
                d={k: v.detach().clone() for k, v in model.netD.state_dict().items()})


def _build(tmp, seed_g, seed_d, gaussian=False, accumulate=False):
    from trainner_amd.models import create_model
    from trainner_amd.options import options
    yml = ref_harness.esrgan_yaml(name="dp_case", out_root=tmp, gpu_ids="[0]", gaussian=gaussian, **KW)
    if accumulate:               # virtual_batch_size = 2 x batch_size: two calls per optimizer step (base_model.py:722-734)
        txt = open(yml).read()
        assert "virtual_batch_size: %d" % KW["batch"] in txt
        with open(yml, "w") as fh:
            fh.write(txt.replace("virtual_batch_size: %d" % KW["batch"], "virtual_batch_size: %d" % (2 * KW["batch"])))
    opt = options.parse(yml, is_train=True)
    model = create_model(opt, verbose=False)
    if gaussian:                 # ESRGAN+ noise: one seed on every rank (train.py seeds all of them alike); each rank draws the
        model.netG.noise_seed = 99      # field of ITS samples of the global batch (SRModel.feed_data sets noise_sample0)
    if seed_g is not None:
        model.netG.load_state_dict(detrand.fill_state_dict_({k: v.clone() for k, v in model.netG.state_dict().items()}, seed_g))
        model.netD.load_state_dict(detrand.fill_state_dict_({k: v.clone() for k, v in model.netD.state_dict().items()}, seed_d))
    netF = [l["function"].network for l in model.generatorlosses.loss_list if "fea" in l["name"]][0]
    sd = netF.state_dict()
    sd.update(FX.vgg_state(77))
    netF.load_state_dict(sd)
    return model


def _run(model, rank=None, world=1, steps=STEPS):
    logs = []
    for s in range(1, steps + 1):
        LR, HR = detrand.synthetic_pair(KW["batch"], KW["crop"], 70 + s)     # every rank is fed the GLOBAL batch
        model.feed_data({"LR": LR, "HR": HR})
        model.optimize_parameters(s)
        logs.append(model.get_current_log())
    return dict(logs=logs, fake=model.fake_H.detach().clone(),
                g={k: v.detach().clone() for k, v in model.netG.state_dict().items()},
                d={k: v.detach().clone() for k, v in model.netD.state_dict().items()})


def _worker(rank, world, port, tmp, q, kind="sr", gaussian=False, accumulate=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    try:
        torch.set_num_threads(4 if world <= 2 else 1)
        emul_backend.install()
        from trainner_amd import dp as dpmod
        dpmod.BUCKET_FLOATS = 100_000             # several buckets per network, fired from inside backward
        # rank 1 holds DIFFERENT weights (as after a per-rank init RNG): sync_replicas must bring it to rank 0's
        # (SRModel's constructor calls it as its last act; here the seeded load happens after construction)
        build, run = (_build, lambda m: _run(m, rank, world, 2 * STEPS if accumulate else STEPS)) if kind == "sr" else (_build_i2i, _run_i2i)
        kw = dict(gaussian=gaussian, accumulate=accumulate) if kind == "sr" else {}

        model = build(os.path.join(tmp, "r%d" % rank), 101 if rank == 0 else 555 + rank, 202 if rank == 0 else 666 + rank, **kw)
        assert model.dp.active and model.dp.world_size == world
        model.sync_replicas()
        out = run(model)
        path = os.path.join(tmp, "rank%d.pt" % rank)
        torch.save(out, path)                     # tensors travel by file (the worker exits before the parent reads)
        q.put((rank, path))
    except Exception as e:                           # pragma: no cover
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))
    finally:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


def test_eight_rank_step_equals_single_process(tmp_path, monkeypatch):
    """World 8 -- the node the scaling run uses (BASELINE configs[2]: batch 128 global = 16 per rank; here 16 global = 2 per rank at
    tiny shapes): eight gloo ranks against one process for the losses, each rank's slice of fake_H, the post-step weights and the
    ESRGAN+ noise field (rank r draws samples [2 r, 2 r + 2) of the global batch's field, `noise_sample0`)."""
    monkeypatch.setenv("TNR_TEST_DP_BATCH", "16")
    monkeypatch.setitem(KW, "batch", 16)
    test_two_rank_step_equals_single_process(tmp_path, monkeypatch, "gaussian", world=8)


@pytest.mark.parametrize("variant", ["plain", "gaussian", "accumulate", "overlap_g"])
def test_two_rank_step_equals_single_process(tmp_path, monkeypatch, variant, world=2):
    """overlap_g: TNR_DP_OVERLAP_G=1 -- the generator's gradient buckets leave from inside its backward (opt-in since round 5; the
    default, which every other variant runs, sends them at its optimizer step).
    gaussian: with the ESRGAN+ noise on (the reference's default) -- every rank must draw the field of its own samples of the
    GLOBAL batch (`noise_pix0`), or the two halves of fake_H would not be the single process's.
    accumulate: virtual_batch_size = 2 x batch_size -- gradients accumulate locally over two calls and are exchanged once, before
    the optimizer step (the buckets do not leave from inside backward then); four calls = two optimizer steps."""
    gaussian, accumulate = variant == "gaussian", variant == "accumulate"
    if variant == "overlap_g":
        monkeypatch.setenv("TNR_DP_OVERLAP_G", "1")          # (the environment reaches the spawned ranks)
    else:
        monkeypatch.delenv("TNR_DP_OVERLAP_G", raising=False)
    steps = 2 * STEPS if accumulate else STEPS
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + (os.getpid() % 2000) + {"plain": 0, "gaussian": 7, "accumulate": 13, "overlap_g": 19}[variant]
    procs = [ctx.Process(target=_worker, args=(r, world, port + world, str(tmp_path), q, "sr", gaussian, accumulate)) for r in range(world)]
    for p in procs:
        p.start()
    # the single-process comparator runs meanwhile: full batch, per-half-batch BatchNorm statistics
    emul_backend.install(monkeypatch)
    from trainner_amd import ops
    f, b = emul_backend.chunked_bn(world)
    monkeypatch.setattr(ops, "bn_train_fwd", f)
    monkeypatch.setattr(ops, "bn_train_bwd", b)
    torch.set_num_threads(4 if world <= 2 else 1)
    single = _build(str(tmp_path / "one"), 101, 202, gaussian, accumulate)
    assert single.netG.noise_sigma == (0.1 if gaussian else 0.0) and single.accumulations == (2 if accumulate else 1)
    one = _run(single, steps=steps)
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for r in range(world):
        assert res[r].endswith(".pt"), res[r]
        res[r] = torch.load(res[r], weights_only=False)
    per = KW["batch"] // world
    lr_steps = 1e-4 * STEPS
    shadow = FX.bn_shadowed_biases([(k, None) for k in one["d"]])
    for r in range(world):
        out = res[r]
        for s in range(steps):
            for k, v in one["logs"][s].items():
                # every log entry is a global-batch quantity on every rank (G losses are all-reduced means)
                # (after the first update the raw mean logits D_real / D_fake carry the +-lr walk of the BatchNorm-shadowed conv
                #  biases, whose true gradient is exactly zero: tests/test_gpu_step.check_logs uses the same looser bound for them)
                tol = 2e-3 if (s >= (2 if accumulate else 1) and k in ("D_real", "D_fake")) else 5e-5
                assert abs(out["logs"][s][k] - v) <= tol * abs(v) + 1e-7, (r, s, k, out["logs"][s][k], v)
        diff = (out["fake"] - one["fake"][r * per:(r + 1) * per]).abs().max().item()
        assert diff <= 1e-5, ("fake_H", r, diff)
        for name, mine, ref in (("G", out["g"], one["g"]), ("D", out["d"], one["d"])):
            worst, tot, cnt = (0.0, None), 0.0, 0
            for k, v in ref.items():
                if k in shadow or not v.is_floating_point():
                    continue
                if ".running_" in k:
                    if r == 0:               # replica 0's running statistics are the ones the reference keeps
                        # (running means carry the +-lr walk of the BN-shadowed conv biases: atol = lr * steps)
                        assert torch.allclose(mine[k], v, rtol=1e-4, atol=2.5e-4), (k,)
                    continue
                d = (mine[k] - v).abs() / lr_steps
                tot, cnt = tot + d.sum().item(), cnt + d.numel()
                if d.max().item() > worst[0]:
                    worst = (d.max().item(), k)
            # Adam's first steps are sign-like: an element whose gradient is at rounding-noise level can move
            # +-lr either way, so the mean is bounded tightly and the worst element loosely
            assert tot / cnt < 2e-3 and worst[0] < 1.0, (name, r, worst, tot / cnt)
    # all replicas hold identical weights after the steps (same reduced gradients, same Adam)
    for r in range(1, world):
        for k, v in res[0]["g"].items():
            assert torch.equal(v, res[r]["g"][k]), k
        for k, v in res[0]["d"].items():
            if ".running_" not in k and v.is_floating_point():
                assert torch.equal(v, res[r]["d"][k]), k


def test_two_rank_pix2pix_step_equals_single_process(tmp_path, monkeypatch):
    """The same equivalence for Pix2PixModel (SURVEY.md 8(f)3 under 8(e)): conditional PatchGAN with per-replica BatchNorm
    statistics, standard-form GAN criterion (no cross-rank coupling: every loss is a mean over the local shard, gradient
    averaging makes it the global mean), InstanceNorm generator, D step before the G step, sharded feed of {'A','B'}."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33000 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), q, "pix2pix")) for r in range(2)]
    for p in procs:
        p.start()
    emul_backend.install(monkeypatch)
    from trainner_amd import ops
    f, b = emul_backend.chunked_bn(2)
    monkeypatch.setattr(ops, "bn_train_fwd", f)
    monkeypatch.setattr(ops, "bn_train_bwd", b)
    torch.set_num_threads(4)
    one = _run_i2i(_build_i2i(str(tmp_path / "one"), 101, 202))
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for r in (0, 1):
        assert res[r].endswith(".pt"), res[r]
        res[r] = torch.load(res[r], weights_only=False)
    per, lr_steps = I2I_KW["batch"] // 2, 2e-4 * STEPS
    shadow = FX.norm_shadowed_biases([(k, None) for k in one["g"]], "instance")
    for r in (0, 1):
        out = res[r]
        for s in range(STEPS):
            assert list(out["logs"][s].keys()) == list(one["logs"][s].keys())
            for k, v in one["logs"][s].items():
                assert abs(out["logs"][s][k] - v) <= 5e-5 * abs(v) + 2e-6, (r, s, k, out["logs"][s][k], v)
        diff = (out["fake"] - one["fake"][r * per:(r + 1) * per]).abs().max().item()
        assert diff <= 2e-5, ("fake_B", r, diff)
        for name, mine, ref in (("G", out["g"], one["g"]), ("D", out["d"], one["d"])):
            tot, cnt, worst = 0.0, 0, 0.0
            for k, v in ref.items():
                if k in shadow or not v.is_floating_point() or ".running_" in k:
                    continue
                d = (mine[k] - v).abs() / lr_steps
                tot, cnt, worst = tot + d.sum().item(), cnt + d.numel(), max(worst, d.max().item())
            assert tot / cnt < 2e-3 and worst < 1.0, (name, r, worst, tot / cnt)
    for k, v in res[0]["g"].items():
        assert torch.equal(v, res[1]["g"][k]), k
