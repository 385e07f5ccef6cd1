"""Worker of test_gpu_step.py::test_dp_collectives_single_rank: runs two ESRGAN steps of the engine and
prints the log dicts as JSON.  Under `torch.distributed.run --nproc-per-node 1` with TNR_DP_SELFTEST=1
every data-parallel collective (bucketed gradient all-reduce on the side stream, relativistic-sum
exchange) really executes through RCCL in a 1-rank group; the numbers must equal the plain run."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import detrand, fixtures as FX, ref_harness  # noqa: E402


def main():
    from trainner_amd import dp as dpmod
    from trainner_amd.models import create_model
    from trainner_amd.options import options
    dpmod.BUCKET_FLOATS = 200000          # several buckets even for the small test network
    kw = dict(nb=1, batch=2, crop=64, d_nf=16)
    tmp = tempfile.mkdtemp(prefix="tnr_dpself_")
    opt = options.parse(ref_harness.esrgan_yaml(name="dpself", out_root=tmp, gpu_ids="[0]", **kw), is_train=True)
    model = create_model(opt, verbose=False)
    g = detrand.fill_state_dict_({k: v.detach().cpu().clone() for k, v in model.netG.state_dict().items()}, 101)
    d = detrand.fill_state_dict_({k: v.detach().cpu().clone() for k, v in model.netD.state_dict().items()}, 202)
    model.netG.load_state_dict(g)
    model.netD.load_state_dict(d)
    netF = [l["function"].network for l in model.generatorlosses.loss_list if "fea" in l["name"]][0]
    sd = netF.state_dict()
    sd.update(FX.vgg_state(77))
    netF.load_state_dict(sd)
    logs = []
    for s in (1, 2):
        LR, HR = detrand.synthetic_pair(2, 64, 50 + s)
        model.feed_data({"LR": LR, "HR": HR})
        model.optimize_parameters(s)
        logs.append(model.get_current_log())
    w = model.netG.state_dict()["model.0.weight"].detach().cpu().flatten()[:8].tolist()
    print("DPSELF " + json.dumps({"logs": logs, "w": w, "active": bool(model.dp.active), "abi": model.dp._abi is not None}))
    model.dp.finalize()


if __name__ == "__main__":
    main()
