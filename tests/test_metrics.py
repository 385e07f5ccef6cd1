"""Validation metrics: the device-side tensor2np / PSNR / SSIM against fixtures produced by the
REFERENCE's own tensor2np + calculate_psnr + calculate_ssim (tests/golden/metrics.pt, oracle/make_golden_metrics.py; the reference's
SSIM code ran with its two OpenCV calls served by scipy: oracle/stubs/cv2) and against the numpy restatement
oracle/metrics_oracle.py.  uint8 images must match byte for byte, PSNR to 1e-9 dB, SSIM to 1e-9."""
import os

import numpy as np
import pytest
import torch

from oracle import metrics_oracle as MO

FX = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metrics.pt"), weights_only=False)


def test_oracle_metrics_match_reference_fixtures():
    for name, c in FX.items():
        for n, im in enumerate(c["images"]):
            a = MO.tensor2np(c["sr"][n].numpy(), denormalize=c["denormalize"])
            b = MO.tensor2np(c["hr"][n].numpy(), denormalize=c["denormalize"])
            assert np.array_equal(a, im["sr_u8"].numpy()) and np.array_equal(b, im["hr_u8"].numpy()), name
            assert abs(MO.calculate_psnr(a, b, 4) - im["psnr4"]) < 1e-12 and abs(MO.calculate_psnr(a, b, 0) - im["psnr0"]) < 1e-12
            # the reference's own ssim() / calculate_ssim() (scipy serving getGaussianKernel / filter2D, then [5:-5, 5:-5])
            assert abs(MO.calculate_ssim(a, b, 4) - im["ssim4"]) < 1e-12 and abs(MO.calculate_ssim(a, b, 0) - im["ssim0"]) < 1e-12
    # SSIM restatement: identical images -> 1, and the window is cv2.getGaussianKernel(11, 1.5)'s published formula
    a = FX["rgb_unit"]["images"][0]["sr_u8"].numpy()
    assert abs(MO.calculate_ssim(a, a, 4) - 1.0) < 1e-12
    w = MO.gaussian_window()
    assert abs(w.sum() - 1.0) < 1e-15 and abs(w[5, 5] / w[5, 4] - np.exp(1 / 4.5)) < 1e-12


@pytest.mark.gpu
def test_device_tensor2np_psnr_ssim():
    from trainner_amd.dataops.common import tensor2np
    from trainner_amd.utils import metrics as M
    for name, c in FX.items():
        sr, hr = c["sr"].cuda(), c["hr"].cuda()
        for n, im in enumerate(c["images"]):
            a = tensor2np(sr[n], denormalize=c["denormalize"])
            b = tensor2np(hr[n], denormalize=c["denormalize"])
            assert a.dtype == torch.uint8 and a.is_cuda and tuple(a.shape) == tuple(im["sr_u8"].shape)
            assert torch.equal(a.cpu(), im["sr_u8"]) and torch.equal(b.cpu(), im["hr_u8"]), name      # byte for byte
            assert abs(M.calculate_psnr(a, b, 4) - im["psnr4"]) < 1e-9 and abs(M.calculate_psnr(a, b, 0) - im["psnr0"]) < 1e-9
            want = MO.calculate_ssim(im["sr_u8"].numpy(), im["hr_u8"].numpy(), 4)
            assert abs(M.calculate_ssim(a, b, 4) - want) < 1e-9, (name, M.calculate_ssim(a, b, 4), want)
            assert abs(M.calculate_ssim(a, b, 4) - im["ssim4"]) < 1e-9 and abs(M.calculate_ssim(a, b, 0) - im["ssim0"]) < 1e-9    # the reference's
        # batch form: every image in one launch; remove_batch keeps image 0 like the reference
        A = tensor2np(sr, remove_batch=False, denormalize=c["denormalize"])
        B = tensor2np(hr, remove_batch=False, denormalize=c["denormalize"])
        assert tuple(A.shape) == (sr.shape[0],) + tuple(c["images"][0]["sr_u8"].shape)
        assert torch.equal(tensor2np(sr, denormalize=c["denormalize"]).cpu(), c["images"][0]["sr_u8"])
        md = M.MetricsDict("psnr,ssim")
        last = md.calculate_metrics(A, B, crop_size=4)
        assert md.count == sr.shape[0] and abs(last["psnr"] - c["images"][-1]["psnr4"]) < 1e-9
        avg = md.get_averages()
        assert abs(avg["psnr"] - np.mean([i["psnr4"] for i in c["images"]])) < 1e-9 and md.count == 0
        # numpy inputs (what train.py hands over) are accepted as well
        assert abs(M.calculate_psnr(c["images"][0]["sr_u8"].numpy(), c["images"][0]["hr_u8"].numpy(), 4) - c["images"][0]["psnr4"]) < 1e-9
    same = tensor2np(FX["gray"]["sr"].cuda())
    assert M.calculate_psnr(same, same, 4) == float("inf") and abs(M.calculate_ssim(same, same, 4) - 1.0) < 1e-12
    with pytest.raises(NotImplementedError):
        M.MetricsDict("psnr,lpips")


@pytest.mark.gpu
def test_validation_psnr_of_model_output(tmp_path):
    """SRModel.test() -> get_current_visuals -> device tensor2np -> MetricsDict(crop = scale), the validation block of
    train.py:331-372, against the oracle's RRDBNet forward scored with the reference's numpy PSNR."""
    import test_gpu_step as TS
    from oracle import detrand, sr_oracle as O
    from trainner_amd.dataops.common import tensor2np
    from trainner_amd.utils.metrics import MetricsDict
    opt, model = TS.build_engine_model(dict(nb=1, batch=1, crop=64, d_nf=16), tmp_path)
    g = detrand.fill_state_dict_({k: v.detach().cpu().clone() for k, v in model.netG.state_dict().items()}, 303)
    model.netG.load_state_dict(g)
    LR, HR = detrand.synthetic_pair(1, 64, 77)
    model.feed_data({"LR": LR, "HR": HR})
    model.test()
    vis = model.get_current_visuals()
    md = MetricsDict("psnr,ssim")
    got = md.calculate_metrics(tensor2np(vis["SR"].cuda()), tensor2np(vis["HR"].cuda()), crop_size=opt["scale"])
    ref_sr = O.rrdbnet_forward(LR, g, nb=1)[0].detach().numpy()
    a, b = MO.tensor2np(ref_sr), MO.tensor2np(HR[0].numpy())
    assert abs(got["psnr"] - MO.calculate_psnr(a, b, 4)) < 0.05          # north star: PSNR within 0.05 dB
    assert abs(got["ssim"] - MO.calculate_ssim(a, b, 4)) < 1e-3
