"""Bit-reproducible pseudo-random fills (integer arithmetic only).

TEST INFRASTRUCTURE ONLY.  torch's CPU RNG stream is not guaranteed identical across
hosts/builds, and the GPU box cannot read the reference's RNG-initialised weights, so every
fixture (tests/golden/*) derives its inputs and initial weights from this generator: the
same call here, in `oracle/make_golden.py` (real reference) and in the `-m gpu` tests gives
the same float32 bits everywhere.
"""
import math
import torch

_M64 = (1 << 64) - 1


def _mix(x):
    # splitmix64 finaliser on int64 tensors (wrap-around arithmetic is exact in int64)
    x = x ^ (x >> 30 & 0x3FFFFFFFF)
    x = x * -4658895280553007687  # 0xBF58476D1CE4E5B9 as signed
    x = x ^ (x >> 27 & 0x1FFFFFFFFF)
    x = x * -7723592293110705685  # 0x94D049BB133111EB as signed
    x = x ^ (x >> 31 & 0x1FFFFFFFF)
    return x


def uniform01(numel, seed):
    """float32 tensor of `numel` values in [0, 1), a pure function of (index, seed)."""
    idx = torch.arange(numel, dtype=torch.int64)
    s = (int(seed) * 0x9E3779B97F4A7C15 + 0x1234567) & _M64
    if s >= 1 << 63:
        s -= 1 << 64
    x = _mix(idx * 2654435761 + s)
    top24 = (x >> 40) & 0xFFFFFF
    return top24.to(torch.float32) / float(1 << 24)


def uniform(shape, seed, lo=-1.0, hi=1.0):
    n = 1
    for d in shape:
        n *= int(d)
    return (uniform01(n, seed) * (hi - lo) + lo).reshape(tuple(shape))


def fill_state_dict_(sd, seed, gain=0.5, bias_amp=0.1):
    """Overwrite every tensor of a network state_dict deterministically (in place).

    conv/linear weights: U(-a, a), a = gain*sqrt(6/fan_in); biases U(-bias_amp, bias_amp);
    BatchNorm weight 1 +- 0.2, bias +- bias_amp; running_mean 0 / running_var 1 / counters 0
    (what a freshly built nn.BatchNorm2d holds, torch.nn.modules.batchnorm).
    """
    with torch.no_grad():
        for i, (k, v) in enumerate(sd.items()):
            s = seed * 1000 + i
            if k.endswith("num_batches_tracked"):
                v.zero_()
            elif k.endswith("running_mean"):
                v.zero_()
            elif k.endswith("running_var"):
                v.fill_(1.0)
            elif v.dim() >= 2:
                fan_in = v[0].numel()
                a = gain * math.sqrt(6.0 / fan_in)
                v.copy_(uniform(v.shape, s, -a, a))
            elif k.endswith("weight"):  # BatchNorm scale
                v.copy_(uniform(v.shape, s, 0.8, 1.2))
            else:
                v.copy_(uniform(v.shape, s, -bias_amp, bias_amp))
    return sd


def synthetic_pair(n, hr_size, seed, scale=4):
    """HR in [0,1) and LR = avg_pool(HR, scale) (SURVEY.md 8(d) input recipe, with the
    reproducible generator instead of torch.rand)."""
    hr = uniform((n, 3, hr_size, hr_size), seed, 0.0, 1.0)
    lr = torch.nn.functional.avg_pool2d(hr, scale)
    return lr.contiguous(), hr.contiguous()
