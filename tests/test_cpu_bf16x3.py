"""The arithmetic of TNR_MMA_BF16X3 (include/trainner_hip.h, csrc/conv_body.h: tnr_split_bf16x3), restated in torch on the CPU
(-m "not gpu"): the three-way bf16 split of an fp32 value is EXACT, every partial product of two bf16 values is exact in fp32, and
the six partial products the kernels keep reproduce the fp32 product to within fp32 rounding -- i.e. the mode is fp32 arithmetic
on the bf16 matrix core, not a reduced-precision mode.  (The kernels themselves are checked against fp64 on the GPU:
tests/test_gpu_kernels.py::test_bf16x3_split_operand_mode.)"""
import torch


def split3(x):
    """hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid); every residual is an fp32 subtraction (what the kernels do)"""
    hi = x.to(torch.bfloat16).to(torch.float32)
    r1 = x - hi
    mid = r1.to(torch.bfloat16).to(torch.float32)
    r2 = r1 - mid
    lo = r2.to(torch.bfloat16).to(torch.float32)
    return hi, mid, lo, r2


def samples(n, seed):
    g = torch.Generator().manual_seed(seed)
    mant = torch.rand(n, generator=g, dtype=torch.float64) + 1.0
    expo = torch.randint(-20, 20, (n,), generator=g).to(torch.float64)
    sign = torch.where(torch.rand(n, generator=g) < 0.5, -1.0, 1.0).to(torch.float64)
    return (sign * mant * torch.pow(torch.tensor(2.0, dtype=torch.float64), expo)).to(torch.float32)


def test_three_bf16_values_hold_an_fp32_value_exactly():
    x = samples(200000, 1)
    hi, mid, lo, r2 = split3(x)
    assert torch.equal(lo, r2)                                                    # the last residual fits bf16: nothing is lost
    assert torch.equal((hi.double() + mid.double() + lo.double()).to(torch.float32), x)
    assert torch.equal(hi.double() + mid.double() + lo.double(), x.double())      # exactly, not just after rounding
    # magnitudes: each split carries 8 more bits than the previous one
    nz = x != 0
    assert (mid[nz].abs() <= x[nz].abs() * 2.0 ** -8).all() and (lo[nz].abs() <= x[nz].abs() * 2.0 ** -16).all()


def test_partial_products_are_exact_in_fp32():
    a, b = samples(100000, 2), samples(100000, 3)
    for pa in split3(a)[:3]:
        for pb in split3(b)[:3]:
            p32 = pa * pb                                     # 8 x 8 significand bits: exact in fp32's 24
            assert torch.equal(p32.double(), pa.double() * pb.double())


def test_six_kept_products_reproduce_the_fp32_product_within_fp32_rounding():
    a, b = samples(200000, 4), samples(200000, 5)
    ah, am, al, _ = split3(a)
    bh, bm, bl, _ = split3(b)
    exact = a.double() * b.double()
    kept = (ah.double() * bl.double() + al.double() * bh.double() + am.double() * bm.double()
            + ah.double() * bm.double() + am.double() * bh.double() + ah.double() * bh.double())
    dropped = am.double() * bl.double() + al.double() * bm.double() + al.double() * bl.double()
    assert torch.equal(kept + dropped, exact)                 # the nine partial products ARE the product
    rel = ((kept - exact).abs() / exact.abs()).max().item()
    assert rel <= 2.0 ** -23, rel                             # the three dropped terms: below one fp32 ulp of the product
    fp32_rounding = ((a * b).double() - exact).abs() / exact.abs()
    assert fp32_rounding.max().item() <= 2.0 ** -24 * 1.0000001
    # a dot product accumulated in fp32 from the six kept terms is as close to the fp64 result as the plain fp32 dot product
    K = 1152
    A, B = samples(64 * K, 6).view(64, K), samples(64 * K, 7).view(64, K)
    A, B = A / A.abs().max(), B / B.abs().max()
    ref = (A.double() * B.double()).sum(1)
    plain = torch.zeros(64)
    split = torch.zeros(64)
    Ah, Am, Al, _ = split3(A)
    Bh, Bm, Bl, _ = split3(B)
    for k in range(K):
        plain = plain + A[:, k] * B[:, k]
        for pa, pb in ((Ah, Bl), (Al, Bh), (Am, Bm), (Ah, Bm), (Am, Bh), (Ah, Bh)):
            split = split + pa[:, k] * pb[:, k]
    e_plain = (plain.double() - ref).abs().max().item()
    e_split = (split.double() - ref).abs().max().item()
    assert e_split <= 4.0 * e_plain + 1e-7 * ref.abs().max().item(), (e_split, e_plain)
