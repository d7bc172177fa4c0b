"""GPU: the split-bf16 form of the K-deep contractions (lcr_gemm_f32_bsplit: fp32 operands as three bf16 terms, six cross products on the
bf16 matrix cores, fp32 accumulation) against an fp64 product of the same fp32 inputs, next to the fp32-MFMA kernel (lcr_gemm_f32)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _split(b):
    from lcrnet_amd import functional as F
    return F.split_bf16x3(b)


def _planar(x):
    from lcrnet_amd import _lib
    n = x.numel()
    planes = torch.empty(3 * n, dtype=torch.int16, device=x.device)
    _lib.check(_lib.lib().lcr_split_bf16x3(_lib.ptr(x), n, _lib.ptr(planes), _lib.stream_ptr(x.device)), "lcr_split_bf16x3")
    return planes


def test_three_bf16_terms_are_the_fp32_number():
    """lcr_split_bf16x3: h1 + h2 + h3 == x exactly (fp32 additions in that order are exact too), h1 = rn_bf16(x); normal range, tiny and
    huge magnitudes, signed zeros, odd length."""
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(100001, device="cuda", generator=g) * torch.exp(torch.randn(100001, device="cuda", generator=g) * 8)
    x[:6] = torch.tensor([0.0, -0.0, 1.0, -1.0, 3.0e38, 1.0e-30], device="cuda")
    pl = _planar(x).view(3, -1).view(torch.bfloat16).float()
    assert torch.equal(pl[0], x.to(torch.bfloat16).float())
    assert torch.equal((pl[0] + pl[1]) + pl[2], x)
    assert (pl[1].abs() <= pl[0].abs() * 2.0 ** -8 + 1e-45).all() and (pl[2].abs() <= pl[0].abs() * 2.0 ** -16 + 1e-45).all()
    # the tiled form the GEMM takes (lcr_split_bf16x3_tiles) holds the same terms: [N,K] with N not a multiple of 64
    from lcrnet_amd import functional as F
    w = x[:100 * 352].view(100, 352).contiguous()
    t = F.unsplit_bf16x3(F.split_bf16x3(w))
    assert t.shape == (3, 100, 352) and torch.equal(t, _planar(w).view(3, 100, 352).view(torch.bfloat16).float())


@pytest.mark.parametrize("M,N,K,rowdiv,groups", [(5000, 96, 352, True, 0), (6479, 256, 3840, True, 32), (19061, 128, 512, False, 32), (700, 64, 480, True, 32),
                                                 (63, 32, 32, False, 32), (8192, 1024, 1024, False, 0)])
def test_split_gemm_against_fp64(M, N, K, rowdiv, groups):
    """Error vs fp64 of the split form is of the size of the fp32-MFMA kernel's own (both bounded here by 3e-6 of the largest output;
    measured ~1e-7): the six kept products carry every a.b to 2^-23, accumulation is fp32 in both.  Epilogue (count division, bias,
    GroupNorm sums per segment) shared with lcr_gemm_f32: statistics agree to 1e-6 (sums relative to sqrt(n * sum of squares))."""
    from lcrnet_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    a = torch.randn(M, K, device=dev, generator=g) * (torch.rand(M, 1, device=dev, generator=g) * 4 + 0.01)
    if rowdiv:
        a = torch.where(torch.rand(M, K, device=dev, generator=g) < 0.35, torch.zeros_like(a), a.abs())
    b = torch.randn(N, K, device=dev, generator=g) * 0.05
    bias = torch.randn(N, device=dev, generator=g)
    div = (torch.rand(M, device=dev, generator=g) * 40 + 1).floor() if rowdiv else None
    S = 3
    seg = torch.tensor([M // 3, M // 3 + 5, M - 2 * (M // 3) - 5], dtype=torch.int64, device=dev)
    planes = _split(b)
    outs, stats = [], []
    for which in (0, 1):
        c = torch.full((M, N), float("nan"), device=dev)
        st = torch.zeros(8, S, max(groups, 1), 2, dtype=torch.float64, device=dev)
        args = (_lib.ptr(bias), _lib.ptr(div) if rowdiv else None, _lib.ptr(seg) if groups else None, S, groups, _lib.ptr(st) if groups else None,
                _lib.stream_ptr(dev))
        if which == 0:
            _lib.check(L.lcr_gemm_f32(_lib.ptr(a), _lib.ptr(b), _lib.ptr(c), M, N, K, 0, 1, *args), "lcr_gemm_f32")
        else:
            _lib.check(L.lcr_gemm_f32_bsplit(_lib.ptr(a), _lib.ptr(planes), _lib.ptr(c), M, N, K, *args), "lcr_gemm_f32_bsplit")
        outs.append(c)
        stats.append(st.sum(0))
    torch.cuda.synchronize()
    ref = a.double() @ b.double().t()
    if rowdiv:
        ref = ref / div.double()[:, None]
    ref = ref + bias.double()
    scale = ref.abs().max().item()
    e0, e1 = (outs[0].double() - ref).abs().max().item() / scale, (outs[1].double() - ref).abs().max().item() / scale
    print("M=%d N=%d K=%d: max err / max |ref|: fp32 MFMA %.2e, split %.2e" % (M, N, K, e0, e1))
    assert torch.isfinite(outs[1]).all()
    assert e1 < 3e-6 and e1 < max(4 * e0, 4e-7)
    if groups:
        # sums against their Cauchy-Schwarz bound sqrt(n * sum of squares) (a group's sum can cancel to ~0), sums of squares relative
        n_el = seg.double()[:, None] * (N // groups)
        rel = max(((stats[0][..., 0] - stats[1][..., 0]).abs() / (n_el * stats[0][..., 1]).sqrt()).max().item(),
                  ((stats[0][..., 1] - stats[1][..., 1]).abs() / stats[0][..., 1]).max().item())
        assert rel < 1e-6, rel


def test_split_gemm_rejects_bad_shapes():
    from lcrnet_amd import _lib
    a = torch.zeros(64, 40, device="cuda")
    pl = torch.zeros(3 * 64 * 64, dtype=torch.int16, device="cuda")
    c = torch.empty(64, 64, device="cuda")
    rc = _lib.lib().lcr_gemm_f32_bsplit(_lib.ptr(a), _lib.ptr(pl), _lib.ptr(c), 64, 64, 40, None, None, None, 0, 0, None, _lib.stream_ptr(a.device))
    assert rc != 0                                            # K % 32 != 0


def _both(a, b):
    """(fp32-MFMA result, split result) of A[M,K] . B[N,K]^T, no epilogue"""
    from lcrnet_amd import functional as F
    c0, _ = F.gemm(a, b, trans_b=True)
    c1, _ = F.gemm_bsplit(a, F.split_bf16x3(b))
    torch.cuda.synchronize()
    return c0, c1


def test_split_gemm_worst_cases():
    """What the split form does at the edges of fp32, next to the fp32-MFMA kernel on the same operands (the form is the default for the
    K-deep contractions since round 5, so its domain is stated and pinned here):
      1. dynamic range 2^40 inside every row of A (and 2^20 inside B): the error against fp64 is bounded RELATIVE TO sum |a.b| (the only
         bound any fp32 accumulation has) by 2e-6 and is not above the fp32 kernel's by more than rounding noise;
      2. subnormal operands: the split flushes them (reconstruction error below 2^-126 per element, measured); products far below the fp32
         range vanish in both forms, and subnormal entries under normal data change nothing beyond 1e-35 absolute;
      3. +Inf / -Inf / NaN in A: the SAME output rows are non-finite in both forms (an infinite term splits into (Inf, NaN, NaN): the row is
         NaN where the fp32 kernel has +-Inf or NaN — non-finite either way), all other rows are bit-identical to the run without them;
      4. the largest magnitudes of the domain, |x| = 3.3895e38 (the largest bf16; above it the first term would round to infinity):
         finite products stay finite and accurate."""
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(20250929)
    M, N, K = 1024, 128, 960
    # 1. dynamic range
    a = torch.randn(M, K, device=dev, generator=g) * torch.exp2(torch.randint(-20, 21, (M, K), device=dev, generator=g).float())
    b = torch.randn(N, K, device=dev, generator=g) * torch.exp2(torch.randint(-10, 11, (N, K), device=dev, generator=g).float())
    c0, c1 = _both(a, b)
    ref = a.double() @ b.double().t()
    bound = a.double().abs() @ b.double().abs().t()
    e0, e1 = ((c0.double() - ref).abs() / bound).max().item(), ((c1.double() - ref).abs() / bound).max().item()
    print("dynamic range 2^40: max |err| / sum|a.b|: fp32 MFMA %.2e, split %.2e" % (e0, e1))
    assert torch.isfinite(c1).all() and e1 < 2e-6 and e1 < max(2 * e0, 2e-7)
    # 2. subnormals
    tiny = torch.randn(M, K, device=dev, generator=g) * 1e-40
    assert (tiny != 0).any() and (tiny.abs() < 1.2e-38).all()
    from lcrnet_amd import functional as F
    w = torch.randn(N, K, device=dev, generator=g) * 1e-40
    t = F.unsplit_bf16x3(F.split_bf16x3(w))
    # below the normal range the terms are NOT exact: the conversions flush subnormal results (h1 of a subnormal x may come out 0 or
    # truncated) — the reconstruction error stays below the smallest normal number per element, i.e. it vanishes in any product
    assert ((t[0] + t[1]) + t[2] - w).abs().max().item() < 1.2e-38
    bn = torch.randn(N, K, device=dev, generator=g)
    z0, z1 = _both(tiny, bn)
    assert z0.abs().max().item() < 1e-35 and z1.abs().max().item() < 1e-35
    an = torch.randn(M, K, device=dev, generator=g)
    n0, n1 = _both(an, bn)
    mix, hole = an.clone(), an.clone()
    mix[:, ::2] = tiny[:, ::2]                                        # every other column subnormal ...
    hole[:, ::2] = 0                                                  # ... against the same columns zeroed
    p0, p1 = _both(mix, bn)
    q0, q1 = _both(hole, bn)
    assert (p0 - q0).abs().max().item() < 1e-35 and (p1 - q1).abs().max().item() < 1e-35
    # 3. non-finite propagation
    bad = an.clone()
    rows = torch.tensor([3, 64, 65, 500, 1023], device=dev)
    bad[3, 7] = float("inf")
    bad[64, 0] = float("-inf")
    bad[65, K - 1] = float("nan")
    bad[500, 100] = float("inf")
    bad[500, 101] = float("-inf")
    bad[1023, 959] = float("inf")
    f0, f1 = _both(bad, bn)
    assert torch.equal(torch.isfinite(f0), torch.isfinite(f1))
    mask = torch.ones(M, dtype=torch.bool, device=dev)
    mask[rows] = False
    assert not torch.isfinite(f1[rows]).any() and torch.isfinite(f1[mask]).all()
    assert torch.equal(f1[mask], n1[mask]) and torch.equal(f0[mask], n0[mask])
    # 4. top of the domain
    big = torch.full((64, K), 3.3895313892515355e38, device=dev)
    big[:, 1::2] *= -1
    sm = torch.full((N, K), 2.0 ** -100, device=dev) * (1 + torch.rand(N, K, device=dev, generator=g))
    h0, h1 = _both(big, sm)
    href = big.double() @ sm.double().t()
    assert torch.isfinite(h1).all()
    hb = big.double().abs() @ sm.double().abs().t()
    assert ((h1.double() - href).abs() / hb).max().item() < 2e-6 and ((h0.double() - href).abs() / hb).max().item() < 2e-6
