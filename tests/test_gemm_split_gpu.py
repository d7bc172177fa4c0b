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
