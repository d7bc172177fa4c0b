"""Thin tensor-level wrappers over the C ABI (include/lcr_hip.h).  No arithmetic happens in Python: every function
allocates the outputs with torch and launches HIP kernels on the current stream.  Inference only (no autograd)."""
import ctypes
import os
import threading

import numpy as np
import torch

from . import _lib

GN_EPS = 1e-5
GN_REPLICAS = 8   # must match csrc/common.h: statistics tables are [GN_REPLICAS, S, groups, 2]


class KernelTimer:
    """Opt-in per-launch timing of lcr_gemm_f32 ("gemm", meta (M,N,K)) and lcr_kpconv_aggregate ("kpconv_aggregate", meta
    (M,Ns,H,C,index bytes)) with HIP events on the launch stream, recorded inside the library (so launches issued by the native
    encoder driver are seen too).  set_timer(t) starts a fresh log, set_timer(None) stops logging, t.summary() synchronises."""
    KINDS = {"gemm": 0, "kpconv_aggregate": 1, "radius_query": 2, "attention": 3, "kpconv_fused": 4, "sinkhorn": 5}

    def __init__(self, names):
        self.names = set(names)
        self._cache = None

    def summary(self):
        """name -> [(seconds between the events bracketing the launch on its stream, meta)]"""
        return {k: [(b, m) for b, _, m in v] for k, v in self.records().items()}

    def records(self):
        """name -> [(bracketed seconds, the kernel's own begin-to-end seconds or None, meta)].  The bracketed time contains whatever
        the launch waited for in its queue (other streams' kernels holding the CUs); the kernel's own time is what a profiler's kernel
        trace reports."""
        if self._cache is None:
            out = {}
            L = _lib.lib()
            for name in self.names:
                kind = self.KINDS[name]
                n = L.lcr_ktimer_read2(kind, 0, None, None, None)
                sec = (ctypes.c_double * max(n, 1))()
                ksec = (ctypes.c_double * max(n, 1))()
                meta = (ctypes.c_int64 * (5 * max(n, 1)))()
                L.lcr_ktimer_read2(kind, n, ctypes.cast(sec, ctypes.c_void_p), ctypes.cast(ksec, ctypes.c_void_p), ctypes.cast(meta, ctypes.c_void_p))
                width = 3 if name == "gemm" else 5            # radius_query: (nq_cap, ns_cap, limit, index bytes, B); attention: (sum Nq*Nk, P, heads, head_dim, 0); sinkhorn: (B, M, N, iters, form)
                out[name] = [(sec[i], ksec[i] if ksec[i] >= 0 else None, tuple(int(meta[5 * i + k]) for k in range(width))) for i in range(n)]
            self._cache = out
        return self._cache


def set_timer(timer, sample_every=1):
    """sample_every = n: only every n-th instrumented launch of each kind is timed (a timed launch is a profiled dispatch; with all
    of them timed the bench pipeline runs 7 % slower)."""
    _lib.lib().lcr_ktimer_sample(int(sample_every))
    mask = 0
    for name in (timer.names if timer is not None else ()):
        mask |= 1 << KernelTimer.KINDS[name]
    _lib.lib().lcr_ktimer_kinds(mask if timer is not None else 0xffffffff)     # only the kinds the timer asked for are clocked
    _lib.lib().lcr_ktimer_enable(1 if timer is not None else 0)


def _timed(name, fn, meta=None):
    return fn()


def _seg(seg_len, n, device):
    """GroupNorm segment lengths: None = the reference's behaviour (one segment = the whole stack)."""
    if seg_len is None:
        # a fill launch, NOT torch.tensor([n], device=...): that is a pageable host-to-device copy, after which torch synchronises the
        # stream — 25 pipeline drains per registration pair (profiles/r06_pair_host_profile.log)
        return torch.full((1,), int(n), dtype=torch.int64, device=device)
    return seg_len


def host_values(values, dtype, device):
    """A small host list as a device tensor without draining the stream: staged in pinned memory (torch's caching host allocator) and
    copied asynchronously on the current stream."""
    return torch.tensor(values, dtype=dtype).pin_memory().to(device, non_blocking=True)


def host_tensor(t, device):
    """same for a host tensor / numpy array"""
    if not torch.is_tensor(t):
        t = torch.from_numpy(t)
    return t.pin_memory().to(device, non_blocking=True)


def _idx_args(idx):
    if idx.dtype == torch.int64:
        return 1
    if idx.dtype == torch.int32:
        return 0
    raise RuntimeError("neighbor indices must be int32 or int64")


class _StatsArena(threading.local):
    buf = None
    off = 0


_ARENA = _StatsArena()


class stats_arena:
    """Context manager: GroupNorm statistics tables of every GEMM / groupnorm_stats call inside are carved out of ONE zeroed
    fp64 buffer (one fill launch per forward pass instead of one per layer).  Per host thread; nests by replacement."""

    def __init__(self, device, entries=1 << 18):
        self.device, self.entries = device, entries

    def __enter__(self):
        self.prev = (_ARENA.buf, _ARENA.off)
        _ARENA.buf = torch.zeros(self.entries, dtype=torch.float64, device=self.device)
        _ARENA.off = 0
        return self

    def __exit__(self, *exc):
        _ARENA.buf, _ARENA.off = self.prev
        return False


def _zero_stats(S, groups, device):
    n = GN_REPLICAS * S * groups * 2
    buf = _ARENA.buf
    if buf is not None and buf.device == device and _ARENA.off + n <= buf.numel():
        v = buf[_ARENA.off:_ARENA.off + n].view(GN_REPLICAS, S, groups, 2)
        _ARENA.off += n
        return v
    return torch.zeros((GN_REPLICAS, S, groups, 2), dtype=torch.float64, device=device)


def gemm(a, b, trans_b=False, trans_a=False, bias=None, rowdiv=None, seg_len=None, groups=0):
    """C = A·B (+ fused epilogue).  a: [M,K] ([K,M] if trans_a); b: [K,N] ([N,K] if trans_b, i.e. an nn.Linear weight).
    Returns (C, stats) where stats is the fp64 [GN_REPLICAS,S,groups,2] GroupNorm accumulator (sum the replicas; None if
    groups == 0)."""
    _lib.require_cuda(a, b)
    assert a.dtype == torch.float32 and b.dtype == torch.float32 and a.is_contiguous() and b.is_contiguous()
    M, K = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
    N = b.shape[0] if trans_b else b.shape[1]
    assert (b.shape[1] if trans_b else b.shape[0]) == K, "inner dimensions differ"
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    stats = None
    S = 0
    if groups:
        seg_len = _seg(seg_len, M, a.device)
        S = seg_len.numel()
        stats = _zero_stats(S, groups, a.device)
    _timed("gemm", lambda: _lib.check(_lib.lib().lcr_gemm_f32(
        _lib.ptr(a), _lib.ptr(b), _lib.ptr(c), M, N, K, int(trans_a), int(trans_b), _lib.ptr(bias), _lib.ptr(rowdiv),
        _lib.ptr(seg_len) if groups else None, S, int(groups), _lib.ptr(stats), _lib.stream_ptr(a.device)), "lcr_gemm_f32"),
        meta=(M, N, K))
    return c, stats


# ---- split-bf16 form of the K-deep GEMMs (csrc/gemm_f32.hip): fp32 operands as three bf16 terms, six products on the bf16 matrix cores.
# DEFAULT since round 5 for the K-deep shapes (K >= 288, K % 32 == 0, N >= 64): every fp32 operand enters as its three bf16 terms (exact), six
# of the nine cross products are accumulated in fp32 — measured error against fp64 at or below the fp32-MFMA kernel's own, non-finite values
# propagate to the same outputs (tests/test_gemm_split_gpu.py).  LCR_GEMM_SPLIT=0 selects the true-fp32 MFMA kernel everywhere;
# `set_gemm_split` is the run-time switch (tests, bench A/B).  Domain of the split form: |x| <= 3.3895e38 (the largest bf16), beyond which
# the first term rounds to infinity.
_GEMM_SPLIT = [os.environ.get("LCR_GEMM_SPLIT", "1") not in ("", "0")]


def gemm_split_enabled():
    return _GEMM_SPLIT[0]


def set_gemm_split(on):
    _GEMM_SPLIT[0] = bool(on)


def gemm_split_ok(N, K):
    return K >= 288 and K % 32 == 0 and N >= 64


# Derived weight tensors (transposed KPConv weights, bf16 planes, the native encoder's table) are built lazily by whichever host thread gets
# there first, on ITS current stream, and then used by every thread on other streams: the builder holds this lock and drains its stream before
# publishing, so a second worker (PairPipeline runs two) never launches a GEMM on a half-written operand.
derived_lock = threading.RLock()


class WeightStamp:
    """Identity of a weight tensor for the derived-tensor caches (transposed / split weights, host copies, the native encoder's table):
    the tensor OBJECT (weak reference), its address, its version counter and its device.  Address + version alone are not an identity: a
    buffer replaced by Module._apply (.cpu() / .cuda() / .to()) is a NEW tensor with version 0 that the caching allocator readily
    places at the address of the one it replaces — found by tools/fuzz_float_parity_gpu.py, which reloads seeded weights through
    .cpu() -> load_state_dict -> .cuda() and got the previous seed's kernel points in one case of eight."""
    __slots__ = ("ref", "ptr", "ver", "dev")

    def __init__(self, t):
        import weakref
        self.ref, self.ptr, self.ver, self.dev = weakref.ref(t), t.data_ptr(), t._version, t.device

    def same(self, t):
        return self.ref is not None and self.ref() is t and self.ptr == t.data_ptr() and self.ver == t._version and self.dev == t.device

    def __reduce__(self):                                   # pickled / copied modules start with caches that match nothing
        return (_dead_stamp, ())


def _dead_stamp():
    s = WeightStamp.__new__(WeightStamp)
    s.ref, s.ptr, s.ver, s.dev = None, 0, -1, None
    return s


def drop_derived(module, *names):
    """forget cached derived tensors of a module (called from its _apply override: every .to() / .cuda() / .cpu() / .float())"""
    for n in names:
        if n in module.__dict__:
            module.__dict__[n] = None



def publish_derived(t):
    """Call on a freshly built derived tensor before storing it where other threads / streams can see it."""
    torch.cuda.current_stream(t.device).synchronize()
    return t


def split_bf16x3(w):
    """The three bf16 terms of a constant fp32 operand [N,K] (K % 32 == 0) in the tiled layout lcr_gemm_f32_bsplit stages them in:
    int16 [ceil(N/64), K/32, 3, 64, 32] (lcr_split_bf16x3_tiles; made once per weight).  The tensor carries N as `.lcr_n`."""
    w = w.detach().contiguous()
    N, K = w.shape
    assert K % 32 == 0
    tiles = torch.empty(((N + 63) // 64, K // 32, 3, 64, 32), dtype=torch.int16, device=w.device)
    _lib.check(_lib.lib().lcr_split_bf16x3_tiles(_lib.ptr(w), int(N), int(K), _lib.ptr(tiles), _lib.stream_ptr(w.device)), "lcr_split_bf16x3_tiles")
    tiles.lcr_n = int(N)
    return tiles


def unsplit_bf16x3(tiles):
    """Inverse view of split_bf16x3 for tests: the three terms as float32 [3, N, K] (un-tiled, un-swizzled)."""
    CT, KS = tiles.shape[0], tiles.shape[1]
    t = tiles.view(CT, KS, 3, 64, 4, 8)
    row = torch.arange(64, device=tiles.device)
    phys = torch.arange(4, device=tiles.device)[None, :] ^ ((row >> 2) & 3)[:, None]          # [row, logical chunk] -> physical chunk
    t = torch.gather(t, 4, phys.view(1, 1, 1, 64, 4, 1).expand(CT, KS, 3, 64, 4, 8))
    t = t.reshape(CT, KS, 3, 64, 32).permute(2, 0, 3, 1, 4).reshape(3, CT * 64, KS * 32)
    return t[:, :tiles.lcr_n].contiguous().view(torch.bfloat16).float()


def gemm_bsplit(a, planes, bias=None, rowdiv=None, seg_len=None, groups=0):
    """C = A[M,K] . B[N,K]^T with B given as the planes of split_bf16x3 (same epilogue and return value as gemm())."""
    _lib.require_cuda(a, planes)
    assert a.dtype == torch.float32 and a.is_contiguous() and planes.dtype == torch.int16 and planes.is_contiguous() and planes.dim() == 5
    M, K = a.shape
    N = planes.lcr_n
    assert planes.shape[1] * 32 == K and planes.shape[0] == (N + 63) // 64, "inner dimensions differ"
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    stats, S = None, 0
    if groups:
        seg_len = _seg(seg_len, M, a.device)
        S = seg_len.numel()
        stats = _zero_stats(S, groups, a.device)
    _lib.check(_lib.lib().lcr_gemm_f32_bsplit(_lib.ptr(a), _lib.ptr(planes), _lib.ptr(c), M, N, K, _lib.ptr(bias), _lib.ptr(rowdiv),
                                              _lib.ptr(seg_len) if groups else None, S, int(groups), _lib.ptr(stats), _lib.stream_ptr(a.device)),
               "lcr_gemm_f32_bsplit")
    return c, stats


ANORM_MAX_K, ANORM_MIN_SEG_ROWS = 256, 64


def gemm_anorm_ok(K, N):
    """Shapes lcr_gemm_f32_anorm takes (the light GEMM form)."""
    return K <= ANORM_MAX_K and K % 4 == 0 and N > 32


def gemm_anorm(a, a_stats, a_gamma, a_beta, a_groups, weight, bias=None, seg_len=None, groups=0, slope=0.1):
    """C = LeakyReLU(GroupNorm(a)) · weight^T (+ bias), a being the RAW output whose sums are a_stats; the normalised tensor is
    never materialised (lcr_gemm_f32_anorm).  The caller guarantees every segment holds >= ANORM_MIN_SEG_ROWS rows.
    Returns (C, stats) like gemm()."""
    _lib.require_cuda(a, weight)
    assert a.dtype == torch.float32 and weight.dtype == torch.float32 and a.is_contiguous() and weight.is_contiguous()
    M, K = a.shape
    N = weight.shape[0]
    assert weight.shape[1] == K and gemm_anorm_ok(K, N)
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    seg_len = _seg(seg_len, M, a.device)
    S = seg_len.numel()
    stats = _zero_stats(S, groups, a.device) if groups else None
    _timed("gemm", lambda: _lib.check(_lib.lib().lcr_gemm_f32_anorm(
        _lib.ptr(a), _lib.ptr(weight), _lib.ptr(c), M, N, K, _lib.ptr(bias), _lib.ptr(a_stats), _lib.ptr(a_gamma), _lib.ptr(a_beta),
        int(a_groups), GN_EPS, float(slope), _lib.ptr(seg_len), S, int(groups), _lib.ptr(stats), _lib.stream_ptr(a.device)),
        "lcr_gemm_f32_anorm"), meta=(M, N, K))
    return c, stats


def groupnorm_stats(x, groups, seg_len=None):
    seg_len = _seg(seg_len, x.shape[0], x.device)
    stats = _zero_stats(seg_len.numel(), groups, x.device)
    _lib.check(_lib.lib().lcr_groupnorm_stats(_lib.ptr(x), x.shape[0], x.shape[1], groups, _lib.ptr(seg_len), seg_len.numel(),
                                              _lib.ptr(stats), _lib.stream_ptr(x.device)), "lcr_groupnorm_stats")
    return stats


def groupnorm_apply(x, stats, gamma, beta, groups, seg_len=None, res=None, res_norm=None, slope=0.1, act=True, want_pos=False):
    """y = act(GN(x) [+ res | + GN(res)]); res_norm = (stats, gamma, beta) of the residual branch or None."""
    seg_len = _seg(seg_len, x.shape[0], x.device)
    y = torch.empty_like(x)
    pos = torch.empty((x.shape[0],), dtype=torch.uint8, device=x.device) if want_pos else None
    rs, rg, rb = res_norm if res_norm is not None else (None, None, None)
    _lib.check(_lib.lib().lcr_groupnorm_apply(_lib.ptr(x), _lib.ptr(stats), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(res), _lib.ptr(rs),
                                              _lib.ptr(rg), _lib.ptr(rb), _lib.ptr(y), x.shape[0], x.shape[1], groups, _lib.ptr(seg_len),
                                              seg_len.numel(), GN_EPS, float(slope), int(act), _lib.ptr(pos),
                                              _lib.stream_ptr(x.device)), "lcr_groupnorm_apply")
    return (y, pos) if want_pos else y


def row_positive(x):
    pos = torch.empty((x.shape[0],), dtype=torch.uint8, device=x.device)
    _lib.check(_lib.lib().lcr_row_positive(_lib.ptr(x), x.shape[0], x.shape[1], _lib.ptr(pos), _lib.stream_ptr(x.device)), "lcr_row_positive")
    return pos


def _kp_host(kernel_points):
    kp = np.ascontiguousarray(kernel_points, dtype=np.float32)
    assert kp.shape == (15, 3)
    return kp


def kpconv_aggregate(s_feats, s_pos, q_points, s_points, idx, kernel_points_host, sigma, order=None):
    """(A [M, 15*C], nn [M]) — the gather/influence/aggregate half of KPConv.forward."""
    _lib.require_cuda(s_feats, q_points, s_points, idx)
    M, H = idx.shape
    Ns, C = s_feats.shape
    assert idx.is_contiguous() and s_feats.is_contiguous() and q_points.is_contiguous() and s_points.is_contiguous()
    A = torch.empty((M, 15 * C), dtype=torch.float32, device=s_feats.device)
    nn = torch.empty((M,), dtype=torch.float32, device=s_feats.device)
    kp = _kp_host(kernel_points_host)
    _timed("kpconv_aggregate", lambda: _lib.check(_lib.lib().lcr_kpconv_aggregate(
        _lib.ptr(s_feats), _lib.ptr(s_pos), _lib.ptr(q_points), _lib.ptr(s_points), _lib.ptr(idx), _idx_args(idx), M, Ns, H, C,
        ctypes.c_void_p(kp.ctypes.data), float(sigma), _lib.ptr(A), _lib.ptr(nn), _lib.ptr(order), _lib.stream_ptr(s_feats.device)),
        "lcr_kpconv_aggregate"), meta=(M, Ns, H, C, idx.element_size()))
    return A, nn


KPCONV_FUSED_C = 32


def kpconv_fused(s_feats, s_pos, q_points, s_points, idx, kernel_points_host, sigma, weights, bias, seg_len=None, groups=0, order=None):
    """Whole rigid KPConv for C_in = C_out = 32 in one launch (lcr_kpconv_fused): the (M, 15*C) aggregate stays in LDS.
    weights: (15, C, C).  Returns (out [M, C], stats) like gemm()."""
    _lib.require_cuda(s_feats, q_points, s_points, idx)
    M, H = idx.shape
    Ns, C = s_feats.shape
    assert C == KPCONV_FUSED_C and tuple(weights.shape) == (15, C, C) and weights.is_contiguous()
    assert idx.is_contiguous() and s_feats.is_contiguous() and q_points.is_contiguous() and s_points.is_contiguous()
    out = torch.empty((M, C), dtype=torch.float32, device=s_feats.device)
    stats, S = None, 0
    if groups:
        seg_len = _seg(seg_len, M, s_feats.device)
        S = seg_len.numel()
        stats = _zero_stats(S, groups, s_feats.device)
    kp = _kp_host(kernel_points_host)
    _timed("kpconv_fused", lambda: _lib.check(_lib.lib().lcr_kpconv_fused(
        _lib.ptr(s_feats), _lib.ptr(s_pos), _lib.ptr(q_points), _lib.ptr(s_points), _lib.ptr(idx), _idx_args(idx), M, Ns, H, C,
        ctypes.c_void_p(kp.ctypes.data), float(sigma), _lib.ptr(weights), _lib.ptr(bias), _lib.ptr(out),
        _lib.ptr(seg_len) if groups else None, S, int(groups), _lib.ptr(stats), _lib.ptr(order), _lib.stream_ptr(s_feats.device)),
        "lcr_kpconv_fused"), meta=(M, Ns, H, C, idx.element_size()))
    return out, stats


def kpconv_cin1(s_feats, q_points, s_points, idx, kernel_points_host, sigma, weights, bias, order=None):
    """Whole KPConv for one input channel: weights (15,1,Cout) -> out [M,Cout]."""
    M, H = idx.shape
    Ns = s_feats.shape[0]
    Cout = weights.shape[-1]
    out = torch.empty((M, Cout), dtype=torch.float32, device=s_feats.device)
    kp = _kp_host(kernel_points_host)
    _lib.check(_lib.lib().lcr_kpconv_cin1(_lib.ptr(s_feats), _lib.ptr(q_points), _lib.ptr(s_points), _lib.ptr(idx), _idx_args(idx), M, Ns, H,
                                          ctypes.c_void_p(kp.ctypes.data), float(sigma), _lib.ptr(weights), _lib.ptr(bias), Cout,
                                          _lib.ptr(out), _lib.ptr(order), _lib.stream_ptr(s_feats.device)), "lcr_kpconv_cin1")
    return out


def maxpool(x, idx, order=None):
    M, H = idx.shape
    out = torch.empty((M, x.shape[1]), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().lcr_maxpool(_lib.ptr(x), _lib.ptr(idx), _idx_args(idx), M, x.shape[0], H, x.shape[1], _lib.ptr(out),
                                      _lib.ptr(order), _lib.stream_ptr(x.device)), "lcr_maxpool")
    return out


class NetvladWeights(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "cluster_weights", "cluster_weights2", "hidden1_weights", "bn1_w", "bn1_b", "bn1_mean", "bn1_var",
        "bn2_w", "bn2_b", "bn2_mean", "bn2_var", "gating_weights", "gbn_w", "gbn_b", "gbn_mean", "gbn_var")]


def netvlad_forward(feats, seg_len_host, weights_struct):
    """feats [sum(seg_len),1024] stacked coarse features -> [S,256] unit-norm descriptors."""
    _lib.require_cuda(feats)
    seg = np.ascontiguousarray(seg_len_host, dtype=np.int64)
    S = int(seg.shape[0])
    assert int(seg.sum()) == feats.shape[0] and feats.shape[1] == 1024 and feats.is_contiguous()
    nbytes = ctypes.c_size_t(0)
    _lib.check(_lib.lib().lcr_netvlad_ws_bytes(feats.shape[0], S, ctypes.byref(nbytes)), "lcr_netvlad_ws_bytes")
    ws = _lib.workspace(nbytes.value, feats.device)
    out = torch.empty((S, 256), dtype=torch.float32, device=feats.device)
    _lib.check(_lib.lib().lcr_netvlad_forward(_lib.ptr(feats), ctypes.c_void_p(seg.ctypes.data), S, ctypes.byref(weights_struct),
                                              _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(feats.device)),
               "lcr_netvlad_forward")
    return out


def linear(x, weight, bias=None, relu=False):
    """nn.Linear (weight [out,in]) on the MFMA GEMM, optional ReLU."""
    y = gemm(x.contiguous(), weight, trans_b=True, bias=bias)[0]
    if relu:
        _lib.check(_lib.lib().lcr_relu_inplace(_lib.ptr(y), y.numel(), _lib.stream_ptr(y.device)), "lcr_relu_inplace")
    return y


def relu_(x):
    _lib.check(_lib.lib().lcr_relu_inplace(_lib.ptr(x), x.numel(), _lib.stream_ptr(x.device)), "lcr_relu_inplace")
    return x


def rotary_embed_(x, theta, heads):
    """In place rotary position embedding (rpetransformer.py:41-54): x [N, heads*32], theta [N, heads*16]."""
    assert x.is_contiguous() and theta.is_contiguous() and x.shape[1] == heads * 32 and theta.shape[1] == heads * 16
    _lib.check(_lib.lib().lcr_rotary_embed(_lib.ptr(x), _lib.ptr(theta), x.shape[0], heads, _lib.stream_ptr(x.device)), "lcr_rotary_embed")
    return x


def attention(q, k, v, heads, q_lens=None, k_lens=None):
    """Fused softmax(q k^T / sqrt(d)) v per head; q [Nq, heads*32], k/v [Nk, heads*32].  With q_lens / k_lens (host sequences of
    equal length P <= 64, summing to Nq / Nk): P independent problems over the stacked rows in one launch."""
    assert q.is_contiguous() and k.is_contiguous() and v.is_contiguous()
    out = torch.empty_like(q)
    if q_lens is not None:
        P = len(q_lens)
        assert len(k_lens) == P and sum(q_lens) == q.shape[0] and sum(k_lens) == k.shape[0]
        ql, kl = (ctypes.c_int64 * P)(*[int(x) for x in q_lens]), (ctypes.c_int64 * P)(*[int(x) for x in k_lens])
        _lib.check(_lib.lib().lcr_attention_seg_f32(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), ctypes.cast(ql, ctypes.c_void_p),
                                                    ctypes.cast(kl, ctypes.c_void_p), P, heads, q.shape[1] // heads, _lib.ptr(out),
                                                    _lib.stream_ptr(q.device)), "lcr_attention_seg_f32")
        return out
    _lib.check(_lib.lib().lcr_attention_f32(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), q.shape[0], k.shape[0], heads, q.shape[1] // heads,
                                            _lib.ptr(out), _lib.stream_ptr(q.device)), "lcr_attention_f32")
    return out


ATTENTION_TOPK_MAX_KEYS = 4096     # ATK_MAXK of csrc/attention.hip: one row of scores per (query, head) in LDS


def attention_topk(q, k, v, heads, q_lens, k_lens, kks):
    """dynamic_attention with k != None (rpetransformer.py:19-39): per problem p (rows stacked like `attention`), query and head, only the
    kks[p] largest scores are soft-maxed.  kks[p] = int(n_queries_p * fraction) is computed by the caller as the reference does."""
    assert q.is_contiguous() and k.is_contiguous() and v.is_contiguous()
    P = len(q_lens)
    assert len(k_lens) == P and len(kks) == P and sum(q_lens) == q.shape[0] and sum(k_lens) == k.shape[0]
    out = torch.empty_like(q)
    ql, kl = (ctypes.c_int64 * P)(*[int(x) for x in q_lens]), (ctypes.c_int64 * P)(*[int(x) for x in k_lens])
    kk = (ctypes.c_int * P)(*[int(x) for x in kks])
    _lib.check(_lib.lib().lcr_attention_topk_f32(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), ctypes.cast(ql, ctypes.c_void_p), ctypes.cast(kl, ctypes.c_void_p),
                                                 ctypes.cast(kk, ctypes.c_void_p), P, heads, q.shape[1] // heads, _lib.ptr(out), _lib.stream_ptr(q.device)),
               "lcr_attention_topk_f32")
    return out


def add_layernorm(a, b, gamma, beta, eps=1e-5):
    y = torch.empty_like(a)
    _lib.check(_lib.lib().lcr_add_layernorm(_lib.ptr(a), _lib.ptr(b), _lib.ptr(gamma), _lib.ptr(beta), a.shape[0], a.shape[1], float(eps),
                                            _lib.ptr(y), _lib.stream_ptr(a.device)), "lcr_add_layernorm")
    return y


# ------------------------------------------------------------------------------------------------ pose tail (a-10)
def _L():
    return _lib.lib()


def _sp(t):
    return _lib.stream_ptr(t.device)


def vote_shift(xyz, offsets, max_range):
    out = torch.empty_like(xyz)
    _lib.check(_L().lcr_vote_shift(_lib.ptr(xyz.contiguous()), _lib.ptr(offsets.contiguous()), xyz.shape[0], float(max_range), _lib.ptr(out),
                                   _sp(xyz)), "lcr_vote_shift")
    return out


def greedy_nms(points, lengths, radius):
    """(keep mask uint8 [N], kept count per cloud i64 [B]) — modules/vote/vote.py:13-70."""
    n, B = points.shape[0], lengths.numel()
    keep = torch.empty((n,), dtype=torch.uint8, device=points.device)
    out_len = torch.empty((B,), dtype=torch.int64, device=points.device)
    nbytes = ctypes.c_size_t(0)
    _lib.check(_L().lcr_greedy_nms_ws_bytes(n, ctypes.byref(nbytes)), "lcr_greedy_nms_ws_bytes")
    ws = _lib.workspace(nbytes.value, points.device)
    _lib.check(_L().lcr_greedy_nms(_lib.ptr(points.contiguous()), _lib.ptr(lengths), B, n, float(radius), _lib.ptr(keep), _lib.ptr(out_len),
                                   _lib.ptr(ws), _sp(points)), "lcr_greedy_nms")
    return keep, out_len


def neighbor_mean(points, idx, pad):
    out = torch.empty((idx.shape[0], 3), dtype=torch.float32, device=points.device)
    _lib.check(_L().lcr_neighbor_mean(_lib.ptr(points.contiguous()), _lib.ptr(idx.contiguous()), _idx_args(idx), idx.shape[0], idx.shape[1], int(pad),
                                      _lib.ptr(out), _sp(points)), "lcr_neighbor_mean")
    return out


def point_to_node_partition(points, nodes, point_limit):
    """(point_to_node i32[N], node_masks bool[M], node_knn_indices i64[M,K], node_knn_masks bool[M,K]) —
    modules/ops/pointcloud_partition.py:60-107."""
    N, M, dev = points.shape[0], nodes.shape[0], points.device
    nbytes = ctypes.c_size_t(0)
    _lib.check(_L().lcr_point_to_node_ws_bytes(N, M, ctypes.byref(nbytes)), "lcr_point_to_node_ws_bytes")
    ws = _lib.workspace(nbytes.value, dev)
    p2n = torch.empty((N,), dtype=torch.int32, device=dev)
    knn = torch.empty((M, point_limit), dtype=torch.int64, device=dev)
    km = torch.empty((M, point_limit), dtype=torch.uint8, device=dev)
    nm = torch.empty((M,), dtype=torch.uint8, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(_L().lcr_point_to_node_partition(_lib.ptr(points.contiguous()), N, _lib.ptr(nodes.contiguous()), M, int(point_limit), _lib.ptr(p2n),
                                                _lib.ptr(knn), _lib.ptr(km), _lib.ptr(nm), _lib.ptr(status), _lib.ptr(ws), ws.numel(), _sp(points)),
               "lcr_point_to_node_partition")
    return p2n, nm.bool(), knn, km.bool()


def point_to_node_partition_stack(points, point_off, nodes, node_off, point_limit):
    """`point_to_node_partition` of every cloud of a stack in ONE launch sequence: cloud c owns points[point_off[c]:point_off[c+1]] and
    nodes[node_off[c]:node_off[c+1]] (host offset lists).  -> (point_to_node i32[N], node_masks bool[M], node_knn_indices i64[M,K],
    node_knn_masks bool[M,K]) stacked in cloud order; indices are local to their cloud, knn rows padded with the cloud's point count —
    slices of these are exactly what the per-cloud call returns."""
    import numpy as np
    C = len(point_off) - 1
    assert len(node_off) == C + 1 and C >= 1
    dev = points.device
    N, M = int(point_off[-1] - point_off[0]), int(node_off[-1] - node_off[0])
    po = np.ascontiguousarray(point_off, dtype=np.int64)
    mo = np.ascontiguousarray(node_off, dtype=np.int64)
    nbytes = ctypes.c_size_t(0)
    _lib.check(_L().lcr_point_to_node_ws_bytes(N, M, ctypes.byref(nbytes)), "lcr_point_to_node_ws_bytes")
    ws = _lib.workspace(nbytes.value, dev)
    p2n = torch.empty((N,), dtype=torch.int32, device=dev)
    knn = torch.empty((M, point_limit), dtype=torch.int64, device=dev)
    km = torch.empty((M, point_limit), dtype=torch.uint8, device=dev)
    nm = torch.empty((M,), dtype=torch.uint8, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(_L().lcr_point_to_node_partition_stack(_lib.ptr(points.contiguous()), po.ctypes.data, _lib.ptr(nodes.contiguous()), mo.ctypes.data, C,
                                                      int(point_limit), _lib.ptr(p2n), _lib.ptr(knn), _lib.ptr(km), _lib.ptr(nm), _lib.ptr(status),
                                                      _lib.ptr(ws), ws.numel(), _sp(points)), "lcr_point_to_node_partition_stack")
    return p2n, nm.bool(), knn, km.bool()


class _PendingStatus(threading.local):
    def __init__(self):
        self.items = []


_pending_ot_status = _PendingStatus()
_PENDING_OT_MAX = 32


def check_transport_status():
    """Raise if a persistent optimal-transport launch issued by this thread reported a timed-out hand-off (its result is invalid).
    One host read; called where the caller synchronises anyway (top1_matching reads a count back right after the transport)."""
    items, _pending_ot_status.items = _pending_ot_status.items, []
    if items and any(int(t.reshape(()).item()) != 0 for t in items):       # one entry per device after folding; a handful otherwise
        raise RuntimeError("lcr_log_sinkhorn: a workgroup hand-off of the persistent form timed out (set LCR_SINKHORN_COOP=0)")


def log_optimal_transport(raw_scores, row_masks, col_masks, alpha, scale=1.0, iters=100, inf=1e12):
    """LearnableLogOptimalTransport on raw products [B,M,N] (scaled by `scale`) -> log scores [B,M+1,N+1]."""
    B, M, N = raw_scores.shape
    dev = raw_scores.device
    rm, cm = row_masks.to(torch.uint8).contiguous(), col_masks.to(torch.uint8).contiguous()
    S = torch.empty((B, M + 1, N + 1), dtype=torch.float32, device=dev)
    _lib.check(_L().lcr_build_padded_scores(_lib.ptr(raw_scores.contiguous()), _lib.ptr(rm), _lib.ptr(cm), B, M, N, float(scale),
                                            _lib.ptr(alpha.reshape(1).float()), float(inf), _lib.ptr(S), _sp(S)), "lcr_build_padded_scores")
    return _transport_padded(S, rm, cm, iters, inf)


PATCH_SCORES_FUSED = [os.environ.get("LCR_PATCH_SCORES_FUSED", "1") != "0"]      # A/B switch (tests, tools): 0 = gather + batched product + padding


def patch_log_optimal_transport(feats_a, idx_a, feats_b, idx_b, mask_a, mask_b, alpha, scale=1.0, iters=100, inf=1e12):
    """The patch-level transport of DenseMatchingHEAD (LCRNet.py:236-250): log scores [P,K+1,K+1] of P patch pairs whose point features are
    rows idx_a[p] of feats_a / idx_b[p] of feats_b (index == number of rows: the zero row of the padded tensor).  The gathers, the batched
    product, the scaling and the dustbin / mask padding are ONE kernel (lcr_patch_scores); LCR_PATCH_SCORES_FUSED=0 runs them apart."""
    P, K = idx_a.shape
    if not PATCH_SCORES_FUSED[0] or K != 128 or feats_a.shape[1] % 32 != 0 or P == 0:
        fa, fb = gather_rows(feats_a, idx_a.contiguous()), gather_rows(feats_b, idx_b.contiguous())
        return log_optimal_transport(bmm_nt(fa, fb), mask_a, mask_b, alpha, scale=scale, iters=iters, inf=inf)
    dev = feats_a.device
    fa, fb = feats_a.contiguous(), feats_b.contiguous()
    ia, ib = idx_a.contiguous(), idx_b.contiguous()
    assert ia.dtype == torch.int64 and ib.dtype == torch.int64 and fa.dtype == torch.float32 and fa.shape[1] == fb.shape[1]
    rm, cm = mask_a.to(torch.uint8).contiguous(), mask_b.to(torch.uint8).contiguous()
    S = torch.empty((P, K + 1, K + 1), dtype=torch.float32, device=dev)
    _lib.check(_L().lcr_patch_scores(_lib.ptr(fa), fa.shape[0], _lib.ptr(fb), fb.shape[0], fa.shape[1], _lib.ptr(ia), _lib.ptr(ib), _lib.ptr(rm),
                                     _lib.ptr(cm), P, K, float(scale), _lib.ptr(alpha.reshape(1).float()), float(inf), _lib.ptr(S), _sp(S)),
               "lcr_patch_scores")
    return _transport_padded(S, rm, cm, iters, inf)


def _transport_padded(S, rm, cm, iters, inf):
    """Sinkhorn on padded scores S [B,M+1,N+1] in place -> log scores (the second half of LearnableLogOptimalTransport)."""
    B, M, N = S.shape[0], S.shape[1] - 1, S.shape[2] - 1
    dev = S.device
    nfl = ctypes.c_size_t(0)
    _lib.check(_L().lcr_log_sinkhorn_ws_floats(B, M, N, ctypes.byref(nfl)), "lcr_log_sinkhorn_ws_floats")
    uv = torch.empty((nfl.value,), dtype=torch.float32, device=dev)
    form = ctypes.c_int(0)
    _lib.check(_L().lcr_log_sinkhorn_form(B, M, N, ctypes.byref(form)), "lcr_log_sinkhorn_form")
    persistent = form.value == 2                      # the only form with hand-offs, i.e. the only one that writes a status word
    if persistent:
        uv[-1:].zero_()                               # status word: bit 0 = a hand-off of the persistent form timed out
    _lib.check(_L().lcr_log_sinkhorn_ex(_lib.ptr(S), _lib.ptr(rm), _lib.ptr(cm), B, M, N, int(iters), float(inf), _lib.ptr(uv), uv.numel(), _sp(S)),
               "lcr_log_sinkhorn")
    if persistent:
        # a stream-ordered 1-element copy (not a view: a view would pin the whole workspace until somebody drains the list)
        items = _pending_ot_status.items
        items.append(uv[-1:].view(torch.int32).clone())
        if len(items) > _PENDING_OT_MAX:              # callers that never reach top1_matching: fold on the device, no host sync
            # bitwise OR per device (the word is a set of flags; one host thread may drive several GPUs)
            by_dev = {}
            for t in items:
                by_dev.setdefault(t.device, []).append(t.reshape(()))
            folded = []
            for ts in by_dev.values():
                acc = ts[0]
                for t in ts[1:]:
                    acc = torch.bitwise_or(acc, t)
                folded.append(acc.reshape(1))
            _pending_ot_status.items = folded
    return S


def top1_matching(log_scores, row_masks=None, col_masks=None, mutual=False, topk=1, use_dustbin=True, confidence_threshold=0.0,
                  global_scores=None):
    """(bij int32 [C,3], scores f32 [C], per-row offsets are internal) — dustbin top-k matching (k = 1 in the shipped configuration),
    row-major order.  mutual: keep a pair only if it is kept from its row AND from its column (local_global_registration.py:84-85)
    instead of either.  topk > 1: the k largest of a row / column, dustbin included, each against the dustbin (:56-82; lcr_topk_matching).
    use_dustbin=False: selection over the interior, kept where the selected value exceeds confidence_threshold (:62-65); global_scores [B]:
    use_global_score (:236-237) — both through lcr_topk_matching_ex.  One host sync for the (data-dependent) number of correspondences."""
    B, M1, N1 = log_scores.shape
    M, N, dev = M1 - 1, N1 - 1, log_scores.device
    topk = int(topk)
    assert topk >= 1
    nbytes = ctypes.c_size_t(0)
    ws_fn = _L().lcr_top1_matching_ws_bytes if (topk == 1 and use_dustbin and global_scores is None) else _L().lcr_topk_matching_ws_bytes
    _lib.check(ws_fn(B, M, N, ctypes.byref(nbytes)), "lcr_top1_matching_ws_bytes")
    ws = _lib.workspace(nbytes.value, dev)
    rm = row_masks.to(torch.uint8).contiguous() if row_masks is not None else None
    cm = col_masks.to(torch.uint8).contiguous() if col_masks is not None else None
    total = torch.zeros(1, dtype=torch.int64, device=dev)
    if not use_dustbin or global_scores is not None:
        gs = global_scores.float().contiguous() if global_scores is not None else None
        assert gs is None or gs.numel() == B
        fn, args = _L().lcr_topk_matching_ex, (_lib.ptr(log_scores.contiguous()), B, M, N, _lib.ptr(rm), _lib.ptr(cm), topk, int(bool(mutual)),
                                               int(bool(use_dustbin)), float(confidence_threshold), _lib.ptr(gs))
    elif topk == 1:
        fn, args = _L().lcr_top1_matching_ex, (_lib.ptr(log_scores.contiguous()), B, M, N, _lib.ptr(rm), _lib.ptr(cm), int(bool(mutual)))
    else:
        fn, args = _L().lcr_topk_matching, (_lib.ptr(log_scores.contiguous()), B, M, N, _lib.ptr(rm), _lib.ptr(cm), topk, int(bool(mutual)))
    _lib.check(fn(*args, _lib.ptr(total), None, None, _lib.ptr(ws), ws.numel(), _sp(log_scores)), "lcr_top1_matching")
    n = int(total.item())
    check_transport_status()
    bij = torch.empty((max(n, 1), 3), dtype=torch.int32, device=dev)
    sc = torch.empty((max(n, 1),), dtype=torch.float32, device=dev)
    if n:
        _lib.check(fn(*args, _lib.ptr(total), _lib.ptr(bij), _lib.ptr(sc), _lib.ptr(ws), ws.numel(), _sp(log_scores)), "lcr_top1_matching")
    return bij[:n], sc[:n]


def upsample_concat(x, idx, skip):
    out = torch.empty((skip.shape[0], x.shape[1] + skip.shape[1]), dtype=torch.float32, device=x.device)
    _lib.check(_L().lcr_upsample_concat(_lib.ptr(x.contiguous()), x.shape[0], x.shape[1], _lib.ptr(idx.contiguous()), _idx_args(idx), idx.shape[1],
                                        _lib.ptr(skip.contiguous()), skip.shape[1], skip.shape[0], _lib.ptr(out), _sp(x)), "lcr_upsample_concat")
    return out


def gather_rows(src, idx):
    """src[idx] with zero rows where idx == src.shape[0] (index_select on the zero-padded tensor); idx int64 any shape."""
    idx = idx.contiguous()
    out = torch.empty(tuple(idx.shape) + (src.shape[1],), dtype=torch.float32, device=src.device)
    _lib.check(_L().lcr_gather_rows(_lib.ptr(src.contiguous()), src.shape[0], src.shape[1], _lib.ptr(idx), idx.numel(), _lib.ptr(out), _sp(src)),
               "lcr_gather_rows")
    return out


def bmm_nt(a, b):
    """[P,M,K] x [P,N,K]^T -> [P,M,N] on the MFMA GEMM (einsum 'bnd,bmd->bnm')."""
    P, M, K = a.shape
    N = b.shape[1]
    c = torch.empty((P, M, N), dtype=torch.float32, device=a.device)
    a, b = a.contiguous(), b.contiguous()
    for z0 in range(0, P, 65535):
        cnt = min(65535, P - z0)
        _lib.check(_L().lcr_gemm_f32_strided_batched(_lib.ptr(a[z0:]), _lib.ptr(b[z0:]), _lib.ptr(c[z0:]), M, N, K, 0, 1, M * K, N * K, M * N, cnt,
                                                     _sp(a)), "lcr_gemm_f32_strided_batched")
    return c


def procrustes(src, ref, w, start=None):
    """Batched weighted Procrustes: problems = ranges [start[p], start[p+1]) (int32 device) or one problem over all rows."""
    dev = src.device
    if start is None:
        start = host_values([0, src.shape[0]], torch.int32, dev)
    P = start.numel() - 1
    T = torch.empty((P, 4, 4), dtype=torch.float32, device=dev)
    _lib.check(_L().lcr_procrustes_batched(_lib.ptr(src.contiguous()), _lib.ptr(ref.contiguous()), _lib.ptr(w.contiguous()), _lib.ptr(start), P, 1e-5,
                                           _lib.ptr(T), _sp(src)), "lcr_procrustes_batched")
    return T


def inlier_count(T, src, ref, radius, start=None, min_count=0):
    P, dev = T.shape[0], T.device
    counts = torch.empty((P,), dtype=torch.int32, device=dev)
    best = torch.empty((1,), dtype=torch.int32, device=dev)
    _lib.check(_L().lcr_inlier_count(_lib.ptr(T), P, _lib.ptr(src), _lib.ptr(ref), src.shape[0], float(radius), _lib.ptr(start), int(min_count),
                                     _lib.ptr(counts), _lib.ptr(best), _sp(T)), "lcr_inlier_count")
    return counts, best


def local_global_registration(src, ref, score, hyp_start, seg_hyp_start, radius, min_count, steps, want_details=False, correspondence_limit=None):
    """local_to_global_registration (local_global_registration.py:134-201) for S pairs in one native call: correspondences stacked
    pair-major, hyp_start int32 [H+1] (one chunk per patch correspondence), seg_hyp_start int32 [S+1] (the chunks of every pair)
    -> T [S,4,4] (and, with want_details, the hypotheses [H,4,4], their inlier counts [H] and the winner per pair [S]).
    correspondence_limit (:152-160): a pair with more correspondences verifies and refits on its highest-scoring `limit` ones only."""
    dev = src.device
    n, H, S = src.shape[0], hyp_start.numel() - 1, seg_hyp_start.numel() - 1
    nbytes = ctypes.c_size_t(0)
    _lib.check(_L().lcr_lgr_ws_bytes(n, H, S, ctypes.byref(nbytes)), "lcr_lgr_ws_bytes")
    ws = _lib.workspace(nbytes.value, dev)
    T = torch.empty((S, 4, 4), dtype=torch.float32, device=dev)
    hyp = torch.empty((H, 4, 4), dtype=torch.float32, device=dev) if want_details else None
    counts = torch.empty((H,), dtype=torch.int32, device=dev) if want_details else None
    best = torch.empty((S,), dtype=torch.int32, device=dev) if want_details else None
    _lib.check(_L().lcr_local_global_registration_ex(_lib.ptr(src.contiguous()), _lib.ptr(ref.contiguous()), _lib.ptr(score.contiguous()), n,
                                                     _lib.ptr(hyp_start.contiguous()), H, _lib.ptr(seg_hyp_start.contiguous()), S, float(radius),
                                                     int(min_count), int(steps), int(correspondence_limit or 0), _lib.ptr(T), _lib.ptr(hyp),
                                                     _lib.ptr(counts), _lib.ptr(best), _lib.ptr(ws), ws.numel(), _sp(src)),
               "lcr_local_global_registration")
    return (T, hyp, counts, best) if want_details else T


def inlier_weights(T_all, sel, src, ref, score, radius):
    w = torch.empty_like(score)
    _lib.check(_L().lcr_inlier_weights(_lib.ptr(T_all), _lib.ptr(sel), _lib.ptr(src), _lib.ptr(ref), _lib.ptr(score), src.shape[0], float(radius),
                                       _lib.ptr(w), _sp(T_all)), "lcr_inlier_weights")
    return w
