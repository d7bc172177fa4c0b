"""KPConv — same parameters/buffers as the reference module (weights (K,Cin,Cout), bias, kernel_points buffer;
experiments/lcrnet/modules/kpconv/kpconv.py:10-122) with the forward on the HIP kernels:
lcr_kpconv_aggregate (gather + influences + aggregation) -> lcr_gemm_f32 (kernel-point contraction with the
neighbour-count division, bias and GroupNorm statistics fused in its epilogue)."""
import math
import os

import torch
import torch.nn as nn

from ... import functional as F
from ...weights import base_kernel_points


# Opt-in: the one-launch KPConv for C = 32 (lcr_kpconv_fused).  It removes the (M, 480) intermediate from memory, and measured
# 1.75-2.1x SLOWER than aggregate + GEMM on MI355X (324 vs 185 us, 190 vs 88 us: LABNOTES.md §4.2) — the gather needs the occupancy
# that the tile in LDS and the weights in registers take away.
_FUSED = bool(os.environ.get("LCR_KPCONV_FUSED"))


class KPConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, radius, sigma, bias=False, dimension=3, inf=1e6, eps=1e-9):
        super().__init__()
        assert kernel_size == 15 and dimension == 3, "the HIP path implements the 15-point rigid 3-D kernel of the reference config"
        self.kernel_size, self.in_channels, self.out_channels = kernel_size, in_channels, out_channels
        self.radius, self.sigma, self.dimension = radius, sigma, dimension
        self.inf, self.eps = inf, eps
        self.weights = nn.Parameter(torch.zeros(kernel_size, in_channels, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter("bias", None)
        nn.init.kaiming_uniform_(self.weights, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = nn.init._calculate_fan_in_and_fan_out(self.weights)
            bound = 1 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)
        # The reference rotates/jitters the disposition randomly at construction (kernel_points.py:426-455) and stores the
        # result in the checkpoint; real values therefore always come from load_state_dict.  Default: un-rotated disposition.
        self.register_buffer("kernel_points", torch.from_numpy(base_kernel_points() * radius).float())
        self._kp_cache = None
        self._wt_cache = None

    def weights_t(self):
        """The (15 Cin, Cout) contraction matrix transposed to [Cout, 15 Cin] (rows contiguous along k), cached until the weights
        change: with both operands k-contiguous the contraction takes the K-deep GEMM form (LDS-direct loads, 16-B fragment reads
        for both operands).  Same products, same summation order: the result does not change."""
        w = self.weights
        c = self._wt_cache
        if c is None or not c[0].same(w):
            with F.derived_lock:
                c = self._wt_cache
                if c is None or not c[0].same(w):
                    wt = w.detach().reshape(self.kernel_size * self.in_channels, self.out_channels).t().contiguous()
                    c = self._wt_cache = (F.WeightStamp(w), F.publish_derived(wt) if wt.is_cuda else wt)
        return c[1]

    def weights_t_split(self):
        """The three bf16 terms of weights_t() (functional.split_bf16x3), cached with it: the contraction on the bf16 matrix cores."""
        wt = self.weights_t()
        c = getattr(self, "_wts_cache", None)
        if c is None or c[0] is not wt:
            with F.derived_lock:
                c = getattr(self, "_wts_cache", None)
                if c is None or c[0] is not wt:
                    c = self._wts_cache = (wt, F.publish_derived(F.split_bf16x3(wt)))
        return c[1]

    def kernel_points_host(self):
        kp = self.kernel_points
        c = self._kp_cache
        if c is None or not c[0].same(kp):
            c = self._kp_cache = (F.WeightStamp(kp), kp.detach().cpu().numpy().copy())   # one D2H per weight load, not per forward
        return c[1]

    def _apply(self, fn, *args, **kwargs):
        F.drop_derived(self, "_kp_cache", "_wt_cache", "_wts_cache")      # .to() / .cuda() / .cpu(): new tensors, possibly at the old addresses
        return super()._apply(fn, *args, **kwargs)

    def forward_raw(self, s_feats, q_points, s_points, neighbor_indices, s_pos=None, seg_len=None, groups=0, order=None):
        """Returns (q_feats (M,Cout), stats) — stats = GroupNorm sums of the output when groups > 0."""
        kp = self.kernel_points_host()
        if self.in_channels == 1:
            out = F.kpconv_cin1(s_feats.contiguous().view(-1), q_points, s_points, neighbor_indices, kp, self.sigma,
                                self.weights, self.bias, order=order)
            stats = F.groupnorm_stats(out, groups, seg_len) if groups else None
            return out, stats
        if s_pos is None:
            s_pos = F.row_positive(s_feats)
        if (_FUSED and self.in_channels == self.out_channels == F.KPCONV_FUSED_C and neighbor_indices.shape[1] <= 128
                and (seg_len is None or seg_len.numel() <= 64)):
            # aggregate + contraction in one launch: the (M, 480) intermediate never reaches memory
            return F.kpconv_fused(s_feats, s_pos, q_points, s_points, neighbor_indices, kp, self.sigma, self.weights, self.bias,
                                  seg_len=seg_len, groups=groups, order=order)
        A, nn_cnt = F.kpconv_aggregate(s_feats, s_pos, q_points, s_points, neighbor_indices, kp, self.sigma, order=order)
        wt = self.weights_t()
        if F.gemm_split_enabled() and F.gemm_split_ok(wt.shape[0], wt.shape[1]):
            return F.gemm_bsplit(A, self.weights_t_split(), bias=self.bias, rowdiv=nn_cnt, seg_len=seg_len, groups=groups)
        return F.gemm(A, wt, trans_b=True, bias=self.bias, rowdiv=nn_cnt, seg_len=seg_len, groups=groups)

    def forward(self, s_feats, q_points, s_points, neighbor_indices):
        return self.forward_raw(s_feats, q_points, s_points, neighbor_indices)[0]

    def __repr__(self):
        return (f"KPConv(kernel_size: {self.kernel_size}, in_channels: {self.in_channels}, out_channels: {self.out_channels}, "
                f"radius: {self.radius:g}, sigma: {self.sigma:g}, bias: {self.bias is not None})")
