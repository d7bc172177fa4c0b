"""KPConv blocks with the reference's module tree (so state_dict keys match: SURVEY Appendix B) —
experiments/lcrnet/modules/kpconv/modules.py:33-225 — executed as fused HIP launches:

    UnaryBlock   : lcr_gemm_f32 (Linear + bias + GN sums)            -> lcr_groupnorm_apply (GN + LeakyReLU [+ pos flags])
    ConvBlock    : KPConv.forward_raw (aggregate + GEMM + GN sums)    -> lcr_groupnorm_apply
    ResidualBlock: unary1 -> KPConv -> norm_conv+LeakyReLU -> unary2 GEMM -> [maxpool] -> [shortcut GEMM]
                   -> ONE lcr_groupnorm_apply doing GN(unary2) + GN(shortcut)/shortcut + LeakyReLU  (modules.py:207-225)

GroupNorm statistics are *segmented*: `StageContext.seg_len` lists the rows of every GroupNorm segment of a stage
(None = the reference's behaviour: one segment = the whole stack, modules.py:46-50).
"""
import os

import torch
import torch.nn as nn

from ... import functional as F
from .kpconv import KPConv


class StageContext:
    """Per-stage execution context: GroupNorm segment lengths (device int64 [S]) or None, and an optional spatially coherent
    processing order of the stage's points (device int32 [N]; data.precompute_batch provides the support grid's cell order)."""

    def __init__(self, seg_len=None, order=None, min_rows=None):
        self.seg_len = seg_len
        self.order = order
        self.min_rows = min_rows        # host int: rows of the shortest segment when known (None with segments = unknown)

    def norm_on_load(self, K, N, M):
        """May ResidualBlock fold norm_conv + LeakyReLU into unary2's GEMM?  (segments of >= 64 rows, the light GEMM form).  The
        same rule as the native driver's (csrc/encoder.hip: min_rows >= 64, min_rows = the stack's rows when there is one segment),
        so the two drivers take the same numeric path for every input."""
        if _NO_NORM_ON_LOAD or not F.gemm_anorm_ok(K, N):
            return False
        rows = M if self.seg_len is None else self.min_rows
        return rows is not None and rows >= F.ANORM_MIN_SEG_ROWS


_NO_NORM_ON_LOAD = bool(os.environ.get("LCR_NO_NORM_ON_LOAD"))     # A/B switch
_WHOLE = StageContext(None)


class GroupNorm(nn.Module):
    def __init__(self, num_groups, num_channels):
        super().__init__()
        self.num_groups, self.num_channels = num_groups, num_channels
        self.norm = nn.GroupNorm(num_groups, num_channels)     # holds weight / bias under the reference key `norm.norm.*`

    def forward(self, x, ctx=_WHOLE, act=False, slope=0.1):
        stats = F.groupnorm_stats(x.contiguous(), self.num_groups, ctx.seg_len)
        return F.groupnorm_apply(x.contiguous(), stats, self.norm.weight, self.norm.bias, self.num_groups, ctx.seg_len, act=act, slope=slope)


class UnaryBlock(nn.Module):
    def __init__(self, in_channels, out_channels, group_norm, has_relu=True, bias=True, layer_norm=False):
        super().__init__()
        assert not layer_norm, "layer_norm=True is not used by the reference configs"
        self.in_channels, self.out_channels, self.group_norm = in_channels, out_channels, group_norm
        self.mlp = nn.Linear(in_channels, out_channels, bias=bias)
        self.norm = GroupNorm(group_norm, out_channels)
        self.leaky_relu = nn.LeakyReLU(0.1) if has_relu else None

    def weight_split(self):
        w = self.mlp.weight
        c = getattr(self, "_ws_cache", None)
        if c is None or not c[0].same(w):
            with F.derived_lock:
                c = getattr(self, "_ws_cache", None)
                if c is None or not c[0].same(w):
                    c = self._ws_cache = (F.WeightStamp(w), F.publish_derived(F.split_bf16x3(w)))
        return c[1]

    def _apply(self, fn, *args, **kwargs):
        F.drop_derived(self, "_ws_cache")
        return super()._apply(fn, *args, **kwargs)

    def raw(self, x, ctx):
        """Linear + GroupNorm sums (no normalisation yet)."""
        if F.gemm_split_enabled() and F.gemm_split_ok(self.out_channels, self.in_channels):
            return F.gemm_bsplit(x.contiguous(), self.weight_split(), bias=self.mlp.bias, seg_len=ctx.seg_len, groups=self.group_norm)
        return F.gemm(x.contiguous(), self.mlp.weight, trans_b=True, bias=self.mlp.bias, seg_len=ctx.seg_len, groups=self.group_norm)

    def forward(self, x, ctx=_WHOLE, want_pos=False):
        y, stats = self.raw(x, ctx)
        return F.groupnorm_apply(y, stats, self.norm.norm.weight, self.norm.norm.bias, self.group_norm, ctx.seg_len,
                                 act=self.leaky_relu is not None, want_pos=want_pos)


class LastUnaryBlock(nn.Module):
    def __init__(self, in_channels, out_channels, bias=True):
        super().__init__()
        self.mlp = nn.Linear(in_channels, out_channels, bias=bias)

    def forward(self, x):
        w = self.mlp.weight
        if F.gemm_split_enabled() and F.gemm_split_ok(w.shape[0], w.shape[1]):
            c = getattr(self, "_ws_cache", None)
            if c is None or not c[0].same(w):
                with F.derived_lock:
                    c = getattr(self, "_ws_cache", None)
                    if c is None or not c[0].same(w):
                        c = self._ws_cache = (F.WeightStamp(w), F.publish_derived(F.split_bf16x3(w)))
            return F.gemm_bsplit(x.contiguous(), c[1], bias=self.mlp.bias)[0]
        return F.gemm(x.contiguous(), w, trans_b=True, bias=self.mlp.bias)[0]

    def _apply(self, fn, *args, **kwargs):
        F.drop_derived(self, "_ws_cache")
        return super()._apply(fn, *args, **kwargs)


class ConvBlock(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, radius, sigma, group_norm, negative_slope=0.1, bias=True, layer_norm=False):
        super().__init__()
        assert not layer_norm
        self.in_channels, self.out_channels, self.group_norm = in_channels, out_channels, group_norm
        self.KPConv = KPConv(in_channels, out_channels, kernel_size, radius, sigma, bias=bias)
        self.norm = GroupNorm(group_norm, out_channels)
        self.leaky_relu = nn.LeakyReLU(negative_slope=negative_slope)
        self.negative_slope = negative_slope

    def forward(self, s_feats, q_points, s_points, neighbor_indices, q_ctx=_WHOLE, s_ctx=_WHOLE):
        x, stats = self.KPConv.forward_raw(s_feats, q_points, s_points, neighbor_indices, seg_len=q_ctx.seg_len, groups=self.group_norm,
                                           order=q_ctx.order)
        return F.groupnorm_apply(x, stats, self.norm.norm.weight, self.norm.norm.bias, self.group_norm, q_ctx.seg_len,
                                 act=True, slope=self.negative_slope)


class ResidualBlock(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, radius, sigma, group_norm, strided=False, bias=True, layer_norm=False):
        super().__init__()
        assert not layer_norm
        self.in_channels, self.out_channels, self.strided, self.group_norm = in_channels, out_channels, strided, group_norm
        mid = out_channels // 4
        self.unary1 = UnaryBlock(in_channels, mid, group_norm, bias=bias) if in_channels != mid else nn.Identity()
        self.KPConv = KPConv(mid, mid, kernel_size, radius, sigma, bias=bias)
        self.norm_conv = GroupNorm(group_norm, mid)
        self.unary2 = UnaryBlock(mid, out_channels, group_norm, has_relu=False, bias=bias)
        self.unary_shortcut = (UnaryBlock(in_channels, out_channels, group_norm, has_relu=False, bias=bias)
                               if in_channels != out_channels else nn.Identity())
        self.leaky_relu = nn.LeakyReLU(0.1)

    def forward(self, s_feats, q_points, s_points, neighbor_indices, q_ctx=_WHOLE, s_ctx=_WHOLE):
        g = self.group_norm
        s_feats = s_feats.contiguous()
        if isinstance(self.unary1, nn.Identity):
            x, pos = s_feats, None
        else:
            x, pos = self.unary1(s_feats, s_ctx, want_pos=True)                       # Linear+GN+LeakyReLU, pos flags for the count
        x, stats = self.KPConv.forward_raw(x, q_points, s_points, neighbor_indices, s_pos=pos, seg_len=q_ctx.seg_len, groups=g,
                                           order=q_ctx.order)
        if q_ctx.norm_on_load(x.shape[1], self.out_channels, x.shape[0]):
            # norm_conv + LeakyReLU applied while unary2's GEMM stages its A tiles: no stand-alone GroupNorm pass over x
            y, ystats = F.gemm_anorm(x, stats, self.norm_conv.norm.weight, self.norm_conv.norm.bias, g, self.unary2.mlp.weight,
                                     bias=self.unary2.mlp.bias, seg_len=q_ctx.seg_len, groups=g)
        else:
            x = F.groupnorm_apply(x, stats, self.norm_conv.norm.weight, self.norm_conv.norm.bias, g, q_ctx.seg_len, act=True)
            y, ystats = self.unary2.raw(x, q_ctx)                                      # normalised below, fused with the shortcut
        shortcut = F.maxpool(s_feats, neighbor_indices, order=q_ctx.order) if self.strided else s_feats
        if isinstance(self.unary_shortcut, nn.Identity):
            res, res_norm = shortcut, None
        else:
            res, rstats = self.unary_shortcut.raw(shortcut, q_ctx)
            res_norm = (rstats, self.unary_shortcut.norm.norm.weight, self.unary_shortcut.norm.norm.bias)
        return F.groupnorm_apply(y, ystats, self.unary2.norm.norm.weight, self.unary2.norm.norm.bias, g, q_ctx.seg_len,
                                 res=res, res_norm=res_norm, act=True, slope=0.1)
