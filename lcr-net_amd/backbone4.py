"""KPEncoder — experiments/lcrnet/backbone4.py:11-89 (4 stages / 11 blocks), same attribute names => same checkpoint keys."""
import os

import torch.nn as nn

from . import functional as F
from . import native_encoder
from .modules.kpconv import ConvBlock, ResidualBlock, StageContext


def segment_min_rows(data_dict, stages=4):
    """Rows of the shortest GroupNorm segment per stage, from HOST data only (None = unknown — the normalise-on-load GEMM, whose
    64-row blocks may span at most two segments, is then not used).  Known exactly in two cases only: the caller states the rows of
    every segment ('segment_rows_host': per stage a list of host ints — LCRNet.forward_pairs does, for its per-pair segments), or
    the segments are the clouds themselves (as many segments as entries of 'lengths_host').  Any other grouping of clouds into
    segments is NOT guessed."""
    seg, host, stated = data_dict.get("segment_lengths"), data_dict.get("lengths_host"), data_dict.get("segment_rows_host")
    if seg is None:
        return [None] * stages
    rows = []
    for i in range(stages):
        n_seg = int(seg[i].numel())
        if stated is not None and len(stated[i]) == n_seg and n_seg > 0:
            rows.append(min(int(v) for v in stated[i]))
        elif host is not None and n_seg > 0 and len(host[i]) == n_seg:
            rows.append(min(int(v) for v in host[i]))
        else:
            rows.append(None)
    return rows


class KPEncoder(nn.Module):
    def __init__(self, input_dim, init_dim, kernel_size, init_radius, init_sigma, group_norm):
        super().__init__()
        self._native_table = None        # native_encoder.table_for's cache (weights as raw pointers + derived tensors)
        d, k, r, s, g = init_dim, kernel_size, init_radius, init_sigma, group_norm
        self.encoder1_1 = ConvBlock(input_dim, d, k, r, s, g)
        self.encoder1_2 = ResidualBlock(d, d * 2, k, r, s, g)
        self.encoder2_1 = ResidualBlock(d * 2, d * 2, k, r, s, g, strided=True)
        self.encoder2_2 = ResidualBlock(d * 2, d * 4, k, r * 2, s * 2, g)
        self.encoder2_3 = ResidualBlock(d * 4, d * 4, k, r * 2, s * 2, g)
        self.encoder3_1 = ResidualBlock(d * 4, d * 4, k, r * 2, s * 2, g, strided=True)
        self.encoder3_2 = ResidualBlock(d * 4, d * 8, k, r * 4, s * 4, g)
        self.encoder3_3 = ResidualBlock(d * 8, d * 8, k, r * 4, s * 4, g)
        self.encoder4_1 = ResidualBlock(d * 8, d * 8, k, r * 4, s * 4, g, strided=True)
        self.encoder4_2 = ResidualBlock(d * 8, d * 16, k, r * 8, s * 8, g)
        self.encoder4_3 = ResidualBlock(d * 16, d * 16, k, r * 8, s * 8, g)
        # forward through lcr_encoder_forward (one native call) when the inputs allow it; LCR_NATIVE_ENCODER=0: the module tree below
        self.native = os.environ.get("LCR_NATIVE_ENCODER", "1") != "0"

    def _apply(self, fn, *args, **kwargs):
        self._native_table = None        # .to() / .cuda() / .cpu(): every tensor is replaced, possibly at its old address
        return super()._apply(fn, *args, **kwargs)

    def __getstate__(self):                     # the table holds raw device pointers: never pickled / deep-copied with the module
        d = dict(self.__dict__)
        d["_native_table"] = None
        return d

    def forward(self, feats, data_dict):
        """data_dict: 'points'[4], 'neighbors'[4], 'subsampling'[3] (+ optional 'segment_lengths'[4]: per-stage device
        int64 GroupNorm segment lengths; absent => one segment = whole stack, the reference's semantics)."""
        P, N, S = data_dict["points"], data_dict["neighbors"], data_dict["subsampling"]
        seg = data_dict.get("segment_lengths")
        order = data_dict.get("order")
        # One native call (csrc/encoder.hip): the same launches in the same order issued by C++ — bit-identical outputs
        # (tests/test_encoder_gpu.py), ~30 % less host time per pass and no interpreter lock held while the pass is issued.  In the
        # descriptor pipeline both drivers measure the same rate today (2 413 vs 2 421 scans/s, LABNOTES.md §4.2; round 1's 10 % deficit
        # went away with the launch-turn gate and the prioritised pre-processing stream), so the native one is the default.
        if self.native and native_encoder.eligible(feats, data_dict):
            return native_encoder.forward(self, feats, data_dict)
        rows = segment_min_rows(data_dict)
        ctx = [StageContext(None if seg is None else seg[i], None if order is None else order[i], rows[i]) for i in range(4)]
        with F.stats_arena(feats.device):
            return self._forward(feats, P, N, S, ctx)

    def _forward(self, feats, P, N, S, ctx):
        f1 = self.encoder1_1(feats, P[0], P[0], N[0], ctx[0], ctx[0])
        f1 = self.encoder1_2(f1, P[0], P[0], N[0], ctx[0], ctx[0])
        f2 = self.encoder2_1(f1, P[1], P[0], S[0], ctx[1], ctx[0])
        f2 = self.encoder2_2(f2, P[1], P[1], N[1], ctx[1], ctx[1])
        f2 = self.encoder2_3(f2, P[1], P[1], N[1], ctx[1], ctx[1])
        f3 = self.encoder3_1(f2, P[2], P[1], S[1], ctx[2], ctx[1])
        f3 = self.encoder3_2(f3, P[2], P[2], N[2], ctx[2], ctx[2])
        f3 = self.encoder3_3(f3, P[2], P[2], N[2], ctx[2], ctx[2])
        f4 = self.encoder4_1(f3, P[3], P[2], S[2], ctx[3], ctx[2])
        f4 = self.encoder4_2(f4, P[3], P[3], N[3], ctx[3], ctx[3])
        f4 = self.encoder4_3(f4, P[3], P[3], N[3], ctx[3], ctx[3])
        return [f1, f2, f3, f4]
