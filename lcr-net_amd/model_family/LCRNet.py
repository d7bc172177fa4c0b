"""LCRNet — the pair model (experiments/lcrnet/model_family/LCRNet.py:25-321) with the reference's module tree, so
`best-model-mixed.tar` loads unchanged (373 tensors).

Everything runs on the HIP path: KeypointDetection (LCRNet.py:124-159: KPEncoder over the pair stack, 3D-RoFormer on the
coarsest stage, Vote_Encoder), GlobalDescritionHEAD on the PRE-transformer features of each cloud (:115-122, :296-297) and
DenseMatchingHEAD (:161-272: point-to-node partition, node-level Sinkhorn + dustbin matching, KPDecoder, patch-level
Sinkhorn, local-to-global registration with device-side 3x3 SVDs) — with no device->host->device tensor round trips (the
reference has four); only a few scalar sizes are read back where output shapes depend on the data.
"""
import torch
import torch.nn as nn

from ..backbone4 import KPEncoder
from ..config import make_cfg
from ..modules.kpconv import LastUnaryBlock, ResidualBlock, UnaryBlock
from ..modules.kpconv.modules import StageContext
from .. import functional as F
from ..modules.netvlad import NetVLADLoupe2
from ..modules.ops import radius_search
from ..modules.thdroformer import ThDRoFormer


class Vote_layer(nn.Module):
    """modules/vote/vote.py:112-182 (output_feats=False): shared MLP 256->512->256 (Linear, LayerNorm, ReLU), ctr_reg 256->3,
    offsets clamped to MAX_TRANSLATE_RANGE."""

    def __init__(self, input_feats_dim=256, max_translate_range=4.2):
        super().__init__()
        c = input_feats_dim
        self.mlp_modules = nn.Sequential(nn.Linear(c, 2 * c), nn.LayerNorm(2 * c), nn.ReLU(), nn.Linear(2 * c, c), nn.LayerNorm(c), nn.ReLU())
        self.ctr_reg = nn.Linear(c, 3)
        self.max_translate_range = max_translate_range

    def forward(self, xyz, features):
        x = features
        for i in (0, 3):
            lin, ln = self.mlp_modules[i], self.mlp_modules[i + 1]
            x = F.relu_(F.add_layernorm(F.linear(x, lin.weight, lin.bias), None, ln.weight, ln.bias, ln.eps))
        off = F.linear(x, self.ctr_reg.weight, self.ctr_reg.bias)
        return F.vote_shift(xyz, off, self.max_translate_range)


class Vote_Encoder(nn.Module):
    """backbone4.py:92-220: vote layer -> greedy NMS -> node centres (mean of in-radius votes) -> two radius searches ->
    encoder6_1..3 (radii 8x/16x/16x init_radius).  All on the device: the reference's three `.cpu()` radius searches
    (:149-157, :191-206) and its Python NMS loop are HIP kernels here."""

    def __init__(self, init_dim, kernel_size, init_radius, init_sigma, group_norm, vote_cfg, neighbor_limits):
        super().__init__()
        self.vote = Vote_layer(256, vote_cfg["MAX_TRANSLATE_RANGE"])
        self.NMS_radius = vote_cfg["NMS_radius"]
        d, k, r, s, g = init_dim, kernel_size, init_radius, init_sigma, group_norm
        self.encoder6_1 = ResidualBlock(d * 4, d * 4, k, r * 8, s * 8, g, strided=True)
        self.encoder6_2 = ResidualBlock(d * 4, d * 8, k, r * 16, s * 16, g)
        self.encoder6_3 = ResidualBlock(d * 8, d * 8, k, r * 16, s * 16, g)
        self.init_radius = init_radius
        self.neighbor_limits = list(neighbor_limits)

    def forward(self, feats, data_dict, neighbor_limit=None, pairs=1):
        """pairs > 1: the stack holds `pairs` registration pairs; GroupNorm statistics are taken per pair (the reference's
        statistics over its one-pair stack), everything else is per cloud anyway."""
        limits = list(neighbor_limit) if neighbor_limit is not None else self.neighbor_limits
        lens_c = data_dict["lengths"][-1]
        points_c = data_dict["points"][-1].contiguous()
        shifted = self.vote(points_c, feats)
        keep, length = F.greedy_nms(shifted, lens_c, self.NMS_radius)
        nms_pts = shifted[keep.bool()].contiguous()                       # compaction (host sync: output size is data dependent)
        pad = shifted.shape[0]
        knn = radius_search(nms_pts, shifted, length, lens_c, 2.4, limits[-1], check=False)                    # backbone4.py:149-156 (literal 2.4)
        centers = F.neighbor_mean(shifted, knn, pad)
        sub = radius_search(centers, points_c, length, lens_c, self.init_radius * 8, limits[-2], check=False)
        nb = radius_search(centers, centers, length, length, self.init_radius * 16, limits[-1], check=False)
        if pairs > 1:                                                  # min_rows unknown on the host here (node counts live on the device):
            q_ctx, s_ctx = StageContext(length.view(pairs, 2).sum(1)), StageContext(lens_c.view(pairs, 2).sum(1))   # plain GroupNorm pass
        else:
            q_ctx = s_ctx = StageContext(None)
        f = self.encoder6_1(feats, centers, points_c, sub, q_ctx, s_ctx)
        f = self.encoder6_2(f, centers, centers, nb, q_ctx, q_ctx)
        f = self.encoder6_3(f, centers, centers, nb, q_ctx, q_ctx)
        return {"shifted_points_c": shifted, "nms_mask": keep, "length": length, "points_c": centers, "feats_c": f}


class KPDecoder(nn.Module):
    """backbone4.py:333-373: nearest-upsample (column 0 of the upsampling lists) + concat + UnaryBlock, three levels."""

    def __init__(self, init_dim, group_norm):
        super().__init__()
        self.decoder3 = UnaryBlock(init_dim * 12, init_dim * 8, group_norm)
        self.decoder2 = UnaryBlock(init_dim * 12, init_dim * 4, group_norm)
        self.decoder1 = LastUnaryBlock(init_dim * 6, init_dim * 2)

    def forward(self, feats, data_dict, pairs=1):
        f1, f2, f3, f4 = feats
        U = data_dict["upsampling"]
        L = data_dict["lengths"]
        c3 = StageContext(L[2].view(pairs, 2).sum(1) if pairs > 1 else None)      # GroupNorm over a pair's two clouds
        c2 = StageContext(L[1].view(pairs, 2).sum(1) if pairs > 1 else None)
        l3 = self.decoder3(F.upsample_concat(f4, U[2], f3), c3)
        l2 = self.decoder2(F.upsample_concat(l3, U[1], f2), c2)
        l1 = self.decoder1(F.upsample_concat(l2, U[0], f1))
        return [l1, l2, l3]


class LearnableLogOptimalTransport(nn.Module):
    def __init__(self, num_iterations):
        super().__init__()
        self.num_iterations = num_iterations
        self.register_parameter("alpha", nn.Parameter(torch.tensor(1.0)))


class LCRNet(nn.Module):
    # The reference has three classes with this body: LCRNet (global descriptors + matching), LCRNet_Matching_infer (the same
    # without the NetVLAD head) and LCRNet_Matching (the evaluation harness' variant: extra outputs, ground-truth labels).  The two
    # others derive from this one (model_family/LCRNet_Matching*.py) and switch these:
    global_head = True              # NetVLAD descriptors of both clouds (LCRNet.py:115-122, :296-297)
    matching_extras = False         # node_matching_scores / node masks / matching_scores in the output (LCRNet_Matching.py:212-214, :274)

    def __init__(self, cfg=None):
        super().__init__()
        cfg = cfg or make_cfg()
        b, g = cfg["backbone"], cfg["GAT"]
        self.encoder = KPEncoder(b["input_dim"], b["init_dim"], b["kernel_size"], b["init_radius"], b["init_sigma"], b["group_norm"])
        self.vote_encoder = Vote_Encoder(b["init_dim"], b["kernel_size"], b["init_radius"], b["init_sigma"], b["group_norm"], cfg["Vote"],
                                         cfg["neighbor_limits"])
        self.num_points_in_patch = cfg["model"]["num_points_in_patch"]
        fm = cfg.get("fine_matching", {})
        self.acceptance_radius = fm.get("acceptance_radius", 0.45)
        self.correspondence_threshold = fm.get("correspondence_threshold", 3)
        self.num_refinement_steps = fm.get("num_refinement_steps", 5)
        self.mutual = bool(fm.get("mutual", False))                  # LocalGlobalRegistration(mutual=...), local_global_registration.py:84-87
        self.topk = int(fm.get("topk", 1))                           # LocalGlobalRegistration(k=...), :56-82
        # the switches the shipped config_model.py:85-93 leaves off, all built (lcr_topk_matching_ex / lcr_local_global_registration_ex)
        self.use_dustbin = bool(fm.get("use_dustbin", True))         # :62-65, :74-77, LCRNet.py:256-257
        self.confidence_threshold = float(fm.get("confidence_threshold", 0.0))
        self.use_global_score = bool(fm.get("use_global_score", False))   # :236-237
        self.correspondence_limit = fm.get("correspondence_limit")   # :152-160
        if self.topk < 1 or (self.correspondence_limit is not None and int(self.correspondence_limit) < 1):
            raise ValueError("fine_matching: topk and correspondence_limit must be positive")
        self.proj_node_overlap_score = nn.Linear(g["output_dim"] * 2, 1)
        self.transformer = ThDRoFormer(g["input_dim"], g["output_dim"], g["hidden_dim"], g["num_heads"], g["num_layers"], g["k"])
        self.kpdecoder = KPDecoder(b["init_dim"], b["group_norm"])
        self.node_optimal_transport = LearnableLogOptimalTransport(cfg["model"]["num_sinkhorn_iterations"])
        self.optimal_transport = LearnableLogOptimalTransport(cfg["model"]["num_sinkhorn_iterations"])
        if self.global_head:
            self.netvlad = NetVLADLoupe2(feature_size=1024, cluster_size=64, output_dim=256, gating=True, add_norm=True, is_training=False)

    # ---- LocalGlobalRegistration (geotransformer/local_global_registration.py:134-246; k=1, dustbin; mutual from the config) --------
    def _local_global_registration_group(self, ref_knn_points, src_knn_points, ref_masks, src_masks, log_scores, patch_off, global_scores=None):
        """The registration tail of S pairs at once.  Patch correspondences of all pairs are stacked (pair s owns patches
        [patch_off[s], patch_off[s+1])): ONE dustbin top-1 matching over all patches, the matched points gathered once, and the
        hypothesis / inlier-count / refit sequence as one native call whose launches do not depend on S (`lcr_local_global_registration`:
        hypotheses only compete inside their own pair).  -> list of (ref_corr_points, src_corr_points, corr_scores, T) per pair.
        One host read-back of the per-pair correspondence counts (output shapes) on top of top-1 matching's own."""
        Pn, K = ref_masks.shape
        S = len(patch_off) - 1
        bij, sc = F.top1_matching(log_scores, ref_masks, src_masks, mutual=self.mutual, topk=self.topk, use_dustbin=self.use_dustbin,
                                  confidence_threshold=self.confidence_threshold, global_scores=global_scores if self.use_global_score else None)
        if bij.shape[0] == 0:
            raise RuntimeError("no dense correspondences (the reference fails here as well)")
        b, i, j = bij[:, 0].long(), bij[:, 1].long(), bij[:, 2].long()
        rp = F.gather_rows(ref_knn_points.reshape(-1, 3), b * K + i)
        sp = F.gather_rows(src_knn_points.reshape(-1, 3), b * K + j)
        start = torch.zeros(Pn + 1, dtype=torch.int32, device=rp.device)
        start[1:] = torch.cumsum(torch.bincount(b, minlength=Pn), 0).int()           # rows are patch-major: chunk p = [start[p], start[p+1])
        seg = F.host_values(list(patch_off), torch.int32, rp.device)                 # pinned + asynchronous: no stream drain
        rows = start[seg.long()].tolist()                                             # host sync: first correspondence row of every pair
        if any(rows[s + 1] == rows[s] for s in range(S)):
            raise RuntimeError("no dense correspondences (the reference fails here as well)")
        T = F.local_global_registration(sp, rp, sc, start, seg, self.acceptance_radius, self.correspondence_threshold, self.num_refinement_steps,
                                        correspondence_limit=self.correspondence_limit)
        return [(rp[rows[s]:rows[s + 1]], sp[rows[s]:rows[s + 1]], sc[rows[s]:rows[s + 1]], T[s]) for s in range(S)]

    def _local_global_registration(self, ref_knn_points, src_knn_points, ref_masks, src_masks, log_scores, global_scores=None):
        return self._local_global_registration_group(ref_knn_points, src_knn_points, ref_masks, src_masks, log_scores, [0, ref_masks.shape[0]],
                                                     global_scores)[0]

    def forward(self, data_dict, pose=True):
        """Pair stack [pos(ref), anc(src)] (data.py:110-113) -> the reference's output_dict (LCRNet.py:274-321).  GroupNorm
        statistics span the pair unless data_dict['segment_lengths'] says otherwise.  pose=False stops after the transformer
        and the global descriptors."""
        return self.forward_pairs(data_dict, pose)[0]

    def forward_pairs(self, data_dict, pose=True):
        """P registration pairs per call: the stack is [pos_0, anc_0, pos_1, anc_1, ...] (2P clouds, P <= 32) -> a list of P
        output dicts, each what `forward` returns for that pair alone (GroupNorm statistics are taken per pair, the reference's
        one-pair stack; attention, matching and registration are per pair by construction).  The reference runs one pair per
        forward (LCRNet.py:274-321, batch_size 1); batching turns its ~1000 small launches per pair into shared ones: encoder,
        3D-RoFormer (segmented attention launch), NetVLAD, vote encoder and decoder run ONCE over the stack, and only the
        matching / registration tail, whose tensor shapes depend on each pair's data, runs pair by pair."""
        if self.training:
            raise RuntimeError("lcr-net_amd implements inference only; call .eval()")
        feats = data_dict["features"].detach()
        lens_dev = data_dict["lengths"]
        B = lens_dev[-1].numel()
        if B % 2 or B < 2:
            raise RuntimeError("the pair model needs an even number of clouds [pos_0, anc_0, pos_1, anc_1, ...]; got %d" % B)
        P = B // 2
        lens_c = data_dict.get("lengths_c_host")
        if lens_c is None:
            lens_c = lens_dev[-1].tolist()                           # host sync, as the reference's .item() calls (LCRNet.py:127-128)
        lens_c = [int(x) for x in lens_c]
        points_c = data_dict["points"][-1]
        dd = data_dict
        if P > 1 and "segment_lengths" not in data_dict:             # GroupNorm per pair at every stage
            dd = dict(data_dict)
            dd["segment_lengths"] = [l.view(P, 2).sum(1) for l in lens_dev]
            host = data_dict.get("lengths_host")
            if host is not None:                                     # rows of every pair segment, stated exactly (normalise-on-load gate)
                dd["segment_rows_host"] = [[int(h[2 * p]) + int(h[2 * p + 1]) for p in range(P)] for h in host]
        feats_list = self.encoder(feats, dd)
        feats_c = feats_list[-1]
        off_c = [0]
        for n in lens_c:
            off_c.append(off_c[-1] + n)
        n_c = off_c[-1]
        pos_lens, anc_lens = lens_c[0::2], lens_c[1::2]
        if P == 1:
            n0, n1 = lens_c
            pos_c, anc_c = feats_c[:n0].contiguous(), feats_c[n0:n0 + n1].contiguous()
            e0, e1, t0, t1 = self.transformer(points_c[:n0].contiguous(), points_c[n0:n0 + n1].contiguous(), pos_c, anc_c, return_pos_emb=True)
            enhanced = torch.cat([e0, e1], 0)
            emb = [(t0, t1)]
        else:
            # rows of all first clouds / all second clouds, stacked: every Linear and LayerNorm of the transformer runs once
            idx0 = F.host_tensor(torch.cat([torch.arange(off_c[2 * p], off_c[2 * p + 1]) for p in range(P)]), feats_c.device)
            idx1 = F.host_tensor(torch.cat([torch.arange(off_c[2 * p + 1], off_c[2 * p + 2]) for p in range(P)]), feats_c.device)
            e0, e1, t0, t1 = self.transformer(points_c[idx0], points_c[idx1], feats_c[idx0], feats_c[idx1], pos_lens, anc_lens, return_pos_emb=True)
            enhanced = torch.empty((n_c, e0.shape[1]), dtype=e0.dtype, device=e0.device)
            enhanced[idx0] = e0
            enhanced[idx1] = e1
            po, ao = [0], [0]
            for p in range(P):
                po.append(po[-1] + pos_lens[p])
                ao.append(ao[-1] + anc_lens[p])
            emb = [(t0[po[p]:po[p + 1]], t1[ao[p]:ao[p + 1]]) for p in range(P)]
        g = self.netvlad.describe(feats_c[:n_c], lens_c) if self.global_head else None      # pre-transformer features (LCRNet.py:296-297)
        outs = []
        for p in range(P):
            a0, a1, a2 = off_c[2 * p], off_c[2 * p + 1], off_c[2 * p + 2]
            o = {"ori_pos_points_c": points_c[a0:a1], "ori_anc_points_c": points_c[a1:a2],
                 "pos_feats_c_enhanced": enhanced[a0:a1], "anc_feats_c_enhanced": enhanced[a1:a2],
                 "pos_emb": emb[p][0][None], "anc_emb": emb[p][1][None]}
            if g is not None:
                o["pos_feature_global"], o["anc_feature_global"] = g[2 * p:2 * p + 1], g[2 * p + 1:2 * p + 2]
            outs.append(o)
        if not pose:
            outs[0]["feats_list"] = feats_list
            if not self.matching_extras:
                for o in outs:
                    o.pop("pos_emb"), o.pop("anc_emb")
            return outs

        # ---- KeypointDetection tail (LCRNet.py:152-159): once over the stack
        vd = self.vote_encoder(enhanced, data_dict, pairs=P)
        m = [int(x) for x in vd["length"].tolist()]                   # host sync (the reference does length[0] indexing too)
        off_m = [0]
        for x in m:
            off_m.append(off_m[-1] + x)
        L0 = data_dict.get("lengths_host")
        L0 = [int(x) for x in (L0[0] if L0 is not None else lens_dev[0].tolist())]
        off_f = [0]
        for x in L0:
            off_f.append(off_f[-1] + x)
        pts_f = data_dict["points"][0]
        fl = list(feats_list)
        fl[-1] = enhanced                                              # LCRNet.py:154-155
        feats_f = self.kpdecoder(fl, data_dict, pairs=P)[0]
        # ---- DenseMatchingHEAD (:161-272)
        sl = lambda off, i: slice(off[i], off[i + 1])
        if P == 1:
            self._dense_matching(outs[0], pts_f[sl(off_f, 0)].contiguous(), pts_f[sl(off_f, 1)].contiguous(),
                                 feats_f[sl(off_f, 0)].contiguous(), feats_f[sl(off_f, 1)].contiguous(),
                                 vd["points_c"][sl(off_m, 0)].contiguous(), vd["points_c"][sl(off_m, 1)].contiguous(),
                                 vd["feats_c"][sl(off_m, 0)].contiguous(), vd["feats_c"][sl(off_m, 1)].contiguous())
        else:
            self._dense_matching_group(outs, P, pts_f, feats_f, off_f, vd, off_m)
        for p in range(P):
            c = 2 * p
            outs[p].update({"shifted_pos_points_c": vd["shifted_points_c"][sl(off_c, c)], "shifted_anc_points_c": vd["shifted_points_c"][sl(off_c, c + 1)],
                            "length": vd["length"][c:c + 2], "feats_c": vd["feats_c"][off_m[c]:off_m[c + 2]]})
            if not self.matching_extras:
                for k in ("node_matching_scores", "pos_node_masks", "anc_node_masks", "pos_emb", "anc_emb"):
                    outs[p].pop(k, None)
        return outs

    def _dense_matching_group(self, outs, P, pts_f, feats_f, off_f, vd, off_m):
        """DenseMatchingHEAD for P pairs, stage by stage: the node-level optimal transport of all pairs in ONE batched problem set
        (matrices padded to the largest pair with masked rows / columns — LearnableLogOptimalTransport excludes masked entries and
        normalises by the valid counts, learnable_sinkhorn.py:20-66, so the valid block is the unpadded result), its top-1 matching
        in one call, the patch gathers over the whole stack, the patch-level transport of all pairs' patches in one call; only the
        local-to-global registration, whose hypotheses compete inside a pair, runs pair by pair.  Per pair that is 200 / P Sinkhorn
        launches instead of 200."""
        K = self.num_points_in_patch
        dev = pts_f.device
        n_all = pts_f.shape[0]
        import numpy as np
        parts, stacks = [], []
        for g0 in range(0, 2 * P, 64):                                  # one launch sequence per 64 clouds of the stack
            g1 = min(2 * P, g0 + 64)
            _, nm, knn, km = F.point_to_node_partition_stack(pts_f, off_f[g0:g1 + 1], vd["points_c"], off_m[g0:g1 + 1], K)
            stacks.append((nm, knn, km))
            for c in range(g0, g1):
                lo, hi = off_m[c] - off_m[g0], off_m[c + 1] - off_m[g0]
                parts.append((nm[lo:hi], knn[lo:hi], km[lo:hi]))
        nm_all, knn_all, km_all = (stacks[0] if len(stacks) == 1 else tuple(torch.cat([st[i] for st in stacks]) for i in range(3)))
        m = [off_m[c + 1] - off_m[c] for c in range(2 * P)]
        Mx, Nx = max(m[0::2]), max(m[1::2])
        fc = vd["feats_c"]
        # ---- the padded (P, Mx | Nx) node batches and everything per node row, for ALL pairs at once: the row maps are built on the host
        # (plain index arithmetic on the offset lists) and uploaded in ONE copy; a per-pair / per-side loop of torch ops was ~120 launches
        # of 5-10 us per 8-pair call (profiles/r05_pair_model_kernel_summary.md)
        cloud = np.repeat(np.arange(2 * P), m)                                        # cloud of every stacked node row
        local = np.arange(off_m[-1] - off_m[0]) - np.asarray(off_m[:-1])[cloud] + off_m[0]
        side = cloud & 1
        dst = (cloud >> 1) * np.where(side == 0, Mx, Nx) + local                      # row in the flattened (P * Mx) / (P * Nx) batch
        n_f = np.asarray(off_f[1:]) - np.asarray(off_f[:-1])
        host = np.stack([dst, np.asarray(off_f[:-1])[cloud], n_f[cloud], side]).astype(np.int64)
        tab = F.host_tensor(host, dev)
        rows_pos = F.host_tensor(np.nonzero(side == 0)[0], dev)
        rows_anc = F.host_tensor(np.nonzero(side == 1)[0], dev)
        fp = torch.zeros((P * Mx, fc.shape[1]), dtype=fc.dtype, device=dev)
        fa = torch.zeros((P * Nx, fc.shape[1]), dtype=fc.dtype, device=dev)
        rm = torch.zeros((P * Mx,), dtype=torch.bool, device=dev)
        cm = torch.zeros((P * Nx,), dtype=torch.bool, device=dev)
        dpos, danc = tab[0][rows_pos], tab[0][rows_anc]
        fp[dpos] = fc[rows_pos]
        fa[danc] = fc[rows_anc]
        rm[dpos] = nm_all[rows_pos]
        cm[danc] = nm_all[rows_anc]
        fp, fa, rm, cm = fp.view(P, Mx, -1), fa.view(P, Nx, -1), rm.view(P, Mx), cm.view(P, Nx)
        ns = F.log_optimal_transport(F.bmm_nt(fp, fa), rm, cm, self.node_optimal_transport.alpha,
                                     scale=1.0 / fc.shape[1] ** 0.5, iters=self.node_optimal_transport.num_iterations)
        nbij, nscore = F.top1_matching(ns)                              # rows (pair, i, j), pair-major; padding never beats a dustbin
        nb = nbij.long()
        per_pair = torch.bincount(nb[:, 0], minlength=P).tolist()                   # host sync: node correspondences per pair
        q_off = [0]
        for x in per_pair:
            q_off.append(q_off[-1] + x)
        # patch point indices of the matched nodes as rows of the WHOLE stack (pad = n_all), every pair and side in one pass
        knn_g = torch.where(knn_all == tab[2][:, None], torch.full_like(knn_all, n_all), knn_all + tab[1][:, None])
        off_even = F.host_values([off_m[2 * p] - off_m[0] for p in range(P)], torch.int64, dev)
        off_odd = F.host_values([off_m[2 * p + 1] - off_m[0] for p in range(P)], torch.int64, dev)
        gi, gj = nb[:, 1] + off_even[nb[:, 0]], nb[:, 2] + off_odd[nb[:, 0]]
        pk_g, ak_g, pkm, akm = knn_g[gi], knn_g[gj], km_all[gi], km_all[gj]
        node_idx = [(nb[q_off[p]:q_off[p + 1], 1], nb[q_off[p]:q_off[p + 1], 2]) for p in range(P)]
        pk_g, ak_g, pkm, akm = pk_g.contiguous(), ak_g.contiguous(), pkm.contiguous(), akm.contiguous()
        pkp, akp = F.gather_rows(pts_f, pk_g), F.gather_rows(pts_f, ak_g)
        # feature gathers + batched product + scaling + dustbin / mask padding: one kernel straight from feats_f (lcr_patch_scores)
        ms = F.patch_log_optimal_transport(feats_f, pk_g, feats_f, ak_g, pkm, akm, self.optimal_transport.alpha,
                                           scale=1.0 / feats_f.shape[1] ** 0.5, iters=self.optimal_transport.num_iterations)
        lgr = self._local_global_registration_group(pkp, akp, pkm, akm, ms, q_off, nscore)      # all pairs: one matching, one registration sequence
        for p in range(P):
            q = slice(q_off[p], q_off[p + 1])
            c = 2 * p
            rp, sp, sc, T = lgr[p]
            pi, ai = node_idx[p]
            outs[p].update({
                "pos_points_c": vd["points_c"][off_m[c]:off_m[c + 1]], "anc_points_c": vd["points_c"][off_m[c + 1]:off_m[c + 2]],
                "pos_feats_c": fc[off_m[c]:off_m[c + 1]], "anc_feats_c": fc[off_m[c + 1]:off_m[c + 2]],
                "pos_points_f": pts_f[off_f[c]:off_f[c + 1]], "anc_points_f": pts_f[off_f[c + 1]:off_f[c + 2]],
                "pos_node_knn_indices": parts[c][1], "pos_node_knn_masks": parts[c][2], "anc_node_knn_indices": parts[c + 1][1],
                "anc_node_knn_masks": parts[c + 1][2], "pos_node_corr_indices": pi, "anc_node_corr_indices": ai,
                "node_corr_scores": nscore[q], "pos_feats_f": feats_f[off_f[c]:off_f[c + 1]], "anc_feats_f": feats_f[off_f[c + 1]:off_f[c + 2]],
                "pos_node_corr_knn_points": pkp[q], "anc_node_corr_knn_points": akp[q], "pos_node_corr_knn_masks": pkm[q],
                "anc_node_corr_knn_masks": akm[q], "matching_scores": ms[q],
                "pos_corr_points": rp, "anc_corr_points": sp, "corr_scores": sc, "estimated_transform": T,
                "node_matching_scores": ns[p, :m[c] + 1, :m[c + 1] + 1] if Mx == m[c] and Nx == m[c + 1] else
                torch.cat([torch.cat([ns[p, :m[c], :m[c + 1]], ns[p, :m[c], Nx:Nx + 1]], 1),
                           torch.cat([ns[p, Mx:Mx + 1, :m[c + 1]], ns[p, Mx:Mx + 1, Nx:Nx + 1]], 1)], 0),
                "pos_node_masks": parts[c][0], "anc_node_masks": parts[c + 1][0]})

    def _dense_matching(self, out, pos_f, anc_f, pos_ff, anc_ff, pos_nodes, anc_nodes, pos_fc, anc_fc):
        K = self.num_points_in_patch
        _, pos_nm, pos_knn, pos_km = F.point_to_node_partition(pos_f, pos_nodes, K)
        _, anc_nm, anc_knn, anc_km = F.point_to_node_partition(anc_f, anc_nodes, K)
        raw = F.gemm(pos_fc, anc_fc, trans_b=True)[0]
        ns = F.log_optimal_transport(raw[None], pos_nm[None], anc_nm[None], self.node_optimal_transport.alpha,
                                     scale=1.0 / pos_fc.shape[1] ** 0.5, iters=self.node_optimal_transport.num_iterations)
        nbij, node_corr_scores = F.top1_matching(ns)                   # masks are not applied at this level (superpoint_matching.py:130-162)
        pi, ai = nbij[:, 1].long(), nbij[:, 2].long()
        pk, ak = pos_knn[pi].contiguous(), anc_knn[ai].contiguous()    # (P, K) point indices of the matched patches
        pkm, akm = pos_km[pi].contiguous(), anc_km[ai].contiguous()
        pkp, akp = F.gather_rows(pos_f, pk), F.gather_rows(anc_f, ak)
        ms = F.patch_log_optimal_transport(pos_ff, pk, anc_ff, ak, pkm, akm, self.optimal_transport.alpha,
                                           scale=1.0 / pos_ff.shape[1] ** 0.5, iters=self.optimal_transport.num_iterations)
        rp, sp, sc, T = self._local_global_registration(pkp, akp, pkm, akm, ms, node_corr_scores)
        out.update({
            "pos_points_c": pos_nodes, "anc_points_c": anc_nodes, "pos_feats_c": pos_fc, "anc_feats_c": anc_fc,
            "pos_points_f": pos_f, "anc_points_f": anc_f,
            "pos_node_knn_indices": pos_knn, "pos_node_knn_masks": pos_km, "anc_node_knn_indices": anc_knn, "anc_node_knn_masks": anc_km,
            "pos_node_corr_indices": pi, "anc_node_corr_indices": ai, "node_corr_scores": node_corr_scores,
            "pos_feats_f": pos_ff, "anc_feats_f": anc_ff, "pos_node_corr_knn_points": pkp, "anc_node_corr_knn_points": akp,
            "pos_node_corr_knn_masks": pkm, "anc_node_corr_knn_masks": akm, "matching_scores": ms,
            "pos_corr_points": rp, "anc_corr_points": sp, "corr_scores": sc, "estimated_transform": T,
            "node_matching_scores": ns[0], "pos_node_masks": pos_nm, "anc_node_masks": anc_nm})


def create_model(cfg=None):
    return LCRNet(cfg)
