"""ORACLE — test infrastructure only.

Plain PyTorch fp32 restatement of the floating-point modules on LCR-Net's per-scan hot path, written against a state
dict in the reference checkpoint layout.  Each function cites what it restates.  Pinned by
tests/test_torch_ref_golden.py against golden outputs of the imported reference model (tests/golden/make_golden_model.py).
The product never imports this file; the GPU tests compare the HIP kernels with it on the same inputs.
"""
import math

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------ KPConv blocks
def kpconv(sd, pfx, s_feats, q_points, s_points, neighbor_indices, sigma):
    """modules/kpconv/kpconv.py:79-122 (rigid KPConv, linear influence, neighbour-count normalisation)."""
    W, bias, kpts = sd[pfx + "weights"], sd.get(pfx + "bias"), sd[pfx + "kernel_points"]
    s_points = torch.cat([s_points, torch.zeros_like(s_points[:1]) + 1e6], 0)
    nb = s_points[neighbor_indices] - q_points[:, None, :]                      # (M,H,3)
    d2 = ((nb[:, :, None, :] - kpts[None, None]) ** 2).sum(-1)                  # (M,H,K)
    w = torch.clamp(1 - torch.sqrt(d2) / sigma, min=0.0).transpose(1, 2)        # (M,K,H)
    s_feats = torch.cat([s_feats, torch.zeros_like(s_feats[:1])], 0)
    nf = s_feats[neighbor_indices]                                              # (M,H,C)
    wf = torch.matmul(w, nf)                                                    # (M,K,C)
    out = torch.einsum("mkc,kco->mo", wf, W)
    num = (nf.sum(-1) > 0).sum(-1).clamp(min=1)
    out = out / num[:, None]
    if bias is not None:
        out = out + bias
    return out


def group_norm(sd, pfx, x, groups, lengths=None):
    """modules/kpconv/modules.py:33-50: nn.GroupNorm over (1,C,N) — statistics span the whole stack.  ``lengths`` (the
    build's extension) restarts the statistics per segment: one segment == what the reference would have stacked."""
    w, b = sd[pfx + "norm.weight"], sd[pfx + "norm.bias"]
    if lengths is None:
        return F.group_norm(x.t()[None], groups, w, b, 1e-5)[0].t()
    out, o = [], 0
    for n in lengths:
        n = int(n)
        out.append(F.group_norm(x[o:o + n].t()[None], groups, w, b, 1e-5)[0].t())
        o += n
    return torch.cat(out, 0)


def unary(sd, pfx, x, groups, relu, lengths=None):
    """modules.py:53-84 UnaryBlock: Linear -> GroupNorm -> LeakyReLU(0.1)."""
    x = F.linear(x, sd[pfx + "mlp.weight"], sd.get(pfx + "mlp.bias"))
    x = group_norm(sd, pfx + "norm.", x, groups, lengths)
    return F.leaky_relu(x, 0.1) if relu else x


def maxpool(x, idx):
    """kpconv/functional.py:54-67 (shadow row of zeros takes part in the max)."""
    x = torch.cat([x, torch.zeros_like(x[:1])], 0)
    return x[idx].max(1)[0]


def conv_block(sd, pfx, s_feats, q_pts, s_pts, idx, sigma, groups, q_lengths=None):
    """modules.py:104-145 ConvBlock: KPConv -> GroupNorm -> LeakyReLU(0.1)."""
    x = kpconv(sd, pfx + "KPConv.", s_feats, q_pts, s_pts, idx, sigma)
    return F.leaky_relu(group_norm(sd, pfx + "norm.", x, groups, q_lengths), 0.1)


def residual_block(sd, pfx, s_feats, q_pts, s_pts, idx, sigma, groups, strided, s_lengths=None, q_lengths=None):
    """modules.py:148-225 ResidualBlock (bottleneck)."""
    x = unary(sd, pfx + "unary1.", s_feats, groups, True, s_lengths) if (pfx + "unary1.mlp.weight") in sd else s_feats
    x = kpconv(sd, pfx + "KPConv.", x, q_pts, s_pts, idx, sigma)
    x = F.leaky_relu(group_norm(sd, pfx + "norm_conv.", x, groups, q_lengths), 0.1)
    x = unary(sd, pfx + "unary2.", x, groups, False, q_lengths)
    sc = maxpool(s_feats, idx) if strided else s_feats
    if (pfx + "unary_shortcut.mlp.weight") in sd:
        sc = unary(sd, pfx + "unary_shortcut.", sc, groups, False, q_lengths)
    return F.leaky_relu(x + sc, 0.1)


ENCODER_PLAN = [  # (name, stage of queries, stage of supports, index list, sigma multiplier, strided)  backbone4.py:15-89
    ("encoder1_1", 0, 0, "neighbors", 1, False), ("encoder1_2", 0, 0, "neighbors", 1, False),
    ("encoder2_1", 1, 0, "subsampling", 1, True), ("encoder2_2", 1, 1, "neighbors", 2, False), ("encoder2_3", 1, 1, "neighbors", 2, False),
    ("encoder3_1", 2, 1, "subsampling", 2, True), ("encoder3_2", 2, 2, "neighbors", 4, False), ("encoder3_3", 2, 2, "neighbors", 4, False),
    ("encoder4_1", 3, 2, "subsampling", 4, True), ("encoder4_2", 3, 3, "neighbors", 8, False), ("encoder4_3", 3, 3, "neighbors", 8, False),
]


def kp_encoder(sd, feats, data_dict, init_sigma=0.6, groups=32, pfx="encoder.", segment_lengths=None, trace=None):
    """backbone4.py:60-89 KPEncoder.forward.  segment_lengths: list (per stage) of per-segment lengths or None (= the
    reference behaviour: one GroupNorm segment = the whole stack)."""
    pts = data_dict["points"]
    outs = {}
    x = feats
    for name, qs, ss, kind, mult, strided in ENCODER_PLAN:
        idx = data_dict[kind][ss if kind == "subsampling" else qs]
        ql = None if segment_lengths is None else segment_lengths[qs]
        sl = None if segment_lengths is None else segment_lengths[ss]
        if name == "encoder1_1":
            x = conv_block(sd, pfx + name + ".", x, pts[qs], pts[ss], idx, init_sigma * mult, groups, ql)
        else:
            x = residual_block(sd, pfx + name + ".", x, pts[qs], pts[ss], idx, init_sigma * mult, groups, strided, sl, ql)
        if trace is not None:
            trace[name] = x
        outs[qs] = x
    return [outs[0], outs[1], outs[2], outs[3]]


# ------------------------------------------------------------------------------------------------ NetVLAD head
def _bn_eval(sd, pfx, x):
    return (x - sd[pfx + "running_mean"]) / torch.sqrt(sd[pfx + "running_var"] + 1e-5) * sd[pfx + "weight"] + sd[pfx + "bias"]


def netvlad(sd, x, pfx="netvlad."):
    """modules/netvlad/NetVlad.py:49-87 (NetVLADLoupe2, eval) + GatingContext :165-201.  x: (B, N, 1024) -> (B, 256)."""
    B, N, C = x.shape
    act = torch.matmul(x, sd[pfx + "cluster_weights"])                              # (B,N,64)
    act = _bn_eval(sd, pfx + "bn1.", act.reshape(-1, act.shape[-1])).reshape(B, N, -1)
    act = torch.softmax(act, dim=-1)
    a = act.sum(-2, keepdim=True) * sd[pfx + "cluster_weights2"]                    # (B,1024,64)
    vlad = torch.matmul(act.transpose(2, 1), x).transpose(2, 1) - a                 # (B,1024,64)
    vlad = F.normalize(vlad, dim=1, p=2, eps=1e-6)
    vlad = F.normalize(vlad.reshape(B, -1), dim=1, p=2, eps=1e-6)
    vlad = torch.matmul(vlad, sd[pfx + "hidden1_weights"])
    vlad = _bn_eval(sd, pfx + "bn2.", vlad)
    gates = torch.sigmoid(_bn_eval(sd, pfx + "context_gating.bn1.", torch.matmul(vlad, sd[pfx + "context_gating.gating_weights"])))
    return vlad * gates


def global_descriptor(sd, feats_c):
    """LCRNet_GlobalDescrition.GlobalDescritionHEAD eval branch (model_family/LCRNet_GlobalDescrition.py:34-38):
    feats_c (N4, 1024) of ONE scan -> (1, 256) unit-norm."""
    x = F.normalize(feats_c[None], dim=2)
    return F.normalize(netvlad(sd, x), dim=1)


# ------------------------------------------------------------------------------------------------ 3D-RoFormer
def _ln(sd, pfx, x):
    return F.layer_norm(x, (x.shape[-1],), sd[pfx + "weight"], sd[pfx + "bias"], 1e-5)


def _lin(sd, pfx, x):
    return F.linear(x, sd[pfx + "weight"], sd[pfx + "bias"])


def rotary(x, theta):
    """rpetransformer.py:41-54: x (H,N,D); theta (H,N,D/2) duplicated onto adjacent channel pairs."""
    H, N, D = x.shape
    xp = x.reshape(H, N, D // 2, 2)
    rot = torch.stack([-xp[..., 1], xp[..., 0]], -1).reshape(H, N, D)
    th = theta.repeat_interleave(2, dim=-1)
    return x * torch.cos(th) + rot * torch.sin(th)


def _heads(x, h):
    return x.reshape(x.shape[0], h, -1).permute(1, 0, 2)       # (N, h*c) -> (h, N, c)


def topk_attention(q, k, v, kk):
    """dynamic_attention with k != None (rpetransformer.py:19-39) on (H,N,D) / (H,M,D) heads: the kk largest scores of every row are
    soft-maxed, everything else is zero.  Ties at the threshold: lowest key index first (torch.topk leaves it unspecified)."""
    sc = torch.einsum("hnd,hmd->hnm", q, k) / math.sqrt(q.shape[-1])
    if kk <= 0:
        return torch.zeros_like(q)
    if kk >= sc.shape[-1]:
        return torch.einsum("hnm,hmd->hnd", torch.softmax(sc, dim=-1), v)
    order = torch.sort(sc, dim=-1, descending=True, stable=True).indices[..., :kk]       # stable: equal scores keep index order
    vals = torch.gather(sc, -1, order)
    prob = torch.zeros_like(sc).scatter_(-1, order, torch.softmax(vals, dim=-1))
    return torch.einsum("hnm,hmd->hnd", prob, v)


def attention_layer(sd, pfx, x, mem, heads, theta_x=None, topk_frac=None):
    """Self (rotary, rpetransformer.py:57-170) when theta_x is given, else vanilla cross (vanilla_transformer.py:30-144).
    topk_frac: the self layer's fraction of cfg.GAT.k (kk = int(n_queries * fraction), rpetransformer.py:27)."""
    a = pfx + "attention.attention."
    q, k, v = _heads(_lin(sd, a + "proj_q.", x), heads), _heads(_lin(sd, a + "proj_k.", mem), heads), _heads(_lin(sd, a + "proj_v.", mem), heads)
    if theta_x is not None:
        th = _heads(theta_x, heads)
        q, k = rotary(q, th), rotary(k, th)
    if topk_frac is not None:
        hdn = topk_attention(q, k, v, int(x.shape[0] * topk_frac)).permute(1, 0, 2).reshape(x.shape[0], -1)
    else:
        s = torch.softmax(torch.einsum("hnd,hmd->hnm", q, k) / math.sqrt(q.shape[-1]), dim=-1)
        hdn = torch.einsum("hnm,hmd->hnd", s, v).permute(1, 0, 2).reshape(x.shape[0], -1)
    hdn = _lin(sd, pfx + "attention.linear.", hdn)
    y = _ln(sd, pfx + "attention.norm.", hdn + x)
    o = pfx + "output."
    z = _lin(sd, o + "squeeze.", F.relu(_lin(sd, o + "expand.", y)))
    return _ln(sd, o + "norm.", y + z)


def thd_roformer(sd, ref_pts, src_pts, ref_feats, src_feats, pfx="transformer.", heads=4, num_layers=4, k=None):
    """thdroformer_linear.py:50-97 + RPEConditionalTransformer.forward rpetransformer.py:198-220 (sequential cross)."""
    emb = lambda p: _lin(sd, pfx + "embedding.encoder2.", _lin(sd, pfx + "embedding.encoder.", p))
    e0, e1 = emb(ref_pts), emb(src_pts)
    f0, f1 = _lin(sd, pfx + "in_proj.", ref_feats), _lin(sd, pfx + "in_proj.", src_feats)
    for i in range(2 * num_layers):
        L = f"{pfx}transformer.layers.{i}."
        if i % 2 == 0:
            fr = None if k is None else k[i // 2]
            f0 = attention_layer(sd, L, f0, f0, heads, e0, fr)
            f1 = attention_layer(sd, L, f1, f1, heads, e1, fr)
        else:
            f0 = attention_layer(sd, L, f0, f1, heads)
            f1 = attention_layer(sd, L, f1, f0, heads)      # attends to the UPDATED f0 (parallel=False)
    return _lin(sd, pfx + "out_proj.", f0), _lin(sd, pfx + "out_proj.", f1)


# ------------------------------------------------------------------------------------------------ retrieval
def retrieval_topk(desc, k=50, exclude=100, start=101, stop=None):
    """experiments/loop_detection/eval_loop_detection_overlap_dataset.py:183-214: for query i in [start, C-1) the database
    is descriptors [0, i-exclude); top-k by squared L2, ascending (ties: ascending index).  faiss IndexIVFFlat with
    nlist=1 is exhaustive, so this is a plain masked exhaustive search (fp64 distances: an exact reference).
    Returns (query ids (Q,), idx (Q,k) with -1 padding, d2 (Q,k) with +inf padding)."""
    C = desc.shape[0]
    stop = C - 1 if stop is None else stop
    qs = torch.arange(start, stop)
    d = desc.double()
    d2 = ((d[qs][:, None, :] - d[None]) ** 2).sum(-1) if C * len(qs) <= 4_000_000 else \
        (d[qs] ** 2).sum(1, keepdim=True) - 2 * d[qs] @ d.t() + (d ** 2).sum(1)[None]
    d2 = d2.clamp(min=0)
    mask = torch.arange(C)[None, :] >= (qs[:, None] - exclude)
    d2 = d2.masked_fill(mask, float("inf"))
    order = torch.argsort(d2, dim=1, stable=True)[:, :k]          # stable: ties keep ascending index
    val = torch.gather(d2, 1, order)
    idx = order.masked_fill(torch.isinf(val), -1)
    if idx.shape[1] < k:
        pad = k - idx.shape[1]
        idx = torch.cat([idx, torch.full((len(qs), pad), -1, dtype=idx.dtype)], 1)
        val = torch.cat([val, torch.full((len(qs), pad), float("inf"), dtype=val.dtype)], 1)
    return qs, idx, val


def recall_at_1(qs, idx, gt):
    """compute_topN with N=1 (eval_loop_detection_overlap_dataset.py:29-62): fraction of GT-bearing query frames whose first
    retrieved frame is a ground-truth loop.  gt: dict frame -> set of frames."""
    hit = tot = 0
    for q, row in zip(qs.tolist(), idx.tolist()):
        g = gt.get(q)
        if not g:
            continue
        tot += 1
        hit += int(row[0] in g)
    return hit / max(tot, 1)


# ------------------------------------------------------------------------------------------------ pose tail (a-10)
def vote_layer(sd, pfx, xyz, feats, max_range=4.2):
    """modules/vote/vote.py:112-182 (output_feats=False): MLP(Linear-LayerNorm-ReLU x2) -> ctr_reg -> offsets clamped to
    `max_range` -> shifted points."""
    x = feats
    for i in (0, 3):
        x = F.linear(x, sd[f"{pfx}mlp_modules.{i}.weight"], sd[f"{pfx}mlp_modules.{i}.bias"])
        x = F.relu(F.layer_norm(x, (x.shape[-1],), sd[f"{pfx}mlp_modules.{i + 1}.weight"], sd[f"{pfx}mlp_modules.{i + 1}.bias"], 1e-5))
    off = F.linear(x, sd[pfx + "ctr_reg.weight"], sd[pfx + "ctr_reg.bias"])
    dis = torch.norm(off, p=2, dim=1)
    alpha = torch.where(dis > max_range, max_range / dis, torch.ones_like(dis))
    return xyz + off * alpha[:, None]


def greedy_nms(nodes, lengths, radius):
    """modules/vote/vote.py:13-70 (NMS.forward without scores): per cloud, node 0 is kept; node i is kept iff its
    nn.PairwiseDistance (||a - b + 1e-6||_2) to EVERY kept node exceeds `radius`.  Order dependent by definition."""
    masks, lens, o = [], [], 0
    for n in lengths:
        n = int(n)
        pts = nodes[o:o + n]
        keep = torch.zeros(n, dtype=torch.bool)
        if n:
            keep[0] = True
        for i in range(1, n):
            d = torch.sqrt(((pts[i][None] - pts[keep] + 1e-6) ** 2).sum(1))
            if bool((d > radius).all()):
                keep[i] = True
        masks.append(keep)
        lens.append(int(keep.sum()))
        o += n
    return torch.cat(masks), torch.tensor(lens, dtype=torch.int64)


def _radius_search(q, s, ql, sl, radius, limit):
    from oracle import ops
    return torch.from_numpy(ops.radius_search(q.numpy(), s.numpy(), ql.numpy(), sl.numpy(), radius, limit))


def vote_encoder(sd, feats, data_dict, limits, init_radius=1.275, init_sigma=0.6, groups=32, pfx="vote_encoder.", nms_radius=2.4):
    """backbone4.py:121-220 Vote_Encoder.forward on a pair stack (GroupNorm over the stack, like the reference)."""
    lens_c = data_dict["lengths"][-1]
    n0, n1 = int(lens_c[0]), int(lens_c[1])
    points_c = data_dict["points"][-1]
    shifted = vote_layer(sd, pfx + "vote.", points_c, feats)
    mask, length = greedy_nms(shifted, lens_c, nms_radius)
    nms_pts = shifted[mask]
    knn = _radius_search(nms_pts, shifted, length, lens_c, 2.4, limits[-1])          # pad = n0 + n1
    pad = n0 + n1
    valid = knn != pad
    sp = torch.cat([shifted, torch.zeros_like(shifted[:1])], 0)
    centers = sp[knn].sum(1) / valid.sum(-1, keepdim=True)                          # mean of the in-radius voted points
    sub = _radius_search(centers, points_c, length, lens_c, init_radius * 8, limits[-2])
    nb = _radius_search(centers, centers, length, length, init_radius * 16, limits[-1])
    f = residual_block(sd, pfx + "encoder6_1.", feats, centers, points_c, sub, init_sigma * 8, groups, True)
    f = residual_block(sd, pfx + "encoder6_2.", f, centers, centers, nb, init_sigma * 16, groups, False)
    f = residual_block(sd, pfx + "encoder6_3.", f, centers, centers, nb, init_sigma * 16, groups, False)
    return {"shifted": shifted, "mask": mask, "length": length, "centers": centers, "feats_c": f,
            "subsampling": sub, "neighbors": nb}


def point_to_node_partition(points, nodes, point_limit):
    """modules/ops/pointcloud_partition.py:60-107: nearest node per point (x^2 - 2xy + y^2 distances, clamp 1e-12), then per
    node the `point_limit` nearest of its own points (ascending), padded with len(points)."""
    d = (nodes ** 2).sum(1)[:, None] - 2 * nodes @ points.t() + (points ** 2).sum(1)[None]
    d = d.clamp(min=1e-12)
    p2n = d.argmin(0)
    node_masks = torch.zeros(nodes.shape[0], dtype=torch.bool)
    node_masks[p2n] = True
    own = torch.zeros_like(d, dtype=torch.bool)
    own[p2n, torch.arange(points.shape[0])] = True
    d = d.masked_fill(~own, 1e12)
    knn = d.topk(point_limit, dim=1, largest=False)[1]
    knn_masks = p2n[knn] == torch.arange(nodes.shape[0])[:, None]
    knn = knn.masked_fill(~knn_masks, points.shape[0])
    return p2n, node_masks, knn, knn_masks


def log_optimal_transport(scores, row_masks, col_masks, alpha, iters=100, inf=1e12):
    """modules/sinkhorn/learnable_sinkhorn.py:20-66 (SuperGlue-style log-domain Sinkhorn with a learnable dustbin score)."""
    B, M, N = scores.shape
    prm = torch.zeros(B, M + 1, dtype=torch.bool)
    prm[:, :M] = ~row_masks
    pcm = torch.zeros(B, N + 1, dtype=torch.bool)
    pcm[:, :N] = ~col_masks
    S = torch.cat([torch.cat([scores, alpha.expand(B, M, 1)], -1), alpha.expand(B, 1, N + 1)], 1)
    S = S.masked_fill(prm[:, :, None] | pcm[:, None, :], -inf)
    nr, nc = row_masks.to(scores.dtype).sum(1), col_masks.to(scores.dtype).sum(1)      # fp64 scores -> an fp64 run (tests)
    norm = -torch.log(nr + nc)
    log_mu = norm[:, None].expand(B, M + 1).clone()
    log_mu[:, M] = torch.log(nc) + norm
    log_mu[prm] = -inf
    log_nu = norm[:, None].expand(B, N + 1).clone()
    log_nu[:, N] = torch.log(nr) + norm
    log_nu[pcm] = -inf
    u, v = torch.zeros_like(log_mu), torch.zeros_like(log_nu)
    for _ in range(iters):
        u = log_mu - torch.logsumexp(S + v[:, None, :], dim=2)
        v = log_nu - torch.logsumexp(S + u[:, :, None], dim=1)
    return S + u[:, :, None] + v[:, None, :] - norm[:, None, None]


def superpoint_matching_ot(log_scores):
    """geotransformer/superpoint_matching.py:54-187 with num_correspondences=None: a pair (i,j) is kept if it is the column
    maximum and beats the dustbin row, OR the row maximum and beats the dustbin column; row-major order."""
    P = torch.exp(log_scores)
    col_top = torch.zeros_like(P)
    ci = P.argmax(0)
    col_top[ci, torch.arange(P.shape[1])] = P.max(0)[0]
    src = col_top > P[-1, :][None]
    row_top = torch.zeros_like(P)
    ri = P.argmax(1)
    row_top[torch.arange(P.shape[0]), ri] = P.max(1)[0]
    ref = row_top > P[:, -1][:, None]
    corr = (ref | src)[:-1, :-1].nonzero()
    return corr[:, 0], corr[:, 1], P[corr[:, 0], corr[:, 1]]


def kp_decoder(sd, feats_list, data_dict, groups=32, pfx="kpdecoder."):
    """backbone4.py:344-373: nearest-upsample (column 0) + concat + UnaryBlock, three times; returns the finest features."""
    def up(x, idx):
        x = torch.cat([x, torch.zeros_like(x[:1])], 0)
        return x[idx[:, 0]]
    f1, f2, f3, f4 = feats_list
    U = data_dict["upsampling"]
    l3 = unary(sd, pfx + "decoder3.", torch.cat([up(f4, U[2]), f3], 1), groups, True)
    l2 = unary(sd, pfx + "decoder2.", torch.cat([up(l3, U[1]), f2], 1), groups, True)
    return F.linear(torch.cat([up(l2, U[0]), f1], 1), sd[pfx + "decoder1.mlp.weight"], sd[pfx + "decoder1.mlp.bias"])


def weighted_procrustes(src, ref, w, eps=1e-5):
    """modules/registration/procrustes.py:6-73 (batched): R = V diag(1,1,sign det(V U^T)) U^T of H = sum w (s - s̄)(r - r̄)^T."""
    if src.dim() == 2:
        return weighted_procrustes(src[None], ref[None], w[None], eps)[0]
    w = torch.where(w < 0.0, torch.zeros_like(w), w)
    w = (w / (w.sum(1, keepdim=True) + eps))[:, :, None]
    sc, rc = (src * w).sum(1, keepdim=True), (ref * w).sum(1, keepdim=True)
    H = (src - sc).transpose(1, 2) @ (w * (ref - rc))
    U, _, V = torch.svd(H)
    Ut = U.transpose(1, 2)
    eye = torch.eye(3).repeat(src.shape[0], 1, 1)
    eye[:, -1, -1] = torch.sign(torch.det(V @ Ut))
    R = V @ eye @ Ut
    t = (rc.transpose(1, 2) - R @ sc.transpose(1, 2))[:, :, 0]
    T = torch.eye(4).repeat(src.shape[0], 1, 1)
    T[:, :3, :3], T[:, :3, 3] = R, t
    return T


def _apply(T, p):
    return p @ T[..., :3, :3].transpose(-1, -2) + T[..., None, :3, 3]


def local_global_registration(ref_knn_pts, src_knn_pts, ref_masks, src_masks, log_scores, acceptance_radius=0.45, threshold=3, steps=5, mutual=False,
                              topk=1, use_dustbin=True, confidence_threshold=0.0, global_scores=None, correspondence_limit=None):
    """geotransformer/local_global_registration.py:204-246 with every switch: mutual (:84-87), k (:56-82), use_dustbin (:62-65, :74-77, :86-87 —
    False: the caller's (K+1) x (K+1) transport output loses its dustbin row / column first, as model_family/LCRNet.py:256-257 does),
    global_scores (use_global_score, :236-237) and correspondence_limit (:152-160).  Equal values are taken in index order (a stable
    descending sort; torch.topk leaves it open)."""
    S = torch.exp(log_scores)
    if not use_dustbin:
        S = S[:, :-1, :-1]
    B = S.shape[0]
    ri = torch.sort(S, dim=2, descending=True, stable=True).indices[:, :, :topk]
    rtop = torch.zeros_like(S).scatter_(2, ri, torch.gather(S, 2, ri))
    si = torch.sort(S, dim=1, descending=True, stable=True).indices[:, :topk, :]
    stop = torch.zeros_like(S).scatter_(1, si, torch.gather(S, 1, si))
    if use_dustbin:
        ref_c, src_c = rtop > S[:, :, -1][:, :, None], stop > S[:, -1, :][:, None, :]
    else:
        ref_c, src_c = rtop > confidence_threshold, stop > confidence_threshold
    corr = (ref_c & src_c) if mutual else (ref_c | src_c)
    if use_dustbin:
        corr, S = corr[:, :-1, :-1], S[:, :-1, :-1]
    corr = corr & (ref_masks[:, :, None] & src_masks[:, None, :])
    if global_scores is not None:
        S = S * global_scores.view(-1, 1, 1)
    S = S * corr.float()
    b, i, j = corr.nonzero(as_tuple=True)
    rp, sp, sc = ref_knn_pts[b, i], src_knn_pts[b, j], S[b, i, j]
    # verification set (:152-160): the `limit` highest scores (stable: ties in row order); hypotheses still come from ALL correspondences
    vr, vs, vc = rp, sp, sc
    if correspondence_limit is not None and sc.shape[0] > correspondence_limit:
        sel = torch.sort(sc, descending=True, stable=True).indices[:correspondence_limit]
        vr, vs, vc = rp[sel], sp[sel], sc[sel]
    # per-patch hypotheses (chunks of >= threshold correspondences), best by inlier count over the verification set
    counts = torch.bincount(b, minlength=B)
    Ts = []
    for p in torch.nonzero(counts >= threshold)[:, 0].tolist():
        m = b == p
        Ts.append(weighted_procrustes(sp[m], rp[m], sc[m]))
    if Ts:
        Ts = torch.stack(Ts)
        res = torch.linalg.norm(vr[None] - _apply(Ts, vs[None]), dim=2)
        inl = res < acceptance_radius
        cur = vc * inl[inl.sum(1).argmax()].float()
    else:
        T = weighted_procrustes(vs, vr, vc)
        cur = vc * (torch.linalg.norm(vr - _apply(T, vs), dim=1) < acceptance_radius).float()
    T = weighted_procrustes(vs, vr, cur)
    for _ in range(steps - 1):
        cur = vc * (torch.linalg.norm(vr - _apply(T, vs), dim=1) < acceptance_radius).float()
        T = weighted_procrustes(vs, vr, cur)
    return rp, sp, sc, T


def pose_tail(sd, data_dict, feats_list, enhanced_feats_c, limits, num_points_in_patch=128, iters=100):
    """LCRNet.forward after the transformer (LCRNet.py:152-272): vote encoder -> partitions -> node Sinkhorn + matching ->
    decoder -> patch Sinkhorn -> local-to-global registration."""
    vd = vote_encoder(sd, enhanced_feats_c, data_dict, limits)
    L0 = data_dict["lengths"][0]
    nf0, nf1 = int(L0[0]), int(L0[1])
    pts_f = data_dict["points"][0]
    pos_f, anc_f = pts_f[:nf0], pts_f[nf0:nf0 + nf1]
    m0 = int(vd["length"][0])
    pos_c, anc_c = vd["centers"][:m0], vd["centers"][m0:]
    pos_fc, anc_fc = vd["feats_c"][:m0], vd["feats_c"][m0:]
    _, pos_nm, pos_knn, pos_km = point_to_node_partition(pos_f, pos_c, num_points_in_patch)
    _, anc_nm, anc_knn, anc_km = point_to_node_partition(anc_f, anc_c, num_points_in_patch)
    ns = (pos_fc @ anc_fc.t() / pos_fc.shape[1] ** 0.5)[None]
    ns = log_optimal_transport(ns, pos_nm[None], anc_nm[None], sd["node_optimal_transport.alpha"], iters)[0]
    pi, ai, node_scores = superpoint_matching_ot(ns)
    fl = list(feats_list)
    fl[-1] = enhanced_feats_c
    feats_f = kp_decoder(sd, fl, data_dict)
    pos_ff, anc_ff = feats_f[:nf0], feats_f[nf0:]
    pad = lambda x: torch.cat([x, torch.zeros_like(x[:1])], 0)
    pk, ak = pos_knn[pi], anc_knn[ai]
    pkp, akp = pad(pos_f)[pk], pad(anc_f)[ak]
    pkf, akf = pad(pos_ff)[pk], pad(anc_ff)[ak]
    ms = torch.einsum("bnd,bmd->bnm", pkf, akf) / feats_f.shape[1] ** 0.5
    ms = log_optimal_transport(ms, pos_km[pi], anc_km[ai], sd["optimal_transport.alpha"], iters)
    rp, sp, sc, T = local_global_registration(pkp, akp, pos_km[pi], anc_km[ai], ms)
    return {"vote": vd, "pos_node_knn_indices": pos_knn, "anc_node_knn_indices": anc_knn, "pos_node_corr_indices": pi,
            "anc_node_corr_indices": ai, "node_scores": ns, "feats_f": feats_f, "matching_scores": ms,
            "pos_corr_points": rp, "anc_corr_points": sp, "corr_scores": sc, "estimated_transform": T}
