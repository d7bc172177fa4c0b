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


def attention_layer(sd, pfx, x, mem, heads, theta_x=None):
    """Self (rotary, rpetransformer.py:57-170) when theta_x is given, else vanilla cross (vanilla_transformer.py:30-144)."""
    a = pfx + "attention.attention."
    q, k, v = _heads(_lin(sd, a + "proj_q.", x), heads), _heads(_lin(sd, a + "proj_k.", mem), heads), _heads(_lin(sd, a + "proj_v.", mem), heads)
    if theta_x is not None:
        th = _heads(theta_x, heads)
        q, k = rotary(q, th), rotary(k, th)
    s = torch.softmax(torch.einsum("hnd,hmd->hnm", q, k) / math.sqrt(q.shape[-1]), dim=-1)
    hdn = torch.einsum("hnm,hmd->hnd", s, v).permute(1, 0, 2).reshape(x.shape[0], -1)
    hdn = _lin(sd, pfx + "attention.linear.", hdn)
    y = _ln(sd, pfx + "attention.norm.", hdn + x)
    o = pfx + "output."
    z = _lin(sd, o + "squeeze.", F.relu(_lin(sd, o + "expand.", y)))
    return _ln(sd, o + "norm.", y + z)


def thd_roformer(sd, ref_pts, src_pts, ref_feats, src_feats, pfx="transformer.", heads=4, num_layers=4):
    """thdroformer_linear.py:50-97 + RPEConditionalTransformer.forward rpetransformer.py:198-220 (sequential cross)."""
    emb = lambda p: _lin(sd, pfx + "embedding.encoder2.", _lin(sd, pfx + "embedding.encoder.", p))
    e0, e1 = emb(ref_pts), emb(src_pts)
    f0, f1 = _lin(sd, pfx + "in_proj.", ref_feats), _lin(sd, pfx + "in_proj.", src_feats)
    for i in range(2 * num_layers):
        L = f"{pfx}transformer.layers.{i}."
        if i % 2 == 0:
            f0 = attention_layer(sd, L, f0, f0, heads, e0)
            f1 = attention_layer(sd, L, f1, f1, heads, e1)
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
