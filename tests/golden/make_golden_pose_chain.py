"""Stage-by-stage golden vectors of the pose tail (a-10) from the IMPORTED reference `LCRNet` (build container only).

    python tests/golden/make_golden_pose_chain.py

The whole-pair golden (make_golden_pose.py) can only be compared loosely: with seeded RANDOM weights the matching scores are
near-uniform and the pose is a consensus over ~4 k arbitrary matches, so any 1e-6 difference upstream moves it.  This script pins
every stage on the reference's OWN intermediates instead (forward hooks on the unmodified reference modules), so that each HIP stage
can be required to meet north_star's 1e-4 on identical inputs:

  A. `vote_encoder` (backbone4.py:121-220)      input  = the reference's transformer output `enhanced_feats_c` (1667 x 256)
                                                  output = shifted points, NMS node counts, node centres, node features
  B. `fine_matching` = LocalGlobalRegistration   input  = the reference's patch points / masks / log matching scores of 40 of its
     (local_global_registration.py:204-246)               639 patch correspondences (a subset keeps the fixture at ~2.6 MB)
                                                  output = correspondences, scores, per-hypothesis inlier counts, best hypothesis,
                                                           estimated transform — the module re-run on that subset, plus the
                                                           intermediate quantities recomputed with the module's own methods
  D. the matching link in between (the part of a-10 that was only checked against the torch restatement before):
     `point_to_node_partition` (ops/pointcloud_partition.py:60-107)   input  = stage-0 points of each cloud + the reference's node centres
                                                  output = node masks, (M, 128) knn indices + masks of both clouds
     `node_optimal_transport` (sinkhorn/learnable_sinkhorn.py:20-66)  input  = the reference's scaled node score matrix + node masks
                                                  output = (M+1, N+1) log matching scores
     `coarse_matching` = SuperPointMatching_OT (geotransformer/superpoint_matching.py:91-187)
                                                  output = node correspondences + scores
     `optimal_transport` (patch level)            input  = the reference's scaled (128 x 128) patch score matrices + masks of
                                                           D_N_PATCH of its patch correspondences;  output = their (129 x 129) log scores
     Function `point_to_node_partition` is captured by wrapping the name the reference model module imported; the three modules
     by forward hooks.
  C. the same module on a seeded WELL-CONDITIONED synthetic case (`synthetic_lgr_case` below: 24 patches, a known rigid motion,
     2 cm noise, peaked scores, 6 outlier patches with a different motion) where the pose is determined by the data — inputs are
     regenerated from the seed by the tests, only the outputs are stored.

Same stubs / seeded weights / `.cuda()` patching as make_golden_pose.py.  Output: tests/golden/pose_chain_golden.npz.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_model as mgm  # noqa: E402

N_SUBSET = 40
D_N_PATCH = 20


def synthetic_lgr_case(seed=0, P=24, K=128, n_out=6):
    """-> ref_knn_points (P,K,3), src_knn_points (P,K,3), ref_masks (P,K), src_masks (P,K), log score matrices (P,K+1,K+1) f32, the
    true transform (4,4).  ref = R src + t for the inlier patches; the last `n_out` patches follow another motion.  Per patch 40-100
    valid points per side, matched through a random permutation; the score of a true match is 0.5-0.9, everything else ~1e-4, the
    dustbin row / column 0.05 (so the dustbin test `score > dustbin` keeps exactly the true matches and a few score-noise ones)."""
    rng = np.random.default_rng(seed)

    def rigid(axis, deg, t):
        a = np.asarray(axis, dtype=np.float64)
        a /= np.linalg.norm(a)
        th = np.deg2rad(deg)
        Kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        T = np.eye(4)
        T[:3, :3] = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        T[:3, 3] = t
        return T

    T_true = rigid([0.1, -0.2, 1.0], 23.0, [3.2, -1.1, 0.4])
    T_bad = rigid([1.0, 0.3, 0.2], 71.0, [-6.0, 9.0, 2.0])
    ref = np.zeros((P, K, 3), np.float32)
    src = np.zeros((P, K, 3), np.float32)
    rm = np.zeros((P, K), bool)
    sm = np.zeros((P, K), bool)
    logs = np.full((P, K + 1, K + 1), np.log(1e-4), np.float32)
    logs[:, -1, :] = np.log(0.05)
    logs[:, :, -1] = np.log(0.05)
    for p in range(P):
        n = int(rng.integers(40, 100))
        centre = rng.uniform(-30, 30, 3) * [1, 1, 0.1]
        s = (centre + rng.standard_normal((n, 3)) * 1.5).astype(np.float32)
        T = T_bad if p >= P - n_out else T_true
        r = (s.astype(np.float64) @ T[:3, :3].T + T[:3, 3] + rng.standard_normal((n, 3)) * 0.02).astype(np.float32)
        perm = rng.permutation(n)
        n_s = n + int(rng.integers(0, 20))                              # the source patch also holds unmatched points
        extra = (centre + rng.standard_normal((n_s - n, 3)) * 1.5).astype(np.float32)
        ref[p, :n], rm[p, :n] = r, True
        src_pts = np.concatenate([s[perm], extra])
        src[p, :n_s], sm[p, :n_s] = src_pts, True
        inv = np.argsort(perm)                                           # ref i <-> src inv[i]
        drop = rng.random(n) < 0.15                                      # 15 % of the true matches lose to the dustbin
        val = rng.uniform(0.5, 0.9, n).astype(np.float32)
        val[drop] = 0.01
        logs[p, np.arange(n), inv] = np.log(val)
        wrong = rng.integers(0, n, 3)                                    # a few confident WRONG matches per patch
        logs[p, wrong, (inv[wrong] + 7) % n_s] = np.log(0.95)
    return ref, src, rm, sm, logs, T_true.astype(np.float32)


def hypothesis_stats(lgr, ref_knn, src_knn, rmask, smask, logs):
    """Intermediates of LocalGlobalRegistration.forward / local_to_global_registration recomputed with the module's OWN methods:
    dense correspondences, chunk table, per-hypothesis inlier counts over all correspondences, index of the best hypothesis."""
    from experiments.lcrnet.modules.ops import apply_transform
    score = torch.exp(logs)
    corr = lgr.compute_correspondence_matrix(score, rmask, smask)
    score = score[:, :-1, :-1] * corr.float()
    b, i, j = torch.nonzero(corr, as_tuple=True)
    rp, sp, sc = ref_knn[b, i], src_knn[b, j], score[b, i, j]
    edges = [0] + (torch.nonzero(b[1:] != b[:-1], as_tuple=True)[0] + 1).tolist() + [b.shape[0]]
    chunks = [(x, y) for x, y in zip(edges[:-1], edges[1:]) if y - x >= lgr.correspondence_threshold]
    br, bs, bw = lgr.convert_to_batch(rp, sp, sc, chunks)
    hyp = lgr.procrustes(bs, br, bw)
    res = torch.linalg.norm(rp.unsqueeze(0) - apply_transform(sp.unsqueeze(0), hyp), dim=2)
    counts = torch.lt(res, lgr.acceptance_radius).sum(1)
    return b, i, j, np.array(chunks, dtype=np.int64), hyp, counts, int(counts.argmax())


def stability(lgr, ref_knn, src_knn, rmask, smask, logs, hyp0, T0, chunks, sp0, seed=1, rel=1e-7, trials=4):
    """Which outputs are determined by the data to better than 1e-4?  Re-run the reference module with the inputs jittered by 1e-7
    (relative, seeded — the size of an fp32 rounding).  Per patch hypothesis: 'matrix stable' when its entries move by < 1e-5 and
    'fit stable' when the images of the patch's OWN source points move by < 1e-5 m (points jittered, scores kept, so the
    correspondences stay the same); the final transform is 'stable' when it moves by < 1e-5 with points AND scores jittered.  With
    random weights many patch hypotheses come from 3-6 near-degenerate correspondences (a rotation about a line is free, or the
    reflection fix of the SVD picks another axis) — those are reported, not pinned."""
    g = torch.Generator().manual_seed(seed)
    n = hyp0.shape[0]
    hyp_move, fit_move, T_move = torch.zeros(n), torch.zeros(n), 0.0

    def images(h):
        return [sp0[x:y].double() @ h[c, :3, :3].double().T + h[c, :3, 3].double() for c, (x, y) in enumerate(chunks)]

    img0 = images(hyp0)
    for _ in range(trials):
        jr = ref_knn * (1 + rel * torch.randn(ref_knn.shape, generator=g))
        js = src_knn * (1 + rel * torch.randn(src_knn.shape, generator=g))
        hyp = hypothesis_stats(lgr, jr, js, rmask, smask, logs)[4]
        assert hyp.shape == hyp0.shape
        hyp_move = torch.maximum(hyp_move, (hyp - hyp0).abs().flatten(1).max(1).values)
        fit_move = torch.maximum(fit_move, torch.tensor([float((a - b).abs().max()) for a, b in zip(images(hyp), img0)]))
        jl = logs + rel * torch.randn(logs.shape, generator=g)
        T = lgr(jr, js, rmask, smask, jl, torch.ones(len(jr)))[3]
        T_move = max(T_move, float((T - T0).abs().max()))
    return (hyp_move < 1e-5).numpy(), (fit_move < 1e-5).numpy(), T_move < 1e-5


def main():
    mgm.install_stubs()
    sys.path.insert(0, mgm.REF)
    mgm.install_ref_ext()
    torch.Tensor.cuda = lambda self, *a, **k: self.contiguous()
    torch.nn.Module.cuda = lambda self, *a, **k: self
    from lcrnet_amd.weights import seeded_state_dict
    from experiments.lcrnet.config_model import make_cfg
    from experiments.lcrnet.data import precompute_data_stack_mode
    from experiments.lcrnet.model_family.LCRNet import LCRNet

    cfg = make_cfg()
    cfg.neighbor_limits = mgm.LIMITS
    cfg.vis = False
    full = LCRNet(cfg).eval()
    full.load_state_dict(seeded_state_dict(full.state_dict(), mgm.SEED), strict=True)
    a = np.load(os.path.join(HERE, "scans", "003854.npy"))
    b = np.load(os.path.join(HERE, "scans", "000958.npy"))
    pts = torch.from_numpy(np.concatenate([a, b]))
    dd = precompute_data_stack_mode(pts, torch.LongTensor([len(a), len(b)]), 4, 0.3, 1.275, mgm.LIMITS)
    dd = {k: [t.contiguous() for t in v] for k, v in dd.items()}
    dd["features"] = torch.ones(len(pts), 1)
    dd["batch_size"] = 1

    cap = {}
    h1 = full.vote_encoder.register_forward_hook(lambda m, i, o: cap.__setitem__("vote", (i[0].clone(), {k: v for k, v in o.items()})))
    h2 = full.fine_matching.register_forward_hook(lambda m, i, o: cap.__setitem__("lgr", ([t.clone() for t in i], o)))
    h3 = full.node_optimal_transport.register_forward_hook(lambda m, i, o: cap.__setitem__("node_ot", ([t.clone() for t in i], o.clone())))
    h4 = full.optimal_transport.register_forward_hook(lambda m, i, o: cap.__setitem__("patch_ot", ([t.clone() for t in i], o.clone())))
    h5 = full.coarse_matching.register_forward_hook(lambda m, i, o: cap.__setitem__("coarse", ([t.clone() for t in i], [t.clone() for t in o])))
    import experiments.lcrnet.model_family.LCRNet as ref_mod
    ref_partition = ref_mod.point_to_node_partition
    cap["partition"] = []

    def partition_spy(points, nodes, limit, *a, **k):
        out = ref_partition(points, nodes, limit, *a, **k)
        cap["partition"].append((points.clone(), nodes.clone(), limit, [t.clone() for t in out]))
        return out

    ref_mod.point_to_node_partition = partition_spy
    with torch.no_grad():
        out = full(dd)
    for h in (h1, h2, h3, h4, h5):
        h.remove()
    ref_mod.point_to_node_partition = ref_partition
    store = {}
    # ---- A: vote encoder on the reference's enhanced features
    enhanced, vo = cap["vote"]
    store["A_enhanced_feats_c"] = enhanced.numpy()
    for k, v in vo.items():
        if torch.is_tensor(v):
            print("vote out:", k, tuple(v.shape), v.dtype)
    store["A_shifted_pos_points_c"] = vo["shifted_pos_points_c"].numpy()
    store["A_shifted_anc_points_c"] = vo["shifted_anc_points_c"].numpy()
    store["A_length"] = vo["length"].numpy().astype(np.int64)
    store["A_pos_points_c"], store["A_anc_points_c"] = vo["pos_points_c"].numpy(), vo["anc_points_c"].numpy()
    store["A_pos_feats_c"], store["A_anc_feats_c"] = vo["pos_feats_c"].numpy(), vo["anc_feats_c"].numpy()
    # ---- B: LocalGlobalRegistration on a subset of the reference's own patch correspondences
    (pkp, akp, pkm, akm, ms, ncs), (rp_all, sp_all, sc_all, T_all) = cap["lgr"]
    P = pkp.shape[0]
    S = torch.from_numpy(np.linspace(0, P - 1, N_SUBSET).round().astype(np.int64))
    lgr = full.fine_matching
    print("LGR settings: k", lgr.k, "mutual", lgr.mutual, "dustbin", lgr.use_dustbin, "global", lgr.use_global_score, "radius", lgr.acceptance_radius,
          "threshold", lgr.correspondence_threshold, "limit", lgr.correspondence_limit, "steps", lgr.num_refinement_steps)
    with torch.no_grad():
        rp, sp, sc, T = lgr(pkp[S], akp[S], pkm[S], akm[S], ms[S], ncs[S])
        bb, ii, jj, chunks, hyp, counts, best = hypothesis_stats(lgr, pkp[S], akp[S], pkm[S], akm[S], ms[S])
        hyp_stable, fit_stable, T_stable = stability(lgr, pkp[S], akp[S], pkm[S], akm[S], ms[S], hyp, T, chunks, sp)
    print("subset: patches", len(S), "correspondences", rp.shape[0], "hypotheses", len(chunks), "best", best, "inliers", int(counts[best]),
          "stable hypotheses", int(hyp_stable.sum()), "fit-stable", int(fit_stable.sum()), "T stable", T_stable)
    print("T subset\n", T.numpy(), "\nT all\n", T_all.numpy())
    store.update(B_patch_ids=S.numpy(), B_ref_knn_points=pkp[S].numpy(), B_src_knn_points=akp[S].numpy(), B_ref_knn_masks=pkm[S].numpy(),
                 B_src_knn_masks=akm[S].numpy(), B_log_scores=ms[S].numpy(), B_ref_corr_points=rp.numpy(), B_src_corr_points=sp.numpy(),
                 B_corr_scores=sc.numpy(), B_corr_bij=np.stack([bb.numpy(), ii.numpy(), jj.numpy()], 1).astype(np.int32), B_chunks=chunks,
                 B_hypotheses=hyp.numpy(), B_inlier_counts=counts.numpy().astype(np.int64), B_best=np.array(best), B_transform=T.numpy(),
                 B_full_transform=T_all.numpy(), B_full_num_corr=np.array(rp_all.shape[0]), B_hyp_stable=hyp_stable, B_fit_stable=fit_stable, B_T_stable=np.array(T_stable))
    # ---- D: partition -> node-level transport -> coarse matching -> patch-level transport, on the reference's own tensors
    assert len(cap["partition"]) == 2
    for side, (p_in, n_in, limit, (p2n, nmask, knn, kmask)) in zip(("pos", "anc"), cap["partition"]):
        assert limit == 128
        assert np.array_equal(n_in.numpy(), store["A_%s_points_c" % side])      # nodes = stage A's centres; points = stage-0 points of the scan
        store["D_%s_point_to_node" % side] = p2n.numpy().astype(np.int32)
        store["D_%s_node_masks" % side] = nmask.numpy()
        store["D_%s_node_knn_indices" % side] = knn.numpy().astype(np.int32)
        store["D_%s_node_knn_masks" % side] = kmask.numpy()
        store["D_%s_num_points" % side] = np.array(p_in.shape[0])
    (ns_in, nrm, ncm), ns_out = cap["node_ot"]
    store.update(D_node_scores_in=ns_in[0].numpy(), D_node_row_masks=nrm[0].numpy(), D_node_col_masks=ncm[0].numpy(),
                 D_node_log_scores=ns_out[0].numpy(), D_node_alpha=np.array(float(full.node_optimal_transport.alpha)))
    (cm_in, _, _), (ci, cj, cs) = cap["coarse"]
    assert torch.equal(cm_in, ns_out[0])
    store.update(D_node_corr_i=ci.numpy().astype(np.int32), D_node_corr_j=cj.numpy().astype(np.int32), D_node_corr_scores=cs.numpy())
    (ps_in, prm, pcm), ps_out = cap["patch_ot"]
    assert torch.equal(ps_out, ms)                                             # what fine_matching received (stage B's input)
    Dsel = S[:: N_SUBSET // D_N_PATCH][:D_N_PATCH]
    store.update(D_patch_ids=Dsel.numpy(), D_patch_scores_in=ps_in[Dsel].numpy(), D_patch_row_masks=prm[Dsel].numpy(),
                 D_patch_col_masks=pcm[Dsel].numpy(), D_patch_log_scores=ps_out[Dsel].numpy(),
                 D_patch_alpha=np.array(float(full.optimal_transport.alpha)))
    # how far the reference's OWN fp32 result is from the same module run in fp64 (the floor any fp32 implementation shares)
    torch.set_default_dtype(torch.float64)
    with torch.no_grad():
        n64 = full.node_optimal_transport.double()(ns_in.double(), nrm, ncm)[0]
        p64 = full.optimal_transport.double()(ps_in[Dsel].double(), prm[Dsel], pcm[Dsel])
    torch.set_default_dtype(torch.float32)
    full.node_optimal_transport.float()
    full.optimal_transport.float()

    def valid_mask(rm_, cm_):
        v = torch.ones(rm_.shape[0], rm_.shape[1] + 1, cm_.shape[1] + 1, dtype=torch.bool)
        v[:, :-1, :] &= rm_[:, :, None]
        v[:, :, :-1] &= cm_[:, None, :]
        return v

    nv = valid_mask(nrm, ncm)[0]
    pv = valid_mask(prm[Dsel], pcm[Dsel])
    e_node = float((ns_out[0].double() - n64)[nv].abs().max())
    e_patch = np.array([float((ps_out[Dsel][k].double() - p64[k])[pv[k]].abs().max()) for k in range(len(Dsel))])
    rng_patch = np.array([float(ps_in[Dsel][k][prm[Dsel][k]][:, pcm[Dsel][k]].max() - ps_in[Dsel][k][prm[Dsel][k]][:, pcm[Dsel][k]].min()) for k in range(len(Dsel))])
    store.update(D_node_log_scores_f64=n64.numpy(), D_node_ref_err_vs_f64=np.array(e_node), D_patch_ref_err_vs_f64=e_patch, D_patch_score_range=rng_patch,
                 D_patch_log_scores_f64=p64.numpy().astype(np.float64))
    print("D: reference fp32 vs fp64: node OT %.2e; patch OT per problem" % e_node, np.array2string(e_patch, precision=1), "score ranges", np.array2string(rng_patch, precision=0))
    print("D: nodes", ns_in.shape, "node corr", ci.shape[0], "patch problems", tuple(ps_in.shape), "kept", len(Dsel),
          "node score range %.3f..%.3f" % (float(ns_in.min()), float(ns_in.max())), "patch score range %.3f..%.3f" % (float(ps_in.min()), float(ps_in.max())))
    # ---- C: the same module on the well-conditioned synthetic case
    ref, src, rm, sm, logs, T_true = synthetic_lgr_case()
    tr, ts, trm, tsm, tl = (torch.from_numpy(x) for x in (ref, src, rm, sm, logs))
    with torch.no_grad():
        rp, sp, sc, T = lgr(tr, ts, trm, tsm, tl, torch.ones(len(ref)))
        bb, ii, jj, chunks, hyp, counts, best = hypothesis_stats(lgr, tr, ts, trm, tsm, tl)
        hyp_stable, fit_stable, T_stable = stability(lgr, tr, ts, trm, tsm, tl, hyp, T, chunks, sp)
    err = np.abs(T.numpy() - T_true).max()
    print("synthetic: stable hypotheses", int(hyp_stable.sum()), "fit-stable", int(fit_stable.sum()), "of", len(hyp_stable), "T stable", T_stable)
    print("synthetic: correspondences", rp.shape[0], "hypotheses", len(chunks), "best", best, "inliers", int(counts[best]), "of", rp.shape[0],
          " |T - T_true|max %.4f" % err)
    assert err < 0.02
    store.update(C_ref_corr_points=rp.numpy(), C_src_corr_points=sp.numpy(), C_corr_scores=sc.numpy(),
                 C_corr_bij=np.stack([bb.numpy(), ii.numpy(), jj.numpy()], 1).astype(np.int32), C_chunks=chunks, C_hypotheses=hyp.numpy(),
                 C_inlier_counts=counts.numpy().astype(np.int64), C_best=np.array(best), C_transform=T.numpy(), C_true_transform=T_true,
                 C_hyp_stable=hyp_stable, C_fit_stable=fit_stable, C_T_stable=np.array(T_stable))
    path = os.path.join(HERE, "pose_chain_golden.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, os.path.getsize(path) / 1e6, "MB")


if __name__ == "__main__":
    main()
