"""GPU: P registration pairs per call (`LCRNet.forward_pairs`, BASELINE configs[4]) give each pair what a call of its own gives.

The reference runs one pair per forward (model_family/LCRNet.py:274-321); batching shares the encoder / 3D-RoFormer (segmented
attention launch) / vote encoder / decoder launches between the pairs.  GroupNorm statistics are per pair in both forms, so the only
differences are fp32 re-association (other GEMM tile shapes for other row counts, atomics order): descriptors and dense features
within 1e-4 (relative to the tensor's magnitude), node counts equal.  The registration tail works on near-uniform random-weight
scores, so correspondences and the pose are only compared loosely here (they are pinned in tests/test_pose_chain_gpu.py)."""
import numpy as np
import pytest
import torch

from conftest import LIMITS, NUM_STAGES, RADIUS, VOXEL, load_scan

pytestmark = pytest.mark.gpu
PAIRS = [("003854", "000958"), ("000026", "000560"), ("003528", "004481")]


def test_segmented_attention_equals_separate_launches():
    from lcrnet_amd import functional as F
    g = torch.Generator().manual_seed(0)
    q_lens, k_lens = [844, 5, 33, 823], [823, 700, 1, 844]
    q = torch.randn(sum(q_lens), 128, generator=g).cuda()
    k = torch.randn(sum(k_lens), 128, generator=g).cuda()
    v = torch.randn(sum(k_lens), 128, generator=g).cuda()
    got = F.attention(q, k, v, 4, q_lens, k_lens)
    qo = ko = 0
    for nq, nk in zip(q_lens, k_lens):
        want = F.attention(q[qo:qo + nq].contiguous(), k[ko:ko + nk].contiguous(), v[ko:ko + nk].contiguous(), 4)
        assert torch.equal(got[qo:qo + nq], want)                  # same tiles, same arithmetic: bit-identical
        qo += nq
        ko += nk


def test_forward_pairs_equals_one_call_per_pair():
    from lcrnet_amd.config import make_cfg
    from lcrnet_amd.model_family import LCRNet
    from lcrnet_amd.pipeline import PairPipeline
    from lcrnet_amd.weights import seeded_state_dict
    cfg = make_cfg()
    cfg["neighbor_limits"] = LIMITS
    m = LCRNet(cfg).eval()
    m.load_state_dict(seeded_state_dict(m.state_dict(), 7351), strict=True)
    m = m.cuda()
    work = []
    for a, b in PAIRS:
        pa, pb = torch.from_numpy(load_scan(a)).cuda(), torch.from_numpy(load_scan(b)).cuda()
        work.append((torch.cat([pa, pb]), torch.tensor([len(pa), len(pb)], dtype=torch.int64, device="cuda")))
    with PairPipeline(m, VOXEL, RADIUS, NUM_STAGES, LIMITS, workers=1, pairs_per_call=1) as single, \
            PairPipeline(m, VOXEL, RADIUS, NUM_STAGES, LIMITS, workers=1, pairs_per_call=3) as batched:
        one = [{k: (v.cpu() if torch.is_tensor(v) else v) for k, v in o.items()} for o in single.run(work)]
        many = [{k: (v.cpu() if torch.is_tensor(v) else v) for k, v in o.items()} for o in batched.run(work)]
    assert len(one) == len(many) == 3

    def rel(a, b):
        return float((a - b).abs().max() / max(1.0, float(b.abs().max())))

    for i, (o, b) in enumerate(zip(one, many)):
        assert rel(b["pos_feature_global"], o["pos_feature_global"]) < 1e-5 and rel(b["anc_feature_global"], o["anc_feature_global"]) < 1e-5
        assert rel(b["pos_feats_c_enhanced"], o["pos_feats_c_enhanced"]) < 1e-4 and rel(b["anc_feats_c_enhanced"], o["anc_feats_c_enhanced"]) < 1e-4
        assert torch.allclose(b["shifted_pos_points_c"], o["shifted_pos_points_c"], atol=1e-4)
        assert b["length"].tolist() == o["length"].tolist(), i                         # same NMS result
        assert torch.allclose(b["pos_points_c"], o["pos_points_c"], atol=1e-4) and torch.allclose(b["anc_points_c"], o["anc_points_c"], atol=1e-4)
        assert rel(b["pos_feats_c"], o["pos_feats_c"]) < 1e-4 and rel(b["pos_feats_f"], o["pos_feats_f"]) < 1e-4 and rel(b["anc_feats_f"], o["anc_feats_f"]) < 1e-4
        assert (b["pos_node_knn_indices"] == o["pos_node_knn_indices"]).float().mean().item() > 0.995   # node centres differ by ~1e-6: rare nearest-node flips
        sb = set(zip(b["pos_node_corr_indices"].tolist(), b["anc_node_corr_indices"].tolist()))
        so = set(zip(o["pos_node_corr_indices"].tolist(), o["anc_node_corr_indices"].tolist()))
        assert len(sb & so) >= 0.97 * len(so)
        n = o["corr_scores"].shape[0]
        assert abs(b["corr_scores"].shape[0] - n) <= 0.05 * n
        T = b["estimated_transform"].numpy()
        assert abs(np.linalg.det(T[:3, :3]) - 1) < 1e-4
        print("pair %d: descriptor diff %.1e, enhanced %.1e, fine feats %.1e, node corr %d/%d, point corr %d vs %d" % (
            i, rel(b["pos_feature_global"], o["pos_feature_global"]), rel(b["pos_feats_c_enhanced"], o["pos_feats_c_enhanced"]),
            rel(b["pos_feats_f"], o["pos_feats_f"]), len(sb & so), len(so), b["corr_scores"].shape[0], n))


def test_nearest_only_upsampling_lists_are_column_zero_and_change_nothing():
    """PairPipeline's default builds the three decoder-only upsampling lists as ONE column (limit-1 search, arg-min path of the kernel):
    that column equals column 0 of the reference collate's full rows (bit for bit, ties by index like the sorted rows), and the pair model's
    outputs are identical with either form — KPDecoder reads column 0 only (backbone4.py:355-367)."""
    from lcrnet_amd.config import make_cfg
    from lcrnet_amd.data import precompute_batch
    from lcrnet_amd.model_family import LCRNet
    from lcrnet_amd.pipeline import PairPipeline
    from lcrnet_amd.weights import seeded_state_dict
    work = []
    for a, b in PAIRS:
        pa, pb = torch.from_numpy(load_scan(a)).cuda(), torch.from_numpy(load_scan(b)).cuda()
        work.append((torch.cat([pa, pb]), torch.tensor([len(pa), len(pb)], dtype=torch.int64, device="cuda")))
    pts, lens = torch.cat([w[0] for w in work]), torch.cat([w[1] for w in work])
    full = precompute_batch(pts, lens, NUM_STAGES, VOXEL, RADIUS, LIMITS, upsampling=True)
    near = precompute_batch(pts, lens, NUM_STAGES, VOXEL, RADIUS, LIMITS, upsampling="nearest")
    for i in range(NUM_STAGES - 1):
        assert near["upsampling"][i].shape == (full["upsampling"][i].shape[0], 1)
        assert torch.equal(near["upsampling"][i][:, 0], full["upsampling"][i][:, 0]), i
        assert torch.equal(near["subsampling"][i], full["subsampling"][i]) and torch.equal(near["neighbors"][i], full["neighbors"][i])
    # a tiny radius: most rows have NO neighbour (padding = the number of support rows) and many exactly one
    from lcrnet_amd.modules.ops import radius_search
    q, s = pts[:5000].contiguous(), pts[5000:9000].contiguous()
    ql, sl = torch.tensor([5000], device="cuda"), torch.tensor([4000], device="cuda")
    one = radius_search(q, s, ql, sl, 0.4, 1, check=False)
    many = radius_search(q, s, ql, sl, 0.4, 16, check=False)
    assert torch.equal(one[:, 0], many[:, 0]) and int((one[:, 0] == 4000).sum()) > 100
    cfg = make_cfg()
    cfg["neighbor_limits"] = LIMITS
    m = LCRNet(cfg).eval()
    m.load_state_dict(seeded_state_dict(m.state_dict(), 7351), strict=True)
    m = m.cuda()
    with PairPipeline(m, VOXEL, RADIUS, NUM_STAGES, LIMITS, workers=1, pairs_per_call=3, upsampling=True) as a, \
            PairPipeline(m, VOXEL, RADIUS, NUM_STAGES, LIMITS, workers=1, pairs_per_call=3) as b:
        assert b.upsampling == "nearest"
        fa = [{k: v.cpu() for k, v in o.items() if torch.is_tensor(v)} for o in a.run(work)]
        fb = [{k: v.cpu() for k, v in o.items() if torch.is_tensor(v)} for o in b.run(work)]
    for oa, ob in zip(fa, fb):
        for k in ("pos_feats_f", "anc_feats_f", "pos_corr_points", "corr_scores", "estimated_transform", "pos_feature_global"):
            assert oa[k].shape == ob[k].shape and torch.allclose(oa[k], ob[k], atol=1e-6), k


def test_many_workers_give_what_one_worker_gives():
    """Four pinned worker threads (one pair per call) and five (four pairs per call) against one call at a time, 45 pairs: outputs arrive in
    input order and every pair's descriptors / node counts / correspondence counts are those of the single-worker run (shared weight tables,
    derived-tensor caches and the native sequencers are used from several host threads at once)."""
    import itertools
    from conftest import GOLDEN
    import os
    from lcrnet_amd.config import make_cfg
    from lcrnet_amd.model_family import LCRNet
    from lcrnet_amd.pipeline import PairPipeline
    from lcrnet_amd.weights import seeded_state_dict
    cfg = make_cfg()
    cfg["neighbor_limits"] = LIMITS
    m = LCRNet(cfg).eval()
    m.load_state_dict(seeded_state_dict(m.state_dict(), 7351), strict=True)
    m = m.cuda()
    names = sorted(f[:-4] for f in os.listdir(os.path.join(GOLDEN, "scans")) if f.endswith(".npy"))
    scans = {n: torch.from_numpy(load_scan(n)).cuda() for n in names}
    combos = list(itertools.combinations(names, 2)) * 3
    work = [(torch.cat([scans[a], scans[b]]), torch.tensor([len(scans[a]), len(scans[b])], dtype=torch.int64, device="cuda")) for a, b in combos]
    keep = lambda o: {k: o[k].cpu() for k in ("pos_feature_global", "anc_feature_global", "length", "corr_scores", "estimated_transform", "pos_feats_c_enhanced")}
    with PairPipeline(m, VOXEL, RADIUS, NUM_STAGES, LIMITS, workers=1, pairs_per_call=1) as one:
        want = [keep(o) for o in one.run(work)]
    for workers, P in ((4, 1), (5, 4)):
        with PairPipeline(m, VOXEL, RADIUS, NUM_STAGES, LIMITS, workers=workers, pairs_per_call=P) as pp:
            got = [keep(o) for o in pp.run(work)]
        assert len(got) == len(want) == 45
        for i, (g, w) in enumerate(zip(got, want)):
            assert (g["pos_feature_global"] - w["pos_feature_global"]).abs().max() < 1e-5 and (g["anc_feature_global"] - w["anc_feature_global"]).abs().max() < 1e-5, (workers, P, i)
            assert g["pos_feats_c_enhanced"].shape == w["pos_feats_c_enhanced"].shape
            assert float((g["pos_feats_c_enhanced"] - w["pos_feats_c_enhanced"]).abs().max()) < 1e-4 * max(1.0, float(w["pos_feats_c_enhanced"].abs().max())), (workers, P, i)
            assert g["length"].tolist() == w["length"].tolist(), (workers, P, i)
            assert abs(g["corr_scores"].shape[0] - w["corr_scores"].shape[0]) <= 0.05 * w["corr_scores"].shape[0], (workers, P, i)
