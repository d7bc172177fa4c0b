"""CPU: the torch restatement of the pose tail (oracle/torch_ref.py, a-10) against the golden tensors of the imported
reference's full `LCRNet.forward` on the demo pair (tests/golden/make_golden_pose.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, LIMITS, NUM_STAGES, RADIUS, VOXEL, load_scan
from oracle import ops as oracle_ops
from oracle import torch_ref
from test_torch_ref_golden import seeded_sd  # noqa: F401  (fixture)


@pytest.fixture(scope="module")
def pose_golden():
    return np.load(os.path.join(GOLDEN, "pose_golden.npz"))


@pytest.fixture(scope="module")
def oracle_run(seeded_sd):  # noqa: F811
    a, b = load_scan("003854"), load_scan("000958")
    st = oracle_ops.precompute_data_stack_mode(np.concatenate([a, b]), np.array([len(a), len(b)]), NUM_STAGES, VOXEL, RADIUS, LIMITS)
    dd = {k: [torch.from_numpy(np.ascontiguousarray(t)) for t in v] for k, v in st.items()}
    with torch.no_grad():
        feats = torch_ref.kp_encoder(seeded_sd, torch.ones(len(a) + len(b), 1), dd)
        n0 = int(dd["lengths"][-1][0])
        pc = dd["points"][-1]
        e0, e1 = torch_ref.thd_roformer(seeded_sd, pc[:n0], pc[n0:], feats[-1][:n0], feats[-1][n0:])
        out = torch_ref.pose_tail(seeded_sd, dd, feats, torch.cat([e0, e1], 0), LIMITS)
    return out


def test_vote_nms_and_node_features(oracle_run, pose_golden):
    vd = oracle_run["vote"]
    n0 = pose_golden["shifted_pos_points_c"].shape[0]
    assert np.allclose(vd["shifted"][:n0].numpy(), pose_golden["shifted_pos_points_c"], atol=2e-4)
    assert vd["length"].tolist() == pose_golden["length"].tolist()
    m0 = int(vd["length"][0])
    assert np.allclose(vd["centers"][:m0].numpy(), pose_golden["pos_points_c"], atol=2e-4)
    assert np.allclose(vd["centers"][m0:].numpy(), pose_golden["anc_points_c"], atol=2e-4)
    r = pose_golden["feats_c_rows"]
    assert np.allclose(vd["feats_c"][r].numpy(), pose_golden["feats_c_vals"], atol=5e-4, rtol=1e-3)


def test_partition_matching_and_pose(oracle_run, pose_golden):
    # node centres agree to ~1e-5, so a point equidistant (to fp32 noise) from two nodes may flip: allow a few entries
    for k in ("pos_node_knn_indices", "anc_node_knn_indices"):
        same = (oracle_run[k].numpy() == pose_golden[k]).mean()
        print(k, "identical entries:", same)
        assert same > 0.995, (k, same)
    got = set(zip(oracle_run["pos_node_corr_indices"].tolist(), oracle_run["anc_node_corr_indices"].tolist()))
    want = set(zip(pose_golden["pos_node_corr_indices"].tolist(), pose_golden["anc_node_corr_indices"].tolist()))
    assert len(got & want) >= 0.98 * len(want) and abs(len(got) - len(want)) <= 0.02 * len(want)
    r = pose_golden["pos_feats_f_rows"]
    n0 = int(pose_golden["pos_feats_f_stats"][2])
    assert np.allclose(oracle_run["feats_f"][:n0][r].numpy(), pose_golden["pos_feats_f_vals"], atol=5e-4, rtol=1e-3)
    n_corr = pose_golden["corr_scores"].shape[0]
    assert abs(oracle_run["corr_scores"].shape[0] - n_corr) <= 0.03 * n_corr
    T, Tw = oracle_run["estimated_transform"].numpy(), pose_golden["estimated_transform"]
    assert np.allclose(T, Tw, atol=5e-2), (T, Tw)


def test_oracle_lgr_switches_vs_reference_module():
    """The oracle's restatement of every LocalGlobalRegistration switch (use_dustbin=False + confidence threshold, use_global_score,
    correspondence_limit, with k and mutual) against the imported reference module's outputs (make_golden_lgr_options.py)."""
    import json
    import sys
    sys.path.insert(0, GOLDEN)
    from make_golden_pose_chain import synthetic_lgr_case
    g = np.load(os.path.join(GOLDEN, "lgr_options_golden.npz"))
    ref, src, rm, sm, logs, _ = synthetic_lgr_case()
    t = lambda x: torch.from_numpy(x)
    for tag, (topk, mutual, dust, thr, use_gs, limit) in json.loads(str(g["cases_json"])).items():
        with torch.no_grad():
            rp, sp, sc, T = torch_ref.local_global_registration(t(ref), t(src), t(rm), t(sm), t(logs), mutual=mutual, topk=topk, use_dustbin=dust,
                                                                confidence_threshold=thr, global_scores=t(g["global_scores"]) if use_gs else None,
                                                                correspondence_limit=limit)
        assert np.array_equal(rp.numpy(), g[tag + "/ref_corr_points"]) and np.array_equal(sp.numpy(), g[tag + "/src_corr_points"]), tag
        assert np.abs(sc.numpy() - g[tag + "/corr_scores"]).max() < 1e-7, tag
        assert np.abs(T.numpy() - g[tag + "/transform"]).max() < 1e-4, tag
