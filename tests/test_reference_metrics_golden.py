"""CPU: evaluation.py / io_formats.py / the retrieval oracle against fixtures produced by the IMPORTED reference
(tests/golden/make_golden_retrieval.py: eval_one_epoch's search loop + compute_topN / compute_PR_overlap / compute_AP / compute_F1 /
plotPRC on the reference's own KITTI-00 ground-truth asset; tests/golden/make_golden_registration.py: compute_registration_error and
the acceptance rule of experiments/registration/eval.py).  These pin a-9's oracle and f-3's metrics."""
import os
import sys

import numpy as np
import pytest
import torch

from lcrnet_amd import evaluation as ev
from lcrnet_amd import io_formats as io
from oracle import torch_ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)


@pytest.fixture(scope="module")
def ret():
    return np.load(os.path.join(GOLD, "retrieval_golden.npz"))


@pytest.fixture(scope="module")
def gt00():
    return np.load(os.path.join(GOLD, "loop_gt_seq00_0.3overlap_inactive.npz"), allow_pickle=True)["arr_0"]


def golden_rows_k00(ret):
    idx, d2 = ret["k00_idx"].astype(np.float64), ret["k00_d2"].astype(np.float64)
    q0, q1 = ret["k00_query_first_last"]
    return io.pair_dist_rows(np.arange(q0, q1 + 1), idx, d2)


def test_metrics_equal_the_reference_on_its_own_rows(ret, gt00):
    rows = golden_rows_k00(ret)
    top1, top45, top5, f1, f1_idx, ap, auc = ret["k00_scalars"]
    assert ev.compute_topN(rows, gt00, 1) == top1
    assert ev.compute_topN(rows, gt00, 45) == top45
    assert ev.compute_topN(rows, gt00, 5) == top5
    P, R = ev.compute_PR_overlap(rows, gt00)
    assert np.array_equal(np.asarray(P, dtype=np.float64), ret["k00_precisions"]) and np.array_equal(np.asarray(R, dtype=np.float64), ret["k00_recalls"])
    got_f1, got_idx = ev.compute_F1(P, R)
    assert got_f1 == f1 and got_idx == int(f1_idx)
    assert ev.compute_AP(P, R) == ap
    assert abs(ev.auc(P, R) - auc) < 1e-9
    assert 0.3 < top1 < 0.9 and len(P) > 20                       # the fixture is not degenerate


def test_oracle_search_equals_the_reference_loop(ret, gt00):
    """fp64 oracle (the checker of the HIP top-k) vs the rows written by the reference's eval_one_epoch loop."""
    from make_golden_retrieval import synthetic_descriptors
    desc = synthetic_descriptors(gt00, seed=0)
    qs, idx, d2 = torch_ref.retrieval_topk(torch.from_numpy(desc), k=50, exclude=100, start=101)
    assert qs[0].item() == ret["k00_query_first_last"][0] and qs[-1].item() == ret["k00_query_first_last"][1]
    widx, wd2 = ret["k00_idx"].astype(np.int64), ret["k00_d2"].astype(np.float64)
    idx, d2 = idx.numpy(), d2.numpy()
    fin = widx >= 0
    assert np.array_equal(idx >= 0, fin)
    assert np.abs(d2[fin] - wd2[fin]).max() < 2e-6                # the stub sums (x-y)^2 in fp32
    diff = (idx != widx) & fin
    gap = np.ones_like(wd2, dtype=bool)                            # index swaps only inside fp32-noise ties
    gap[:, 1:] &= np.abs(wd2[:, 1:] - wd2[:, :-1]) > 2e-6
    gap[:, :-1] &= np.abs(wd2[:, 1:] - wd2[:, :-1]) > 2e-6
    assert not (diff & gap).any() and diff.mean() < 1e-3
    rows = io.pair_dist_rows(qs.numpy(), idx, d2)
    assert ev.compute_topN(rows, gt00, 1) == ret["k00_scalars"][0]


def test_small_case_rows_with_ties_and_short_databases(ret):
    from make_golden_retrieval import small_case
    desc, gt = small_case()
    qs, idx, d2 = torch_ref.retrieval_topk(torch.from_numpy(desc), k=50, exclude=100, start=101)
    rows = io.pair_dist_rows(qs.numpy(), idx.numpy(), d2.numpy())
    want = ret["small_rows"]
    assert rows.shape == want.shape
    assert np.array_equal(rows[:, 0], want[:, 0])                 # query ids
    fill = want[:, 1] < 0
    assert np.array_equal(rows[:, 1] < 0, fill) and np.array_equal(rows[fill, 2], want[fill, 2])     # the -1 / FLT_MAX fill of short databases
    assert np.abs(rows[~fill, 2] - want[~fill, 2]).max() < 2e-6
    # match ids: identical, except inside runs whose fp32 distances (the stub's) are closer than the fp32 noise; EXACT ties
    # (the duplicated descriptors 0..39 / 40..79) keep ascending id in both
    d = want[:, 2]
    near = np.zeros(len(d), dtype=bool)
    close = (np.abs(np.diff(d)) < 2e-6) & (want[1:, 0] == want[:-1, 0])
    near[1:] |= close
    near[:-1] |= close
    diff = rows[:, 1] != want[:, 1]
    assert not (diff & ~near).any() and diff.sum() < 20
    dup = (~fill[:-1]) & (want[1:, 2] == want[:-1, 2]) & (want[1:, 0] == want[:-1, 0]) & (np.abs(want[1:, 1] - want[:-1, 1]) == 40)
    assert dup.sum() > 1000 and (want[1:, 1][dup] > want[:-1, 1][dup]).all()
    same = dup & ~diff[1:] & ~diff[:-1]
    assert same.sum() > 1000 and (rows[1:, 1][same] > rows[:-1, 1][same]).all()
    top1, top3, f1, f1_idx, ap, auc = ret["small_scalars"]
    assert ev.compute_topN(want, gt, 1) == top1 and ev.compute_topN(want, gt, 3) == top3
    P, R = ev.compute_PR_overlap(want, gt)
    assert np.array_equal(np.asarray(P, dtype=np.float64), ret["small_precisions"]) and np.array_equal(np.asarray(R, dtype=np.float64), ret["small_recalls"])
    gf1, gidx = ev.compute_F1(P, R)
    assert (gf1 == f1 or (np.isnan(gf1) and np.isnan(f1))) and gidx == int(f1_idx)
    assert ev.compute_AP(P, R) == ap and abs(ev.auc(P, R) - auc) < 1e-9


def test_registration_errors_equal_the_reference():
    g = np.load(os.path.join(GOLD, "registration_golden.npz"))
    for gt, est, want in zip(g["gt"], g["est"], g["errors"]):
        got = np.array(ev.compute_registration_error(gt, est), dtype=np.float64)
        assert np.array_equal(got, want), (got, want)
    s = ev.registration_summary(g["gt"], g["est"])
    got = np.array([s["RR"], s["RRE"], s["RTE"], s["Rx"], s["Ry"], s["Rz"]])
    assert np.allclose(got, g["summary"], rtol=1e-13, atol=0) and 0 < s["accepted"] < s["pairs"]


def test_registration_npz_has_the_reference_keys(tmp_path):
    rng = np.random.default_rng(0)
    out = {k: torch.from_numpy(rng.standard_normal((7, 3)).astype(np.float32)) for k in
           ("pos_points_f", "anc_points_f", "pos_points_c", "anc_points_c", "pos_corr_points", "anc_corr_points")}
    out.update(pos_node_corr_indices=torch.arange(5), anc_node_corr_indices=torch.arange(5), corr_scores=torch.rand(7),
               estimated_transform=torch.eye(4), pos_feature_global=torch.rand(1, 256), anc_feature_global=torch.rand(1, 256))
    path = io.save_registration(str(tmp_path), 8, 15, 1200, out, np.eye(4, dtype=np.float32))
    assert os.path.basename(path) == "8_15_1200.npz"                # demo.py:84: f'{seq_id}_{anchor_idx}_{positive_idx}.npz'
    d = io.load_registration(path)
    want_keys = {"pos_points_f", "anc_points_f", "pos_points_c", "anc_points_c", "pos_node_corr_indices", "anc_node_corr_indices",
                 "pos_corr_points", "anc_corr_points", "corr_scores", "gt_node_corr_indices", "gt_node_corr_overlaps",
                 "estimated_transform", "transform", "pos_feature_global", "anc_feature_global"}
    assert set(d) == want_keys
    assert d["gt_node_corr_indices"].shape == (0, 2) and np.array_equal(d["estimated_transform"], np.eye(4, dtype=np.float32))
    rre, rte, *_ = ev.compute_registration_error(d["transform"], d["estimated_transform"])
    assert rre == 0 and rte == 0
