"""GPU: exhaustive masked squared-L2 top-k (a-9) through the C ABI vs the fp64 oracle."""
import numpy as np
import pytest
import torch

from oracle import torch_ref

pytestmark = pytest.mark.gpu


def _desc(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.nn.functional.normalize(torch.randn(n, 256, generator=g), dim=1)


@pytest.mark.parametrize("C,k", [(700, 50), (4541, 50), (150, 50)])
def test_topk_matches_oracle(C, k):
    from lcrnet_amd.retrieval import retrieval_topk
    desc = _desc(C)
    qs, widx, wd2 = torch_ref.retrieval_topk(desc, k=k, exclude=100, start=101)
    d = desc.cuda()
    q0, q1 = 101, C - 1
    idx, d2 = retrieval_topk(d[q0:q1], q0, d, k=k, exclude=100)
    idx, d2 = idx.cpu().long(), d2.cpu().double()
    fin = torch.isfinite(wd2)
    assert torch.equal(torch.isfinite(d2), fin)
    assert (d2[fin] - wd2[fin]).abs().max().item() < 1e-5
    assert torch.equal(idx == -1, widx == -1)
    # indices: identical wherever the oracle's neighbouring distances are separated by more than the fp32 noise
    gap = torch.ones_like(wd2, dtype=torch.bool)
    gap[:, 1:] &= (wd2[:, 1:] - wd2[:, :-1]).abs() > 1e-5
    gap[:, :-1] &= (wd2[:, 1:] - wd2[:, :-1]).abs() > 1e-5
    sel = gap & fin
    assert torch.equal(idx[sel], widx[sel])
    assert sel.sum().item() >= 0.9 * fin.sum().item()


def test_exact_ties_break_by_index():
    from lcrnet_amd.retrieval import retrieval_topk
    base = _desc(40, seed=3)
    desc = base.repeat(8, 1)                      # frames i and i+40 are identical: exact distance ties
    d = desc.cuda()
    idx, d2 = retrieval_topk(d[250:260], 250, d, k=20, exclude=100)
    qs, widx, wd2 = torch_ref.retrieval_topk(desc, k=20, exclude=100, start=250, stop=260)
    assert torch.equal(idx.cpu().long(), widx)


def _check_range(desc, d, q0, q1, k=50, exclude=100):
    from lcrnet_amd.retrieval import retrieval_topk
    qs, widx, wd2 = torch_ref.retrieval_topk(desc, k=k, exclude=exclude, start=q0, stop=q1)
    idx, d2 = retrieval_topk(d[q0:q1], q0, d, k=k, exclude=exclude)
    idx, d2 = idx.cpu().long(), d2.cpu().double()
    fin = torch.isfinite(wd2)
    assert torch.equal(torch.isfinite(d2), fin) and torch.equal(idx == -1, widx == -1)
    err = (d2[fin] - wd2[fin]).abs().max().item() if fin.any() else 0.0
    assert err < 1e-5, err
    # identical indices wherever the oracle's neighbouring distances are separated by more than the fp32 noise, or tie EXACTLY
    # (exact ties — duplicated descriptors — must come out in ascending index order, like the oracle's stable sort)
    gap = (wd2[:, 1:] - wd2[:, :-1]).abs()
    ok = (gap > 1e-5) | (gap == 0)
    sel = torch.ones_like(wd2, dtype=torch.bool)
    sel[:, 1:] &= ok
    sel[:, :-1] &= ok
    sel &= fin
    assert torch.equal(idx[sel], widx[sel])
    assert sel.sum().item() >= 0.9 * fin.sum().item()
    return idx, widx


@pytest.mark.parametrize("C", [16500, 23201, 40000])
def test_chunked_rows_beyond_16384_columns(C):
    """Rows longer than one LDS chunk (TK_CHUNK = 16384 columns): the carried best-k path that BASELINE config 4 (C = 23 201 =
    KITTI 00-10) needs.  Query ranges: the first frames (database shorter than k), the frames whose window ends at the chunk
    boundary, and the last frames (two or three chunks); exact duplicates placed in DIFFERENT chunks tie and must keep index order."""
    desc = _desc(C, seed=C)
    dup = [(5, 16390), (7, 16383), (9, 16384), (11, min(C - 300, 32770))]
    for a, b in dup:
        desc[b] = desc[a]
    g = torch.Generator().manual_seed(1)
    for t, (a, _) in enumerate(dup):                            # late queries that sit next to a duplicated pair: the tie is in their top-2
        desc[C - 20 - t] = torch.nn.functional.normalize(desc[a] + 0.02 * torch.randn(256, generator=g), dim=0)
    d = desc.cuda()
    _check_range(desc, d, 101, 170)
    _check_range(desc, d, 16384 + 100 - 20, min(16384 + 100 + 20, C - 1))   # window end j < i - 100 crosses column 16384
    idx, widx = _check_range(desc, d, C - 120, C - 1)
    for t, (a, b) in enumerate(dup):
        row = (C - 20 - t) - (C - 120)
        assert idx[row, 0].item() == a
        if b < (C - 20 - t) - 100:                              # the duplicate lies inside this query's window (not for C = 16500)
            assert idx[row, 1].item() == b, (idx[row, :3], a, b)


def test_hip_search_reproduces_the_reference_loop_on_kitti00_ground_truth():
    """The HIP top-k on the seeded descriptors of tests/golden/make_golden_retrieval.py vs the rows the reference's own
    eval_one_epoch loop wrote for them (retrieval_golden.npz), then the reference's metrics on the HIP rows: Recall@1 / @45, the PR
    sweep, F1max, AP and AUC must come out as the reference computed them."""
    import os
    import sys
    from conftest import GOLDEN
    sys.path.insert(0, GOLDEN)
    from make_golden_retrieval import synthetic_descriptors
    from lcrnet_amd import evaluation as ev
    from lcrnet_amd import io_formats as io
    from lcrnet_amd.retrieval import retrieval_topk
    ret = np.load(os.path.join(GOLDEN, "retrieval_golden.npz"))
    gt = np.load(os.path.join(GOLDEN, "loop_gt_seq00_0.3overlap_inactive.npz"), allow_pickle=True)["arr_0"]
    desc = synthetic_descriptors(gt, seed=0)
    d = torch.from_numpy(desc).cuda()
    C = len(desc)
    idx, d2 = retrieval_topk(d[101:C - 1], 101, d, k=50, exclude=100)
    idx, d2 = idx.cpu().numpy().astype(np.int64), d2.cpu().numpy().astype(np.float64)
    widx, wd2 = ret["k00_idx"].astype(np.int64), ret["k00_d2"].astype(np.float64)
    fin = widx >= 0
    assert np.array_equal(idx >= 0, fin)
    assert np.abs(d2[fin] - wd2[fin]).max() < 1e-5
    gap = np.ones_like(wd2, dtype=bool)
    gap[:, 1:] &= np.abs(wd2[:, 1:] - wd2[:, :-1]) > 1e-5
    gap[:, :-1] &= np.abs(wd2[:, 1:] - wd2[:, :-1]) > 1e-5
    assert not ((idx != widx) & fin & gap).any()
    rows = io.pair_dist_rows(np.arange(101, C - 1), idx, np.where(fin, d2, np.inf))
    top1, top45, top5, f1, f1_idx, ap, auc = ret["k00_scalars"]
    assert ev.compute_topN(rows, gt, 1) == top1 and ev.compute_topN(rows, gt, 45) == top45
    P, R = ev.compute_PR_overlap(rows, gt)
    P, R, wP, wR = np.asarray(P, np.float64), np.asarray(R, np.float64), ret["k00_precisions"].astype(np.float64), ret["k00_recalls"].astype(np.float64)
    assert P.shape == wP.shape
    # The HIP distances are within 1e-5 of the reference's, so at a sweep threshold ONE frame whose top-1 distance sits that close to it may
    # change class.  What that is worth, per sweep point: 1 / (frames predicted positive there) of precision, 1 / (frames with a loop) of
    # recall — and of the AUC (x 100, trapezoid over the sweep) the moved point's share of its two neighbouring segments.
    first = rows.reshape(-1, 50, 3)[:, 0, :]                                    # the top-1 row of every query frame 101..C-2
    q = first[:, 0].astype(np.int64)
    sweep = q >= 150                                                            # compute_PR_overlap's start
    top1_d = first[sweep, 2].astype(np.float32)
    n_loop = sum(1 for i in q[sweep] if np.asarray(gt[i]).any())
    thr = np.arange(0, 1, 0.01)[:len(P)]
    n_pos = np.array([(top1_d <= t).sum() for t in thr])
    dP, dR = 1.0 / np.maximum(n_pos - 1, 1), 1.0 / max(n_loop - 1, 1)
    assert (np.abs(P - wP) <= dP + 1e-12).all() and (np.abs(R - wR) <= dR + 1e-12).all()
    moved = (P != wP) | (R != wR)
    span = lambda a: np.abs(np.concatenate([a[1:], a[-1:]]) - np.concatenate([a[:1], a[:-1]])) / 2
    auc_worth = 100.0 * float(((span(wR) + dR) * dP + (span(wP) + dP) * dR)[moved].sum())
    got_auc = ev.auc(P, R)
    print("PR sweep: %d of %d points moved by one frame; AUC %.6f vs reference %.6f (worth of the moved points: %.2e)" % (moved.sum(), len(P), got_auc, auc, auc_worth))
    assert abs(got_auc - auc) <= auc_worth + 1e-9
    assert abs(ev.compute_F1(P, R)[0] - f1) <= 2 * max(dP[moved].max(initial=0.0), dR if moved.any() else 0.0) + 1e-12
