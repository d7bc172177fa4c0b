"""Golden vectors for descriptor retrieval + loop-detection metrics (a-9, f-3) from the IMPORTED reference (build container only).

    python tests/golden/make_golden_retrieval.py

What runs, unmodified, from /root/reference (never copied):
  experiments/loop_detection/eval_loop_detection_overlap_dataset.py
    eval_one_epoch (:148-260)  — the per-query search loop :183-214 and the row layout of predicted_des_L2_dis.npz :217-219
    compute_topN (:29-62), compute_PR_overlap (:66-121), compute_AP (:13-17), compute_F1 (:19-27), plotPRC (:124-145, sklearn auc)
on (a) the reference's own ground-truth asset assets/data/kitti/loop_detection/overlap/loop_gt_seq00_0.3overlap_inactive.npz
(4541 frames; committed as a data fixture) with seeded synthetic descriptors (`synthetic_descriptors` below: a frame with a
ground-truth loop sits at a noisy copy of its first loop partner, so that the sweep produces non-trivial precision / recall), and
(b) a small 400-frame case with exact duplicates (distance ties) and a database shorter than k.

`faiss` is not installed (un-vendored dependency): for the import it is replaced by `_FaissStub` — an exhaustive
float32 squared-L2 search with faiss's result conventions (ascending distance, ties by ascending id, `-1` / FLT_MAX fill when the
database holds fewer than k vectors; IndexIVFFlat with nlist=1 is exhaustive).  So the LOOP STRUCTURE, the row layout and every
metric are the reference's own code; the arithmetic inside the distance is the stub's ((x-y)^2 summed in fp32, like faiss's
`fvec_L2sqr` for a single query) and faiss's tie order stays unpinned, as DESIGN.md §2 says.

Output: tests/golden/retrieval_golden.npz (+ the GT asset copied byte for byte to tests/golden/loop_gt_seq00_0.3overlap_inactive.npz).
"""
import os
import shutil
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference"
GT_ASSET = os.path.join(REF, "assets/data/kitti/loop_detection/overlap/loop_gt_seq00_0.3overlap_inactive.npz")
GT_COPY = os.path.join(HERE, "loop_gt_seq00_0.3overlap_inactive.npz")


def synthetic_descriptors(ground_truth, seed=0, dim=256):
    """[C, dim] unit-norm f32.  Frames without a loop: random directions.  Frames with ground-truth loops: their first loop
    partner's descriptor plus noise whose size cycles through 5 levels (0.3 .. 6 x unit norm), renormalised — the nearest
    neighbour is a true loop for the quiet ones and a random frame for the loud ones."""
    rng = np.random.default_rng(seed)
    C = len(ground_truth)
    d = rng.standard_normal((C, dim)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    levels = np.array([0.3, 1.0, 2.5, 4.0, 6.0], dtype=np.float32)
    for i in range(C):
        gt = np.asarray(ground_truth[i])
        if gt.any():
            j = int(gt[0])
            if j < i:
                n = rng.standard_normal(dim).astype(np.float32)
                n /= np.linalg.norm(n)
                v = d[j] + levels[i % 5] * n
                d[i] = v / np.linalg.norm(v)
    return d


def small_case(seed=5, C=400, dim=256):
    """400 frames: duplicates (exact ties), loops every 3rd frame after 150, database shorter than k for the first queries."""
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((C, dim)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[40:80] = d[0:40]                      # exact duplicates inside every late query's database
    gt = np.empty(C, dtype=object)
    for i in range(C):
        gt[i] = np.array([float(i - 120), float(i - 121)]) if (i >= 150 and i % 3 == 0) else np.array([])
    for i in range(150, C):
        if i % 3 == 0 and i % 2 == 0:       # half of the loop frames really are close to their partner
            v = d[i - 120] + 0.1 * rng.standard_normal(dim).astype(np.float32)
            d[i] = v / np.linalg.norm(v)
    return d, gt


class _FaissStub(types.ModuleType):
    METRIC_L2 = 1

    class IndexFlatL2:
        def __init__(self, d):
            self.d = d

    class IndexIVFFlat:
        def __init__(self, quantizer, d, nlist, metric):
            assert nlist == 1 and metric == 1
            self.d, self.is_trained, self.x = d, False, None

        def train(self, x):
            self.is_trained = True

        def add(self, x):
            self.x = np.ascontiguousarray(x, dtype=np.float32)

        def search(self, q, k):
            q = np.ascontiguousarray(q, dtype=np.float32)
            D = np.full((len(q), k), np.finfo(np.float32).max, dtype=np.float32)
            I = np.full((len(q), k), -1, dtype=np.int64)
            for r in range(len(q)):
                diff = self.x - q[r][None, :]
                d2 = np.einsum("ij,ij->i", diff, diff, dtype=np.float32)
                order = np.argsort(d2, kind="stable")[:k]
                D[r, :len(order)], I[r, :len(order)] = d2[order], order
            return D, I


def run_reference(desc, gt_file, topns):
    """Descriptors -> per-frame npz files -> the reference's eval_one_epoch (search loop, npz rows) -> its metric functions."""
    import experiments.loop_detection.eval_loop_detection_overlap_dataset as ev
    from easydict import EasyDict
    tmp = tempfile.mkdtemp(prefix="lcr_ret_")
    feat = os.path.join(tmp, "kitti")
    data = os.path.join(tmp, "data")
    os.makedirs(feat)
    os.makedirs(os.path.join(data, "overlap"))
    shutil.copy(gt_file, os.path.join(data, "overlap", "loop_gt_seq00_0.3overlap_inactive.npz"))
    from lcrnet_amd.io_formats import save_descriptor
    for i in range(len(desc)):
        save_descriptor(feat, 0, i, desc[i])            # `{seq}_{idx}.npz` like test_loop_detection.py:65; the reader sorts by int('0_12') = 12
    cfg = EasyDict(ld_feature_dir=tmp + "/", dataset="kitti", data=EasyDict(dataset_root=data))
    cwd = os.getcwd()
    os.chdir(tmp)                                       # plotPRC saves ./PRC.png
    try:
        ev.eval_one_epoch(cfg, seqlist=[0])
        pred = os.path.join(feat, "predicted_des_L2_dis.npz")
        rows = np.load(pred)["arr_0"]
        pair = np.asarray(rows, dtype="float32").reshape((len(rows), 3))
        gtf = os.path.join(data, "overlap", "loop_gt_seq00_0.3overlap_inactive.npz")
        P, R = ev.compute_PR_overlap(pair, gtf, [0, 1], 0.01)
        ap = ev.compute_AP(P, R)
        f1, f1_idx = ev.compute_F1(P, R)
        tops = [ev.compute_topN(pred, gtf, int(n)) for n in topns]
        auc = ev.plotPRC(P, R, f1, tops[:2], False, "kitti")
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp)
    return rows, np.array(P, dtype=np.float64), np.array(R, dtype=np.float64), ap, f1, f1_idx, tops, auc


def main():
    import make_golden_model as mgm
    mgm.install_stubs()
    sys.modules.pop("IPython", None)                    # matplotlib inspects a loaded IPython; the eval script does not need the stub
    sys.modules["faiss"] = _FaissStub("faiss")
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, REF)
    shutil.copy(GT_ASSET, GT_COPY)
    os.chmod(GT_COPY, 0o644)
    store = {}
    # (a) KITTI 00 ground truth, C = 4541
    gt = np.load(GT_COPY, allow_pickle=True)["arr_0"]
    desc = synthetic_descriptors(gt, seed=0)
    rows, P, R, ap, f1, f1_idx, tops, auc = run_reference(desc, GT_COPY, [1, 45, 5])
    print("kitti00: rows", rows.shape, "top1 %.4f top45 %.4f top5 %.4f  F1 %.4f@%d  AP %.4f  AUC %.3f  PR points %d" % (tops[0], tops[1], tops[2], f1, f1_idx, ap, auc, len(P)))
    k = 50
    assert rows.shape[1:] == (1, 3)                                # the reference stacks (1,3) rows: predicted_des_L2_dis.npz holds [R,1,3]
    rows = rows.reshape(-1, 3)
    q = rows[:, 0].reshape(-1, k)
    assert (q == q[:, :1]).all() and q[0, 0] == 101 and q[-1, 0] == len(gt) - 2
    store["k00_query_first_last"] = np.array([q[0, 0], q[-1, 0]], dtype=np.int64)
    store["k00_idx"] = rows[:, 1].reshape(-1, k).astype(np.int16)
    store["k00_d2"] = rows[:, 2].reshape(-1, k).astype(np.float32)
    store["k00_precisions"], store["k00_recalls"] = P, R
    store["k00_scalars"] = np.array([tops[0], tops[1], tops[2], f1, f1_idx, ap, auc], dtype=np.float64)
    # (b) small case with ties and short databases
    d2, gt2 = small_case()
    tmp = tempfile.mkdtemp(prefix="lcr_gt_")
    gtf = os.path.join(tmp, "gt.npz")
    np.savez(gtf, gt2)
    rows, P, R, ap, f1, f1_idx, tops, auc = run_reference(d2, gtf, [1, 3])
    shutil.rmtree(tmp)
    print("small: rows", rows.shape, "top1 %.4f top3 %.4f  F1 %.4f@%d  AP %.4f  AUC %.3f" % (tops[0], tops[1], f1, f1_idx, ap, auc))
    store["small_rows"] = rows.reshape(-1, 3).astype(np.float64)   # full [R,3] rows incl. the -1 / FLT_MAX fill (298 queries x 50)
    store["small_precisions"], store["small_recalls"] = P, R
    store["small_scalars"] = np.array([tops[0], tops[1], f1, f1_idx, ap, auc], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "retrieval_golden.npz"), **store)
    print("wrote retrieval_golden.npz", os.path.getsize(os.path.join(HERE, "retrieval_golden.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
