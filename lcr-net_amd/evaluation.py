"""Loop-detection metrics with the reference's definitions (experiments/loop_detection/eval_loop_detection_overlap_dataset.py):
Recall@N `compute_topN` (:29-62), PR sweep `compute_PR_overlap` (:66-121), F1max (:13-27), AUC (sklearn-free trapezoid of
`plotPRC` :124-145).  Inputs are the in-memory forms of the reference's files: rows [R,3] (i, j, d2) and the ground-truth
object array (ground_truth[i] = array of loop frame ids, empty if none)."""
import numpy as np


def _first_rows(rows):
    """index of the first row of every query id (rows are grouped per query, ascending distance)."""
    q = rows[:, 0].astype(np.int64)
    first = {}
    for r, i in enumerate(q):
        if i not in first:
            first[int(i)] = r
    return first


def compute_topN(rows, ground_truth, topn):
    rows = np.asarray(rows, dtype=np.float32).reshape(-1, 3)
    first = _first_rows(rows)
    all_have_gt = tps = 0
    for idx in range(0, len(ground_truth) - 1):
        gt = np.asarray(ground_truth[idx])
        if not gt.any():
            continue
        all_have_gt += 1
        if idx not in first:
            raise IndexError("query frame %d has ground truth but no retrieval rows (the reference raises here too)" % idx)
        r0 = first[idx]
        for t in range(topn):
            if rows[r0 + t, 0] == idx and rows[r0 + t, 1] in gt:
                tps += 1
                break
    return tps / max(all_have_gt, 1)


def compute_PR_overlap(rows, ground_truth, thre_range=(0, 1), interval=0.01, start=150):
    rows = np.asarray(rows, dtype=np.float32).reshape(-1, 3)
    first = _first_rows(rows)
    precisions, recalls = [], []
    for thres in np.arange(thre_range[0], thre_range[1], interval):
        tps = fps = tns = fns = 0
        for idx in range(start, len(ground_truth) - 1):
            gt = np.asarray(ground_truth[idx])
            r0 = first[idx]
            if rows[r0, 2] > thres:
                if not gt.any():
                    tns += 1
                else:
                    fns += 1
            elif rows[r0, 1] in gt:
                tps += 1
            else:
                fps += 1
        precision = 1 if fps == 0 else tps / (tps + fps)
        recall = 1 if fns == 0 else tps / (tps + fns)
        precisions.append(precision)
        recalls.append(recall)
        if recall == 1:
            break
    return precisions, recalls


def compute_F1(precisions, recalls):
    p, r = np.asarray(precisions, dtype=np.float64), np.asarray(recalls, dtype=np.float64)
    f1 = 2 * p * r / (p + r + 1e-12)
    return float(f1.max()), int(f1.argmax())


def auc(precisions, recalls):
    order = sorted(zip(recalls, precisions), reverse=True)
    r = np.array([o[0] for o in order], dtype=np.float64)
    p = np.array([o[1] for o in order], dtype=np.float64)
    return float(abs(np.trapz(p, r)) * 100)
