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


def compute_AP(precisions, recalls):
    """:13-17 — sum of (recall step) x precision."""
    ap = 0.
    for i in range(1, len(precisions)):
        ap += (recalls[i] - recalls[i - 1]) * precisions[i]
    return ap


def compute_F1(precisions, recalls):
    """:19-27 — max over the sweep of 2pr/(p+r), and its position.  Like the reference there is no epsilon: a sweep point with
    p + r == 0 yields NaN and NaN wins max/argmax (numpy semantics), exactly as there."""
    p, r = np.asarray(precisions), np.asarray(recalls)
    with np.errstate(invalid="ignore", divide="ignore"):
        f1 = 2 * p * r / (p + r)
    return f1.max(), f1.argmax()


def auc(precisions, recalls):
    """plotPRC :124-145: points sorted by recall descending, sklearn.metrics.auc (trapezoid; sign follows the direction) x 100."""
    order = sorted(zip(recalls, precisions), reverse=True)
    r = np.array([o[0] for o in order], dtype=np.float64)
    p = np.array([o[1] for o in order], dtype=np.float64)
    trapezoid = getattr(np, "trapezoid", None) or np.trapz
    return float(abs(trapezoid(p, r)) * 100)


# ---- registration metrics (BASELINE config 5; utils/utils/registration.py:13-113, experiments/registration/eval.py:222-236) ----------
def relative_rotation_error(gt_rotation, est_rotation):
    """Isotropic RRE in degrees: acos((trace(R_est^T R_gt) - 1) / 2), argument clipped to [-1, 1] (registration.py:13-28)."""
    c = 0.5 * (np.trace(np.asarray(est_rotation).T @ np.asarray(gt_rotation)) - 1.0)
    return 180.0 * np.arccos(np.clip(c, -1.0, 1.0)) / np.pi


def euler_angles_deg(R):
    """Roll / pitch / yaw in degrees with the reference's conventions (registration.py:30-47): the gimbal-lock branch
    (sqrt(R00^2 + R10^2) < 1e-6) sets yaw = 0, and degrees use the literal 3.141592653589793."""
    import math
    sy = math.sqrt(R[0, 0] * R[0, 0] + R[1, 0] * R[1, 0])
    if sy >= 1e-6:
        ang = (math.atan2(R[2, 1], R[2, 2]), math.atan2(-R[2, 0], sy), math.atan2(R[1, 0], R[0, 0]))
    else:
        ang = (math.atan2(-R[1, 2], R[1, 1]), math.atan2(-R[2, 0], sy), 0)
    return tuple(a * 180.0 / 3.141592653589793 for a in ang)


def relative_rotation_error_rpy(gt_rotation, est_rotation):
    """Per-axis |difference| of the Euler angles, folded into [0, 180] (registration.py:50-80)."""
    d = [abs(g - e) for g, e in zip(euler_angles_deg(gt_rotation), euler_angles_deg(est_rotation))]
    return tuple(360 - x if x > 180 else x for x in d)


def relative_translation_error(gt_translation, est_translation):
    """RTE = |t_gt - t_est|_2 (registration.py:82-93)."""
    return np.linalg.norm(np.asarray(gt_translation) - np.asarray(est_translation))


def compute_registration_error(gt_transform, est_transform):
    """(4,4), (4,4) -> (rre deg, rte m, rx, ry, rz) like registration.py:97-113."""
    gt, est = np.asarray(gt_transform), np.asarray(est_transform)
    rre = relative_rotation_error(gt[:3, :3], est[:3, :3])
    rx, ry, rz = relative_rotation_error_rpy(gt[:3, :3], est[:3, :3])
    return rre, relative_translation_error(gt[:3, 3], est[:3, 3]), rx, ry, rz


def registration_partial(gt_transforms, est_transforms, rre_threshold=5.0, rte_threshold=2.0):
    """This rank's share of the registration block as SUMS (float64 [7]: pairs, accepted, then RRE / RTE / Rx / Ry / Rz summed over
    the accepted pairs) — what the sharded evaluation (pairs dealt to the ranks, BASELINE configs[4]) exchanges; the reference
    reduces its per-rank meters the same way (one all-reduce of scalars, utils/utils/torch.py:16-34)."""
    v = np.zeros(7, dtype=np.float64)
    for gt, est in zip(gt_transforms, est_transforms):
        e = compute_registration_error(gt, est)
        v[0] += 1.0
        if e[0] < rre_threshold and e[1] < rte_threshold:
            v[1] += 1.0
            v[2:] += np.asarray(e, dtype=np.float64)
    return v


def registration_reduce(partial, group=None, device=None):
    """Sum the ranks' registration_partial vectors (one all-reduce when torch.distributed is initialised; the identity otherwise)
    and return the summary dict of registration_summary.  `device`: where the all-reduce runs (the rank's GPU for RCCL)."""
    import torch
    import torch.distributed as dist
    v = np.asarray(partial, dtype=np.float64)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        t = torch.from_numpy(v.copy())
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        v = t.cpu().numpy()
    n, k = int(round(v[0])), int(round(v[1]))
    m = v[2:] / k if k else np.full(5, np.nan)
    return {"RR": float(k / n) if n else 0.0, "RRE": float(m[0]), "RTE": float(m[1]), "Rx": float(m[2]), "Ry": float(m[3]),
            "Rz": float(m[4]), "pairs": n, "accepted": k}


def registration_summary(gt_transforms, est_transforms, rre_threshold=5.0, rte_threshold=2.0):
    """The registration block of experiments/registration/eval.py:222-236,269-277: a pair is accepted when RRE < 5 deg and
    RTE < 2 m (config_reg.py:66-67); RR = mean acceptance over all pairs, RRE / RTE / Rx / Ry / Rz = means over the ACCEPTED pairs
    (NaN when none is accepted: the reference's meter is np.mean of an empty list, utils/utils/average_meter.py:28-29)."""
    acc, kept = [], []
    for gt, est in zip(gt_transforms, est_transforms):
        e = compute_registration_error(gt, est)
        ok = bool(e[0] < rre_threshold and e[1] < rte_threshold)
        acc.append(float(ok))
        if ok:
            kept.append(e)
    m = np.mean(np.asarray(kept, dtype=np.float64), axis=0) if kept else np.full(5, np.nan)
    return {"RR": float(np.mean(acc)) if acc else 0.0, "RRE": float(m[0]), "RTE": float(m[1]), "Rx": float(m[2]), "Ry": float(m[3]),
            "Rz": float(m[4]), "pairs": len(acc), "accepted": len(kept)}
