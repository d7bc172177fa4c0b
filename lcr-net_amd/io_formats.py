"""On-disk formats of the reference's loop-detection / registration scripts, so that its own eval*.py
consume this package's outputs unchanged.

  * per-frame descriptor  `{seq_id}_{idx}.npz`, key `anc_global` (1,256) f32   — test_loop_detection.py:60-69
  * retrieval rows        `predicted_des_L2_dis.npz`, `arr_0` float64 [R,1,3] = (query i, match j, squared L2), 50 rows per
    query frame 101..C-2 in ascending distance; queries whose database holds fewer than k frames are filled the way faiss
    fills them (j = -1, d = FLT_MAX)                                            — eval_loop_detection_overlap_dataset.py:183-219
  * demo text line        `pos anc L2 r11 … t3`                                 — demo/demo.py:80-81
  * raw scans             KITTI velodyne `.bin` (little-endian f32 [N,4]: x, y, z, intensity — data/Kitti/downsample_pcd.py:21) and the
    reference's down-sampled `.npy` ([N,4] xyzi, :29-38; dataset_overlap_online.py:245 slices [:, :3] on the host) as PINNED host rows for
    the pipeline's ingest leg (pipeline.HostIngest), which consumes the four columns unsliced
"""
import glob
import os

import numpy as np

FAISS_EMPTY_DISTANCE = np.float32(3.4028234663852886e38)


def load_scan_rows(path, pin=True):
    """A scan file -> contiguous float32 host rows [N, C] (C = 4 for velodyne `.bin` and xyzi `.npy`, 3 for xyz `.npy`) as a torch tensor,
    pinned when `pin` and a GPU runtime is there: ready for `DescriptorPipeline.run` (host batches), no `[:, :3]` copy."""
    import torch
    if path.endswith(".bin"):
        a = np.fromfile(path, dtype="<f4")
        if a.size % 4:
            raise ValueError("%s: %d floats is not a whole number of velodyne rows (x, y, z, intensity)" % (path, a.size))
        a = a.reshape(-1, 4)
    else:
        a = np.load(path)
        if a.ndim != 2 or a.shape[1] < 3:
            raise ValueError("%s: expected [N, C >= 3] points, got %s" % (path, a.shape))
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    if pin and torch.cuda.is_available():
        t = t.pin_memory()
    return t


def stack_scan_rows(rows):
    """[t_0 [N_0,C], ...] -> (points [sum N, C], lengths i64 [B]) host batch for `DescriptorPipeline.run` (pinned if the inputs are)."""
    import torch
    if len({int(r.shape[1]) for r in rows}) != 1:
        raise ValueError("scans of one batch must have the same number of columns")
    pts = torch.cat(list(rows))
    if all(r.is_pinned() for r in rows) and not pts.is_pinned():
        pts = pts.pin_memory()
    return pts, torch.tensor([int(r.shape[0]) for r in rows], dtype=torch.int64)


def save_descriptor(output_dir, seq_id, idx, anc_global):
    a = np.asarray(anc_global, dtype=np.float32).reshape(1, -1)
    np.savez_compressed(os.path.join(output_dir, f"{seq_id}_{idx}.npz"), anc_global=a)


def load_descriptors(features_root, seq):
    """All `{seq}*.npz` of a directory, sorted by the integer file stem like the reference (:167-170) -> [C,256] f32."""
    names = sorted(glob.glob(os.path.join(features_root, "%d*.npz" % seq)), key=lambda x: int(os.path.splitext(os.path.basename(x))[0].replace("_", "")))
    return np.concatenate([np.load(n)["anc_global"].astype(np.float32) for n in names])


def pair_dist_rows(query_ids, idx, d2):
    """(Q,), (Q,k), (Q,k) -> float64 [Q*k, 3] rows (i, j, d2) in the reference's order."""
    q = np.asarray(query_ids, dtype=np.float64)
    idx = np.asarray(idx, dtype=np.float64)
    d2 = np.asarray(d2, dtype=np.float64).copy()
    d2[idx < 0] = float(FAISS_EMPTY_DISTANCE)
    rows = np.stack([np.repeat(q, idx.shape[1]), idx.reshape(-1), d2.reshape(-1)], 1)
    return rows


def save_pair_dist(features_root, rows):
    """`predicted_des_L2_dis.npz`, `arr_0` float64 [R,1,3]: the reference stacks (1,3) rows (eval_..._dataset.py:203-217), and both
    of its readers reshape to (R,3) (:32-33, :226-227)."""
    np.savez_compressed(os.path.join(features_root, "predicted_des_L2_dis"), np.asarray(rows, dtype=np.float64).reshape(-1, 1, 3))


def lcr_output_line(pos_idx, anc_idx, pos_global, anc_global, estimated_transform):
    feat_dis = float(np.sqrt(np.sum((np.asarray(pos_global) - np.asarray(anc_global)) ** 2)))
    m = np.asarray(estimated_transform, dtype=np.float64).reshape(-1)[:12]
    return f"{pos_idx} {anc_idx} {feat_dis:.2f} " + " ".join(f"{v:.6f}" for v in m) + " \n"


# keys of the per-pair registration file, in the reference's order (demo/demo.py:84-105; experiments/registration/test_registration.py
# writes the same file and experiments/registration/eval.py reads it back)
REGISTRATION_KEYS = ("pos_points_f", "anc_points_f", "pos_points_c", "anc_points_c", "pos_node_corr_indices", "anc_node_corr_indices",
                     "pos_corr_points", "anc_corr_points", "corr_scores", "gt_node_corr_indices", "gt_node_corr_overlaps",
                     "estimated_transform")


def _to_numpy(v):
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.asarray(v)


def save_registration(output_dir, seq_id, anchor_idx, positive_idx, output_dict, transform):
    """`{seq_id}_{anchor_idx}_{positive_idx}.npz` with the 15 arrays of demo/demo.py:86-105 (the model's output dict + the
    ground-truth `transform` of the data dict + both global descriptors).  Returns the path."""
    # gt_node_corr_indices / gt_node_corr_overlaps are written by demo.py:97-98 but are NOT produced by the shipped
    # LCRNet.forward (LCRNet.py:161-321 never sets them; only the training-time loss path reads them, loss_reg.py:175,256): when the
    # output dict has none they are stored empty, (0,2) int64 / (0,) f32, so that experiments/registration/eval.py:97-98 can open the file
    empty = {"gt_node_corr_indices": np.zeros((0, 2), dtype=np.int64), "gt_node_corr_overlaps": np.zeros((0,), dtype=np.float32)}
    arrays = {k: (_to_numpy(output_dict[k]) if k in output_dict else empty[k]) for k in REGISTRATION_KEYS}
    arrays["transform"] = _to_numpy(transform)
    arrays["pos_feature_global"] = _to_numpy(output_dict["pos_feature_global"])
    arrays["anc_feature_global"] = _to_numpy(output_dict["anc_feature_global"])
    path = os.path.join(output_dir, f"{seq_id}_{anchor_idx}_{positive_idx}.npz")
    np.savez_compressed(path, **arrays)
    return path


def load_registration(path):
    """-> dict of arrays; `estimated_transform` falls back to `estimated_transform_lgr` like eval.py:171-175."""
    d = dict(np.load(path))
    if "estimated_transform" not in d and "estimated_transform_lgr" in d:
        d["estimated_transform"] = d["estimated_transform_lgr"]
    return d


# ---- loop detection -> registration hand-off (experiments/inference/infer_loop_detection_find_top1.py, infer_registration.py) -----------
def renormalise_descriptors(desc):
    """`emb_list_map / np.linalg.norm(emb_list_map, axis=1, keepdims=True)` (infer_loop_detection_find_top1.py:75): the inference flow
    re-normalises the stored descriptors on the host before the search."""
    d = np.asarray(desc, dtype=np.float32)
    return d / np.linalg.norm(d, axis=1, keepdims=True)


def top1_with_threshold(rows, n_frames, thres):
    """find_top1 (:13-26): for every query frame idx in [0, n_frames - 1) ALL of its rows (i, j, d2) whose distance is below `thres`, in row
    order — despite its name the reference keeps every candidate under the threshold, not only the nearest.  rows: the float32 [R,3] view of
    predicted_des_L2_dis.npz (:108-109).  -> float32 [K,3]."""
    r = np.asarray(rows, dtype=np.float32).reshape(-1, 3)
    keep = (r[:, 2] < thres) & (r[:, 0] >= 0) & (r[:, 0] < n_frames - 1)
    r = r[keep]
    return r[np.argsort(r[:, 0], kind="stable")]           # the reference walks idx upwards and appends each query's rows in file order


def top1_lines(rows3):
    """The text of `top1_with_thres_%.2f/%02d.txt` (:36-39): `{int(i)} {int(j)} {d}  \\n` with d a numpy float32 printed by an f-string."""
    return "".join(f"{int(r[0])} {int(r[1])} {(r[2])}  \n" for r in np.asarray(rows3, dtype=np.float32).reshape(-1, 3))


def save_top1_with_threshold(dataset_root, seq, rows3, thres):
    """-> path of `{dataset_root}/result/top1_with_thres_{thres:.2f}/{seq:02d}.txt` (appended to, like the reference's open(..., 'a'))."""
    path = "%s/result/top1_with_thres_%.2f" % (dataset_root, thres)
    os.makedirs(path, exist_ok=True)
    name = "%s/%02d.txt" % (path, seq)
    with open(name, "a") as f:
        f.write(top1_lines(rows3))
    return name


def load_loop_pairs(path):
    """The reference's reader of that file (datasets/loop_closure/kitti/dataset.py:48-57): per line anc_idx = field 0 (the query frame),
    pos_idx = field 1 (its match) -> [(pos_idx, anc_idx)] = (ref frame, src frame) of the registration pair."""
    out = []
    with open(path) as f:
        for line in f.readlines():
            s = line.split()
            if s:
                out.append((int(s[1]), int(s[0])))
    return out


def pose_line(pos_idx, anc_idx, estimated_transform):
    """One line of `{seq}_pose` (infer_registration.py:77-78): pos anc and the first 12 entries of the 4x4 transform at 6 decimals."""
    m = np.asarray(estimated_transform, dtype=np.float32).reshape(-1)[:12]
    return f"{pos_idx} {anc_idx} " + " ".join(f"{v:.6f}" for v in m) + " \n"
