"""On-disk formats of the reference's loop-detection / registration scripts, so that its own eval*.py
consume this package's outputs unchanged.

  * per-frame descriptor  `{seq_id}_{idx}.npz`, key `anc_global` (1,256) f32   — test_loop_detection.py:60-69
  * retrieval rows        `predicted_des_L2_dis.npz`, `arr_0` float64 [R,3] = (query i, match j, squared L2), 50 rows per
    query frame 101..C-2 in ascending distance; queries whose database holds fewer than k frames are filled the way faiss
    fills them (j = -1, d = FLT_MAX)                                            — eval_loop_detection_overlap_dataset.py:183-219
  * demo text line        `pos anc L2 r11 … t3`                                 — demo/demo.py:80-81
"""
import glob
import os

import numpy as np

FAISS_EMPTY_DISTANCE = np.float32(3.4028234663852886e38)


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
    np.savez_compressed(os.path.join(features_root, "predicted_des_L2_dis"), np.asarray(rows, dtype=np.float64))


def lcr_output_line(pos_idx, anc_idx, pos_global, anc_global, estimated_transform):
    feat_dis = float(np.sqrt(np.sum((np.asarray(pos_global) - np.asarray(anc_global)) ** 2)))
    m = np.asarray(estimated_transform, dtype=np.float64).reshape(-1)[:12]
    return f"{pos_idx} {anc_idx} {feat_dis:.2f} " + " ".join(f"{v:.6f}" for v in m) + " \n"
