"""Loop detection chained into registration on the device — the reference's inference flow
    experiments/inference/infer_loop_detection_descriptor_generation.py  (descriptors, one `{seq}_{idx}.npz` per frame)
 -> experiments/inference/infer_loop_detection_find_top1.py              (re-normalise :75, per-query search, `top1_with_thres_%.2f/NN.txt`)
 -> experiments/inference/infer_registration.py                           (every listed pair through the pair model, `{seq}_pose` lines)
as three offline scripts with files between them.  Here the same three stages share the voxelised clouds in HBM: descriptors through
`DescriptorPipeline`, the masked exhaustive search through `lcr_retrieval_topk`, the threshold rule and the file formats of
`io_formats` (pinned against the imported reference, tests/test_loop_closure_handoff.py), and the detected loops through
`PairPipeline.forward_pairs`.  The files are still written — they are the reference's interface to its evaluation scripts."""
import os

import numpy as np
import torch

from . import io_formats as io
from .data import voxelize_raw_scans
from .pipeline import DescriptorPipeline, PairPipeline
from .retrieval import retrieval_topk

VOXEL, RADIUS, NUM_STAGES = 0.3, 1.275, 4
K, EXCLUDE, START = 50, 100, 101


def voxelise_frames(raw_frames, voxel=VOXEL, batch=8):
    """raw scans (device f32 [N_i, >= 3]) -> the 0.3 m voxelised clouds the reference reads from `downsampled_xyzi/*.npy` (device f32 [n_i,3])."""
    out = []
    for f0 in range(0, len(raw_frames), batch):
        chunk = [f[:, :3].contiguous() for f in raw_frames[f0:f0 + batch]]
        pts, _, lh = voxelize_raw_scans(torch.cat(chunk), torch.tensor([len(c) for c in chunk], dtype=torch.int64, device=chunk[0].device), voxel)
        off = 0
        for n in lh:
            out.append(pts[off:off + n].contiguous())
            off += n
    return out


def sequence_descriptors(desc_model, clouds, limits, batch=8):
    """voxelised clouds -> [C,256] descriptors (device), `batch` scans per step."""
    dev = clouds[0].device
    groups = [(torch.cat(clouds[f0:f0 + batch]), torch.tensor([len(c) for c in clouds[f0:f0 + batch]], dtype=torch.int64, device=dev))
              for f0 in range(0, len(clouds), batch)]
    with DescriptorPipeline(desc_model, VOXEL, RADIUS, NUM_STAGES, limits, upsampling=False, raw_voxel=None) as pipe:
        pipe.enable_dual_encoder(2)
        return torch.cat([d.clone() for d in pipe.run(groups)])


def detect_loops(desc, thres, k=K, exclude=EXCLUDE):
    """descriptors [C,256] (device) -> (rows float64 [R,3] as predicted_des_L2_dis.npz holds them, kept float32 [K,3] = the lines of NN.txt).
    The descriptors take the reference's host round trip: stored as float32, re-normalised with numpy (:75), searched in float32."""
    C = desc.shape[0]
    d = torch.from_numpy(io.renormalise_descriptors(desc.cpu().numpy())).to(desc.device)
    if C - 1 <= START:
        return np.zeros((0, 3)), np.zeros((0, 3), np.float32)
    idx, d2 = retrieval_topk(d[START:C - 1], START, d, k, exclude)
    idx_h, d2_h = idx.cpu().numpy(), d2.cpu().numpy()
    rows = io.pair_dist_rows(np.arange(START, C - 1), idx_h, np.where(idx_h >= 0, d2_h, np.inf))
    return rows, io.top1_with_threshold(rows, C, thres)


def register_pairs(pair_model, clouds, pairs, limits, pairs_per_call=16, workers=None):
    """pairs [(pos_idx, anc_idx)] -> list of output dicts (ref = clouds[pos], src = clouds[anc]; the stack order of the reference's
    registration collate: [ref, src])."""
    dev = clouds[0].device
    work = [(torch.cat([clouds[p], clouds[a]]), torch.tensor([len(clouds[p]), len(clouds[a])], dtype=torch.int64, device=dev)) for p, a in pairs]
    P = max(1, min(pairs_per_call, len(work)))
    with PairPipeline(pair_model, neighbor_limits=limits, workers=workers or (4 if P == 1 else 5), pairs_per_call=P) as pp:
        return list(pp.run(work))


def run(desc_model, pair_model, raw_frames, thres, out_dir, seq=0, desc_limits=(64, 65, 74, 80), pair_limits=(74, 68, 70, 67), pairs_per_call=16,
        max_pairs=None, write_descriptors=True):
    """The whole chain; writes `{out_dir}/features/{seq}_{idx}.npz`, `{out_dir}/features/predicted_des_L2_dis.npz`,
    `{out_dir}/result/top1_with_thres_%.2f/%02d.txt` and `{out_dir}/registration/{seq}_pose`.  -> dict with the in-memory results."""
    clouds = voxelise_frames(raw_frames)
    desc = sequence_descriptors(desc_model, clouds, list(desc_limits))
    feat_dir = os.path.join(out_dir, "features")
    os.makedirs(feat_dir, exist_ok=True)
    if write_descriptors:
        dh = desc.cpu().numpy()
        for i in range(len(dh)):
            io.save_descriptor(feat_dir, seq, i, dh[i])
    rows, kept = detect_loops(desc, thres)
    io.save_pair_dist(feat_dir, rows)
    name = io.save_top1_with_threshold(out_dir, seq, kept, thres)
    pairs = io.load_loop_pairs(name)                       # through the file, like the reference's loader
    if max_pairs is not None:
        pairs = pairs[:max_pairs]
    outs = register_pairs(pair_model, clouds, pairs, list(pair_limits), pairs_per_call) if pairs else []
    reg_dir = os.path.join(out_dir, "registration")
    os.makedirs(reg_dir, exist_ok=True)
    pose_file = os.path.join(reg_dir, "%s_pose" % seq)
    with open(pose_file, "a") as f:
        for (pos, anc), o in zip(pairs, outs):
            f.write(io.pose_line(pos, anc, o["estimated_transform"].cpu().numpy()))
    return {"clouds": clouds, "descriptors": desc, "rows": rows, "kept": kept, "top1_file": name, "pairs": pairs, "outputs": outs, "pose_file": pose_file}
