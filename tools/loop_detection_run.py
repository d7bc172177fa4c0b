#!/usr/bin/env python
"""End-to-end loop detection over a whole sequence (BASELINE configs[2] on one GPU, configs[3] sharded over N GPUs), on synthetic
stand-in data, writing and reading the reference's own file formats; one JSON line.

    python tools/loop_detection_run.py [--frames 4541] [--gpus N] [--out DIR] [--unique 16]

What runs, per rank (frames are split into contiguous ranges, one process per GPU like bench.py):
  raw scans --DescriptorPipeline--> 256-D descriptors                      (the hot path bench.py measures)
    --> `{seq}_{idx}.npz` per frame, key `anc_global`                       (test_loop_detection.py:60-69; rank-local frames)
    --> all-gather of the descriptor blocks (RCCL, N > 1)                   
    --> masked exhaustive squared-L2 top-50 of the rank's own query frames  (eval_loop_detection_overlap_dataset.py:183-214)
  rank 0: rows --> `predicted_des_L2_dis.npz` ([R,1,3], the reference's layout) --> read back --> Recall@1 / top-1 % (45),
  the PR sweep, F1max, AP and AUC with the reference's definitions (lcrnet_amd.evaluation, pinned by fixtures from the imported
  reference) against the reference's ground truth for KITTI 00 (committed fixture; used when --frames 4541, else a synthetic one).

The KITTI dataset is not available offline: frames are `--unique` synthetic 64-beam scans (lcrnet_amd.synthetic), re-used with a
rigid motion per frame (yaw i * 2.39996 rad, +-2 m shifts) so that every frame is a different cloud.  Revisits (frame i == frame
i - K) make the metrics non-trivial: with --revisit-every K (default 1500) frame i >= K repeats the pose of frame i - K with 1 cm
noise, and the ground truth marks it.  With random weights the numbers mean nothing for place recognition; the point is the path,
its formats and its timing (scans/s over a whole sequence, retrieval ms, file I/O excluded from the scan rate like bench.py)."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VOXEL, RADIUS, NUM_STAGES, LIMITS, BATCH = 0.3, 1.275, 4, [64, 65, 74, 80], 8


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4541)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--unique", type=int, default=16, help="distinct synthetic scans (1.2 s each to generate)")
    ap.add_argument("--revisit-every", type=int, default=1500)
    ap.add_argument("--host-rows", action="store_true", help="feed the scans as host [N,4] velodyne rows through the ingest leg (upload inside the timed region)")
    ap.add_argument("--out", default=None, help="directory for the npz files (default: a temporary one, removed afterwards)")
    ap.add_argument("--dump", default=None, help="rank 0 writes the gathered descriptors and the rows to this .npz (strong-scaling identity check of tools/scale_run.sh; small corpora only)")
    return ap.parse_args()


def spawn(args):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                              env=dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                                       MASTER_PORT=str(port))) for r in range(args.gpus)]
    sys.exit(max(p.wait() for p in procs))


def frame_cloud(base, i, revisit):
    """Frame i: unique scan (i mod U) of the pose of frame i, or — on a revisit — of frame i - revisit plus 1 cm noise."""
    src = i - revisit if (revisit and i >= revisit and (i // 7) % 3 == 0) else i
    rng = np.random.default_rng(src)
    yaw = src * 2.39996
    R = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]], dtype=np.float32)
    pts = base[src % len(base)] @ R.T + np.append(rng.uniform(-2, 2, 2), 0).astype(np.float32)
    if src != i:
        pts = pts + np.random.default_rng(i).standard_normal(pts.shape).astype(np.float32) * 0.01
    return pts.astype(np.float32), src != i


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn(args)
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = 0 if os.environ.get("LCR_BENCH_SINGLE_DEVICE") else int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("LCR_BENCH_BACKEND", "gloo" if os.environ.get("LCR_BENCH_SINGLE_DEVICE") else "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    import lcrnet_amd.synthetic as synthetic
    from lcrnet_amd import evaluation as ev
    from lcrnet_amd import io_formats as io
    from lcrnet_amd.model_family import create_model
    from lcrnet_amd.pipeline import DescriptorPipeline
    from lcrnet_amd.retrieval import all_gather_descriptors, retrieval_topk, search_range, shard_range
    from lcrnet_amd.weights import seeded_state_dict

    C = args.frames
    lo, hi = shard_range(C, world, rank)
    base = [synthetic.synthetic_scan(1000 + u) for u in range(args.unique)]
    model = create_model().eval()
    model.load_state_dict(seeded_state_dict(model.state_dict(), 7351))
    model = model.to(dev)
    out_dir = args.out or tempfile.mkdtemp(prefix="lcr_ld_")
    os.makedirs(out_dir, exist_ok=True)

    def batches():
        for f0 in range(lo, hi, BATCH):
            clouds = [frame_cloud(base, i, args.revisit_every)[0] for i in range(f0, min(f0 + BATCH, hi))]
            if args.host_rows:
                # velodyne rows [N,4] (x, y, z, intensity) in pageable HOST memory, as a loader reading .bin files leaves them: the pipeline's
                # ingest leg stages, uploads (copy stream) and voxelises them unsliced — the upload is then INSIDE the timed region
                xyz = np.concatenate(clouds)
                rows = np.concatenate([xyz, np.zeros((len(xyz), 1), np.float32)], axis=1)
                yield (torch.from_numpy(rows), torch.tensor([len(c) for c in clouds], dtype=torch.int64))
            else:
                yield (torch.from_numpy(np.concatenate(clouds)).to(dev, non_blocking=True),
                       torch.tensor([len(c) for c in clouds], dtype=torch.int64, device=dev))

    staged = list(batches())                     # inputs resident in HBM before the clock starts (file I/O / synthesis excluded, like bench.py)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    with DescriptorPipeline(model, VOXEL, RADIUS, NUM_STAGES, LIMITS, upsampling=False, raw_voxel=VOXEL) as pipe:
        pipe.enable_dual_encoder(2)
        for _ in pipe.run(staged[:4]):
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        desc = torch.cat([d for d in pipe.run(staged)])
        torch.cuda.synchronize()
        t_desc = time.perf_counter() - t0
    assert desc.shape == (hi - lo, 256)
    t0 = time.perf_counter()
    dn = desc.cpu().numpy()
    for k, i in enumerate(range(lo, hi)):
        io.save_descriptor(out_dir, 0, i, dn[k])
    t_write = time.perf_counter() - t0
    # ---- exchange + retrieval of this rank's query frames
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    full = all_gather_descriptors(desc, C) if world > 1 else desc
    q_lo, q_hi = search_range(C, world, rank, 101, 100)      # contiguous, equal causal work per rank
    idx, d2 = retrieval_topk(full[q_lo:q_hi], q_lo, full, 50, 100) if q_hi > q_lo else (torch.empty((0, 50), dtype=torch.int32, device=dev),
                                                                                         torch.empty((0, 50), device=dev))
    torch.cuda.synchronize()
    t_ret = time.perf_counter() - t0
    rows = io.pair_dist_rows(np.arange(q_lo, q_hi), idx.cpu().numpy(), np.where(idx.cpu().numpy() >= 0, d2.cpu().numpy(), np.inf))
    if world > 1:                                # gather the row blocks on rank 0 for the file (not on the data path)
        blocks = [None] * world
        dist.gather_object(rows, blocks if rank == 0 else None, dst=0)
        rows = np.concatenate(blocks) if rank == 0 else rows
        per = [None] * world
        dist.all_gather_object(per, (round(t_ret * 1e3, 2), int(q_hi - q_lo)))
        t_ret_all, q_rows_all = [p[0] for p in per], [p[1] for p in per]
        t = torch.tensor([t_desc, t_ret], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t_desc, t_ret = float(t[0]), float(t[1])
    else:
        t_ret_all, q_rows_all = [round(t_ret * 1e3, 2)], [int(q_hi - q_lo)]
    if rank == 0:
        import hashlib
        full_host = full.cpu().numpy()
        desc_sha1, rows_sha1 = hashlib.sha1(full_host.tobytes()).hexdigest(), hashlib.sha1(np.ascontiguousarray(rows).tobytes()).hexdigest()
        if args.dump:
            np.savez(args.dump, desc=full_host, rows=rows)
        io.save_pair_dist(out_dir, rows)
        back = np.load(os.path.join(out_dir, "predicted_des_L2_dis.npz"))["arr_0"]
        assert back.shape == ((C - 102) * 50, 1, 3)
        gt_file = os.path.join(ROOT, "tests", "golden", "loop_gt_seq00_0.3overlap_inactive.npz")
        if C == 4541 and os.path.exists(gt_file):
            gt, gt_name = np.load(gt_file, allow_pickle=True)["arr_0"], "reference asset loop_gt_seq00_0.3overlap_inactive.npz (KITTI 00)"
        else:
            gt = np.empty(C, dtype=object)
            for i in range(C):
                rev = frame_cloud(base[:1], i, args.revisit_every)[1]
                gt[i] = np.array([float(i - args.revisit_every)]) if rev else np.array([])
            gt_name = "synthetic revisits (frame i repeats frame i - %d)" % args.revisit_every
        pair = back.reshape(-1, 3)
        top1, top45 = ev.compute_topN(pair, gt, 1), ev.compute_topN(pair, gt, 45)
        P, R = ev.compute_PR_overlap(pair, gt)
        f1, _ = ev.compute_F1(P, R)
        print(json.dumps({"metric": "loop detection over a sequence, end to end", "frames": C, "n_gpus": world,
                          "descriptor_scans_per_s": round(C / t_desc, 1), "descriptor_s": round(t_desc, 3), "inputs": "host [N,4] rows through the ingest leg" if args.host_rows else "resident in HBM",
                          "retrieval_ms_slowest_rank": round(t_ret * 1e3, 2), "retrieval_ms_per_rank": t_ret_all, "query_rows_per_rank": q_rows_all,
                          "wall_s_descriptors_plus_retrieval": round(t_desc + t_ret, 3), "gathered_descriptors_sha1": desc_sha1, "rows_sha1": rows_sha1,
                          "npz_write_s_rank0": round(t_write, 2),
                          "rows": int(pair.shape[0]), "ground_truth": gt_name, "recall_at_1": round(float(top1), 4),
                          "recall_at_45": round(float(top45), 4), "f1_max": round(float(f1), 4), "ap": round(float(ev.compute_AP(P, R)), 4),
                          "auc": round(ev.auc(P, R), 3), "files": "%d x {seq}_{idx}.npz + predicted_des_L2_dis.npz in %s" % (hi - lo, "a temporary directory" if not args.out else out_dir),
                          "data": "synthetic (%d unique scans, a rigid motion per frame); seeded random weights" % args.unique}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not args.out:
        import shutil
        shutil.rmtree(out_dir, ignore_errors=True)


if __name__ == "__main__":
    main()
