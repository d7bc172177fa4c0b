#!/usr/bin/env python
"""Micro-benchmark of lcr_kpconv_aggregate / lcr_radius_query on the bench workload's real neighbourhoods (batch of 8 synthetic
scans): µs per launch and achieved algorithmic GB/s (idx + xyz + support rows + A output), HIP events.
    python tools/agg_bench.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from lcrnet_amd import functional as F
    from lcrnet_amd.data import precompute_batch, voxelize_raw_scans
    from lcrnet_amd.weights import base_kernel_points
    dev = torch.device("cuda:0")
    scans = bench.make_batch(0)
    pts = torch.from_numpy(np.concatenate(scans)).to(dev)
    lens = torch.tensor([len(s) for s in scans], dtype=torch.int64, device=dev)
    p, l, _ = voxelize_raw_scans(pts, lens, bench.VOXEL)
    dd = precompute_batch(p.contiguous(), l, bench.NUM_STAGES, bench.VOXEL, bench.RADIUS, bench.LIMITS, upsampling=False)
    kp = base_kernel_points()
    cases = [(0, 32), (1, 32), (1, 64), (2, 64), (2, 128), (3, 128), (3, 256)]   # (stage, C_mid) pairs the encoder runs
    ref_out = {}
    for stage, C in cases:
        q = dd["points"][stage]
        idx = dd["neighbors"][stage]
        order = dd["order"][stage] if "order" in dd else None
        M, H = idx.shape
        feats = torch.randn(M, C, device=dev)
        pos = F.row_positive(feats)
        sigma = bench.VOXEL * 2 ** stage * 2.0
        kpts = kp * (bench.VOXEL * 2 ** stage * 2.5)
        for _ in range(3):
            A, nn = F.kpconv_aggregate(feats, pos, q, q, idx, kpts, sigma, order=order)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        reps = 20
        for _ in range(reps):
            A, nn = F.kpconv_aggregate(feats, pos, q, q, idx, kpts, sigma, order=order)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps * 1e-3
        byts = M * H * idx.element_size() + 2 * M * 12 + M * C * 4 + M * 15 * C * 4
        valid = float((idx < M).sum()) / M
        print("stage %d C=%3d M=%6d H=%d (%.1f valid)  %8.1f us   %7.1f GB/s algorithmic   checksum %.6e" %
              (stage, C, M, H, valid, t * 1e6, byts / t / 1e9, float(A.double().abs().sum())))


if __name__ == "__main__":
    main()
