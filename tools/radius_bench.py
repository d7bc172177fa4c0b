#!/usr/bin/env python
"""Micro-benchmark of the 10 radius searches of one bench batch (8 synthetic scans): µs per query launch, HIP events.
    python tools/radius_bench.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


REPS = int(os.environ.get("LCR_RB_REPS", "20"))


def main():
    import bench
    from lcrnet_amd.data import voxelize_raw_scans
    from lcrnet_amd.modules.ops import grid_subsample_device, SupportGrid
    dev = torch.device("cuda:0")
    scans = bench.make_batch(0)
    pts = torch.from_numpy(np.concatenate(scans)).to(dev)
    lens = torch.tensor([len(s) for s in scans], dtype=torch.int64, device=dev)
    p, l, lh = voxelize_raw_scans(pts, lens, bench.VOXEL)
    P, L = [p.contiguous()], [l]
    v = bench.VOXEL
    for i in range(1, bench.NUM_STAGES):
        v *= 2
        q, ql, _ = grid_subsample_device(P[-1], L[-1], v)
        n = int(ql.sum())
        P.append(q[:n].contiguous())
        L.append(ql)
    r = bench.RADIUS
    grids = []
    for i in range(bench.NUM_STAGES):
        grids.append(SupportGrid(P[i], L[i], r))
        r *= 2
    total = 0.0

    def timed(tag, fn):
        nonlocal total
        for _ in range(3):
            out = fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(REPS):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / REPS * 1e3
        total += t
        nq, H = out.shape
        print("%-16s nq=%7d limit=%3d  %8.1f us  %6.1f GB/s out   sum %d" % (tag, nq, H, t, nq * H * 4 / t / 1e3, int(out.long().sum())))

    for i in range(bench.NUM_STAGES):
        timed("neighbors[%d]" % i, lambda: grids[i].query(P[i], L[i], bench.LIMITS[i]))
        if i < bench.NUM_STAGES - 1:
            timed("subsampling[%d]" % i, lambda: grids[i].query(P[i + 1], L[i + 1], bench.LIMITS[i]))
            timed("upsampling[%d]" % i, lambda: grids[i + 1].query(P[i], L[i], bench.LIMITS[i + 1]))
    print("total %.1f us" % total)
    # the production configuration (csrc/precompute.hip): every search walks its queries in the QUERY set's own cell order (q_order)
    total = 0.0
    orders = [g.order() for g in grids]
    outs = {}

    def timed_o(tag, fn):
        timed(tag, fn)
        outs[tag] = fn()
    for i in range(bench.NUM_STAGES):
        timed_o("ordered neighbors[%d]" % i, lambda: grids[i].query(P[i], L[i], bench.LIMITS[i], q_order=orders[i]))
        if i < bench.NUM_STAGES - 1:
            timed_o("ordered subsampling[%d]" % i, lambda: grids[i].query(P[i + 1], L[i + 1], bench.LIMITS[i], q_order=orders[i + 1]))
            timed_o("ordered upsampling[%d]" % i, lambda: grids[i + 1].query(P[i], L[i], bench.LIMITS[i + 1], q_order=orders[i]))
    print("total with q_order (production) %.1f us" % total)
    if os.environ.get("LCR_RB_CHECK"):
        # rows against the un-ordered call (always the wave form) — bit-identical whatever kernel served the ordered call
        bad = 0
        for i in range(bench.NUM_STAGES):
            ref = {"ordered neighbors[%d]" % i: lambda: grids[i].query(P[i], L[i], bench.LIMITS[i])}
            if i < bench.NUM_STAGES - 1:
                ref["ordered subsampling[%d]" % i] = lambda: grids[i].query(P[i + 1], L[i + 1], bench.LIMITS[i])
                ref["ordered upsampling[%d]" % i] = lambda: grids[i + 1].query(P[i], L[i], bench.LIMITS[i + 1])
            for tag, fn in ref.items():
                want = fn()
                same = torch.equal(want, outs[tag])
                nbad = int((want != outs[tag]).any(dim=1).sum())
                print("check %-24s rows equal: %s (%d of %d rows differ)" % (tag, same, nbad, want.shape[0]))
                bad += nbad
        print("CHECK", "OK" if bad == 0 else "FAILED (%d rows)" % bad)
    if os.environ.get("LCR_RB_NO_CELL_ORDER"):
        return
    # the same searches with the QUERIES permuted into their own grid's cell order (spatially coherent wavefronts)
    total = 0.0
    Pc = [P[i][grids[i].order().long()].contiguous() for i in range(bench.NUM_STAGES)]
    for i in range(bench.NUM_STAGES):
        timed("cell-order neighbors[%d]" % i, lambda: grids[i].query(Pc[i], L[i], bench.LIMITS[i]))
        if i < bench.NUM_STAGES - 1:
            timed("cell-order subsampling[%d]" % i, lambda: grids[i].query(Pc[i + 1], L[i + 1], bench.LIMITS[i]))
            timed("cell-order upsampling[%d]" % i, lambda: grids[i + 1].query(Pc[i], L[i], bench.LIMITS[i + 1]))
    print("total with cell-ordered queries %.1f us" % total)


if __name__ == "__main__":
    main()
