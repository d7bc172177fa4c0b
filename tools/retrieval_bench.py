#!/usr/bin/env python
"""Descriptor retrieval (a-9) at BASELINE configs[2] / configs[3] sizes on one GPU: C = 4541 (KITTI 00) and C = 23 201 (KITTI 00-10)
unit-norm 256-D descriptors, every query frame 101..C-2 against frames [0, i-100), k = 50; one JSON line.

Phases timed with HIP events: the whole `lcr_retrieval_topk` call (row norms + Q·D^T on the fp32 MFMA GEMM + mask + per-row top-k).
Flops 2·Q·C·256 (the GEMM); the reference does this with a faiss index rebuilt per query on the CPU
(eval_loop_detection_overlap_dataset.py:183-214).  With --shards N the queries are split into N contiguous ranges the way
`distributed_retrieval` gives them to N ranks, and the slowest shard is reported (what one rank of an N-GPU job would do)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", type=int, nargs="+", default=[4541, 23201])
    ap.add_argument("--shards", type=int, nargs="+", default=[1, 8])
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    from lcrnet_amd.retrieval import retrieval_topk, shard_range
    out = {}
    for C in args.sizes:
        g = torch.Generator().manual_seed(C)
        d = torch.nn.functional.normalize(torch.randn(C, 256, generator=g), dim=1).cuda()
        for n in args.shards:
            worst = 0.0
            for r in range(n):
                lo, hi = shard_range(C, n, r)
                q_lo, q_hi = max(lo, 101), min(hi, C - 1)
                if q_hi <= q_lo:
                    continue
                for _ in range(2):
                    retrieval_topk(d[q_lo:q_hi], q_lo, d, 50, 100)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(args.reps):
                    idx, d2 = retrieval_topk(d[q_lo:q_hi], q_lo, d, 50, 100)
                e1.record()
                torch.cuda.synchronize()
                worst = max(worst, e0.elapsed_time(e1) / args.reps)
            Q = C - 102
            out["C=%d,shards=%d" % (C, n)] = {"ms_slowest_shard": round(worst, 3), "queries_per_s": round((Q / n) / worst * 1e3, 1),
                                              "gemm_gflop_whole_job": round(2.0 * Q * C * 256 / 1e9, 2)}
    print(json.dumps({"metric": "descriptor retrieval, masked exhaustive squared-L2 top-50", "unit": "ms", "results": out}))


if __name__ == "__main__":
    main()
