#!/usr/bin/env python
"""Soak of the multi-worker pair pipeline: for `--seconds` the 45 demo pairs go through PairPipeline with 4 workers at one pair per call and
5 workers at 4 / 16 pairs per call, every output compared with the one-worker result (descriptors 1e-5, coarse node counts equal, output
order = input order).  One JSON line.  (Shared weight tables, derived caches and native sequencers used from several pinned host threads.)"""
import argparse
import itertools
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_blocks as bb  # noqa: E402
from lcrnet_amd.pipeline import PairPipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    m = bb.pair_model(dev)
    work = bb.demo_pairs(dev, 45)
    keep = lambda o: (o["pos_feature_global"].cpu(), o["anc_feature_global"].cpu(), o["length"].tolist(), int(o["corr_scores"].shape[0]))
    with PairPipeline(m, neighbor_limits=bb.PAIR_LIMITS, workers=1, pairs_per_call=1) as one:
        want = [keep(o) for o in one.run(work)]
    t0, passes, pairs, worst, bad = time.time(), 0, 0, 0.0, []
    cfgs = [(4, 1), (5, 4), (5, 16), (8, 1)]
    pipes = [PairPipeline(m, neighbor_limits=bb.PAIR_LIMITS, workers=w, pairs_per_call=P) for w, P in cfgs]
    try:
        while time.time() - t0 < args.seconds:
            for (w, P), pp in zip(cfgs, pipes):
                got = [keep(o) for o in pp.run(work)]
                passes += 1
                pairs += len(got)
                if len(got) != len(want):
                    bad.append({"cfg": [w, P], "what": "count %d" % len(got)})
                    continue
                for i, (g, x) in enumerate(zip(got, want)):
                    e = max(float((g[0] - x[0]).abs().max()), float((g[1] - x[1]).abs().max()))
                    worst = max(worst, e)
                    if e > 1e-5 or g[2] != x[2] or abs(g[3] - x[3]) > 0.05 * x[3]:
                        bad.append({"cfg": [w, P], "pair": i, "err": e, "lengths": [g[2], x[2]], "corr": [g[3], x[3]]})
    finally:
        for pp in pipes:
            pp.close()
    print(json.dumps({"tool": "soak_pairs", "seconds": round(time.time() - t0, 1), "passes": passes, "pairs": pairs, "configs_workers_pairs_per_call": cfgs,
                      "worst_descriptor_abs_diff_vs_one_worker": worst, "failures": bad[:20], "n_failures": len(bad)}))


if __name__ == "__main__":
    main()
