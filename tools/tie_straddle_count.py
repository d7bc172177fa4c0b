#!/usr/bin/env python
"""How often can a row differ from the reference's AS A SET?  (CPU, oracle only.)

The search emits equal-distance runs in canonical (d², index) order; the reference (nanoflann) orders them by kd-leaf artefact.  Rows are
then equal up to a permutation inside equal-d² runs — unless such a run STRADDLES the limit cut, in which case the two sides may keep
different members of the run.  This tool counts those rows on the bench's synthetic scans and on the committed demo scans, for all ten
searches at the bench's limits: it runs the oracle at limit + 1 and compares d² of columns limit - 1 and limit.

    python tools/tie_straddle_count.py [--scans N] > profiles/rNN_tie_straddle.json"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ops  # noqa: E402
import lcrnet_amd.synthetic as synthetic  # noqa: E402

VOXEL, RADIUS, NUM_STAGES = 0.3, 1.275, 4


def d2(q, s, idx):
    pad = idx >= len(s)
    p = s[np.minimum(idx, len(s) - 1)]
    d = ((q[:, None, 0] - p[..., 0]) ** 2 + (q[:, None, 1] - p[..., 1]) ** 2) + (q[:, None, 2] - p[..., 2]) ** 2      # fp32, the reference's order
    return np.where(pad, np.float32(np.inf), d.astype(np.float32))


def count(points, limits):
    lens = np.array([len(points)])
    st = ops.precompute_data_stack_mode(points, lens, NUM_STAGES, VOXEL, RADIUS, [l + 1 for l in limits])
    P = st["points"]
    out = {}

    def one(name, q, s, idx, limit):
        dd = d2(q, s, idx.astype(np.int64))
        full = np.isfinite(dd[:, limit])                                   # rows with more than `limit` supports in range
        tie = full & (dd[:, limit - 1] == dd[:, limit])
        any_tie = (np.isfinite(dd[:, 1:limit]) & (dd[:, 1:limit] == dd[:, :limit - 1])).any(1)
        out[name] = {"rows": int(len(q)), "rows_over_limit": int(full.sum()), "rows_with_a_tie_inside": int(any_tie.sum()),
                     "rows_with_a_tie_across_the_cut": int(tie.sum())}
    for i in range(NUM_STAGES):
        one("neighbors[%d]" % i, P[i], P[i], st["neighbors"][i], limits[i])
        if i < NUM_STAGES - 1:
            one("subsampling[%d]" % i, P[i + 1], P[i], st["subsampling"][i], limits[i])
            one("upsampling[%d]" % i, P[i], P[i + 1], st["upsampling"][i], limits[i + 1])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=8)
    args = ap.parse_args()
    res = {"tool": "tools/tie_straddle_count.py", "what": "rows whose limit cut splits an equal-d2 run (the only rows that can differ from the reference as a SET)", "cases": {}}
    tot = {"rows": 0, "rows_over_limit": 0, "rows_with_a_tie_inside": 0, "rows_with_a_tie_across_the_cut": 0}
    cases = [("synthetic scan %d, bench limits [64,65,74,80]" % k, ops.grid_subsample(synthetic.synthetic_scan(k), np.array([len(synthetic.synthetic_scan(k))]), VOXEL)[0],
              [64, 65, 74, 80]) for k in range(args.scans)]
    sd = os.path.join(ROOT, "tests", "golden", "scans")
    cases += [("demo scan %s, calibrated limits [74,68,70,67]" % f[:-4], np.load(os.path.join(sd, f)).astype(np.float32), [74, 68, 70, 67]) for f in sorted(os.listdir(sd))]
    for name, pts, limits in cases:
        c = count(np.ascontiguousarray(pts), limits)
        res["cases"][name] = c
        for v in c.values():
            for k in tot:
                tot[k] += v[k]
    res["total"] = tot
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
