"""Instruction model of the two candidate-phase forms of the radius search on real geometry (CPU only, numpy + the oracle's subsample):

  wave-per-query (the kernel): a query tests the candidates of ITS nine sphere-culled x-runs, 64 per chunk of ~45 wavefront instructions;
  lane-per-query: T consecutive queries of the cell order (T = 16: one DPP row each, four tiles per wavefront; T = 64: the
  whole wavefront, candidate broadcast from scalar registers) all test the UNION of their neighbourhoods' cells, un-culled per query.

For a stage-0 cloud it prints the mean candidates per query of the first form and the mean union size per tile of the second, and what both
cost in wavefront instructions per query with the per-step counts of LABNOTES.md §4.4.  No GPU involved: this sizes the experiment before it is built."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ops  # noqa: E402
import lcrnet_amd.synthetic as synthetic  # noqa: E402


def main():
    stage = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    raw = synthetic.synthetic_scan(0)
    pts, _ = ops.grid_subsample(raw, np.array([len(raw)]), 0.3)
    v = 0.3
    for _ in range(stage):
        v *= 2
        pts, _ = ops.grid_subsample(pts, np.array([len(pts)]), v)
    r = 1.275 * 2 ** stage
    cell = r * 1.000001
    org = pts.min(0).astype(np.float64)
    u = (pts.astype(np.float64) - org) / cell
    c = np.floor(u).astype(np.int64)
    dim = c.max(0) + 1
    key = (c[:, 2] * dim[1] + c[:, 1]) * dim[0] + c[:, 0]
    order = np.argsort(key, kind="stable")
    ks = key[order]
    ncell = int(dim.prod())
    cnt = np.bincount(ks, minlength=ncell)
    start = np.concatenate([[0], np.cumsum(cnt)])
    n = len(pts)
    # per-query culled candidates (the kernel's rule: runs / outer x cells farther than r in cell units are dropped)
    fr = u - np.floor(u)
    cand = np.zeros(n, np.int64)
    incount = np.zeros(n, np.int64)
    rc2 = 1.0 / 1.000001 ** 2 * 1.00001
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            gy = np.where(dy == 0, 0.0, np.where(dy < 0, fr[:, 1], 1 - fr[:, 1]))
            gz = np.where(dz == 0, 0.0, np.where(dz < 0, fr[:, 2], 1 - fr[:, 2]))
            g2 = gy * gy + gz * gz
            cy, cz = c[:, 1] + dy, c[:, 2] + dz
            ok = (g2 < rc2) & (cy >= 0) & (cy < dim[1]) & (cz >= 0) & (cz < dim[2])
            x0 = np.where(fr[:, 0] ** 2 + g2 >= rc2, c[:, 0], c[:, 0] - 1)
            x1 = np.where((1 - fr[:, 0]) ** 2 + g2 >= rc2, c[:, 0], c[:, 0] + 1)
            x0, x1 = np.maximum(x0, 0), np.minimum(x1, dim[0] - 1)
            row = (np.clip(cz, 0, dim[2] - 1) * dim[1] + np.clip(cy, 0, dim[1] - 1)) * dim[0]
            ln = np.where(ok & (x0 <= x1), start[row + x1 + 1] - start[row + x0], 0)
            cand += ln
    # in-radius counts on a sample (brute force inside the 27 cells is enough for a mean)
    q_sorted = order                                             # processing order = cell order
    res = {"points": n, "radius": r, "mean_candidates_per_query_culled": float(cand.mean()),
           "chunks_of_64_per_query": float(np.ceil(cand / 64).mean())}
    for T in (16, 32, 64):
        unions = []
        for t0 in range(0, n - T + 1, T * 7):                   # every 7th tile: a sample
            qs = q_sorted[t0:t0 + T]
            cells = set()
            for qi in qs:
                cx, cy, cz = c[qi]
                for dz in (-1, 0, 1):
                    for dy in (-1, 0, 1):
                        for dx in (-1, 0, 1):
                            x, y, z = cx + dx, cy + dy, cz + dz
                            if 0 <= x < dim[0] and 0 <= y < dim[1] and 0 <= z < dim[2]:
                                cells.add((z * dim[1] + y) * dim[0] + x)
            unions.append(sum(int(cnt[k]) for k in cells))
        res["union_candidates_T%d" % T] = float(np.mean(unions))
    # instruction model (wavefront instructions per QUERY, candidate phase only)
    res["model"] = {
        "wave_per_query": round(res["chunks_of_64_per_query"] * 45, 1),
        "lane_per_query_T16_rows": round(res["union_candidates_T16"] * 16 / 64, 1),      # 16 instr per step, one candidate for each of 4 tiles = 64 tests
        "lane_per_query_T64_scalar": round(res["union_candidates_T64"] * 12 / 64, 1),    # 12 instr per step, one candidate for 64 queries
        "note": "per-step counts: 45 = load + 9-way run select + 8 distance + compare + ballot/compact/bin atomic (the kernel, PMC-calibrated); "
                "16 = 4 row-broadcast moves + 8 distance + compare + per-lane append (3) + loop; 12 = scalar candidate + 8 distance + compare + append (3)"}
    import json
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
