"""GPU twin of tools/fuzz_oracle_vs_ref.py: the same random cases (op-fuzz clouds + planes / lines / points one cell below the voxel origin +
offsets of thousands of metres) through the HIP subsample and search, bit-exact against the oracle (canonical tie order).
    python tools/fuzz_degenerate_gpu.py FIRST LAST [--json FILE] [--max-seconds S]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import ops  # noqa: E402
from fuzz_oracle_vs_ref import make_case  # noqa: E402
from lcrnet_amd.modules.ops import grid_subsample, radius_search  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("first", type=int)
ap.add_argument("last", type=int)
ap.add_argument("--json", default=None)
ap.add_argument("--max-seconds", type=float, default=0.0)
args = ap.parse_args()
dev = torch.device("cuda:0")
t0, bad, n, seed = time.time(), [], 0, args.first - 1
for seed in range(args.first, args.last):
    xyz, lens, voxel, rng = make_case(seed)
    try:
        want_p, want_l = ops.grid_subsample(xyz, lens, voxel)
        got_p, got_l = grid_subsample(torch.from_numpy(xyz).to(dev), torch.from_numpy(lens).to(dev), voxel)
        assert np.array_equal(got_l.cpu().numpy(), want_l), "subsample lengths"
        assert np.array_equal(got_p.cpu().numpy().view(np.uint32), want_p.view(np.uint32)), "subsample points / order"
        radius, limit = voxel * float(rng.choice([2.5, 4.25])), int(rng.integers(8, 70))
        want = ops.radius_search(want_p, xyz, want_l, lens, radius, limit, ref_width=True)
        got = radius_search(got_p.contiguous(), torch.from_numpy(xyz).to(dev), got_l, torch.from_numpy(lens).to(dev), radius, limit)
        assert np.array_equal(got.cpu().numpy(), want), "cross search"
        want = ops.radius_search(want_p, want_p, want_l, want_l, radius, limit, ref_width=True)
        got = radius_search(got_p.contiguous(), got_p.contiguous(), got_l, got_l, radius, limit)
        assert np.array_equal(got.cpu().numpy(), want), "self search"
        n += 1
    except (AssertionError, RuntimeError) as e:
        bad.append({"seed": seed, "error": str(e)[:200]})
        print("FAIL", bad[-1], flush=True)
    if args.max_seconds and time.time() - t0 > args.max_seconds:
        break
rec = {"tool": "fuzz_degenerate_gpu", "first": args.first, "last_done": seed, "cases_exact": n, "failures": bad, "seconds": round(time.time() - t0, 1)}
print("degenerate op fuzz: " + json.dumps(rec))
if args.json:
    with open(args.json, "a") as f:
        f.write(json.dumps(rec) + "\n")
