"""GPU: descriptors of the HIP path (DescriptorPipeline, pre-voxelised input, the stack as ONE scan per step) against the float oracle
(oracle/torch_ref.py on the oracle's neighbour lists) on random inputs: decimated / cropped / rigidly moved demo scans, a fresh weight
seed per case.  Tolerance 1e-4 (north_star), the worst case is reported.
    python tools/fuzz_float_parity_gpu.py FIRST LAST [--json FILE] [--max-seconds S]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ops as oracle_ops  # noqa: E402
from oracle import torch_ref  # noqa: E402
from lcrnet_amd.model_family import create_model  # noqa: E402
from lcrnet_amd.pipeline import DescriptorPipeline  # noqa: E402
from lcrnet_amd.weights import seeded_state_dict  # noqa: E402

LIMITS, NUM_STAGES, VOXEL, RADIUS = [74, 68, 70, 67], 4, 0.3, 1.275
ap = argparse.ArgumentParser()
ap.add_argument("first", type=int)
ap.add_argument("last", type=int)
ap.add_argument("--json", default=None)
ap.add_argument("--max-seconds", type=float, default=0.0)
args = ap.parse_args()
scans_dir = os.path.join(ROOT, "tests", "golden", "scans")
scans = [np.load(os.path.join(scans_dir, f)) for f in sorted(os.listdir(scans_dir))]
model = create_model().eval()
t0, bad, n, seed, worst = time.time(), [], 0, args.first - 1, 0.0
for seed in range(args.first, args.last):
    rng = np.random.default_rng(4200 + seed)
    model = model.cpu()
    model.load_state_dict(seeded_state_dict(model.state_dict(), int(rng.integers(1, 1 << 30))))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.cuda()
    xyz = scans[int(rng.integers(0, len(scans)))]
    xyz = xyz[:: int(rng.integers(1, 5))]
    if rng.random() < 0.4:
        nrm = rng.standard_normal(3)
        xyz = xyz[xyz @ nrm > 0]
    a = rng.uniform(0, 2 * np.pi)
    R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], dtype=np.float32)
    xyz = np.ascontiguousarray((xyz @ R.T + rng.normal(0, [50, 50, 3]).astype(np.float32)).astype(np.float32))
    if len(xyz) < 2000:
        continue
    lens = np.array([len(xyz)], dtype=np.int64)
    with DescriptorPipeline(model, VOXEL, RADIUS, NUM_STAGES, LIMITS, upsampling=False, raw_voxel=None) as pipe:
        got = [d.cpu() for d in pipe.run([(torch.from_numpy(xyz).cuda(), torch.from_numpy(lens).cuda())])][0]
    st = oracle_ops.precompute_data_stack_mode(xyz, lens, NUM_STAGES, VOXEL, RADIUS, LIMITS)
    od = {k: [torch.from_numpy(np.ascontiguousarray(t)) for t in v] for k, v in st.items()}
    with torch.no_grad():
        want = torch_ref.global_descriptor(sd, torch_ref.kp_encoder(sd, torch.ones(len(xyz), 1), od)[-1])
    e = float((got[0] - want[0]).abs().max())
    worst = max(worst, e)
    if e > 1e-4:
        bad.append({"seed": seed, "points": int(len(xyz)), "descriptor_abs": e})
        print("FAIL", bad[-1], flush=True)
    else:
        n += 1
    if args.max_seconds and time.time() - t0 > args.max_seconds:
        break
rec = {"tool": "fuzz_float_parity_gpu", "first": args.first, "last_done": seed, "cases_within_1e-4": n, "failures": bad, "worst_descriptor_abs": worst,
       "seconds": round(time.time() - t0, 1)}
print("float parity fuzz: " + json.dumps(rec))
if args.json:
    with open(args.json, "a") as f:
        f.write(json.dumps(rec) + "\n")
