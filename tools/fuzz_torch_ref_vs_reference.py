"""CPU, build container only: the floating-point oracle (oracle/torch_ref.py: encoder + NetVLAD head) against the REFERENCE's own Python model
(experiments.lcrnet.model_family.LCRNet_GlobalDescrition, imported from /root/reference exactly as tests/golden/make_golden_model.py does,
seeded weights) on random inputs the six golden scans do not cover: decimated / cropped / rigidly moved demo scans, random seeds of the weights.
Compares the 256-D descriptor and the coarse features.
    python tools/fuzz_torch_ref_vs_reference.py FIRST LAST [--json FILE] [--max-seconds S]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden_model as G  # noqa: E402  (stubs for the reference's third-party imports, utils.ext served by oracle/_ref)

LIMITS, NUM_STAGES, VOXEL, RADIUS = [74, 68, 70, 67], 4, 0.3, 1.275


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("first", type=int)
    ap.add_argument("last", type=int)
    ap.add_argument("--json", default=None)
    ap.add_argument("--max-seconds", type=float, default=0.0)
    args = ap.parse_args()
    if not os.path.isdir(G.REF):
        sys.exit("needs /root/reference")
    G.install_stubs()
    sys.path.insert(0, G.REF)
    G.install_ref_ext()
    from lcrnet_amd.weights import seeded_state_dict
    from experiments.lcrnet.config_model import make_cfg
    from experiments.lcrnet.data import precompute_data_stack_mode
    from experiments.lcrnet.model_family.LCRNet_GlobalDescrition import LCRNet_GlobalDescrition
    from oracle import ops as oracle_ops
    from oracle import torch_ref

    cfg = make_cfg()
    cfg.neighbor_limits = LIMITS
    cfg.vis = False
    gd = LCRNet_GlobalDescrition(cfg).eval()
    trace = {}
    gd.encoder.encoder4_3.register_forward_hook(lambda m, i, o: trace.__setitem__("c", o.detach()))
    scans_dir = os.path.join(ROOT, "tests", "golden", "scans")
    scans = [np.load(os.path.join(scans_dir, f)) for f in sorted(os.listdir(scans_dir))]
    t0, bad, n, seed, worst_d, worst_f = time.time(), [], 0, args.first - 1, 0.0, 0.0
    for seed in range(args.first, args.last):
        rng = np.random.default_rng(4200 + seed)
        sd = seeded_state_dict(gd.state_dict(), int(rng.integers(1, 1 << 30)))
        gd.load_state_dict(sd, strict=True)
        xyz = scans[int(rng.integers(0, len(scans)))]
        xyz = xyz[:: int(rng.integers(1, 5))]
        if rng.random() < 0.4:                                         # a crop: one half-space through the sensor
            nrm = rng.standard_normal(3)
            xyz = xyz[xyz @ nrm > 0]
        a = rng.uniform(0, 2 * np.pi)
        R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], dtype=np.float32)
        xyz = np.ascontiguousarray((xyz @ R.T + rng.normal(0, [50, 50, 3]).astype(np.float32)).astype(np.float32))
        if len(xyz) < 2000:
            continue
        with torch.no_grad():
            pts = torch.from_numpy(xyz)
            dd = precompute_data_stack_mode(pts, torch.LongTensor([len(xyz)]), NUM_STAGES, VOXEL, RADIUS, LIMITS)
            dd = {k: [t.contiguous() for t in v] for k, v in dd.items()}
            dd["features"] = torch.ones(len(xyz), 1)
            dd["batch_size"] = 1
            want = gd(dd)["anc_global"]
            want_c = trace["c"]
            st = oracle_ops.precompute_data_stack_mode(xyz, np.array([len(xyz)], dtype=np.int64), NUM_STAGES, VOXEL, RADIUS, LIMITS)
            od = {k: [torch.from_numpy(np.ascontiguousarray(t)) for t in v] for k, v in st.items()}
            feats = torch_ref.kp_encoder(sd, torch.ones(len(xyz), 1), od)
            got = torch_ref.global_descriptor(sd, feats[-1])
        ed = float((got - want).abs().max())
        ef = float((feats[-1] - want_c).abs().max() / max(1.0, float(want_c.abs().max()))) if feats[-1].shape == want_c.shape else float("inf")
        worst_d, worst_f = max(worst_d, ed), max(worst_f, ef)
        if ed > 1e-5 or ef > 2e-4:
            bad.append({"seed": seed, "points": int(len(xyz)), "descriptor_abs": ed, "coarse_rel": ef})
            print("FAIL", bad[-1], flush=True)
        else:
            n += 1
        if args.max_seconds and time.time() - t0 > args.max_seconds:
            break
    rec = {"tool": "fuzz_torch_ref_vs_reference", "first": args.first, "last_done": seed, "cases_within_tolerance": n, "failures": bad,
           "tolerance": {"descriptor_abs": 1e-5, "coarse_features_rel": 2e-4}, "worst_descriptor_abs": worst_d, "worst_coarse_rel": worst_f,
           "seconds": round(time.time() - t0, 1)}
    print("torch_ref vs imported reference model: " + json.dumps(rec))
    if args.json:
        with open(args.json, "a") as f:
            f.write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
