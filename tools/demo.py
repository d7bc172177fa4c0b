#!/usr/bin/env python
"""The reference's demo (demo/demo.py; BASELINE configs[0]) on the HIP path: two scans in, descriptor distance + relative pose out,
printed in the reference's format (demo.py:76-81) and appended to `lcr_output` like `python3 demo/demo.py` does.

    python tools/demo.py [--pos 003854] [--anc 000958] [--weights weights/best-model-mixed.tar] [--out-dir .]

Scans: `<name>.npy` (xyz in columns 0-2, already 0.3 m-voxelised like demo/data_demo/*.npy) looked up in --data-dir (default: the
committed fixtures tests/golden/scans).  Neighbour limits are calibrated on the pair like the demo's data loader
(`calibrate_neighbors_stack_mode`, data.py:408-433).  Without --weights the seeded random weights of the test-suite are used (the
numbers then mean nothing); with the trained checkpoint the README's known answer is `L2 feature distance: 0.809192`."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pos", default="003854")
    ap.add_argument("--anc", default="000958")
    ap.add_argument("--data-dir", default=os.path.join(ROOT, "tests", "golden", "scans"))
    ap.add_argument("--weights", default=os.path.join(ROOT, "weights", "best-model-mixed.tar"))
    ap.add_argument("--out-dir", default=".")
    args = ap.parse_args()
    from lcrnet_amd import io_formats as io
    from lcrnet_amd.config import make_cfg
    from lcrnet_amd.data import calibrate_neighbors_stack_mode, precompute_data_stack_mode
    from lcrnet_amd.model_family import LCRNet
    from lcrnet_amd.weights import load_snapshot, seeded_state_dict
    dev = torch.device("cuda", 0)
    a = torch.from_numpy(np.load(os.path.join(args.data_dir, args.pos + ".npy"))[:, :3].astype(np.float32)).to(dev)
    b = torch.from_numpy(np.load(os.path.join(args.data_dir, args.anc + ".npy"))[:, :3].astype(np.float32)).to(dev)
    pts = torch.cat([a, b]).contiguous()
    lens = torch.tensor([len(a), len(b)], dtype=torch.int64, device=dev)
    limits = [int(x) for x in calibrate_neighbors_stack_mode([(pts, lens)], 4, 0.3, 1.275)]      # one dataset item = the pair
    print("Calibrate neighbors: %s." % limits)
    cfg = make_cfg()
    cfg["neighbor_limits"] = limits
    model = LCRNet(cfg).eval()
    if os.path.isfile(args.weights):
        load_snapshot(model, args.weights)
    else:
        print("(no checkpoint at %s: seeded random weights)" % args.weights)
        model.load_state_dict(seeded_state_dict(model.state_dict(), 7351))
    model = model.to(dev)
    dd = precompute_data_stack_mode(pts, lens, 4, 0.3, 1.275, limits)
    dd["features"] = torch.ones(len(pts), 1, device=dev)
    with torch.no_grad():
        out = model(dd)
    T = out["estimated_transform"].cpu().numpy()
    pg, ag = out["pos_feature_global"].cpu().numpy(), out["anc_feature_global"].cpu().numpy()
    feat_dis = np.sqrt(np.sum((pg - ag) ** 2))
    print("Test pos_idx: %i and anc_idx: %i\nL2 feature distance: %f\nEstimated transformation:\n%s" % (int(args.pos), int(args.anc), feat_dis, T))
    print("nCorr: %d" % out["corr_scores"].shape[0])
    with open(os.path.join(args.out_dir, "lcr_output"), "a") as f:
        f.write(io.lcr_output_line(int(args.pos), int(args.anc), pg, ag, T))


if __name__ == "__main__":
    main()
