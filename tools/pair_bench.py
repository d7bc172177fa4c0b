"""Pairs/s of the full pair model (BASELINE configs[4]-style: encoder over the pair stack + 3D-RoFormer + vote encoder +
matching + LGR pose) on one GPU, with a per-kernel-family time breakdown from HIP events.  Not the headline metric."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from lcrnet_amd.config import make_cfg
    from lcrnet_amd.data import precompute_batch
    from lcrnet_amd.model_family import LCRNet
    from lcrnet_amd.weights import seeded_state_dict
    import lcrnet_amd.synthetic as synthetic
    from lcrnet_amd.data import voxelize_raw_scans
    dev = torch.device("cuda", 0)
    limits = [74, 68, 70, 67]
    cfg = make_cfg()
    cfg["neighbor_limits"] = limits
    m = LCRNet(cfg).eval()
    m.load_state_dict(seeded_state_dict(m.state_dict(), 7351))
    m = m.to(dev)
    gold = os.path.join(ROOT, "tests", "golden", "scans")
    if os.path.exists(os.path.join(gold, "003854.npy")):
        a, b = np.load(os.path.join(gold, "003854.npy")), np.load(os.path.join(gold, "000958.npy"))
    else:
        a, b = synthetic.synthetic_scan(0), synthetic.synthetic_scan(1)
    pts = torch.from_numpy(np.concatenate([a, b])).to(dev)
    lens = torch.tensor([len(a), len(b)], dtype=torch.int64, device=dev)

    def one():
        dd = precompute_batch(pts, lens, 4, 0.3, 1.275, limits, upsampling=True)
        del dd["segment_lengths"]            # pair semantics of the reference: GroupNorm statistics over BOTH clouds
        dd["features"] = torch.ones(pts.shape[0], 1, device=dev)
        dd["lengths_c_host"] = dd["lengths_host"][-1]
        with torch.no_grad():
            return m(dd)

    for _ in range(3):
        out = one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        out = one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    from lcrnet_amd.pipeline import PairPipeline
    pp = PairPipeline(m, neighbor_limits=limits, workers=2)
    list(pp.run([(pts, lens)] * 4))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    n2 = 40
    for _ in pp.run([(pts, lens)] * n2):
        pass
    torch.cuda.synchronize()
    dt2 = (time.perf_counter() - t1) / n2
    print(f"PairPipeline, 2 pairs in flight: {dt2*1e3:.2f} ms/pair = {1/dt2:.1f} pairs/s")
    print(f"pair model end to end: {dt*1e3:.2f} ms/pair = {1/dt:.1f} pairs/s; nodes {out['length'].tolist()}, "
          f"node corr {out['pos_node_corr_indices'].shape[0]}, point corr {out['corr_scores'].shape[0]}")
    print("estimated_transform\n", out["estimated_transform"].cpu().numpy())
    gp = os.path.join(ROOT, "tests", "golden", "pose_golden.npz")
    if os.path.exists(gp):
        Tw = np.load(gp)["estimated_transform"]
        T = out["estimated_transform"].cpu().numpy()
        ang = np.degrees(np.arccos(np.clip((np.trace(T[:3, :3].T @ Tw[:3, :3]) - 1) / 2, -1, 1)))
        print(f"vs reference golden: rotation diff {ang:.4f} deg, translation diff {np.linalg.norm(T[:3,3]-Tw[:3,3]):.4f} m")


if __name__ == "__main__":
    main()
