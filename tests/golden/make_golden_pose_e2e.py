"""A STABLE end-to-end pose case from the imported reference `LCRNet.forward` (build container only).

    python tests/golden/make_golden_pose_e2e.py

The demo pair's pose under seeded random weights is a consensus over near-uniform matches — the reference's own `estimated_transform`
moves under one fp32 rounding of its inputs, so tests could only hold it to 2 degrees / 0.5 m.  Here the second cloud is a
PLANTED rigid motion of the first (demo scan 003854 rotated by 3 degrees about z, moved by (1.6, -0.9, 0.12) m, 5 mm Gaussian noise, 12 % of
the points dropped): corresponding neighbourhoods look alike, so even random-weight features match them and the local-to-global registration
recovers the planted motion.  The reference runs three times — on the inputs and on two copies whose coordinates are perturbed by a relative
2^-23 (one fp32 rounding) — and the fixture records how far ITS outputs move: what a 1e-4 comparison can be held to.

Same stubs / seeded weights / `.cuda()` patching as make_golden_pose.py.  Output: tests/golden/pose_e2e_golden.npz — the second cloud
(float32, as fed), the planted transform, the reference's estimated_transform, correspondences (points + scores), node correspondences,
and the perturbation spread of each."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_model as mgm  # noqa: E402

SEED_MODEL = 7351


def planted_pair():
    a = np.load(os.path.join(HERE, "scans", "003854.npy")).astype(np.float32)
    rng = np.random.default_rng(20250929)
    ang = np.radians(3.0)
    R = np.array([[np.cos(ang), -np.sin(ang), 0.0], [np.sin(ang), np.cos(ang), 0.0], [0.0, 0.0, 1.0]])
    t = np.array([1.6, -0.9, 0.12])
    keep = rng.random(len(a)) > 0.12
    b = a[keep].astype(np.float64) @ R.T + t + rng.normal(0.0, 0.005, (int(keep.sum()), 3))
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    return a, np.ascontiguousarray(b.astype(np.float32)), T


def main():
    mgm.install_stubs()
    sys.path.insert(0, mgm.REF)
    mgm.install_ref_ext()
    torch.Tensor.cuda = lambda self, *a, **k: self.contiguous()
    torch.nn.Module.cuda = lambda self, *a, **k: self
    from lcrnet_amd.weights import seeded_state_dict
    from experiments.lcrnet.config_model import make_cfg
    from experiments.lcrnet.data import precompute_data_stack_mode
    from experiments.lcrnet.model_family.LCRNet import LCRNet

    cfg = make_cfg()
    cfg.neighbor_limits = mgm.LIMITS
    cfg.vis = False
    full = LCRNet(cfg).eval()
    full.load_state_dict(seeded_state_dict(full.state_dict(), SEED_MODEL), strict=True)
    a, b, T_planted = planted_pair()

    def run(a_, b_):
        pts = torch.from_numpy(np.concatenate([a_, b_]))
        dd = precompute_data_stack_mode(pts, torch.LongTensor([len(a_), len(b_)]), 4, 0.3, 1.275, mgm.LIMITS)
        dd = {k: [t.contiguous() for t in v] for k, v in dd.items()}
        dd["features"] = torch.ones(len(pts), 1)
        dd["batch_size"] = 1
        with torch.no_grad():
            out = full(dd)
        return {k: (v[0] if isinstance(v, tuple) else v) for k, v in out.items()}

    out = run(a, b)
    for k, v in out.items():
        if torch.is_tensor(v):
            print(k, tuple(v.shape), v.dtype)
    T = out["estimated_transform"].numpy().astype(np.float64)
    print("planted\n", T_planted, "\nreference estimated_transform\n", T)
    # which way does the reference's transform map?  (pos -> anc or anc -> pos: record both residuals)
    d_fwd = np.abs(T - T_planted).max()
    d_inv = np.abs(T - np.linalg.inv(T_planted)).max()
    print("max |T - planted| %.3e, max |T - planted^-1| %.3e" % (d_fwd, d_inv))
    # stability of the REFERENCE's own outputs under one fp32 rounding of the inputs
    spreads = {"estimated_transform": 0.0, "num_corr": [int(out["corr_scores"].shape[0])], "num_node_corr": [int(out["pos_node_corr_indices"].shape[0])]}
    rng = np.random.default_rng(7)
    others = []
    for rep in range(2):
        ja = (a.astype(np.float64) * (1.0 + rng.choice([-1.0, 0.0, 1.0], a.shape) * 2.0 ** -23)).astype(np.float32)
        jb = (b.astype(np.float64) * (1.0 + rng.choice([-1.0, 0.0, 1.0], b.shape) * 2.0 ** -23)).astype(np.float32)
        o2 = run(ja, jb)
        others.append(o2)
        spreads["estimated_transform"] = max(spreads["estimated_transform"], float((o2["estimated_transform"] - out["estimated_transform"]).abs().max()))
        spreads["num_corr"].append(int(o2["corr_scores"].shape[0]))
        spreads["num_node_corr"].append(int(o2["pos_node_corr_indices"].shape[0]))
    print("reference under one-rounding input jitter:", spreads)

    def pairs(o):
        return set(zip(o["pos_node_corr_indices"].tolist(), o["anc_node_corr_indices"].tolist()))

    def corr_set(o):
        p = np.concatenate([o["pos_corr_points"].numpy(), o["anc_corr_points"].numpy()], axis=1)
        return set(map(tuple, np.round(p.astype(np.float64), 3)))

    base_pairs, base_corr = pairs(out), corr_set(out)
    node_sym = max(len(base_pairs ^ pairs(o)) for o in others)
    corr_sym = max(len(base_corr ^ corr_set(o)) for o in others)
    print("node pairs %d (sym. diff under jitter %d), correspondences %d (sym. diff under jitter %d)" % (len(base_pairs), node_sym, len(base_corr), corr_sym))
    store = {
        "cloud_b": b, "planted_transform": T_planted, "estimated_transform": out["estimated_transform"].numpy(),
        "pos_corr_points": out["pos_corr_points"].numpy(), "anc_corr_points": out["anc_corr_points"].numpy(), "corr_scores": out["corr_scores"].numpy(),
        "pos_node_corr_indices": out["pos_node_corr_indices"].numpy().astype(np.int32),
        "anc_node_corr_indices": out["anc_node_corr_indices"].numpy().astype(np.int32),
        "node_corr_scores": out["node_corr_scores"].numpy() if "node_corr_scores" in out else np.zeros(0, np.float32),
        "length": np.asarray(out["length"]) if "length" in out else np.zeros(0),
        "jitter_transform_spread": np.float64(spreads["estimated_transform"]),
        "jitter_num_corr": np.array(spreads["num_corr"]), "jitter_num_node_corr": np.array(spreads["num_node_corr"]),
        "jitter_node_pairs_symdiff": np.int64(node_sym), "jitter_corr_symdiff": np.int64(corr_sym),
        "model_seed": np.int64(SEED_MODEL), "limits": np.array(mgm.LIMITS),
    }
    # ---- the two registration-model entry points on the planted pair (tests/test_matching_models_gpu.py)
    from experiments.lcrnet.model_family.LCRNet_Matching import LCRNet_Matching as RefEval
    from experiments.lcrnet.model_family.LCRNet_Matching_infer import LCRNet_Matching as RefInfer
    T_gt = np.linalg.inv(T_planted).astype(np.float32)               # the direction `estimated_transform` has (see the residuals above)
    pts = torch.from_numpy(np.concatenate([a, b]))
    dd = precompute_data_stack_mode(pts, torch.LongTensor([len(a), len(b)]), 4, 0.3, 1.275, mgm.LIMITS)
    dd = {k: [t.contiguous() for t in v] for k, v in dd.items()}
    dd["features"] = torch.ones(len(pts), 1)
    dd["batch_size"] = 1
    dd["transform"] = torch.from_numpy(T_gt)
    store["transform_gt"] = T_gt
    for tag, cls in (("eval", RefEval), ("infer", RefInfer)):
        model = cls(cfg).eval()
        model.load_state_dict(seeded_state_dict(model.state_dict(), SEED_MODEL), strict=True)
        with torch.no_grad():
            o = model(dd)
        store[tag + "_estimated_transform"] = o["estimated_transform"].numpy()
        store[tag + "_num_corr"] = np.array(o["corr_scores"].shape[0])
        store[tag + "_node_corr"] = np.stack([o["pos_node_corr_indices"].numpy(), o["anc_node_corr_indices"].numpy()], 1).astype(np.int32)
        store[tag + "_corr_scores"] = o["corr_scores"].numpy()
        store[tag + "_pos_corr_points"], store[tag + "_anc_corr_points"] = o["pos_corr_points"].numpy(), o["anc_corr_points"].numpy()
        if tag == "eval":
            store["eval_node_matching_scores"] = o["node_matching_scores"].numpy()
            store["eval_pos_node_masks"], store["eval_anc_node_masks"] = o["pos_node_masks"].numpy(), o["anc_node_masks"].numpy()
            store["eval_score"] = o["score"].numpy()
        print(tag, "planted pair: max |T - T(LCRNet)| %.3e, correspondences %d" %
              (float((o["estimated_transform"] - out["estimated_transform"]).abs().max()), o["corr_scores"].shape[0]))
    # ---- how far the REFERENCE's own pose of the DEMO pair (random weights: a consensus over near-uniform matches) moves under the same
    # one-rounding jitter: the yardstick for the demo-pair pose checks that stay in the tests
    da = np.load(os.path.join(HERE, "scans", "003854.npy")).astype(np.float32)
    db = np.load(os.path.join(HERE, "scans", "000958.npy")).astype(np.float32)
    base = run(da, db)["estimated_transform"].numpy().astype(np.float64)
    worst_deg = worst_m = 0.0
    for rep in range(3):
        ja = (da.astype(np.float64) * (1.0 + rng.choice([-1.0, 0.0, 1.0], da.shape) * 2.0 ** -23)).astype(np.float32)
        jb = (db.astype(np.float64) * (1.0 + rng.choice([-1.0, 0.0, 1.0], db.shape) * 2.0 ** -23)).astype(np.float32)
        Tj = run(ja, jb)["estimated_transform"].numpy().astype(np.float64)
        # small-angle form: acos((tr - 1) / 2) of fp32 matrices has a floor of ~0.03 degrees (sqrt of the entries' rounding)
        worst_deg = max(worst_deg, float(np.degrees(np.linalg.norm(Tj[:3, :3].T @ base[:3, :3] - np.eye(3)) / np.sqrt(2.0))))
        worst_m = max(worst_m, float(np.linalg.norm(Tj[:3, 3] - base[:3, 3])))
        print("demo pair, jitter %d: the reference's own pose moves by %.3e deg / %.3e m (max entry %.3e)" % (rep, worst_deg, worst_m, np.abs(Tj - base).max()))
    store["demo_pair_reference_jitter_deg"], store["demo_pair_reference_jitter_m"] = np.float64(worst_deg), np.float64(worst_m)
    np.savez_compressed(os.path.join(HERE, "pose_e2e_golden.npz"), **store)
    print("size MB", os.path.getsize(os.path.join(HERE, "pose_e2e_golden.npz")) / 1e6)


if __name__ == "__main__":
    main()
