"""A STABLE end-to-end pose case from the imported reference `LCRNet.forward` (build container only).

    python tests/golden/make_golden_pose_e2e.py

The demo pair's pose under seeded random weights is a consensus over near-uniform matches, so tests held it to 2 degrees / 0.5 m
(VERDICT r4).  Here the second cloud is a PLANTED rigid motion of the first (demo scan 003854), in two cases (planted_pair below):
`shift` — whole-voxel translation, thousands of inlier correspondences, the well-conditioned case the 1e-4 bound is held on — and
`rot3` — a 3 degree rotation, where random-weight features match in a few patches only and the pose rests on ~17 inliers (held to what
that conditioning allows).  Per case the reference runs three times — on the inputs and on two copies whose coordinates are perturbed by a
relative 2^-23 (one fp32 rounding) — and the fixture records how far ITS outputs move, plus the number of inliers behind its pose.

Same stubs / seeded weights / `.cuda()` patching as make_golden_pose.py.  Output: tests/golden/pose_e2e_{shift,rot3}_golden.npz — the second
cloud (float32, as fed), the planted transform, the reference's estimated_transform, correspondences (points + scores), node
correspondences, the same from the two registration-model entry points, and the perturbation spreads; the `shift` file also carries the
jitter spread of the reference's pose on the DEMO pair (the yardstick of the demo-pair checks that stay in the tests)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_model as mgm  # noqa: E402

SEED_MODEL = 7351


def planted_pair(case="rot3"):
    """rot3:  3 degrees about z, (1.6, -0.9, 0.12) m, 5 mm noise, 12 % of the points dropped — random-weight features survive the rotation in
              only a few patches: the reference's pose rests on ~17 inlier correspondences of 4 095 (recorded as *_inliers);
       shift: a pure translation by whole coarse voxels (2.4, -4.8, 0) m with 1 mm noise, nothing dropped — KPConv features depend on
              relative positions only, so corresponding points carry (almost) the same features and thousands of correspondences are inliers:
              the well-conditioned case the 1e-4 end-to-end bound is held on."""
    a = np.load(os.path.join(HERE, "scans", "003854.npy")).astype(np.float32)
    rng = np.random.default_rng(20250929)
    if case == "rot3":
        ang, t, noise, drop = np.radians(3.0), np.array([1.6, -0.9, 0.12]), 0.005, 0.12
    else:
        ang, t, noise, drop = 0.0, np.array([2.4, -4.8, 0.0]), 0.001, 0.0
    R = np.array([[np.cos(ang), -np.sin(ang), 0.0], [np.sin(ang), np.cos(ang), 0.0], [0.0, 0.0, 1.0]])
    keep = rng.random(len(a)) >= drop
    b = a[keep].astype(np.float64) @ R.T + t + rng.normal(0.0, noise, (int(keep.sum()), 3))
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
    from experiments.lcrnet.model_family.LCRNet_Matching import LCRNet_Matching as RefEval
    from experiments.lcrnet.model_family.LCRNet_Matching_infer import LCRNet_Matching as RefInfer

    cfg = make_cfg()
    cfg.neighbor_limits = mgm.LIMITS
    cfg.vis = False
    models = {}
    for tag, cls in (("", LCRNet), ("eval", RefEval), ("infer", RefInfer)):
        m = cls(cfg).eval()
        m.load_state_dict(seeded_state_dict(m.state_dict(), SEED_MODEL), strict=True)
        models[tag] = m

    def run(model, a_, b_, transform=None):
        pts = torch.from_numpy(np.concatenate([a_, b_]))
        dd = precompute_data_stack_mode(pts, torch.LongTensor([len(a_), len(b_)]), 4, 0.3, 1.275, mgm.LIMITS)
        dd = {k: [t.contiguous() for t in v] for k, v in dd.items()}
        dd["features"] = torch.ones(len(pts), 1)
        dd["batch_size"] = 1
        if transform is not None:
            dd["transform"] = torch.from_numpy(transform)
        with torch.no_grad():
            out = model(dd)
        return {k: (v[0] if isinstance(v, tuple) else v) for k, v in out.items()}

    def jitter(x, rng):
        return (x.astype(np.float64) * (1.0 + rng.choice([-1.0, 0.0, 1.0], x.shape) * 2.0 ** -23)).astype(np.float32)

    def inliers(o):
        T = o["estimated_transform"].numpy().astype(np.float64)
        p, q = o["pos_corr_points"].numpy().astype(np.float64), o["anc_corr_points"].numpy().astype(np.float64)
        r1 = np.linalg.norm(p - (q @ T[:3, :3].T + T[:3, 3]), axis=1)
        r2 = np.linalg.norm(q - (p @ T[:3, :3].T + T[:3, 3]), axis=1)
        r = r1 if np.median(r1) < np.median(r2) else r2
        return int((r < 0.45).sum()), float(np.sort(np.abs(r - 0.45))[0])

    for case in ("shift", "rot3"):
        a, b, T_planted = planted_pair(case)
        out = run(models[""], a, b)
        T = out["estimated_transform"].numpy().astype(np.float64)
        n_in, margin = inliers(out)
        print("==== case %s: %d correspondences, %d inliers (closest residual to the 0.45 m threshold: %.3f m)" % (case, out["corr_scores"].shape[0], n_in, margin))
        print("max |T - planted^-1| %.3e   max |T - planted| %.3e" % (np.abs(T - np.linalg.inv(T_planted)).max(), np.abs(T - T_planted).max()))
        rng = np.random.default_rng(7)
        spread, ncs, nns, node_sym = 0.0, [], [], 0
        base_pairs = set(zip(out["pos_node_corr_indices"].tolist(), out["anc_node_corr_indices"].tolist()))
        for rep in range(2):
            o2 = run(models[""], jitter(a, rng), jitter(b, rng))
            spread = max(spread, float((o2["estimated_transform"] - out["estimated_transform"]).abs().max()))
            ncs.append(int(o2["corr_scores"].shape[0]))
            nns.append(int(o2["pos_node_corr_indices"].shape[0]))
            node_sym = max(node_sym, len(base_pairs ^ set(zip(o2["pos_node_corr_indices"].tolist(), o2["anc_node_corr_indices"].tolist()))))
        print("reference under one-rounding input jitter: transform moves %.3e, correspondences %s, node pairs %s (sym. diff %d)" % (spread, ncs, nns, node_sym))
        store = {
            "cloud_b": b, "planted_transform": T_planted, "estimated_transform": out["estimated_transform"].numpy(),
            "pos_corr_points": out["pos_corr_points"].numpy(), "anc_corr_points": out["anc_corr_points"].numpy(), "corr_scores": out["corr_scores"].numpy(),
            "pos_node_corr_indices": out["pos_node_corr_indices"].numpy().astype(np.int32),
            "anc_node_corr_indices": out["anc_node_corr_indices"].numpy().astype(np.int32),
            "inliers": np.int64(n_in), "inlier_margin_m": np.float64(margin),
            "jitter_transform_spread": np.float64(spread), "jitter_num_corr": np.array(ncs), "jitter_node_pairs_symdiff": np.int64(node_sym),
            "model_seed": np.int64(SEED_MODEL), "limits": np.array(mgm.LIMITS),
        }
        T_gt = np.linalg.inv(T_planted).astype(np.float32)           # the direction `estimated_transform` has (see the residuals above)
        store["transform_gt"] = T_gt
        for tag in ("eval", "infer"):
            o = run(models[tag], a, b, T_gt)
            store[tag + "_estimated_transform"] = o["estimated_transform"].numpy()
            store[tag + "_node_corr"] = np.stack([o["pos_node_corr_indices"].numpy(), o["anc_node_corr_indices"].numpy()], 1).astype(np.int32)
            store[tag + "_corr_scores"] = o["corr_scores"].numpy()
            store[tag + "_pos_corr_points"], store[tag + "_anc_corr_points"] = o["pos_corr_points"].numpy(), o["anc_corr_points"].numpy()
            if tag == "eval":
                store["eval_node_matching_scores"] = o["node_matching_scores"].numpy()
                store["eval_pos_node_masks"], store["eval_anc_node_masks"] = o["pos_node_masks"].numpy(), o["anc_node_masks"].numpy()
                store["eval_score"] = o["score"].numpy()
            print(tag, "max |T - T(LCRNet)| %.3e, correspondences %d" %
                  (float((o["estimated_transform"] - out["estimated_transform"]).abs().max()), o["corr_scores"].shape[0]))
        if case == "shift":
            # how far the REFERENCE's own pose of the DEMO pair moves under the same one-rounding jitter
            da = np.load(os.path.join(HERE, "scans", "003854.npy")).astype(np.float32)
            db = np.load(os.path.join(HERE, "scans", "000958.npy")).astype(np.float32)
            od = run(models[""], da, db)
            base = od["estimated_transform"].numpy().astype(np.float64)
            worst_deg = worst_m = 0.0
            for rep in range(3):
                Tj = run(models[""], jitter(da, rng), jitter(db, rng))["estimated_transform"].numpy().astype(np.float64)
                # small-angle form: acos((tr - 1) / 2) of fp32 matrices has a floor of ~0.03 degrees (sqrt of the entries' rounding)
                worst_deg = max(worst_deg, float(np.degrees(np.linalg.norm(Tj[:3, :3].T @ base[:3, :3] - np.eye(3)) / np.sqrt(2.0))))
                worst_m = max(worst_m, float(np.linalg.norm(Tj[:3, 3] - base[:3, 3])))
            n_d, _ = inliers(od)
            print("demo pair: %d inliers of %d; the reference's own pose moves by %.3e deg / %.3e m under the jitter" % (n_d, od["corr_scores"].shape[0], worst_deg, worst_m))
            store["demo_pair_reference_jitter_deg"], store["demo_pair_reference_jitter_m"] = np.float64(worst_deg), np.float64(worst_m)
            store["demo_pair_inliers"] = np.int64(n_d)
        path = os.path.join(HERE, "pose_e2e_%s_golden.npz" % case)
        np.savez_compressed(path, **store)
        print("wrote", path, os.path.getsize(path) / 1e6, "MB")


if __name__ == "__main__":
    main()
