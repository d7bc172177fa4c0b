"""Golden vectors of the two registration-model entry points, from the IMPORTED reference classes (build container only).

    python tests/golden/make_golden_matching.py

  experiments.lcrnet.model_family.LCRNet_Matching.LCRNet_Matching          (test_loop_closure.py:13 — the evaluation harness)
  experiments.lcrnet.model_family.LCRNet_Matching_infer.LCRNet_Matching    (infer_registration.py:11)

Both run UNMODIFIED in eval mode on CPU on the demo pair 003854 / 000958 (stubs, seeded weights and `.cuda()` patching as in
make_golden_pose.py; seeded_state_dict is a pure function of key name / shape / seed, so the shared keys hold the same values as in
the `LCRNet` goldens).  `data_dict['transform']` = the README's known-answer pose of this pair (README.md:80-85), projected onto
SO(3) — only the ground-truth labels depend on it.

Output: tests/golden/matching_golden.npz (what LCRNet_Matching returns beyond LCRNet: score, pos_emb / anc_emb, node_matching_scores,
node masks, gt_node_corr_indices / overlaps; plus the tensors the two classes share with the pose goldens, to pin that the
entry points are the same computation) and the two state-dict manifests merged into model_manifest.json.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_model as mgm  # noqa: E402

README_POSE = np.array([[0.3864, -0.9223, -0.0017, -5.186], [0.9222, 0.3863, 0.0198, 5.141], [-0.0176, -0.0092, 0.9998, -0.088], [0, 0, 0, 1.0]])


def demo_transform():
    T = README_POSE.copy()
    u, _, vt = np.linalg.svd(T[:3, :3])
    T[:3, :3] = u @ vt
    return T.astype(np.float32)


def main():
    mgm.install_stubs()
    sys.path.insert(0, mgm.REF)
    mgm.install_ref_ext()
    torch.Tensor.cuda = lambda self, *a, **k: self.contiguous()
    torch.nn.Module.cuda = lambda self, *a, **k: self
    from lcrnet_amd.weights import seeded_state_dict
    from experiments.lcrnet.config_model import make_cfg
    from experiments.lcrnet.data import precompute_data_stack_mode
    from experiments.lcrnet.model_family.LCRNet_Matching import LCRNet_Matching as RefEval
    from experiments.lcrnet.model_family.LCRNet_Matching_infer import LCRNet_Matching as RefInfer

    cfg = make_cfg()
    cfg.neighbor_limits = mgm.LIMITS
    cfg.vis = False
    a = np.load(os.path.join(HERE, "scans", "003854.npy"))
    b = np.load(os.path.join(HERE, "scans", "000958.npy"))
    pts = torch.from_numpy(np.concatenate([a, b]))
    dd = precompute_data_stack_mode(pts, torch.LongTensor([len(a), len(b)]), 4, 0.3, 1.275, mgm.LIMITS)
    dd = {k: [t.contiguous() for t in v] for k, v in dd.items()}
    dd["features"] = torch.ones(len(pts), 1)
    dd["batch_size"] = 1
    dd["transform"] = torch.from_numpy(demo_transform())

    manifest_path = os.path.join(HERE, "model_manifest.json")
    manifest = json.load(open(manifest_path))
    store = {"transform": demo_transform()}
    for tag, cls in (("eval", RefEval), ("infer", RefInfer)):
        model = cls(cfg).eval()
        model.load_state_dict(seeded_state_dict(model.state_dict(), mgm.SEED), strict=True)
        manifest["LCRNet_Matching" if tag == "eval" else "LCRNet_Matching_infer"] = {k: [list(v.shape), str(v.dtype)] for k, v in model.state_dict().items()}
        with torch.no_grad():
            out = model(dd)
        keys = sorted(k for k in out.keys())
        print(tag, "output keys:", keys)
        store[tag + "_keys"] = np.array(keys)
        un = lambda v: v[0] if isinstance(v, tuple) else v
        store[tag + "_estimated_transform"] = out["estimated_transform"].numpy()
        store[tag + "_length"] = out["length"].numpy().astype(np.int64)
        store[tag + "_pos_points_c"], store[tag + "_anc_points_c"] = out["pos_points_c"].numpy(), out["anc_points_c"].numpy()
        store[tag + "_node_corr"] = np.stack([out["pos_node_corr_indices"].numpy(), out["anc_node_corr_indices"].numpy()], 1).astype(np.int32)
        store[tag + "_num_corr"] = np.array(out["corr_scores"].shape[0])
        r = mgm.rows(out["pos_feats_f"].shape[0], 64, seed=3)
        store[tag + "_pos_feats_f_rows"], store[tag + "_pos_feats_f_vals"] = r, out["pos_feats_f"][r].numpy()
        if tag == "eval":
            store["eval_score"] = out["score"].numpy()
            store["eval_pos_emb"], store["eval_anc_emb"] = out["pos_emb"].numpy(), out["anc_emb"].numpy()
            store["eval_node_matching_scores"] = out["node_matching_scores"].numpy()
            store["eval_pos_node_masks"], store["eval_anc_node_masks"] = out["pos_node_masks"].numpy(), out["anc_node_masks"].numpy()
            store["eval_gt_node_corr_indices"] = out["gt_node_corr_indices"].numpy().astype(np.int32)
            store["eval_gt_node_corr_overlaps"] = out["gt_node_corr_overlaps"].numpy()
            ms = out["matching_scores"]
            store["eval_matching_scores_shape"] = np.array(ms.shape)
            print("eval: score", tuple(out["score"].shape), "pos_emb", tuple(out["pos_emb"].shape), "gt corr", tuple(out["gt_node_corr_indices"].shape),
                  "overlap max %.3f" % float(out["gt_node_corr_overlaps"].max()) if out["gt_node_corr_overlaps"].numel() else "none")
            # the ground-truth labels again from the module function on the reference's own node / patch tensors (the inputs the
            # HIP model is tested with, so that the label code is pinned independently of upstream float differences)
            store["eval_pos_node_knn_indices"] = un(out["pos_node_knn_indices"]).numpy().astype(np.int32)
            store["eval_anc_node_knn_indices"] = un(out["anc_node_knn_indices"]).numpy().astype(np.int32)
        print(tag, "T\n", out["estimated_transform"].numpy())
    json.dump(manifest, open(manifest_path, "w"), indent=0)
    path = os.path.join(HERE, "matching_golden.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, os.path.getsize(path) / 1e6, "MB; manifest sizes", {k: len(v) for k, v in manifest.items() if isinstance(v, dict)})


if __name__ == "__main__":
    main()
