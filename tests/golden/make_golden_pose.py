"""Golden vectors for the pose tail (a-10) from the imported reference `LCRNet.forward` (build container only).

    python tests/golden/make_golden_pose.py

Same stubs / seeded weights as make_golden_model.py.  The reference hard-codes `.cuda()` in the pose tail
(modules/ops/pointcloud_partition.py:87, sinkhorn/learnable_sinkhorn.py:34-58, registration/procrustes.py:54-63, …); there is
no GPU here, so for this run only `torch.Tensor.cuda` / `nn.Module.cuda` are patched to identity and the model
runs on CPU.  Output: tests/golden/pose_golden.npz with the intermediate tensors of the demo pair 003854/000958.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_model as mgm  # noqa: E402


def main():
    mgm.install_stubs()
    sys.path.insert(0, mgm.REF)
    mgm.install_ref_ext()
    torch.Tensor.cuda = lambda self, *a, **k: self.contiguous()   # .cuda() densifies the [:, :limit] views (utils/utils/torch.py:113-123)
    torch.nn.Module.cuda = lambda self, *a, **k: self
    from lcrnet_amd.weights import seeded_state_dict
    from experiments.lcrnet.config_model import make_cfg
    from experiments.lcrnet.data import precompute_data_stack_mode
    from experiments.lcrnet.model_family.LCRNet import LCRNet

    cfg = make_cfg()
    cfg.neighbor_limits = mgm.LIMITS
    cfg.vis = False
    full = LCRNet(cfg).eval()
    full.load_state_dict(seeded_state_dict(full.state_dict(), mgm.SEED), strict=True)
    a = np.load(os.path.join(HERE, "scans", "003854.npy"))
    b = np.load(os.path.join(HERE, "scans", "000958.npy"))
    pts = torch.from_numpy(np.concatenate([a, b]))
    dd = precompute_data_stack_mode(pts, torch.LongTensor([len(a), len(b)]), 4, 0.3, 1.275, mgm.LIMITS)
    dd = {k: [t.contiguous() for t in v] for k, v in dd.items()}
    dd["features"] = torch.ones(len(pts), 1)
    dd["batch_size"] = 1
    with torch.no_grad():
        out = full(dd)
    store = {}
    big = {"pos_feats_f", "anc_feats_f", "feats_c", "pos_feats_c", "anc_feats_c"}            # sampled rows + checksums only
    skip = {"pos_points_f", "anc_points_f", "ori_pos_points_c", "ori_anc_points_c", "pos_node_corr_knn_points",
            "anc_node_corr_knn_points", "pos_node_corr_knn_masks", "anc_node_corr_knn_masks"}  # recomputable from the inputs
    for k, v in out.items():
        if isinstance(v, tuple):
            v = v[0]
        if not torch.is_tensor(v) or k in skip:
            continue
        print(k, tuple(v.shape), v.dtype)
        if k in big:
            r = mgm.rows(v.shape[0], 64, seed=3)
            store[k + "_rows"], store[k + "_vals"] = r, v[r].numpy()
            store[k + "_stats"] = np.array([v.mean().item(), v.abs().mean().item(), v.shape[0], v.shape[1]])
        elif v.dtype == torch.int64:
            store[k] = v.numpy().astype(np.int32)
        else:
            store[k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "pose_golden.npz"), **store)
    print("estimated_transform\n", out["estimated_transform"].numpy())
    print("size MB", os.path.getsize(os.path.join(HERE, "pose_golden.npz")) / 1e6)


if __name__ == "__main__":
    main()
