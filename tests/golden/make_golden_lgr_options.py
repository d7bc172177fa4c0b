"""Golden for the LocalGlobalRegistration switches the shipped configuration leaves off — `use_dustbin=False` (with a confidence
threshold), `use_global_score=True`, `correspondence_limit=L` (local_global_registration.py:62-90, :153-160, :234-237) — from the
IMPORTED reference module (build container only):  python tests/golden/make_golden_lgr_options.py

Input: the seeded synthetic case of make_golden_pose_chain.py (24 patch correspondences, a known rigid motion, 6 outlier patches, confident
wrong matches) + seeded per-patch global scores.  For `use_dustbin=False` the module receives the score matrices with the dustbin row /
column stripped, as LCRNet.forward does (model_family/LCRNet.py:256-257).  Output: tests/golden/lgr_options_golden.npz — per case the
correspondences (patch, i, j), their points and scores, and the refined transform."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_model as mgm  # noqa: E402
from make_golden_pose_chain import synthetic_lgr_case  # noqa: E402

# tag -> (topk, mutual, use_dustbin, confidence_threshold, use_global_score, correspondence_limit)
CASES = {
    "nodust_thr05": (1, False, False, 0.05, False, None),
    "nodust_thr0_top2_mutual": (2, True, False, 0.0, False, None),
    "nodust_thr60_top3": (3, False, False, 0.6, False, None),
    "gscore": (1, False, True, 0.0, True, None),
    "limit300": (1, False, True, 0.0, False, 300),
    "limit200_gscore_nodust": (2, False, False, 0.3, True, 200),
    "limit5000": (1, False, True, 0.0, False, 5000),          # above the count: no effect
}


def global_scores(P, seed=3):
    return np.random.default_rng(seed).uniform(0.2, 1.0, P).astype(np.float32)


def main():
    mgm.install_stubs()
    sys.path.insert(0, mgm.REF)
    mgm.install_ref_ext()
    torch.Tensor.cuda = lambda self, *a, **k: self.contiguous()
    torch.nn.Module.cuda = lambda self, *a, **k: self
    from experiments.lcrnet.config_model import make_cfg
    from experiments.lcrnet.modules.geotransformer.local_global_registration import LocalGlobalRegistration
    fm = make_cfg().fine_matching
    store = {}
    ref, src, rm, sm, logs, T_true = synthetic_lgr_case()
    gs = global_scores(len(ref))
    tr, ts, trm, tsm, tl, tg = (torch.from_numpy(x) for x in (ref, src, rm, sm, logs, gs))
    for tag, (topk, mutual, dust, thr, use_gs, limit) in CASES.items():
        lgr = LocalGlobalRegistration(topk, fm.acceptance_radius, mutual=mutual, confidence_threshold=thr, use_dustbin=dust, use_global_score=use_gs,
                                      correspondence_threshold=fm.correspondence_threshold, correspondence_limit=limit,
                                      num_refinement_steps=fm.num_refinement_steps)
        scores = tl if dust else tl[:, :-1, :-1]
        with torch.no_grad():
            rp, sp, sc, T = lgr(tr, ts, trm, tsm, scores, tg)
            corr = lgr.compute_correspondence_matrix(torch.exp(scores), trm, tsm)
        b, i, j = torch.nonzero(corr, as_tuple=True)
        store.update({tag + "/corr_bij": torch.stack([b, i, j], 1).numpy().astype(np.int32), tag + "/ref_corr_points": rp.numpy(),
                      tag + "/src_corr_points": sp.numpy(), tag + "/corr_scores": sc.numpy(), tag + "/transform": T.numpy()})
        print("%-26s %5d correspondences, |T - T_true|max %.4f" % (tag, rp.shape[0], np.abs(T.numpy() - T_true).max()))
    import json
    store["cases_json"] = np.array(json.dumps(CASES))          # tag -> (topk, mutual, use_dustbin, confidence_threshold, use_global_score, limit)
    store["global_scores"] = gs
    store["true_transform"] = T_true
    np.savez_compressed(os.path.join(HERE, "lgr_options_golden.npz"), **store)


if __name__ == "__main__":
    main()
