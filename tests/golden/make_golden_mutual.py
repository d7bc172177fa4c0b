"""Golden for the `mutual=True` and `k > 1` branches of LocalGlobalRegistration (local_global_registration.py:56-87), from the IMPORTED reference module
(build container only):  python tests/golden/make_golden_mutual.py

The shipped configuration has mutual=False; round 5 builds the other value of the switch (lcr_top1_matching_ex).  Input: the seeded
well-conditioned synthetic case of make_golden_pose_chain.py (24 patches, a known rigid motion, outlier patches, confident wrong matches).
Output: tests/golden/mutual_golden.npz — the module's correspondences (patch, i, j), points, scores and refined transform with mutual=True,
and the number of correspondences the mutual test removes."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_model as mgm  # noqa: E402
from make_golden_pose_chain import synthetic_lgr_case  # noqa: E402


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
    tr, ts, trm, tsm, tl = (torch.from_numpy(x) for x in (ref, src, rm, sm, logs))
    for mutual, topk in ((False, 1), (True, 1), (False, 2), (True, 3)):
        lgr = LocalGlobalRegistration(topk, fm.acceptance_radius, mutual=mutual, confidence_threshold=fm.confidence_threshold, use_dustbin=fm.use_dustbin,
                                      use_global_score=fm.use_global_score, correspondence_threshold=fm.correspondence_threshold,
                                      correspondence_limit=fm.correspondence_limit, num_refinement_steps=fm.num_refinement_steps)
        with torch.no_grad():
            rp, sp, sc, T = lgr(tr, ts, trm, tsm, tl, torch.ones(len(ref)))
            corr = lgr.compute_correspondence_matrix(torch.exp(tl), trm, tsm)
        b, i, j = torch.nonzero(corr, as_tuple=True)
        tag = ("mutual_" if mutual else "either_") + ("" if topk == 1 else "top%d_" % topk)
        store.update({tag + "corr_bij": torch.stack([b, i, j], 1).numpy().astype(np.int32), tag + "ref_corr_points": rp.numpy(), tag + "src_corr_points": sp.numpy(),
                      tag + "corr_scores": sc.numpy(), tag + "transform": T.numpy()})
        print("mutual=%s topk=%d: %d correspondences, |T - T_true|max %.4f" % (mutual, topk, rp.shape[0], np.abs(T.numpy() - T_true).max()))
    store["true_transform"] = T_true
    np.savez_compressed(os.path.join(HERE, "mutual_golden.npz"), **store)


if __name__ == "__main__":
    main()
