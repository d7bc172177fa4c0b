"""LCRNet_Matching — experiments/lcrnet/model_family/LCRNet_Matching.py:25-355, the class the registration evaluation harness
builds (`experiments/registration/test_loop_closure.py:13`, BASELINE configs[4]).  Relative to `LCRNet` it has no NetVLAD head and,
in eval mode, additionally returns what the harness' Evaluator / loss terms read:

  pos_emb, anc_emb                     rotary angles of both clouds, (1, N, 64)         (:130-136, :331-332)
  score                                sigmoid(proj_node_overlap_score(node feats)) clamped to [0, 1], one per voted node (:146-147)
  node_matching_scores, pos/anc_node_masks, matching_scores                              (:212-214, :274)
  gt_node_corr_indices / _overlaps     ground-truth node correspondences under data_dict['transform'] (:189-204)

Same module tree -> same `state_dict` keys as the reference class (354 tensors, tests/test_checkpoint_layout.py); inference only, like the rest of this package
(the training branch — `coarse_target`, the `mask` for the vote loss — raises).  Forward = the HIP path of `LCRNet`."""
import torch

from .. import functional as F
from ..modules.registration import get_node_correspondences
from .LCRNet import LCRNet


class LCRNet_Matching(LCRNet):
    global_head = False
    matching_extras = True

    def __init__(self, cfg=None):
        super().__init__(cfg)
        from ..config import make_cfg
        m = (cfg or make_cfg())["model"]
        self.matching_radius = m.get("ground_truth_matching_radius", 0.45)

    def forward(self, data_dict, pose=True):
        return self.forward_pairs(data_dict, pose)[0]

    def forward_pairs(self, data_dict, pose=True):
        """data_dict['transform']: (4, 4) for one pair, (P, 4, 4) or a list for P pairs per call."""
        if "transform" not in data_dict:
            raise KeyError("transform")                              # the reference reads it first thing (LCRNet_Matching.py:301)
        T = data_dict["transform"]
        T = torch.stack(list(T)) if isinstance(T, (list, tuple)) else T
        T = T.detach().reshape(-1, 4, 4).float()
        outs = super().forward_pairs(data_dict, pose)
        if T.shape[0] != len(outs):
            raise RuntimeError("one transform per pair: got %d for %d pairs" % (T.shape[0], len(outs)))
        if not pose:
            return outs
        w, b = self.proj_node_overlap_score.weight, self.proj_node_overlap_score.bias
        for p, o in enumerate(outs):
            o["score"] = torch.sigmoid(F.linear(o["feats_c"].contiguous(), w, b).view(-1)).clamp(0, 1)
            N_pos, N_anc = o["pos_points_f"].shape[0], o["anc_points_f"].shape[0]
            pad = lambda x: torch.cat([x, torch.zeros_like(x[:1])], 0)
            pos_knn_pts = pad(o["pos_points_f"])[o["pos_node_knn_indices"].clamp(max=N_pos)]
            anc_knn_pts = pad(o["anc_points_f"])[o["anc_node_knn_indices"].clamp(max=N_anc)]
            gi, go = get_node_correspondences(o["pos_points_c"], o["anc_points_c"], pos_knn_pts, anc_knn_pts, T[p].to(pos_knn_pts.device),
                                              self.matching_radius, o["pos_node_masks"], o["anc_node_masks"], o["pos_node_knn_masks"],
                                              o["anc_node_knn_masks"])
            o["gt_node_corr_indices"], o["gt_node_corr_overlaps"] = gi, go
        return outs


def create_model(cfg=None):
    return LCRNet_Matching(cfg)
