"""LCRNet — the pair model (experiments/lcrnet/model_family/LCRNet.py:25-321) with the reference's module tree, so
`best-model-mixed.tar` loads unchanged (373 tensors).

Implemented on the HIP path (this round): KeypointDetection up to the transformer (LCRNet.py:124-151: KPEncoder over the
pair stack, 3D-RoFormer on the coarsest stage) and GlobalDescritionHEAD on the PRE-transformer features of each cloud
(:115-122, :296-297).  The pose tail (Vote_Encoder forward, KPDecoder, Sinkhorn matching, LocalGlobalRegistration; :161-272)
holds its parameters here — so checkpoints round-trip — but its forward is SURVEY §8f-2 "next" work and raises.
"""
import torch
import torch.nn as nn

from ..backbone4 import KPEncoder
from ..config import make_cfg
from ..modules.kpconv import LastUnaryBlock, ResidualBlock, UnaryBlock
from ..modules.netvlad import NetVLADLoupe2
from ..modules.thdroformer import ThDRoFormer


class Vote_layer(nn.Module):
    """Parameters of modules/vote/vote.py:112-140 (shared MLP 256->512->256 with LayerNorm+ReLU, ctr_reg 256->3)."""

    def __init__(self, input_feats_dim=256, max_translate_range=4.2):
        super().__init__()
        c = input_feats_dim
        self.mlp_modules = nn.Sequential(nn.Linear(c, 2 * c), nn.LayerNorm(2 * c), nn.ReLU(), nn.Linear(2 * c, c), nn.LayerNorm(c), nn.ReLU())
        self.ctr_reg = nn.Linear(c, 3)
        self.max_translate_range = max_translate_range


class Vote_Encoder(nn.Module):
    """Parameters of backbone4.py:92-118 (vote layer + encoder6_1..3 at radii 8x/16x/16x init_radius)."""

    def __init__(self, init_dim, kernel_size, init_radius, init_sigma, group_norm, vote_cfg):
        super().__init__()
        self.vote = Vote_layer(256, vote_cfg["MAX_TRANSLATE_RANGE"])
        self.NMS_radius = vote_cfg["NMS_radius"]
        d, k, r, s, g = init_dim, kernel_size, init_radius, init_sigma, group_norm
        self.encoder6_1 = ResidualBlock(d * 4, d * 4, k, r * 8, s * 8, g, strided=True)
        self.encoder6_2 = ResidualBlock(d * 4, d * 8, k, r * 16, s * 16, g)
        self.encoder6_3 = ResidualBlock(d * 8, d * 8, k, r * 16, s * 16, g)

    def forward(self, *a, **k):
        raise NotImplementedError("Vote_Encoder.forward (vote -> NMS -> radius searches -> encoder6_x) is SURVEY §8f-2 'next' work")


class KPDecoder(nn.Module):
    """Parameters of backbone4.py:333-343."""

    def __init__(self, init_dim, group_norm):
        super().__init__()
        self.decoder3 = UnaryBlock(init_dim * 12, init_dim * 8, group_norm)
        self.decoder2 = UnaryBlock(init_dim * 12, init_dim * 4, group_norm)
        self.decoder1 = LastUnaryBlock(init_dim * 6, init_dim * 2)


class LearnableLogOptimalTransport(nn.Module):
    def __init__(self, num_iterations):
        super().__init__()
        self.num_iterations = num_iterations
        self.register_parameter("alpha", nn.Parameter(torch.tensor(1.0)))


class LCRNet(nn.Module):
    def __init__(self, cfg=None):
        super().__init__()
        cfg = cfg or make_cfg()
        b, g = cfg["backbone"], cfg["GAT"]
        self.encoder = KPEncoder(b["input_dim"], b["init_dim"], b["kernel_size"], b["init_radius"], b["init_sigma"], b["group_norm"])
        self.vote_encoder = Vote_Encoder(b["init_dim"], b["kernel_size"], b["init_radius"], b["init_sigma"], b["group_norm"], cfg["Vote"])
        self.proj_node_overlap_score = nn.Linear(g["output_dim"] * 2, 1)
        self.transformer = ThDRoFormer(g["input_dim"], g["output_dim"], g["hidden_dim"], g["num_heads"], g["num_layers"], g["k"])
        self.kpdecoder = KPDecoder(b["init_dim"], b["group_norm"])
        self.node_optimal_transport = LearnableLogOptimalTransport(cfg["model"]["num_sinkhorn_iterations"])
        self.optimal_transport = LearnableLogOptimalTransport(cfg["model"]["num_sinkhorn_iterations"])
        self.netvlad = NetVLADLoupe2(feature_size=1024, cluster_size=64, output_dim=256, gating=True, add_norm=True, is_training=False)

    def forward(self, data_dict):
        """Pair stack [pos(ref), anc(src)] (data.py:110-113).  GroupNorm statistics span the pair unless
        data_dict['segment_lengths'] says otherwise.  Needs lengths[-1] on the host: data_dict['lengths_c_host'] or a sync.
        Returns the keys of the reference output_dict that this round implements."""
        if self.training:
            raise RuntimeError("lcr-net_amd implements inference only; call .eval()")
        feats = data_dict["features"].detach()
        lens_c = data_dict.get("lengths_c_host")
        if lens_c is None:
            lens_c = data_dict["lengths"][-1].tolist()              # host sync, as the reference's .item() calls (LCRNet.py:127-128)
        n0, n1 = int(lens_c[0]), int(lens_c[1])
        points_c = data_dict["points"][-1]
        feats_list = self.encoder(feats, data_dict)
        feats_c = feats_list[-1]
        pos_c, anc_c = feats_c[:n0].contiguous(), feats_c[n0:n0 + n1].contiguous()
        e0, e1 = self.transformer(points_c[:n0].contiguous(), points_c[n0:n0 + n1].contiguous(), pos_c, anc_c)
        g = self.netvlad.describe(feats_c[:n0 + n1], [n0, n1])       # pre-transformer features (LCRNet.py:296-297)
        return {"pos_feature_global": g[0:1], "anc_feature_global": g[1:2], "pos_points_c": points_c[:n0],
                "anc_points_c": points_c[n0:n0 + n1], "pos_feats_c_enhanced": e0, "anc_feats_c_enhanced": e1,
                "feats_list": feats_list}


def create_model(cfg=None):
    return LCRNet(cfg)
