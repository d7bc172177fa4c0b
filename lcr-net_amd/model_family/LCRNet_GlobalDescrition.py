"""LCRNet_GlobalDescrition — descriptor-only model (experiments/lcrnet/model_family/LCRNet_GlobalDescrition.py:10-108),
inference branch.  Same sub-module names (`encoder`, `netvlad`) => `best-model-ld.tar` / `best-model-mixed.tar` load
unchanged with strict=False (0 missing keys; the mixed tar has 203 unexpected ones)."""
import torch.nn as nn

from ..backbone4 import KPEncoder
from ..modules.netvlad import NetVLADLoupe2
from ..config import make_cfg


class LCRNet_GlobalDescrition(nn.Module):
    def __init__(self, cfg=None):
        super().__init__()
        cfg = cfg or make_cfg()
        b = cfg["backbone"]
        self.encoder = KPEncoder(b["input_dim"], b["init_dim"], b["kernel_size"], b["init_radius"], b["init_sigma"], b["group_norm"])
        self.netvlad = NetVLADLoupe2(feature_size=1024, cluster_size=64, output_dim=256, gating=True, add_norm=True, is_training=False)

    def forward(self, data_dict):
        """Reference eval forward (:66-74): one stack -> {'anc_global': (1,256)} treating the WHOLE stack as one scan
        (the reference runs test batch_size=1).  With data_dict['segment_lengths'] (per-stage per-scan lengths, device) and
        data_dict['lengths_c_host'] (coarse-stage lengths on the host) every scan of the stack gets its own descriptor:
        {'anc_global': (S,256)}."""
        if self.training:
            raise RuntimeError("lcr-net_amd implements inference only; call .eval()")
        feats = data_dict["features"].detach()
        feats_list = self.encoder(feats, data_dict)
        feats_c = feats_list[-1]
        lens = data_dict.get("lengths_c_host")
        if lens is None:
            lens = [feats_c.shape[0]]
        return {"anc_global": self.netvlad.describe(feats_c, lens), "feats_c": feats_c}


def create_model(cfg=None):
    return LCRNet_GlobalDescrition(cfg)
