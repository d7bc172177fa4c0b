"""LCRNet_Matching (inference form) — experiments/lcrnet/model_family/LCRNet_Matching_infer.py:24-288, the class
`experiments/inference/infer_registration.py:11` builds: KeypointDetection + DenseMatchingHEAD of the pair model WITHOUT the
NetVLAD head.  Same module tree as the reference class, hence the same `state_dict` keys (LCRNet's 373 tensors minus the 19
`netvlad.*` ones = 354; `best-model-mixed.tar` loads with strict=False exactly as `utils/engine/base_tester.py:111-122` does), same
`create_model(cfg)` / `forward(data_dict)`, same output keys.  The forward is the HIP path of `LCRNet` (model_family/LCRNet.py in
this package) with the descriptor head switched off; P pairs per call through `forward_pairs`."""
from .LCRNet import LCRNet


class LCRNet_Matching(LCRNet):
    global_head = False


def create_model(cfg=None):
    return LCRNet_Matching(cfg)
