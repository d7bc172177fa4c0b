"""GPU, skipped unless the trained checkpoint is present: the reference's only published known answer for this path.

README.md:78-86 of the reference — `python3 demo/demo.py` with `weights/best-model-mixed.tar` on the demo pair 003854 / 000958:
    L2 feature distance: 0.809192
    Estimated transformation: [[ 0.38640183 -0.92232913 -0.00168825 -5.1863933 ] [ 0.92216253 0.38629568 0.01978903 5.1413069 ]
                               [-0.01759983 -0.00920335 0.99980271 -0.0880447 ] [0 0 0 1]]
The checkpoint is a OneDrive / Baidu download (README.md:57-68) and exists in neither container, so this test is dormant here; it
runs the moment the file is dropped at `weights/best-model-mixed.tar` (or LCR_WEIGHTS points at it).  Neighbour limits are
calibrated on the pair like the demo loader does ([74, 68, 70, 67], tests/test_ops_gpu.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import LIMITS, NUM_STAGES, RADIUS, VOXEL, load_scan

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CANDIDATES = [os.environ.get("LCR_WEIGHTS", ""), os.path.join(ROOT, "weights", "best-model-mixed.tar")]
README_L2 = 0.809192
README_T = np.array([[3.8640183e-01, -9.2232913e-01, -1.6882520e-03, -5.1863933e+00],
                     [9.2216253e-01, 3.8629568e-01, 1.9789029e-02, 5.1413069e+00],
                     [-1.7599827e-02, -9.2033548e-03, 9.9980271e-01, -8.8044703e-02],
                     [0.0, 0.0, 0.0, 1.0]])


def _checkpoint():
    for p in CANDIDATES:
        if p and os.path.isfile(p):
            return p
    return None


@pytest.mark.skipif(_checkpoint() is None, reason="best-model-mixed.tar not present (README.md:57-68: external download)")
def test_readme_known_answer_with_the_trained_checkpoint():
    from lcrnet_amd.config import make_cfg
    from lcrnet_amd.data import precompute_data_stack_mode
    from lcrnet_amd.evaluation import compute_registration_error
    from lcrnet_amd.model_family import LCRNet
    from lcrnet_amd.weights import load_snapshot
    cfg = make_cfg()
    cfg["neighbor_limits"] = LIMITS
    model = LCRNet(cfg).eval()
    missing, unexpected = load_snapshot(model, _checkpoint(), strict=False)
    assert not missing, missing                                   # the reference layout loads unchanged
    model = model.cuda()
    a, b = load_scan("003854"), load_scan("000958")               # pos, anc (demo/demo.py: pos_idx 3854, anc_idx 958)
    pts = torch.from_numpy(np.concatenate([a, b])).cuda()
    lens = torch.tensor([len(a), len(b)], device="cuda")
    dd = precompute_data_stack_mode(pts, lens, NUM_STAGES, VOXEL, RADIUS, LIMITS)
    dd["features"] = torch.ones(len(pts), 1, device="cuda")
    with torch.no_grad():
        out = model(dd)
    l2 = float(torch.sqrt(((out["pos_feature_global"] - out["anc_feature_global"]) ** 2).sum()))
    assert abs(l2 - README_L2) < 1e-4, l2                         # north_star: descriptors within 1e-4
    T = out["estimated_transform"].cpu().numpy().astype(np.float64)
    rre, rte, *_ = compute_registration_error(README_T, T)
    # the README prints 8 significant digits of an fp32 pose from a consensus over ~1 k matches; same pose = far inside the
    # reference's own success criterion (5 deg / 2 m), and within 1e-2 deg / 1e-2 m of the printed one
    assert rre < 1e-2 and rte < 1e-2, (rre, rte, T)
