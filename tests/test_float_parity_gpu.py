"""GPU: floating-point parity at the contract tolerance (BASELINE.json north_star: "descriptors/poses within 1e-4 fp32"), measured
and REPORTED, on (1) the headline workload itself — the batch of 8 synthetic 120k-point scans of bench.py through the production
pipeline (raw-scan voxelisation + native collate + encoder + NetVLAD) against the oracle run scan by scan — and (2) the dense node
features of the pair model against the goldens of the imported reference.

Descriptors are unit vectors: absolute 1e-4.  Node features are O(1..20) activations (no normalisation at the end of the encoder /
decoder): they are checked RELATIVE to the largest magnitude of the tensor (1e-4), and the absolute maxima are printed so that
DESIGN.md §8 can quote measured numbers instead of bounds.  `pytest -s` shows the table; it is also written to
gpurun_out/float_parity.json when that directory exists."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, LIMITS, NUM_STAGES, RADIUS, ROOT, VOXEL, load_scan
from oracle import ops as oracle_ops
from oracle import torch_ref

pytestmark = pytest.mark.gpu
REPORT = {}
_HEADLINE_ORACLE = {}


def _note(key, err, scale):
    REPORT[key] = {"max_abs_err": float(err), "max_abs_value": float(scale), "relative": float(err / max(scale, 1e-30))}
    print("%-40s max|err| %.3e   max|x| %8.3f   relative %.2e" % (key, err, scale, err / max(scale, 1e-30)))
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        json.dump(REPORT, open(os.path.join(out, "float_parity.json"), "w"), indent=1)


@pytest.mark.parametrize("split", [True, False])
def test_headline_batch_descriptors_vs_oracle(split):
    """Both forms of the K-deep GEMMs (split=True: the default, bf16 x 3 split on the bf16 matrix cores; False: fp32 MFMA everywhere).
    configs[1]: the 8 synthetic scans of bench.py, raw, through DescriptorPipeline (the code path bench.py times) vs the oracle:
    C++ restatement for voxelisation / subsampling / neighbours, torch fp32 restatement for encoder + NetVLAD, one scan at a time
    (GroupNorm segments are per scan)."""
    import bench
    from lcrnet_amd.model_family import create_model
    from lcrnet_amd.pipeline import DescriptorPipeline
    from lcrnet_amd.weights import seeded_state_dict
    model = create_model().eval()
    model.load_state_dict(seeded_state_dict(model.state_dict(), 7351))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.cuda()
    scans = bench.make_batch(0)
    raw = torch.from_numpy(np.concatenate(scans)).cuda()
    lens = torch.tensor([len(s) for s in scans], dtype=torch.int64, device="cuda")
    from lcrnet_amd import functional as F
    was = F.gemm_split_enabled()
    F.set_gemm_split(split)
    try:
        with DescriptorPipeline(model, bench.VOXEL, bench.RADIUS, bench.NUM_STAGES, bench.LIMITS, upsampling=True, raw_voxel=bench.VOXEL) as pipe:
            got = [d.cpu() for d in pipe.run([(raw, lens), (raw, lens)])]
            dd = pipe.preprocess(raw, lens)
            with torch.no_grad():
                feats_c = model.encoder(dd["features"], dd)[-1].cpu()
    finally:
        F.set_gemm_split(was)
    assert torch.equal(got[0], got[1]) and got[0].shape == (8, 256)
    n_c = dd["lengths_host"][-1]
    worst_d = worst_f = scale_f = 0.0
    off = 0
    for i, s in enumerate(scans):
        if i not in _HEADLINE_ORACLE:                       # the oracle side does not depend on the GEMM form: once for both parameters
            p, l = oracle_ops.grid_subsample(s, np.array([len(s)]), bench.VOXEL)
            st = oracle_ops.precompute_data_stack_mode(p, l, bench.NUM_STAGES, bench.VOXEL, bench.RADIUS, bench.LIMITS)
            tdd = {k: [torch.from_numpy(np.ascontiguousarray(t)) for t in v] for k, v in st.items()}
            with torch.no_grad():
                f = torch_ref.kp_encoder(sd, torch.ones(len(p), 1), tdd)
                _HEADLINE_ORACLE[i] = (f[-1], torch_ref.global_descriptor(sd, f[-1]))
        f, want = [_HEADLINE_ORACLE[i][0]], _HEADLINE_ORACLE[i][1]
        n = int(n_c[i])
        assert f[-1].shape[0] == n
        worst_d = max(worst_d, (got[0][i] - want[0]).abs().max().item())
        worst_f = max(worst_f, (feats_c[off:off + n] - f[-1]).abs().max().item())
        scale_f = max(scale_f, f[-1].abs().max().item())
        off += n
    tag = "split-bf16 K-deep GEMMs" if split else "fp32 MFMA everywhere"
    _note("headline batch: descriptors (8 scans), %s" % tag, worst_d, 1.0)
    _note("headline batch: coarse features [N4,1024], %s" % tag, worst_f, scale_f)
    assert worst_d < 1e-4
    assert worst_f < 1e-4 * max(1.0, scale_f)
    # north_star names descriptors / poses at 1e-4; the un-normalised coarse features (|x| up to ~80) are held RELATIVE to their magnitude
    # above and pinned in absolute terms here: measured 1.9e-4 (fp32 re-association through 11 blocks), twice that is the bound
    assert worst_f < 4e-4, "coarse features: absolute error %.3e (pinned: 1.9e-4 measured at magnitude %.1f)" % (worst_f, scale_f)


def test_pair_model_dense_features_vs_reference_goldens():
    """Transformer outputs, vote-encoder node features and decoder point features of the demo pair vs the imported reference's
    tensors (model_golden.npz / pose_golden.npz), relative to each tensor's magnitude."""
    from lcrnet_amd.config import make_cfg
    from lcrnet_amd.model_family import LCRNet
    from lcrnet_amd.weights import seeded_state_dict
    golden = np.load(os.path.join(GOLDEN, "model_golden.npz"))
    pose = np.load(os.path.join(GOLDEN, "pose_golden.npz"))
    seed = json.load(open(os.path.join(GOLDEN, "model_manifest.json")))["seed"]
    cfg = make_cfg()
    cfg["neighbor_limits"] = LIMITS
    m = LCRNet(cfg).eval()
    m.load_state_dict(seeded_state_dict(m.state_dict(), seed), strict=True)
    m = m.cuda()
    a, b = load_scan("003854"), load_scan("000958")
    st = oracle_ops.precompute_data_stack_mode(np.concatenate([a, b]), np.array([len(a), len(b)]), NUM_STAGES, VOXEL, RADIUS, LIMITS)
    dd = {k: [torch.from_numpy(np.ascontiguousarray(t)).cuda() for t in v] for k, v in st.items()}
    dd["features"] = torch.ones(len(a) + len(b), 1, device="cuda")
    with torch.no_grad():
        out = m(dd)
    worst = 0.0
    for tag, e in (("pos", out["pos_feats_c_enhanced"]), ("anc", out["anc_feats_c_enhanced"])):
        rows, want = golden[f"pair/{tag}_tf_rows"], golden[f"pair/{tag}_tf_vals"]
        err, scale = np.abs(e.cpu().numpy()[rows] - want).max(), np.abs(want).max()
        _note("3D-RoFormer features (%s, 256-D)" % tag, err, scale)
        worst = max(worst, err / scale)
    r = golden["pair/feats_c_rows"]
    fc = m.encoder(dd["features"], dd)[-1].cpu().numpy()
    _note("pair-stack coarse features [N4,1024]", np.abs(fc[r] - golden["pair/feats_c_vals"]).max(), np.abs(golden["pair/feats_c_vals"]).max())
    worst = max(worst, REPORT["pair-stack coarse features [N4,1024]"]["relative"])
    for key, mine in (("feats_c", out["feats_c"]), ("pos_feats_f", out["pos_feats_f"]), ("anc_feats_f", out["anc_feats_f"])):
        rows, want = pose[key + "_rows"], pose[key + "_vals"]
        if mine.shape[0] <= rows.max():
            continue
        err, scale = np.abs(mine.cpu().numpy()[rows] - want).max(), np.abs(want).max()
        _note("pair model %s" % key, err, scale)        # downstream of the transformer + vote NMS of THIS run (not pinned inputs)
    assert worst < 1e-4, worst
