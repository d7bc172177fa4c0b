"""CPU: the torch fp32 restatement (oracle/torch_ref.py) against golden outputs of the imported REFERENCE model
(tests/golden/make_golden_model.py).  Tolerances: features 2e-4 abs on O(1) activations, descriptors 1e-5."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, LIMITS, NUM_STAGES, RADIUS, VOXEL, load_scan
from oracle import ops as oracle_ops
from oracle import torch_ref


@pytest.fixture(scope="module")
def model_golden():
    return np.load(os.path.join(GOLDEN, "model_golden.npz"))


@pytest.fixture(scope="module")
def seeded_sd():
    from lcrnet_amd.weights import seeded_tensor, kpconv_radius_of
    man = json.load(open(os.path.join(GOLDEN, "model_manifest.json")))
    sd = {}
    for k, (shape, dtype) in man["LCRNet"].items():
        dt = getattr(torch, dtype.split(".")[1])
        r = kpconv_radius_of(k, 4.25 * 0.3) if k.endswith("kernel_points") else None
        sd[k] = seeded_tensor(k, shape, dt, man["seed"], radius=r)
    return sd


def stack(xyz_list):
    xyz = np.concatenate(xyz_list)
    lens = np.array([len(x) for x in xyz_list], dtype=np.int64)
    st = oracle_ops.precompute_data_stack_mode(xyz, lens, NUM_STAGES, VOXEL, RADIUS, LIMITS)
    return {k: [torch.from_numpy(np.ascontiguousarray(t)) for t in v] for k, v in st.items()}


def test_encoder_blocks_and_descriptor_single_scan(model_golden, seeded_sd):
    dd = stack([load_scan("003854")])
    trace = {}
    with torch.no_grad():
        feats = torch_ref.kp_encoder(seeded_sd, torch.ones(dd["points"][0].shape[0], 1), dd, trace=trace)
        g = torch_ref.global_descriptor(seeded_sd, feats[-1])
    for name, o in trace.items():
        r = model_golden[f"003854/{name}_rows"]
        want = torch.from_numpy(model_golden[f"003854/{name}_vals"])
        assert o.shape[0] == int(model_golden[f"003854/{name}_stats"][2])
        assert torch.allclose(o[r], want, atol=2e-4, rtol=1e-4), (name, (o[r] - want).abs().max())
    assert torch.allclose(feats[-1], torch.from_numpy(model_golden["003854/feats_c"]), atol=2e-4, rtol=1e-4)
    assert torch.allclose(g, torch.from_numpy(model_golden["003854/anc_global"]), atol=1e-5)


@pytest.mark.parametrize("name", ["000026", "004481"])
def test_descriptor_other_scans(name, model_golden, seeded_sd):
    dd = stack([load_scan(name)])
    with torch.no_grad():
        feats = torch_ref.kp_encoder(seeded_sd, torch.ones(dd["points"][0].shape[0], 1), dd)
        g = torch_ref.global_descriptor(seeded_sd, feats[-1])
    assert torch.allclose(g, torch.from_numpy(model_golden[f"{name}/anc_global"]), atol=1e-5)


def test_pair_stack_transformer_and_descriptors(model_golden, seeded_sd):
    dd = stack([load_scan("003854"), load_scan("000958")])
    with torch.no_grad():
        feats = torch_ref.kp_encoder(seeded_sd, torch.ones(dd["points"][0].shape[0], 1), dd)   # GroupNorm over the pair
        fc = feats[-1]
        n0 = int(dd["lengths"][-1][0])
        assert [n0, fc.shape[0] - n0] == model_golden["pair/n_c"].tolist()
        r = model_golden["pair/feats_c_rows"]
        assert torch.allclose(fc[r], torch.from_numpy(model_golden["pair/feats_c_vals"]), atol=2e-4, rtol=1e-4)
        g0, g1 = torch_ref.global_descriptor(seeded_sd, fc[:n0]), torch_ref.global_descriptor(seeded_sd, fc[n0:])
        assert torch.allclose(g0, torch.from_numpy(model_golden["pair/pos_global"]), atol=1e-5)
        assert torch.allclose(g1, torch.from_numpy(model_golden["pair/anc_global"]), atol=1e-5)
        pc = dd["points"][-1]
        e0, e1 = torch_ref.thd_roformer(seeded_sd, pc[:n0], pc[n0:], fc[:n0], fc[n0:])
    for tag, e in (("pos", e0), ("anc", e1)):
        rr = model_golden[f"pair/{tag}_tf_rows"]
        want = torch.from_numpy(model_golden[f"pair/{tag}_tf_vals"])
        assert torch.allclose(e[rr], want, atol=2e-4, rtol=1e-4), (tag, (e[rr] - want).abs().max())
        assert abs(e.abs().mean().item() - model_golden[f"pair/{tag}_tf_stats"][1]) < 1e-4


def test_segmented_groupnorm_equals_separate_stacks(seeded_sd):
    """The build's per-segment GroupNorm on a 2-scan batch == running each scan alone (reference batch_size=1)."""
    a, b = load_scan("003854")[:6000], load_scan("000958")[:5000]
    lim = [40, 40, 40, 40]
    def prep(lst):
        xyz = np.concatenate(lst)
        lens = np.array([len(x) for x in lst], dtype=np.int64)
        st = oracle_ops.precompute_data_stack_mode(xyz, lens, NUM_STAGES, VOXEL, RADIUS, lim)
        return {k: [torch.from_numpy(np.ascontiguousarray(t)) for t in v] for k, v in st.items()}
    both, da, db = prep([a, b]), prep([a]), prep([b])
    with torch.no_grad():
        fb = torch_ref.kp_encoder(seeded_sd, torch.ones(len(a) + len(b), 1), both,
                                  segment_lengths=[l.tolist() for l in both["lengths"]])[-1]
        fa = torch_ref.kp_encoder(seeded_sd, torch.ones(len(a), 1), da)[-1]
        fbb = torch_ref.kp_encoder(seeded_sd, torch.ones(len(b), 1), db)[-1]
    assert torch.allclose(fb, torch.cat([fa, fbb]), atol=1e-4, rtol=1e-4)


def test_topk_attention_oracle_vs_reference_golden():
    """oracle/torch_ref.thd_roformer(k=...) — the top-k sparsified attention branch (rpetransformer.py:19-39) — against the imported
    reference ThDRoFormer with k = [0.5, 0.4, 0.3, 0.25] on seeded inputs and weights (tests/golden/make_golden_topk_attention.py)."""
    import os
    import sys
    import numpy as np
    import torch
    from conftest import GOLDEN
    sys.path.insert(0, GOLDEN)
    from make_golden_topk_attention import K_FRAC, topk_inputs
    from oracle import torch_ref
    from lcrnet_amd.modules.thdroformer.thdroformer_linear import ThDRoFormer
    from lcrnet_amd.weights import seeded_state_dict
    gold = np.load(os.path.join(GOLDEN, "topk_attention_golden.npz"))
    p0, p1, f0, f1 = topk_inputs()
    sd = {"transformer." + k: v for k, v in seeded_state_dict(ThDRoFormer(1024, 256, 128, 4, 4, k=K_FRAC).state_dict(), int(gold["seed"])).items()}
    t = torch.from_numpy
    for tag, k in (("topk", K_FRAC), ("dense", None)):
        with torch.no_grad():
            e0, e1 = torch_ref.thd_roformer(sd, t(p0), t(p1), t(f0), t(f1), k=k)
        err = max(np.abs(e0.numpy() - gold[tag + "_out0"]).max(), np.abs(e1.numpy() - gold[tag + "_out1"]).max())
        assert err < 1e-4, (tag, err)
    assert np.abs(gold["topk_out0"] - gold["dense_out0"]).max() > 1e-2          # the branch does something on this input
