"""GPU: loop detection chained into registration on the device (lcrnet_amd.loop_closure, tools/loop_closure_run.py) gives what the two
stages give when run separately — descriptors -> re-normalisation -> masked top-50 -> the find_top1 rule and its NN.txt -> the listed
pairs through PairPipeline — and writes the reference's files (infer_loop_detection_find_top1.py:9-116, infer_registration.py:66-80)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_chained_detection_and_registration_equal_the_separate_stages(tmp_path):
    import lcrnet_amd.synthetic as synthetic
    import loop_closure_run as tool
    from lcrnet_amd import io_formats as io
    from lcrnet_amd import loop_closure as lc
    from lcrnet_amd.config import make_cfg
    from lcrnet_amd.model_family import LCRNet, create_model
    from lcrnet_amd.retrieval import retrieval_topk
    from lcrnet_amd.weights import seeded_state_dict
    dev = torch.device("cuda", 0)
    dm = create_model().eval()
    dm.load_state_dict(seeded_state_dict(dm.state_dict(), 7351))
    dm = dm.to(dev)
    cfg = make_cfg()
    limits = [74, 68, 70, 67]
    cfg["neighbor_limits"] = limits
    pm = LCRNet(cfg).eval()
    pm.load_state_dict(seeded_state_dict(pm.state_dict(), 7351))
    pm = pm.to(dev)
    base = [synthetic.synthetic_scan(2000 + u, n_azimuth=400) for u in range(6)]           # small scans: ~25 k returns
    C = 230
    frames, rev = tool.make_frames(base, C, 120, dev)
    assert len(rev) > 10
    # ---- the stages on their own
    clouds = lc.voxelise_frames(frames)
    desc = lc.sequence_descriptors(dm, clouds, [64, 65, 74, 80])
    d = torch.from_numpy(io.renormalise_descriptors(desc.cpu().numpy())).to(dev)
    idx, d2 = retrieval_topk(d[101:C - 1], 101, d, 50, 100)
    ih, dh = idx.cpu().numpy(), d2.cpu().numpy()
    rows = io.pair_dist_rows(np.arange(101, C - 1), ih, np.where(ih >= 0, dh, np.inf))
    first = np.sort(rows.reshape(-1, 50, 3)[:, 0, 2])
    thres = float(first[9]) * (1 + 1e-6) + 1e-12                                           # lets ten queries' nearest rows through (and what else is that close)
    kept = io.top1_with_threshold(rows, C, thres)
    assert 10 <= len(kept) <= 200
    # ---- the chain
    got = lc.run(dm, pm, frames, thres, str(tmp_path), seq=0, pair_limits=limits, pairs_per_call=4, max_pairs=8)
    assert torch.allclose(got["descriptors"], desc, atol=1e-6)
    assert got["rows"].shape == rows.shape == ((C - 102) * 50, 3)
    fin = np.isfinite(rows[:, 2]) & (rows[:, 1] >= 0)
    assert np.array_equal(got["rows"][:, :2], rows[:, :2]) and np.abs(got["rows"][fin, 2] - rows[fin, 2]).max() < 1e-6
    assert np.array_equal(got["kept"][:, :2], kept[:, :2])
    text = open(got["top1_file"]).read()
    assert got["top1_file"].endswith("result/top1_with_thres_%.2f/00.txt" % thres) and text.count("\n") == len(kept)
    for line, r in zip(text.splitlines(), kept):
        i, j, dd = line.split()
        assert int(i) == int(r[0]) and int(j) == int(r[1]) and abs(float(dd) - float(r[2])) < 1e-6
    pairs = [(int(r[1]), int(r[0])) for r in kept][:8]                                     # ref = match, src = query (dataset.py:48-57)
    assert got["pairs"] == pairs and all(p < a - 100 for p, a in pairs)
    back = np.load(os.path.join(str(tmp_path), "features", "predicted_des_L2_dis.npz"))["arr_0"]
    assert back.shape == ((C - 102) * 50, 1, 3)
    assert len([n for n in os.listdir(os.path.join(str(tmp_path), "features")) if n.startswith("0_")]) == C
    # ---- registration of the same pairs on its own, same grouping
    sep = lc.register_pairs(pm, clouds, pairs, limits, pairs_per_call=4)
    lines = open(got["pose_file"]).read().splitlines()
    assert len(lines) == len(pairs) == len(got["outputs"]) == len(sep)
    n_rev = 0
    for (pos, anc), a, b, line in zip(pairs, got["outputs"], sep, lines):
        Ta, Tb = a["estimated_transform"].cpu().numpy(), b["estimated_transform"].cpu().numpy()
        assert np.abs(Ta - Tb).max() < 1e-4 and a["corr_scores"].shape == b["corr_scores"].shape
        assert line + "\n" == io.pose_line(pos, anc, Ta)
        if rev.get(anc) == pos:                                                            # a planted revisit: the pose is the planted motion
            n_rev += 1
            Tp = tool.planted()
            print("pair (%d, %d): |R - R_planted| %.2e, |t - t_planted| %.3f m" % (pos, anc, np.abs(Ta[:3, :3] - Tp[:3, :3]).max(), np.linalg.norm(Ta[:3, 3] - Tp[:3, 3])))
    print("%d loop rows under %.4f, %d pairs registered, %d of them planted revisits" % (len(kept), thres, len(pairs), n_rev))
