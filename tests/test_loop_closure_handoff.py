"""CPU: the loop-detection -> registration hand-off formats (io_formats.top1_with_threshold / top1_lines / load_loop_pairs / pose_line) against
the files the imported reference wrote and read (tests/golden/make_golden_top1.py: infer_loop_detection_find_top1.py:inference_one_epoch,
datasets/loop_closure/kitti/dataset.py:make_dataset_kitti, the f-string of infer_registration.py:78)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from lcrnet_amd import io_formats as io


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "top1_golden.npz"))


@pytest.mark.parametrize("thres", [0.11, 0.5])
def test_top1_file_text_and_pairs_equal_the_reference(gold, thres, tmp_path):
    rows = np.asarray(gold["rows"], dtype="float32").reshape(-1, 3)                 # the reference's own cast (:108-109)
    n_frames = len(gold["stored_descriptors"])
    kept = io.top1_with_threshold(rows, n_frames, thres)
    want = str(gold["thres_%.2f/text" % thres])
    assert len(kept) == want.count("\n") > 0
    assert io.top1_lines(kept) == want
    name = io.save_top1_with_threshold(str(tmp_path), 0, kept, thres)
    assert name.endswith("result/top1_with_thres_%.2f/00.txt" % thres) and open(name).read() == want
    pairs = io.load_loop_pairs(name)
    assert np.array_equal(np.array(pairs, dtype=np.int64).reshape(-1, 2), gold["thres_%.2f/pairs_pos_anc" % thres])
    assert all(pos < anc - 100 for pos, anc in pairs)                                # ref = the earlier frame, src = the query


def test_rows_from_renormalised_descriptors_match_the_reference(gold):
    """Stored (un-normalised) descriptors -> host re-normalisation (:75) -> the oracle's masked exhaustive search -> the reference's rows."""
    import torch
    from oracle import torch_ref
    d = io.renormalise_descriptors(gold["stored_descriptors"])
    assert np.abs(np.linalg.norm(d, axis=1) - 1).max() < 1e-6
    qs, idx, d2 = torch_ref.retrieval_topk(torch.from_numpy(d))
    rows = io.pair_dist_rows(qs.numpy(), idx.numpy(), d2.numpy())
    want = np.asarray(gold["rows"]).reshape(-1, 3)
    fin = want[:, 1] >= 0
    assert rows.shape == want.shape and np.array_equal(rows[:, 0], want[:, 0])
    assert np.abs(rows[fin, 2] - want[fin, 2]).max() < 1e-5
    # ids must agree wherever the distances are not tied (the 50th entry of a query may tie with the unseen 51st: exact duplicates exist)
    wd, wi, gi = want[:, 2].reshape(-1, 50), want[:, 1].reshape(-1, 50), rows[:, 1].reshape(-1, 50)
    gap = np.ones_like(wd, dtype=bool)
    gap[:, 1:] &= np.abs(wd[:, 1:] - wd[:, :-1]) > 1e-5
    gap[:, :-1] &= np.abs(wd[:, 1:] - wd[:, :-1]) > 1e-5
    gap[:, -1] = False
    assert not ((gi != wi) & (wi >= 0) & gap).any()


def test_pose_line(gold):
    assert io.pose_line(17, 250, gold["pose_T"]) == str(gold["pose_line"])
