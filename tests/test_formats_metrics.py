"""CPU: on-disk formats and metric definitions (host logic) — round trips and hand-checkable cases."""
import numpy as np
import pytest

from lcrnet_amd import evaluation as ev
from lcrnet_amd import io_formats as io


def test_descriptor_npz_roundtrip(tmp_path):
    g = np.random.default_rng(0).standard_normal((5, 256)).astype(np.float32)
    for i in range(5):
        io.save_descriptor(str(tmp_path), 0, i, g[i])
    back = io.load_descriptors(str(tmp_path), 0)
    assert back.shape == (5, 256) and np.array_equal(back, g)
    one = np.load(tmp_path / "0_3.npz")["anc_global"]
    assert one.shape == (1, 256) and one.dtype == np.float32


def test_pair_dist_rows_and_metrics(tmp_path):
    # 400 frames; frame i >= 200 revisits frame i-200 (ground truth), descriptors = place id + noise
    rng = np.random.default_rng(1)
    place = rng.standard_normal((200, 256)).astype(np.float32)
    desc = np.concatenate([place, place + 0.01 * rng.standard_normal((200, 256)).astype(np.float32)])
    desc /= np.linalg.norm(desc, axis=1, keepdims=True)
    import torch
    from oracle import torch_ref
    qs, idx, d2 = torch_ref.retrieval_topk(torch.from_numpy(desc), k=50, exclude=100, start=101)
    rows = io.pair_dist_rows(qs.numpy(), idx.numpy(), d2.numpy())
    assert rows.shape == ((400 - 1 - 101) * 50, 3) and rows.dtype == np.float64
    assert rows[0, 0] == 101 and rows[0, 1] == 0                      # frame 101 has a 1-frame database
    assert rows[1, 1] == -1 and rows[1, 2] == float(io.FAISS_EMPTY_DISTANCE)
    io.save_pair_dist(str(tmp_path), rows)
    back = np.load(tmp_path / "predicted_des_L2_dis.npz")["arr_0"]
    assert back.shape == (len(rows), 1, 3) and np.array_equal(back.reshape(-1, 3), rows)       # the reference's [R,1,3] layout
    gt = np.empty(400, dtype=object)
    for i in range(400):
        gt[i] = np.array([i - 200]) if i >= 200 else np.array([])
    # frames 1..100 have no rows and no GT; frame 0 has gt [] -> .any() False (the reference's test, :47)
    assert ev.compute_topN(rows, gt, 1) == 1.0
    p, r = ev.compute_PR_overlap(rows, gt)
    f1, _ = ev.compute_F1(p, r)
    assert f1 > 0.99 and ev.auc(p, r) >= 0.0


def test_lcr_output_line():
    T = np.eye(4)
    T[:3, 3] = [1.5, -2.25, 0.125]
    line = io.lcr_output_line(3854, 958, np.ones(256), np.zeros(256), T)
    parts = line.split()
    assert parts[0] == "3854" and parts[1] == "958" and parts[2] == "16.00" and len(parts) == 15
    assert parts[3] == "1.000000" and parts[6] == "1.500000"


def test_collates_mirror_the_reference_layout_without_precompute():
    """registration / loop-detection collates (data.py:77-127, :350-406): stacking order, features, unwrapping, batch_size."""
    import torch
    from lcrnet_amd.data import registration_collate_fn_stack_mode, test_loop_detection_collate_fn_stack_mode_online
    rng = np.random.default_rng(0)
    def sample(nr, ns, tag):
        return {"ref_points": rng.random((nr, 3)).astype(np.float32), "src_points": rng.random((ns, 3)).astype(np.float32),
                "ref_feats": np.ones((nr, 1), np.float32), "src_feats": np.full((ns, 1), 2, np.float32),
                "transform": np.eye(4, dtype=np.float32) * tag, "seq_id": tag}
    s0, s1 = sample(5, 7, 1), sample(3, 4, 2)
    one = registration_collate_fn_stack_mode([s0], 4, 0.3, 1.275, [8, 8, 8, 8], precompute_data=False)
    assert one["batch_size"] == 1 and one["lengths"].tolist() == [5, 7] and one["points"].shape == (12, 3)
    assert torch.equal(one["points"][:5], torch.from_numpy(s0["ref_points"])) and one["features"].flatten().tolist() == [1.0] * 5 + [2.0] * 7
    assert one["seq_id"] == 1 and one["transform"].shape == (4, 4)                 # unwrapped for a single sample
    two = registration_collate_fn_stack_mode([s0, s1], 4, 0.3, 1.275, [8, 8, 8, 8], precompute_data=False)
    assert two["lengths"].tolist() == [5, 3, 7, 4]                                  # [ref_1, ref_2, src_1, src_2]
    assert two["features"].flatten().tolist() == [1.0] * 8 + [2.0] * 11 and two["seq_id"] == [1, 2] and two["batch_size"] == 2
    a = {"anc_points": rng.random((6, 3)).astype(np.float32), "anc_feats": np.ones((6, 1), np.float32), "frame": 17}
    got = test_loop_detection_collate_fn_stack_mode_online([a], 4, 0.3, 1.275, [8, 8, 8, 8], precompute_data=False)
    assert got["lengths"].tolist() == [6] and got["features"].shape == (6, 1) and got["frame"] == 17 and got["batch_size"] == 1


def test_scan_row_readers(tmp_path):
    """KITTI velodyne .bin (f32 [N,4]) and xyzi / xyz .npy files -> contiguous float32 host rows, stacked into a host batch."""
    import torch
    from lcrnet_amd import io_formats as io
    rng = np.random.default_rng(3)
    a = rng.standard_normal((1234, 4)).astype(np.float32)
    b = rng.standard_normal((77, 4)).astype(np.float32)
    a.astype("<f4").tofile(tmp_path / "000000.bin")
    np.save(tmp_path / "000001.npy", b)
    ra, rb = io.load_scan_rows(str(tmp_path / "000000.bin"), pin=False), io.load_scan_rows(str(tmp_path / "000001.npy"), pin=False)
    assert ra.dtype == torch.float32 and ra.is_contiguous() and np.array_equal(ra.numpy(), a) and np.array_equal(rb.numpy(), b)
    pts, lens = io.stack_scan_rows([ra, rb])
    assert pts.shape == (1311, 4) and lens.tolist() == [1234, 77] and lens.dtype == torch.int64
    np.save(tmp_path / "xyz.npy", a[:, :3].astype(np.float64))             # other dtypes are converted, 3 columns pass through
    assert io.load_scan_rows(str(tmp_path / "xyz.npy"), pin=False).shape == (1234, 3)
    (tmp_path / "bad.bin").write_bytes(b"\0" * 20)
    with pytest.raises(ValueError):
        io.load_scan_rows(str(tmp_path / "bad.bin"))
    with pytest.raises(ValueError):
        io.stack_scan_rows([ra, io.load_scan_rows(str(tmp_path / "xyz.npy"), pin=False)])
