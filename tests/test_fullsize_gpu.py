"""GPU, BASELINE full size (configs[1]): a batch of 8 raw synthetic 64-beam scans (~120 k points each) through the
throughput path (voxelize_raw_scans + precompute_batch: capacity buffers, shared grids, int32 indices, per-scan lengths on
the device) — compared EXACTLY with the C++ oracle run scan by scan, plus size-independent properties of the index tensors."""
import numpy as np
import pytest
import torch

from oracle import ops as oracle_ops

pytestmark = pytest.mark.gpu
LIMITS = [64, 65, 74, 80]


@pytest.fixture(scope="module")
def full_batch():
    import lcrnet_amd.synthetic as synthetic
    from lcrnet_amd.data import precompute_batch, voxelize_raw_scans
    scans = [synthetic.synthetic_scan(100 + i) for i in range(8)]
    raw = torch.from_numpy(np.concatenate(scans)).cuda()
    lens = torch.tensor([len(s) for s in scans], dtype=torch.int64, device="cuda")
    pts, lens_dev, lens_host = voxelize_raw_scans(raw, lens, 0.3)
    dd = precompute_batch(pts.contiguous(), lens_dev, 4, 0.3, 1.275, LIMITS, upsampling=True)
    return scans, dd, lens_host


def test_full_batch_equals_oracle_scan_by_scan(full_batch):
    scans, dd, lens_host = full_batch
    off = [np.concatenate([[0], np.cumsum(l)]) for l in dd["lengths_host"]]
    tot = [int(o[-1]) for o in off]
    assert 8 * 13000 < tot[0] < 8 * 19000                                   # 16 k +- a few k voxels per scan
    P = [p.cpu().numpy() for p in dd["points"]]
    NB = {k: [t.cpu().numpy().astype(np.int64) for t in dd[k]] for k in ("neighbors", "subsampling", "upsampling")}
    for b, raw in enumerate(scans):
        v, vl = oracle_ops.grid_subsample(raw, np.array([len(raw)]), 0.3)
        assert len(v) == lens_host[b]
        st = oracle_ops.precompute_data_stack_mode(v, vl, 4, 0.3, 1.275, LIMITS)
        for i in range(4):
            a, e = off[i][b], off[i][b + 1]
            assert np.array_equal(P[i][a:e].view(np.uint32), st["points"][i].view(np.uint32)), (b, i)

            def local(x, s_stage, q_stage):          # stacked indices -> per-scan indices (pad = support size)
                sa, se = off[s_stage][b], off[s_stage][b + 1]
                rows = x[off[q_stage][b]:off[q_stage][b + 1]]
                return np.where(rows == tot[s_stage], se - sa, rows - sa)
            assert np.array_equal(local(NB["neighbors"][i], i, i), st["neighbors"][i]), (b, i)
            if i < 3:
                assert np.array_equal(local(NB["subsampling"][i], i, i + 1), st["subsampling"][i]), (b, i)
                assert np.array_equal(local(NB["upsampling"][i], i + 1, i), st["upsampling"][i]), (b, i)


def test_full_batch_properties(full_batch):
    """Size-independent invariants: rows ascending in distance, strictly inside the radius, self first, never across scans,
    and (for rows that were not truncated) symmetric."""
    _, dd, _ = full_batch
    pts = dd["points"][0]
    nb = dd["neighbors"][0].long()
    n = pts.shape[0]
    valid = nb != n
    assert torch.equal(nb[:, 0], torch.arange(n, device=nb.device))          # d2 = 0 to itself, lowest key
    pp = torch.cat([pts, torch.full((1, 3), 1e6, device=pts.device)])
    d2 = ((pp[nb] - pts[:, None, :]) ** 2).sum(-1)
    assert bool((d2[valid] < 1.275 ** 2 + 1e-4).all())
    dm = torch.where(valid, d2, torch.full_like(d2, float("inf")))
    assert bool((dm[:, 1:] >= dm[:, :-1] - 1e-6).all())
    off = torch.tensor(np.concatenate([[0], np.cumsum(dd["lengths_host"][0])]), device=nb.device)
    seg = torch.bucketize(torch.arange(n, device=nb.device), off[1:], right=True)
    assert bool((torch.bucketize(nb[valid], off[1:], right=True) == seg[:, None].expand_as(nb)[valid]).all())
    full = valid.all(1)                                                       # truncated rows may lose the reverse edge
    rows = torch.nonzero(~full)[:2000, 0]
    for r in rows[:200].tolist():
        for j in nb[r][valid[r]].tolist():
            if not bool(full[j]):
                assert r in nb[j].tolist()
