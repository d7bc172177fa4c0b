"""CPU: the oracle (oracle/lcr_oracle.cpp) reproduces the golden digests generated from the compiled reference."""
import numpy as np
import pytest

from conftest import DEMO_SCANS, LIMITS, NUM_STAGES, RADIUS, VOXEL, load_scan, search_specs, sha
from oracle import ops


@pytest.mark.parametrize("name", DEMO_SCANS + ["syn0", "syn1"])
def test_oracle_matches_reference_digests(name, ops_golden):
    xyz = load_scan(name)
    if name.startswith("syn"):
        assert sha(xyz) == str(ops_golden[f"{name}/voxel03_sha"])       # raw 120k -> 0.3 m step, bit-exact vs reference op
        assert len(xyz) == int(ops_golden[f"{name}/voxel03_n"])
    lens = np.array([len(xyz)], dtype=np.int64)
    pts, ls = [xyz], [lens]
    v = VOXEL
    for i in range(1, NUM_STAGES):
        v *= 2
        p, l = ops.grid_subsample(pts[-1], ls[-1], v)
        pts.append(p)
        ls.append(l)
    for i in range(NUM_STAGES):
        assert sha(pts[i]) == str(ops_golden[f"{name}/points{i}_sha"]), f"stage {i} points differ from the reference"
        assert np.array_equal(ls[i], ops_golden[f"{name}/lengths{i}"])
    for sname, q, s, ql, sl, r, lim in search_specs(pts, ls):
        out, cnt = ops.radius_search(q, s, ql, sl, r, lim, return_counts=True)
        k = f"{name}/{sname}"
        assert sha(cnt) == str(ops_golden[k + "_counts_sha"]), k
        assert int(cnt.max()) == int(ops_golden[k + "_max_count"]), k
        assert sha(out) == str(ops_golden[k + "_sha_canon"]), k            # canonical (d2, idx) order == reference up to ties
        assert int(ops_golden[k + "_n_rows_set_diff"]) == 0               # no tie run straddles the limit cut on these inputs


def test_oracle_pair_stack(ops_golden):
    a, b = load_scan("003854"), load_scan("000958")
    st = ops.precompute_data_stack_mode(np.concatenate([a, b]), np.array([len(a), len(b)]), NUM_STAGES, VOXEL, RADIUS, LIMITS)
    for i in range(NUM_STAGES):
        assert sha(st["points"][i]) == str(ops_golden[f"pair_003854_000958/points{i}_sha"])
        assert sha(st["neighbors"][i]) == str(ops_golden[f"pair_003854_000958/neighbors{i}_sha_canon"])
    for i in range(NUM_STAGES - 1):
        assert sha(st["subsampling"][i]) == str(ops_golden[f"pair_003854_000958/subsampling{i}_sha_canon"])
        assert sha(st["upsampling"][i]) == str(ops_golden[f"pair_003854_000958/upsampling{i}_sha_canon"])


def test_oracle_small_full_tensors(ops_golden):
    """Full tensors (not only digests) on the small 2-cloud stack, including the reference's narrower-than-limit width."""
    pts_in = ops_golden["small/points_in"]
    st = ops.precompute_data_stack_mode(pts_in, np.array([2048, 1500]), NUM_STAGES, VOXEL, RADIUS, LIMITS)
    for i in range(NUM_STAGES):
        assert np.array_equal(st["points"][i].view(np.uint32), ops_golden[f"small/points{i}"].view(np.uint32))
    for key in ["neighbors0", "neighbors3", "subsampling1", "upsampling2"]:
        kind, i = key[:-1], int(key[-1])
        got = st[kind][i]
        want = ops_golden[f"small/{key}"].astype(np.int64)          # reference width = min(limit, max count)
        w = want.shape[1]
        assert np.array_equal(got[:, :w], want)
        assert (got[:, w:] == got.max()).all()                      # extra columns are pure padding


def test_radius_count_bruteforce_matches_grid():
    xyz = load_scan("004481")[:3000]
    lens = np.array([1800, 1200])
    out, cnt = ops.radius_search(xyz, xyz, lens, lens, RADIUS, 40, return_counts=True)
    cnt2, mx = ops.radius_count(xyz, xyz, lens, lens, RADIUS)
    assert np.array_equal(cnt, cnt2)
    # same-cloud only: cloud 0 rows never reference cloud 1 supports
    valid = out[:1800][out[:1800] != 3000]
    assert valid.max() < 1800
