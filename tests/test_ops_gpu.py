"""GPU parity tests for the native ops, through the C ABI (ctypes): HIP path vs the oracle and vs the golden digests
generated from the compiled reference.  Bit-exact: stage points (fp32 bit patterns, order included), neighbour indices."""
import numpy as np
import pytest
import torch

from conftest import DEMO_SCANS, LIMITS, NUM_STAGES, RADIUS, VOXEL, load_scan, search_specs, sha
from oracle import ops as oracle_ops

pytestmark = pytest.mark.gpu


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def test_fp32_div_floor_exact():
    """The voxel index arithmetic relies on IEEE fp32 subtract/divide on the device."""
    from lcrnet_amd.modules.ops import grid_subsample
    rng = np.random.default_rng(0)
    xyz = (rng.standard_normal((200000, 3)) * 40).astype(np.float32)
    lens = np.array([len(xyz)], dtype=np.int64)
    for v in (0.3, 0.123, 1.7):
        want, wl = oracle_ops.grid_subsample(xyz, lens, v)
        got, gl = grid_subsample(dev(xyz), dev(lens), v)
        assert np.array_equal(gl.cpu().numpy(), wl)
        assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("name", DEMO_SCANS + ["syn0", "syn1"])
def test_precompute_stages_bit_exact(name, ops_golden):
    from lcrnet_amd.modules.ops import grid_subsample, radius_search, radius_count
    xyz = load_scan(name)
    lens = np.array([len(xyz)], dtype=np.int64)
    pts, ls = [dev(xyz)], [dev(lens)]
    v = VOXEL
    for i in range(1, NUM_STAGES):
        v *= 2
        p, l = grid_subsample(pts[-1], ls[-1], v)
        pts.append(p.contiguous())
        ls.append(l)
    for i in range(NUM_STAGES):
        assert sha(pts[i].cpu().numpy()) == str(ops_golden[f"{name}/points{i}_sha"]), f"stage {i}"
        assert np.array_equal(ls[i].cpu().numpy(), ops_golden[f"{name}/lengths{i}"])
    for sname, q, s, ql, sl, r, lim in search_specs(pts, ls):
        k = f"{name}/{sname}"
        out = radius_search(q, s, ql, sl, r, lim)
        assert out.dtype == torch.int64 and out.is_contiguous()
        assert sha(out.cpu().numpy()) == str(ops_golden[k + "_sha_canon"]), k
        cnt = radius_count(q, s, ql, sl, r)
        assert sha(cnt.cpu().numpy()) == str(ops_golden[k + "_counts_sha"]), k
        o32 = radius_search(q, s, ql, sl, r, lim, dtype=torch.int32)
        assert torch.equal(o32.long(), out)


def test_pair_stack_and_uncapped_width(ops_golden):
    from lcrnet_amd.modules.ops import grid_subsample, radius_search
    a, b = load_scan("003854"), load_scan("000958")
    xyz = np.concatenate([a, b])
    lens = np.array([len(a), len(b)], dtype=np.int64)
    p1, l1 = grid_subsample(dev(xyz), dev(lens), 0.6)
    assert sha(p1.cpu().numpy()) == str(ops_golden["pair_003854_000958/points1_sha"])
    out = radius_search(dev(xyz), dev(xyz), dev(lens), dev(lens), RADIUS, LIMITS[0])
    assert sha(out.cpu().numpy()) == str(ops_golden["pair_003854_000958/neighbors0_sha_canon"])
    full = radius_search(p1.contiguous(), dev(xyz), l1, dev(lens), RADIUS, -1)        # reference semantics: width = max count
    assert full.shape[1] == int(ops_golden["pair_003854_000958/subsampling0_max_count"])
    want = oracle_ops.radius_search(p1.cpu().numpy(), xyz, l1.cpu().numpy(), lens, RADIUS, -1)
    assert np.array_equal(full.cpu().numpy(), want)


def test_ragged_and_empty_clouds():
    from lcrnet_amd.modules.ops import grid_subsample, radius_search
    rng = np.random.default_rng(5)
    xyz = (rng.random((5000, 3)) * np.array([30, 30, 4])).astype(np.float32)
    lens = np.array([1, 0, 2999, 2000], dtype=np.int64)              # single-point cloud, empty cloud
    want_p, want_l = oracle_ops.grid_subsample(xyz, lens, 0.8)
    got_p, got_l = grid_subsample(dev(xyz), dev(lens), 0.8)
    assert np.array_equal(got_l.cpu().numpy(), want_l)
    assert np.array_equal(got_p.cpu().numpy().view(np.uint32), want_p.view(np.uint32))
    want = oracle_ops.radius_search(want_p, xyz, want_l, lens, 2.0, 30, ref_width=True)
    got = radius_search(got_p.contiguous(), dev(xyz), got_l, dev(lens), 2.0, 30)
    assert np.array_equal(got.cpu().numpy(), want)


def test_dense_ball_exceeds_lds_capacity():
    """More than 512 in-radius supports per query: exercises the storage-free fallback ranking."""
    from lcrnet_amd.modules.ops import radius_search
    rng = np.random.default_rng(9)
    s = (rng.standard_normal((3000, 3)) * 0.4).astype(np.float32)
    q = s[:200].copy()
    ql, sl = np.array([200]), np.array([3000])
    want, cnt = oracle_ops.radius_search(q, s, ql, sl, 1.0, 100, return_counts=True, ref_width=True)
    assert cnt.max() > 512
    got = radius_search(dev(q), dev(s), dev(ql), dev(sl), 1.0, 100)
    assert np.array_equal(got.cpu().numpy(), want)


def _d2_forms(x, y, z):
    """The reference's d2 = ((x*x + y*y) + z*z), every operation rounded to fp32 (nanoflann.hpp:432-440), and what a compiler that
    contracts a product into a neighbouring add (hipcc's default, -ffp-contract=fast) can make of it: one product left exact inside
    an FMA, possibly after commuting the adds."""
    x, y, z = (np.asarray(a, dtype=np.float32) for a in (x, y, z))
    f32, f64 = np.float32, np.float64
    xx, yy, zz = x * x, y * y, z * z                                   # rounded products
    ex, ey, ez = x.astype(f64) * x.astype(f64), y.astype(f64) * y.astype(f64), z.astype(f64) * z.astype(f64)   # exact (48 bits)
    return {"plain": (xx + yy) + zz,
            "fma(y,y,xx)+zz": (xx.astype(f64) + ey).astype(f32) + zz,
            "fma(x,x,yy)+zz": (ex + yy.astype(f64)).astype(f32) + zz,
            "fma(z,z,xx+yy)": ((xx + yy).astype(f64) + ez).astype(f32),
            "xx+fma(y,y,zz)": xx + (ey + zz.astype(f64)).astype(f32),
            "xx+fma(z,z,yy)": xx + (yy.astype(f64) + ez).astype(f32)}


def test_distance_arithmetic_is_not_contracted():
    """d2 must be ((dx*dx + dy*dy) + dz*dz) with five fp32 roundings.  hipcc contracts a product into a neighbouring add by default
    (`__fmul_rn` / `__fadd_rn` do not stop it): one rounding less, d2 off by an ulp for a tenth of all offsets, which shows only when a
    support lies within that ulp of r2 (about 0.3 cases per 8-scan batch).  The library is built with -ffp-contract=off and the exact
    helpers carry `#pragma clang fp contract(off)`; this test places supports exactly where each possible contraction would change the
    membership, so a build that loses either protection fails here, not once in a few batches."""
    from lcrnet_amd.modules.ops import radius_search, radius_count
    rng = np.random.default_rng(17)
    radius = np.float32(1.275)
    r2 = radius * radius
    fams = {}
    while min([len(v) for v in fams.values()] or [0]) < 24 or len(fams) < 5:
        v = rng.standard_normal((400000, 3))
        v = (v / np.linalg.norm(v, axis=1, keepdims=True) * float(radius) * (1 + rng.uniform(-2e-7, 2e-7, (400000, 1)))).astype(np.float32)
        forms = _d2_forms(v[:, 0], v[:, 1], v[:, 2])
        for name, d2 in forms.items():
            if name != "plain":
                fams.setdefault(name, [])
                fams[name] += list(v[(forms["plain"] < r2) != (d2 < r2)][:24])
    for name, pts in fams.items():
        s = np.stack(pts[:24]).astype(np.float32)
        q = np.zeros((1, 3), dtype=np.float32)        # query at the origin: dx = -s exactly
        ql, sl = np.array([1]), np.array([len(s)])
        want, cnt = oracle_ops.radius_search(q, s, ql, sl, float(radius), len(s), return_counts=True, ref_width=True)
        forms = _d2_forms(s[:, 0], s[:, 1], s[:, 2])
        assert (forms["plain"] < r2).sum() == cnt[0] and (forms[name] < r2).sum() != cnt[0], name      # the generator did its job
        got_cnt = radius_count(dev(q), dev(s), dev(ql), dev(sl), float(radius)).cpu().numpy()
        assert got_cnt[0] == cnt[0], "in-radius count follows the contracted form %s: %d vs %d" % (name, got_cnt[0], cnt[0])
        got = radius_search(dev(q), dev(s), dev(ql), dev(sl), float(radius), len(s))
        assert np.array_equal(got.cpu().numpy(), want), name


def test_exact_ties_order_by_index():
    from lcrnet_amd.modules.ops import radius_search
    # lattice points: many exactly equal distances -> order must be (d2, idx)
    g = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(6), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    rng = np.random.default_rng(2)
    g = g[rng.permutation(len(g))]
    n = np.array([len(g)])
    want = oracle_ops.radius_search(g, g, n, n, 2.5, 40, ref_width=True)
    got = radius_search(dev(g), dev(g), dev(n), dev(n), 2.5, 40)
    assert np.array_equal(got.cpu().numpy(), want)


def test_raw_scan_voxel03(ops_golden):
    """BASELINE configs[1] front end: raw ~120k-pt synthetic scans -> 0.3 m voxels, batch of 2 stacked."""
    import lcrnet_amd.synthetic as synthetic
    from lcrnet_amd.modules.ops import grid_subsample
    raws = [synthetic.synthetic_scan(s) for s in (0, 1)]
    assert sha(raws[0]) == str(ops_golden["syn0/raw_sha"])
    xyz = np.concatenate(raws)
    lens = np.array([len(r) for r in raws], dtype=np.int64)
    got, gl = grid_subsample(dev(xyz), dev(lens), VOXEL)
    got = got.cpu().numpy()
    n0 = int(ops_golden["syn0/voxel03_n"])
    assert gl.cpu().tolist() == [n0, int(ops_golden["syn1/voxel03_n"])]
    assert sha(got[:n0]) == str(ops_golden["syn0/voxel03_sha"])
    assert sha(got[n0:]) == str(ops_golden["syn1/voxel03_sha"])


def test_neighbor_limit_calibration_matches_reference_demo_values():
    """calibrate_neighbors_stack_mode (data.py:408-433) from device-side counts: the reference demo pair calibrates to
    [74, 68, 70, 67] (measured with the reference itself)."""
    from lcrnet_amd.data import calibrate_neighbors_stack_mode
    clouds = [dev(load_scan("003854")), dev(load_scan("000958"))]
    limits = calibrate_neighbors_stack_mode(clouds, NUM_STAGES, VOXEL, RADIUS)
    assert limits.tolist() == LIMITS
    pair = (torch.cat(clouds), torch.tensor([len(c) for c in clouds], dtype=torch.int64, device="cuda"))
    assert calibrate_neighbors_stack_mode([pair], NUM_STAGES, VOXEL, RADIUS).tolist() == LIMITS      # the demo's item: the stacked pair


def test_key_bits_promise_broken_then_retried():
    """A cloud with two far-away outliers needs more voxel-key bits than the 32 the raw-scan ingest promises: the device raises
    LCR_STATUS_KEY_OVERFLOW and voxelize_raw_scans retries with 64-bit-safe keys — result == oracle, bit for bit."""
    from lcrnet_amd.data import voxelize_raw_scans
    from lcrnet_amd.modules.ops import grid_subsample_device
    rng = np.random.default_rng(3)
    xyz = (rng.random((4000, 3)) * np.array([40, 40, 3]) - 20).astype(np.float32)
    xyz[7] = (30000.0, -25000.0, 9000.0)                      # 2e5 x 2e5 x 3e4 voxels of 0.3 m: ~50 key bits
    xyz[1234] = (-31000.0, 28000.0, -2000.0)
    lens = np.array([1500, 2500], dtype=np.int64)
    _, _, st = grid_subsample_device(dev(xyz), dev(lens), 0.3, key_bits_hint=32)
    assert int(st.item()) & 1, "the broken promise must be reported, not silently mis-sorted"
    got_p, got_l, got_lh = voxelize_raw_scans(dev(xyz), dev(lens), 0.3)
    want_p, want_l = oracle_ops.grid_subsample(xyz, lens, 0.3)
    assert got_lh == want_l.tolist() and np.array_equal(got_l.cpu().numpy(), want_l)
    assert np.array_equal(got_p.cpu().numpy().view(np.uint32), want_p.view(np.uint32))


def test_negative_coordinates_and_queries_outside_the_support_box():
    from lcrnet_amd.modules.ops import radius_search
    rng = np.random.default_rng(11)
    s = (rng.random((3000, 3)) * 20 - 30).astype(np.float32)              # all-negative support box
    q = np.concatenate([s[:50] + 0.01, (rng.random((50, 3)) * 400 - 200).astype(np.float32)])   # half of the queries far outside
    ql, sl = np.array([60, 40]), np.array([1800, 1200])
    want, cnt = oracle_ops.radius_search(q, s, ql, sl, 2.5, 40, return_counts=True, ref_width=True)
    got = radius_search(dev(q), dev(s), dev(ql), dev(sl), 2.5, 40)
    assert np.array_equal(got.cpu().numpy(), want) and (cnt == 0).any() and (cnt > 0).any()


def test_sixty_four_clouds_and_the_limit_beyond():
    from lcrnet_amd.modules.ops import grid_subsample, radius_search
    rng = np.random.default_rng(2)
    B = 64
    sizes = rng.integers(0, 200, B)
    xyz = (rng.random((int(sizes.sum()), 3)) * 12).astype(np.float32)
    lens = sizes.astype(np.int64)
    want_p, want_l = oracle_ops.grid_subsample(xyz, lens, 1.0)
    got_p, got_l = grid_subsample(dev(xyz), dev(lens), 1.0)
    assert np.array_equal(got_l.cpu().numpy(), want_l) and np.array_equal(got_p.cpu().numpy().view(np.uint32), want_p.view(np.uint32))
    want = oracle_ops.radius_search(xyz, xyz, lens, lens, 1.5, 16, ref_width=True)
    got = radius_search(dev(xyz), dev(xyz), dev(lens), dev(lens), 1.5, 16)
    assert np.array_equal(got.cpu().numpy(), want)
    got_p, got_l = grid_subsample(dev(xyz), dev(np.concatenate([lens, [0]])), 1.0)        # 65 clouds: two native calls (round 3)
    assert np.array_equal(got_l.cpu().numpy(), np.concatenate([want_l, [0]])) and np.array_equal(got_p.cpu().numpy().view(np.uint32), want_p.view(np.uint32))


@pytest.mark.parametrize("seed", range(4))
def test_output_order_at_every_rehash_boundary(seed):
    """Clouds whose voxel counts sit on, just below and just above the bucket counts of libstdc++'s rehash schedule (13, 29,
    ..., 2357 | 5087, 10273): the first eight replay phases run in LDS, the later ones in global memory, and the clouds of one
    call cross the hand-over at different phases.  Bit-exact points, order included, against std::unordered_map (the oracle)."""
    from lcrnet_amd.modules.ops import grid_subsample
    rng = np.random.default_rng(100 + seed)
    sizes = [1, 12, 13, 14, 29, 30, 58, 59, 60, 127, 128, 257, 541, 542, 1109, 1110, 2356, 2357, 2358, 5086, 5087, 5088, 10273, 10274]
    pick = list(rng.permutation(sizes)[:12]) + [2357, 2358, 5088][seed % 3:seed % 3 + 1]
    clouds = []
    for n_vox in pick:
        # n_vox distinct voxels of a 64^3 lattice, 1-3 points each, in random order (so insertion order != key order)
        cells = rng.choice(64 ** 3, size=int(n_vox), replace=False)
        ijk = np.stack([cells % 64, (cells // 64) % 64, cells // 4096], 1).astype(np.float32)
        reps = rng.integers(1, 4, size=int(n_vox))
        base = np.repeat(ijk, reps, axis=0)
        pts = (base + rng.random(base.shape).astype(np.float32) * 0.9 + 0.05) * np.float32(0.5) + rng.normal(0, 20, 3).astype(np.float32)
        clouds.append(pts[rng.permutation(len(pts))].astype(np.float32))
    xyz = np.concatenate(clouds)
    lens = np.array([len(c) for c in clouds], dtype=np.int64)
    want, wl = oracle_ops.grid_subsample(xyz, lens, 0.5)
    got, gl = grid_subsample(dev(xyz), dev(lens), 0.5)
    assert np.array_equal(gl.cpu().numpy(), wl)
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("voxel,cs", [(0.3, [3.299999952316284, 5.099999904632568, 6.599999904632568]), (0.77, [-923.2300415039062, -922.4600219726562, -920.9200439453125])])
def test_whole_clouds_one_cell_below_the_voxel_origin(voxel, cs):
    """Coordinates c with floor(c * fl(1/v)) * v > c in fp32: a plane / line / point at such a constant coordinate sits, as a whole,
    one cell BELOW its own voxel origin; the reference's extent on that axis is (size_t)(-1) + 1 = 0 and every key carries the wrapped
    index 2^64 - 1 (a plane x = c even collapses into ONE voxel: the y and z strides are multiples of NX = 0).  Found by the op
    fuzz (seed 63843); pinned here on all axis combinations, next to an ordinary cloud in the same stack."""
    from lcrnet_amd.modules.ops import grid_subsample
    c = [np.float32(x) for x in cs]
    v32, inv = np.float32(voxel), np.float32(1.0) / np.float32(voxel)
    for x in c:
        assert np.float32(np.floor(np.float32(x * inv)) * v32) > x             # the premise, in the reference's own arithmetic
    rng = np.random.default_rng(7)
    a = (rng.random((500, 3)) * 20).astype(np.float32)
    pl_z, pl_x, pl_y, ln, ln2 = a.copy(), a.copy(), a.copy(), a.copy(), a.copy()
    pl_z[:, 2] = c[0]
    pl_x[:, 0] = c[1]
    pl_y[:, 1] = c[2]
    ln[:, 1], ln[:, 2] = c[2], c[0]
    ln2[:, 0], ln2[:, 1] = c[1], c[2]
    pt = np.array([[c[0], c[1], c[2]]], np.float32)
    clouds = [a, pl_z, pl_x, pl_y, ln, ln2, pt]
    xyz = np.concatenate(clouds)
    lens = np.array([len(x) for x in clouds], dtype=np.int64)
    want, wl = oracle_ops.grid_subsample(xyz, lens, voxel)
    got, gl = grid_subsample(dev(xyz), dev(lens), voxel)
    assert np.array_equal(gl.cpu().numpy(), wl)
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert wl[2] == 1                                                            # the collapsed plane


@pytest.mark.parametrize("seed", list(range(6)) + [1379, 63843])     # 1379: a point one cell below the voxel origin (wrapped 64-bit key); 63843: a whole plane one cell below it (extent 0 on that axis)
def test_random_clouds_subsample_and_search(seed):
    """Randomised shapes the fixed fixtures do not have: duplicates, points exactly on voxel faces, a plane, a line, tight
    clusters, far-apart clouds in one stack; subsample and both kinds of search bit-exact against the oracle."""
    from lcrnet_amd.modules.ops import grid_subsample, radius_search
    rng = np.random.default_rng(500 + seed)
    clouds = []
    for _ in range(int(rng.integers(2, 7))):
        n = int(rng.integers(1, 4000))
        kind = int(rng.integers(0, 6))
        if kind == 0:
            p = rng.random((n, 3)) * rng.uniform(1, 60, 3)
        elif kind == 1:
            p = np.round(rng.random((n, 3)) * 40) * 0.25                     # lattice: duplicates and points on voxel faces
        elif kind == 2:
            p = np.concatenate([rng.random((n, 2)) * 50, np.zeros((n, 1))], 1)   # plane
        elif kind == 3:
            p = np.outer(rng.random(n) * 80, rng.standard_normal(3))             # line
        elif kind == 4:
            p = rng.standard_normal((n, 3)) * 0.3 + rng.integers(0, 5, (n, 1)) * 7.0   # tight clusters
        else:
            p = rng.standard_normal((n, 3)) * 15
        clouds.append((p + rng.normal(0, 200, 3)).astype(np.float32))
    xyz = np.concatenate(clouds)
    lens = np.array([len(c) for c in clouds], dtype=np.int64)
    voxel = float(rng.choice([0.25, 0.5, 0.77, 1.3]))
    want_p, want_l = oracle_ops.grid_subsample(xyz, lens, voxel)
    got_p, got_l = grid_subsample(dev(xyz), dev(lens), voxel)
    assert np.array_equal(got_l.cpu().numpy(), want_l)
    assert np.array_equal(got_p.cpu().numpy().view(np.uint32), want_p.view(np.uint32))
    radius, limit = voxel * 4.25, int(rng.integers(8, 70))
    want = oracle_ops.radius_search(want_p, xyz, want_l, lens, radius, limit, ref_width=True)               # coarse queries, fine supports
    got = radius_search(got_p.contiguous(), dev(xyz), got_l, dev(lens), radius, limit)
    assert np.array_equal(got.cpu().numpy(), want)
    want = oracle_ops.radius_search(want_p, want_p, want_l, want_l, radius, limit, ref_width=True)          # self search
    got = radius_search(got_p.contiguous(), got_p.contiguous(), got_l, got_l, radius, limit)
    assert np.array_equal(got.cpu().numpy(), want)


def test_registration_collate_precomputes_the_pair_stack(ops_golden):
    """The pair collate with precompute_data=True: the reference's dictionary for the demo pair, computed on the device."""
    from lcrnet_amd.data import registration_collate_fn_stack_mode
    a, b = load_scan("003854"), load_scan("000958")
    sample = {"ref_points": a, "src_points": b, "ref_feats": np.ones((len(a), 1), np.float32), "src_feats": np.ones((len(b), 1), np.float32),
              "transform": np.eye(4, dtype=np.float32)}
    dd = registration_collate_fn_stack_mode([sample], NUM_STAGES, VOXEL, RADIUS, LIMITS)
    assert dd["batch_size"] == 1 and dd["features"].is_cuda and dd["features"].shape == (len(a) + len(b), 1)
    assert [len(dd[k]) for k in ("points", "lengths", "neighbors", "subsampling", "upsampling")] == [4, 4, 4, 3, 3]
    assert sha(dd["points"][1].cpu().numpy()) == str(ops_golden["pair_003854_000958/points1_sha"])
    assert sha(dd["neighbors"][0].cpu().numpy()) == str(ops_golden["pair_003854_000958/neighbors0_sha_canon"])
    want = oracle_ops.precompute_data_stack_mode(np.concatenate([a, b]), np.array([len(a), len(b)]), NUM_STAGES, VOXEL, RADIUS, LIMITS)
    for key in ("points", "lengths", "neighbors", "subsampling", "upsampling"):
        for got_t, want_t in zip(dd[key], want[key]):
            assert np.array_equal(got_t.cpu().numpy(), want_t), key


def test_radius_query_processing_order_does_not_change_the_result():
    """lcr_radius_query_ordered: any permutation of the queries as processing order (cell order, reversed, random) gives the rows
    of the unordered call, counts included; ragged clouds with an empty one."""
    from lcrnet_amd.modules.ops import SupportGrid
    rng = np.random.default_rng(11)
    lens = np.array([3000, 0, 1, 2500], dtype=np.int64)
    s = (rng.random((int(lens.sum()), 3)) * np.array([40, 40, 5])).astype(np.float32)
    qlens = np.array([700, 0, 3, 900], dtype=np.int64)
    q = np.concatenate([s[:700] + 0.01, s[3000:3001].repeat(3, 0), s[3001:3901] - 0.02]).astype(np.float32)
    grid = SupportGrid(dev(s), dev(lens), 2.0)
    base, base_cnt = grid.query(dev(q), dev(qlens), 40, want_counts=True)
    want = oracle_ops.radius_search(q, s, qlens, lens, 2.0, 40)
    assert np.array_equal(base.cpu().numpy(), want)
    nq = len(q)
    qgrid = SupportGrid(dev(q), dev(qlens), 2.0)
    orders = [qgrid.order()[:nq].contiguous(), torch.arange(nq - 1, -1, -1, dtype=torch.int32, device="cuda"),
              torch.from_numpy(rng.permutation(nq).astype(np.int32)).cuda()]
    for od in orders:
        got, cnt = grid.query(dev(q), dev(qlens), 40, want_counts=True, q_order=od)
        assert torch.equal(got, base) and torch.equal(cnt, base_cnt)


@pytest.mark.parametrize("seed", range(12))
def test_radius_search_fuzz_against_the_oracle(seed):
    """Randomised configurations through every path of the query kernel — 1 / 2 / 4 queries set up per wavefront turn (query counts
    below / above 16 384 / 65 536), sphere culling with coordinates quantised to a lattice commensurate with nothing in particular
    AND with the cell size, balls denser than the 512-key LDS capacity (exact fallback), empty and one-point clouds, queries far
    outside the support box, limits below and above the densest ball — bit-exact rows and counts vs the C++ oracle."""
    from lcrnet_amd.modules.ops import SupportGrid
    rng = np.random.default_rng(1000 + seed)
    B = int(rng.integers(1, 6))
    radius = float(rng.choice([0.4, 0.75, 1.0, 1.275, 2.55]))
    big = seed % 4 == 3
    s_sizes = rng.integers(0, 30000 if big else 4000, B)
    s_sizes[rng.integers(0, B)] = max(s_sizes.max(), 1)
    q_sizes = rng.integers(0, 30000 if big else 3000, B)
    if seed % 6 == 5:
        q_sizes[:] = 0
        q_sizes[0] = 70000                                   # > 65 536 queries: four per turn
    extent = np.array([rng.uniform(5, 60), rng.uniform(5, 60), rng.uniform(0.5, 6)])
    s_list, q_list = [], []
    dense_seen = False                                       # some query certainly sees a planted dense ball (premise of the last assert)
    for b in range(B):
        s = rng.random((int(s_sizes[b]), 3)) * extent - extent / 2
        if seed % 3 == 0:
            s = np.round(s / 0.25) * 0.25                    # lattice: exact ties, points on common planes
        planted = seed % 3 == 1 and len(s) > 700             # per CLOUD: the ball goes into every cloud that is large enough
        if planted:
            s[:700] = s[0] + rng.standard_normal((700, 3)) * 0.05 * radius      # a ball with > 512 points inside the radius
        q = rng.random((int(q_sizes[b]), 3)) * extent * 1.3 - extent * 0.65       # some queries outside the support box
        if len(q) and len(s):
            k = min(len(q), len(s)) // 2
            q[:k] = s[rng.integers(0, len(s), k)]           # half of the queries ON support points
        s_list.append(s.astype(np.float32))
        q_list.append(q.astype(np.float32))
        if planted and len(q):
            # a query of THIS cloud within r / 2 of the ball's centre (points 0 .. 699 lie within ~0.25 r of it: 5 sigma) has more than
            # 512 of them inside its radius
            far = np.linalg.norm(s_list[-1][:700].astype(np.float64) - s_list[-1][0], axis=1).max()
            near = np.linalg.norm(q_list[-1].astype(np.float64) - s_list[-1][0], axis=1).min()
            dense_seen |= bool(near + far < 0.999 * radius)
    s, q = np.concatenate(s_list), np.concatenate(q_list)
    sl, ql = s_sizes.astype(np.int64), q_sizes.astype(np.int64)
    if len(q) == 0 or len(s) == 0:
        pytest.skip("degenerate draw")
    limit = int(rng.choice([1, 16, 40, 80]))
    want, cnt = oracle_ops.radius_search(q, s, ql, sl, radius, limit, return_counts=True)
    grid = SupportGrid(dev(s), dev(sl), radius)
    got, gcnt = grid.query(dev(q), dev(ql), limit, want_counts=True)
    assert np.array_equal(gcnt.cpu().numpy(), cnt), "in-radius counts"
    assert np.array_equal(got.cpu().numpy().astype(np.int64), want), "neighbour rows"
    if dense_seen:
        assert cnt.max() > 512, "generator premise: a query inside a planted dense ball must take the storage-free fallback"


def test_more_than_64_clouds_per_call():
    """The native calls take up to 64 clouds; the drop-in ops cut longer stacks into groups (clouds are independent) and return
    what one call over the whole stack would: 150 small clouds through grid_subsample and radius_search vs the oracle."""
    from lcrnet_amd.modules.ops import grid_subsample, radius_search
    rng = np.random.default_rng(11)
    scan = load_scan("003528")
    sizes = [int(x) for x in rng.integers(30, 400, 150)]
    clouds = [scan[rng.choice(len(scan), n, replace=False)] for n in sizes]
    pts = np.concatenate(clouds).astype(np.float32)
    lens = np.array(sizes, dtype=np.int64)
    sp, sl = grid_subsample(torch.from_numpy(pts).cuda(), torch.from_numpy(lens).cuda(), 0.6)
    wp, wl = oracle_ops.grid_subsample(pts, lens, 0.6)
    assert sl.cpu().tolist() == wl.tolist() and np.array_equal(sp.cpu().numpy().view(np.uint32), wp.view(np.uint32))
    idx = radius_search(sp, torch.from_numpy(pts).cuda(), sl, torch.from_numpy(lens).cuda(), 1.275, 40)
    want = oracle_ops.radius_search(wp, pts, wl, lens, 1.275, 40)
    want = want[:, :idx.shape[1]]
    assert np.array_equal(idx.cpu().numpy(), want)
