"""CPU, build container only: the oracle against the REFERENCE's own compiled C++ (oracle/_ref/libref_ops.so)."""
import numpy as np
import pytest

from conftest import load_scan
from oracle import ops

pytestmark = pytest.mark.skipif(not ops.have_ref(), reason="oracle/_ref/libref_ops.so not built (needs /root/reference)")


@pytest.mark.parametrize("voxel", [0.3, 0.6, 1.2, 2.4, 0.45])
def test_grid_subsample_bit_exact(voxel):
    for name in ["003854", "000026"]:
        xyz = load_scan(name)
        lens = np.array([7000, len(xyz) - 7000])
        a, al = ops.grid_subsample(xyz, lens, voxel)
        b, bl = ops.grid_subsample(xyz, lens, voxel, impl="ref")
        assert np.array_equal(al, bl)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_radius_self_search_identical():
    xyz = load_scan("000560")
    lens = np.array([len(xyz)])
    a = ops.radius_search(xyz, xyz, lens, lens, 1.275, -1)
    b = ops.radius_search(xyz, xyz, lens, lens, 1.275, -1, impl="ref")
    assert a.shape == b.shape            # width == max count
    # self lists have no exact ties on these scans -> identical including order
    assert np.array_equal(a, b)


def test_radius_cross_search_equal_up_to_ties():
    xyz = load_scan("000560")
    lens = np.array([len(xyz)])
    sub, sl = ops.grid_subsample(xyz, lens, 0.6)
    a = ops.radius_search(sub, xyz, sl, lens, 1.275, -1)
    b = ops.radius_search(sub, xyz, sl, lens, 1.275, -1, impl="ref")
    assert a.shape == b.shape
    assert np.array_equal(np.sort(a, 1), np.sort(b, 1))      # same sets; order differs only inside equal-d2 runs


@pytest.mark.parametrize("voxel,cs", [(0.3, [3.299999952316284, 5.099999904632568, 6.599999904632568]), (0.77, [-923.2300415039062, -922.4600219726562, -920.9200439453125])])
def test_grid_subsample_clouds_below_the_voxel_origin(voxel, cs):
    """Planes / lines / a point at coordinates c with floor(c * fl(1/v)) * v > c: the whole cloud lands one cell below its own origin, the
    reference's extent on that axis is (size_t)(-1) + 1 = 0 and its keys wrap mod 2^64 (a plane x = c collapses into one voxel).  The
    oracle must restate exactly that — the compiled reference decides (the GPU twin of this test: tests/test_ops_gpu.py)."""
    c = [np.float32(x) for x in cs]
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
    p, l = ops.grid_subsample(xyz, lens, voxel)
    q, m = ops.grid_subsample(xyz, lens, voxel, impl="ref")
    assert np.array_equal(l, m) and l[2] == 1
    assert np.array_equal(p.view(np.uint32), q.view(np.uint32))
