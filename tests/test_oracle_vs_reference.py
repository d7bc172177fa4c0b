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
