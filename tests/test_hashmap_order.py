"""CPU: the product's closed-form replay of libstdc++'s unordered_map iteration order (host mirror of the HIP
kernel k_gs_hashorder, exported as lcr_hashmap_order_host) against the real container (oracle)."""
import ctypes

import numpy as np
import pytest

from oracle import ops


def _host_order(keys):
    import lcrnet_amd._lib as L
    lib = ctypes.CDLL(L.LIB_PATH)
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    order = np.empty(len(keys), dtype=np.int64)
    rc = lib.lcr_hashmap_order_host(ctypes.c_void_p(keys.ctypes.data), ctypes.c_int64(len(keys)), ctypes.c_void_p(order.ctypes.data))
    assert rc == 0
    return order


@pytest.mark.parametrize("n", [0, 1, 2, 12, 13, 14, 28, 29, 30, 59, 60, 541, 542, 6270, 42043, 42044, 100000])
def test_host_mirror_matches_unordered_map(n):
    rng = np.random.default_rng(n)
    keys = rng.choice(1 << 34, size=n, replace=False).astype(np.uint64)
    assert np.array_equal(_host_order(keys), ops.hashmap_order(keys))


def test_dense_voxel_like_keys():
    rng = np.random.default_rng(7)
    keys = rng.permutation(60000)[:17000].astype(np.uint64)
    assert np.array_equal(_host_order(keys), ops.hashmap_order(keys))


def test_huge_keys_wrap():
    rng = np.random.default_rng(3)
    keys = (rng.integers(0, 1 << 62, size=5000, dtype=np.uint64) * np.uint64(4) + np.arange(5000, dtype=np.uint64) % np.uint64(4))
    keys = np.unique(keys)
    rng.shuffle(keys)
    assert np.array_equal(_host_order(keys), ops.hashmap_order(keys))
