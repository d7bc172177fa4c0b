"""GPU: the `utils.ext` replacement of INTEGRATION.md (tools/integration/utils_ext.py — ctypes on the C ABI, the reference's three
function names and signatures, utils/extensions/pybind.cpp:7-24) against the oracle."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from conftest import load_scan
from oracle import ops as oracle_ops

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ext():
    spec = importlib.util.spec_from_file_location("utils_ext_stub", os.path.join(ROOT, "tools", "integration", "utils_ext.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_grid_subsampling_and_radius_neighbors(ext):
    a, b = load_scan("003854"), load_scan("000958")
    pts = torch.from_numpy(np.concatenate([a, b]))
    lens = torch.tensor([len(a), len(b)], dtype=torch.int64)
    sp, sl = ext.grid_subsampling(pts, lens, 0.6)                                     # CPU tensors in, like the reference call sites
    wp, wl = oracle_ops.grid_subsample(pts.numpy(), lens.numpy(), 0.6)
    assert sl.cpu().tolist() == wl.tolist() and np.array_equal(sp.cpu().numpy().view(np.uint32), wp.view(np.uint32))
    idx = ext.radius_neighbors(sp, pts, sl, lens, 1.275)                              # full width, padded with the support count
    want = oracle_ops.radius_search(wp, pts.numpy(), wl, lens.numpy(), 1.275, -1)
    assert idx.dtype == torch.int64 and tuple(idx.shape) == want.shape and np.array_equal(idx.cpu().numpy(), want)
    with pytest.raises(RuntimeError):
        ext.radius_neighbors(sp, pts, sl, torch.tensor([len(a), len(b) + 5]), 1.275)  # lengths beyond the rows: refused, not truncated


def test_radius_filter(ext):
    g = torch.Generator().manual_seed(0)
    pts = torch.rand(900, 3, generator=g) * torch.tensor([40.0, 40.0, 3.0])
    lens = [500, 400]
    masks, kept = ext.radius_filter(pts, torch.tensor(lens), 2.4)
    o = 0
    for m, k, n in zip(masks, kept, lens):                                            # the reference's loop (radius_filter.cpp:19-29), in numpy fp64
        p = pts[o:o + n].double().numpy()
        keep = np.zeros(n, bool)
        keep[0] = True
        for i in range(1, n):
            keep[i] = bool((np.linalg.norm(p[i] - p[keep], axis=1) > 2.4).all())
        assert np.array_equal(m.cpu().numpy(), keep) and int(k) == int(keep.sum())
        o += n
