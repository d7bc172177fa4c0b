import hashlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
LIMITS = [74, 68, 70, 67]
NUM_STAGES, VOXEL, RADIUS = 4, 0.3, 1.275
DEMO_SCANS = ["000026", "000560", "000958", "003528", "003854", "004481"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="session")
def ops_golden():
    return np.load(os.path.join(GOLDEN, "ops_golden.npz"), allow_pickle=False)


def load_scan(name):
    """Demo scans are committed fixtures; syn0/syn1 are regenerated (deterministic) and voxelised by the ORACLE."""
    if name.startswith("syn"):
        import lcrnet_amd.synthetic as synthetic
        from oracle import ops
        raw = synthetic.synthetic_scan(int(name[3:]))
        ds, _ = ops.grid_subsample(raw, np.array([len(raw)]), VOXEL)
        return ds
    return np.load(os.path.join(GOLDEN, "scans", name + ".npy"))


def search_specs(pts, lens, limits=LIMITS, radius=RADIUS):
    """The 10 radius searches of precompute_data_stack_mode (data.py:28-66) as (name, q, s, ql, sl, radius, limit)."""
    specs = []
    r = radius
    n = len(pts)
    for i in range(n):
        specs.append((f"neighbors{i}", pts[i], pts[i], lens[i], lens[i], r, limits[i]))
        if i < n - 1:
            specs.append((f"subsampling{i}", pts[i + 1], pts[i], lens[i + 1], lens[i], r, limits[i]))
            specs.append((f"upsampling{i}", pts[i], pts[i + 1], lens[i], lens[i + 1], r * 2, limits[i + 1]))
        r *= 2
    return specs
