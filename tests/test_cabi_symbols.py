"""CPU: liblcr_hip.so loads (no GPU needed) and exports every function include/lcr_hip.h declares."""
import ctypes
import os
import re

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    import lcrnet_amd._lib as L
    assert os.path.exists(L.LIB_PATH), "build first: python lcr-net_amd/csrc/build.py"
    lib = ctypes.CDLL(L.LIB_PATH)
    for header, least in (("lcr_hip.h", 70), ("lcr_hip_debug.h", 5)):        # the product ABI, and the test / tuning hooks beside it
        hdr = open(os.path.join(ROOT, "include", header)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        names = re.findall(r"\b(lcr_[a-z0-9_]+)\s*\(", hdr)
        assert len(set(names)) >= least, (header, len(set(names)))
        for n in sorted(set(names)):
            assert hasattr(lib, n), f"{n} declared in include/{header} but not exported"
        if header == "lcr_hip.h":
            assert not [n for n in names if "debug" in n], "debug hooks belong in include/lcr_hip_debug.h"
    assert lib.lcr_version() >= 1


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from lcrnet_amd.modules.ops import grid_subsample, radius_search
    p = torch.zeros(4, 3)
    l = torch.tensor([4])
    with pytest.raises(RuntimeError):
        grid_subsample(p, l, 0.3)
    with pytest.raises(RuntimeError):
        radius_search(p, p, l, l, 1.0, 4)


def test_precompute_layout_is_a_host_function_with_a_consistent_arena():
    """lcr_precompute_layout needs no GPU: offsets are 256-B aligned, ordered, non-overlapping and sized for the capacities;
    the ctypes mirror of LcrPrecomputeLayout has the C struct's size (a mismatch would corrupt the offsets silently)."""
    import lcrnet_amd._lib as L
    from lcrnet_amd.data import MAX_STAGES, PrecomputeLayout
    lib = L.lib()
    for n0, B, limits, ups in [(16963, 1, [74, 68, 70, 67], 1), (127812, 8, [64, 65, 74, 80], 1), (500, 3, [8, 9], 0), (0, 2, [4, 4, 4], 1)]:
        lay = PrecomputeLayout()
        lim = (ctypes.c_int * len(limits))(*limits)
        rc = lib.lcr_precompute_layout(n0, B, len(limits), ctypes.cast(lim, ctypes.c_void_p), ups, 0, ctypes.addressof(lay))
        assert rc == 0
        assert lay.num_stages == len(limits) and lay.B == B and lay.upsampling == ups
        cap = max(n0, 1)
        spans = []
        for i in range(len(limits)):
            assert lay.cap[i] == cap and lay.limits[i] == limits[i]
            if i > 0:
                spans += [(lay.off_points[i], cap * 12), (lay.off_lengths[i], B * 8)]
            spans += [(lay.off_order[i], cap * 4), (lay.off_neighbors[i], cap * limits[i] * 4)]
            if i + 1 < len(limits):
                spans.append((lay.off_subsampling[i], cap * limits[i] * 4))
                if ups:
                    spans.append((lay.off_upsampling[i], cap * limits[i + 1] * 4))
        spans.sort()
        for (o, n), (o2, _) in zip(spans, spans[1:]):
            assert o % 256 == 0 and o + n <= o2
        assert spans[-1][0] + spans[-1][1] <= lay.out_bytes and lay.ws_bytes > 0
    bad = PrecomputeLayout()
    lim = (ctypes.c_int * 2)(4, 0)
    assert lib.lcr_precompute_layout(10, 1, 2, ctypes.cast(lim, ctypes.c_void_p), 1, 0, ctypes.addressof(bad)) != 0     # limit < 1
    assert lib.lcr_precompute_layout(10, 65, 2, ctypes.cast(lim, ctypes.c_void_p), 1, 0, ctypes.addressof(bad)) != 0    # B > 64
    raw = PrecomputeLayout()
    lim4 = (ctypes.c_int * 4)(8, 8, 8, 8)
    assert lib.lcr_precompute_layout(3000, 2, 4, ctypes.cast(lim4, ctypes.c_void_p), 1, 20000, ctypes.addressof(raw)) == 0
    assert raw.n_raw == 20000 and raw.off_points[0] % 256 == 0 and raw.off_lengths[0] >= raw.off_points[0] + 20000 * 12   # stage 0 in the arena
    assert MAX_STAGES == 8


def test_stats_arena_hands_out_disjoint_zeroed_tables():
    """functional.stats_arena on the CPU (no kernel involved): consecutive tables do not overlap, nesting restores the outer
    arena, an exhausted arena falls back to a fresh zero tensor."""
    import torch
    from lcrnet_amd import functional as F
    dev = torch.device("cpu")
    with F.stats_arena(dev, entries=F.GN_REPLICAS * 2 * 32 * 2 * 3):
        a = F._zero_stats(2, 32, dev)
        b = F._zero_stats(2, 32, dev)
        a.fill_(1.0)
        assert float(b.sum()) == 0.0 and a.shape == (F.GN_REPLICAS, 2, 32, 2)
        with F.stats_arena(dev, entries=16):
            c = F._zero_stats(1, 32, dev)           # does not fit: fresh tensor
            assert float(c.sum()) == 0.0
        d = F._zero_stats(2, 32, dev)               # third table of the OUTER arena
        assert float(d.sum()) == 0.0 and d.data_ptr() != a.data_ptr() and d.data_ptr() != b.data_ptr()
        e = F._zero_stats(2, 32, dev)               # outer arena exhausted
        assert float(e.sum()) == 0.0
    assert F._ARENA.buf is None


def test_library_is_built_without_fp_contraction():
    """Bit-exact neighbour rows need d2 = ((dx*dx + dy*dy) + dz*dz) with every operation rounded (nanoflann.hpp:432-440); hipcc's default
    -ffp-contract=fast would fuse a product into the add.  The build recipe must keep contraction off (the exact helpers of common.h
    also carry the pragma; the GPU test test_distance_arithmetic_is_not_contracted checks the result)."""
    import importlib.util
    import os
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("lcr_build", os.path.join(here, "lcr-net_amd", "csrc", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert "-ffp-contract=off" in mod.FLAGS
    src = open(os.path.join(here, "lcr-net_amd", "csrc", "common.h")).read()
    assert src.count("#pragma clang fp contract(off)") >= 4
