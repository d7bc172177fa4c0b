"""CPU: liblcr_hip.so loads (no GPU needed) and exports every function include/lcr_hip.h declares."""
import ctypes
import os
import re

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    import lcrnet_amd._lib as L
    assert os.path.exists(L.LIB_PATH), "build first: python lcr-net_amd/csrc/build.py"
    lib = ctypes.CDLL(L.LIB_PATH)
    hdr = open(os.path.join(ROOT, "include", "lcr_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = re.findall(r"\b(lcr_[a-z0-9_]+)\s*\(", hdr)
    assert len(names) >= 8
    for n in sorted(set(names)):
        assert hasattr(lib, n), f"{n} declared in include/lcr_hip.h but not exported"
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
