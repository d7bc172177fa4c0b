"""Runs a few K-deep GEMM shapes in both forms (register-staged / LDS-direct) for a rocprofv3 --pmc pass."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcrnet_amd import _lib, functional as F  # noqa: E402

SHAPES = [(19061, 128, 1920), (6479, 256, 3840), (8192, 1024, 1024), (51547, 64, 960)]
dev = torch.device("cuda")
lib = ctypes.CDLL(_lib.LIB_PATH)
for M, N, K in SHAPES:
    a = torch.randn(M, K, device=dev)
    b = torch.randn(N, K, device=dev)
    for mode in (0, 1):
        lib.lcr_gemm_debug_deep(mode)
        for _ in range(3):
            F.gemm(a, b, trans_b=True)
    torch.cuda.synchronize()
