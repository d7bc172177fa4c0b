"""Micro-benchmark of lcr_gemm_f32 on the encoder's shapes (batch of 8 scans): TFLOP/s per launch, HIP events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcrnet_amd import functional as F  # noqa: E402

SHAPES = [  # (tag, M, N, K, transB, stats, rowdiv)
    ("1_2 unary1", 128000, 32, 64, 1, 1, 0), ("1_2 kpconv", 128000, 32, 480, 0, 1, 1), ("1_2 unary2", 128000, 128, 32, 1, 1, 0),
    ("1_2 shortcut", 128000, 128, 64, 1, 1, 0), ("2_2 kpconv", 51000, 64, 960, 0, 1, 1), ("2_2 unary2", 51000, 256, 64, 1, 1, 0),
    ("3_2 kpconv", 19000, 128, 1920, 0, 1, 1), ("3_2 unary2", 19000, 512, 128, 1, 1, 0), ("4_2 kpconv", 6500, 256, 3840, 0, 1, 1),
    ("4_2 unary2", 6500, 1024, 256, 1, 1, 0), ("4_3 unary1", 6500, 256, 1024, 1, 1, 0), ("netvlad assign", 6500, 64, 1024, 0, 0, 0),
    ("big square", 8192, 1024, 1024, 0, 0, 0),
]


def main():
    import ctypes
    from lcrnet_amd import _lib
    dev = torch.device("cuda")
    tiles = [0] if "--tiles" not in sys.argv else [0, 1, 2, 3, 4, 5]
    names = {0: "auto", 1: "128x128", 2: "128x64", 3: "128x32", 4: "64x64", 5: "64x128"}
    for tile in tiles:
      ctypes.CDLL(_lib.LIB_PATH).lcr_gemm_debug_force_tile(tile)
      print("---- tile", names[tile])
      for tag, M, N, K, tb, stats, rd in SHAPES:
        if tile and ((tile == 3 and N > 32) or (tile in (2, 4) and N > 64 and False)):
            continue
        a = torch.randn(M, K, device=dev)
        b = torch.randn((N, K) if tb else (K, N), device=dev)
        bias = torch.randn(N, device=dev)
        div = torch.rand(M, device=dev) + 1 if rd else None
        seg = torch.tensor([M // 8 - 3] * 7 + [M - 7 * (M // 8 - 3)], dtype=torch.int64, device=dev)
        if "--nostats" in sys.argv:
            stats = 0
        kw = dict(trans_b=bool(tb), bias=bias, rowdiv=div, seg_len=seg if stats else None, groups=32 if stats else 0)
        for _ in range(3):
            F.gemm(a, b, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        reps = 20
        for _ in range(reps):
            F.gemm(a, b, **kw)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps * 1e-3
        print(f"{tag:16s} M={M:7d} N={N:5d} K={K:5d}  {t*1e6:9.1f} us  {2.0*M*N*K/t/1e12:7.2f} TFLOP/s  A-stream {M*K*4/t/1e9:8.1f} GB/s")


if __name__ == "__main__":
    main()
