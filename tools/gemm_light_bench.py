"""Kernel-own times of the light (K <= 256) GEMM form on the encoder's unary / shortcut shapes, per tile shape, with and without the
GroupNorm statistics epilogue; beside each the HBM floor (A + C once at 4.9 TB/s) and the 100 TFLOP/s time."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcrnet_amd import functional as F  # noqa: E402
from lcrnet_amd import _lib  # noqa: E402

SHAPES = [(127812, 32, 64), (127812, 128, 32), (127812, 128, 64), (127812, 32, 128), (51547, 128, 32), (51547, 64, 128), (51547, 256, 64),
          (51547, 256, 128), (51547, 64, 256), (19061, 256, 64), (19061, 128, 256), (19061, 512, 128), (19061, 512, 256), (6479, 512, 128),
          (6479, 1024, 256), (6479, 64, 1024)]
names = {0: "auto", 1: "128x128", 2: "128x64", 3: "128x32", 4: "64x64", 5: "64x128"}


def main():
    dev = torch.device("cuda")
    L = ctypes.CDLL(_lib.LIB_PATH)
    tiles = [0, 2, 3, 4, 5] if "--tiles" in sys.argv else [0]
    print("%-22s %8s %8s | %s" % ("shape", "hbm us", "100TF us", "  ".join("%7s/st" % names[t] for t in tiles) + "  (us without / with statistics)"))
    for M, N, K in SHAPES:
        a = torch.randn(M, K, device=dev)
        b = torch.randn(N, K, device=dev)
        bias = torch.randn(N, device=dev)
        seg = torch.tensor([M // 8 - 3] * 7 + [M - 7 * (M // 8 - 3)], dtype=torch.int64, device=dev)
        cells = []
        for tile in tiles:
            if tile == 3 and N > 32:
                cells.append("      -      ")
                continue
            L.lcr_gemm_debug_force_tile(tile)
            res = []
            for stats in (0, 1):
                kw = dict(trans_b=True, bias=bias, seg_len=seg if stats else None, groups=32 if stats else 0)
                for _ in range(3):
                    F.gemm(a, b, **kw)
                torch.cuda.synchronize()
                t = F.KernelTimer({"gemm"})
                F.set_timer(t)
                for _ in range(10):
                    F.gemm(a, b, **kw)
                torch.cuda.synchronize()
                F.set_timer(None)
                own = [kk for _, kk, _ in t.records()["gemm"] if kk is not None]
                res.append(sorted(own)[len(own) // 2] * 1e6)
            cells.append("%6.1f/%6.1f" % tuple(res))
        L.lcr_gemm_debug_force_tile(0)
        print("%7d x%5d x%5d %8.1f %8.1f | %s" % (M, N, K, (M * K + M * N) * 4 / 4.9e12 * 1e6, 2.0 * M * N * K / 1e14 * 1e6, "  ".join(cells)))


if __name__ == "__main__":
    main()
