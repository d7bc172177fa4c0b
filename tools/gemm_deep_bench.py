"""A/B of the K-deep GEMM form (lcr_gemm_debug_deep 1/0) on the encoder's deep shapes with k-contiguous operands (trans_b=1):
outputs compared bit for bit, then both timed with the kernel's own begin/end clock, interleaved rounds."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcrnet_amd import _lib, functional as F  # noqa: E402

SHAPES = [("1_2 kpconv", 127812, 32, 480, 1), ("2_1 kpconv", 51547, 32, 480, 1), ("2_2 kpconv", 51547, 64, 960, 1), ("3_1 kpconv", 19061, 64, 960, 1),
          ("3_2 kpconv", 19061, 128, 1920, 1), ("4_1 kpconv", 6479, 128, 1920, 1), ("4_2 kpconv", 6479, 256, 3840, 1),
          ("3_x unary", 19061, 128, 512, 0), ("3_2 short", 19061, 512, 256, 0), ("4_x unary1", 6479, 256, 1024, 0), ("4_2 short", 6479, 1024, 512, 0),
          ("4_2 unary1", 6479, 256, 512, 0), ("netvlad", 6479, 64, 1024, 0), ("square", 8192, 1024, 1024, 0), ("ragged", 5000, 96, 352, 1)]
if "--short" in sys.argv:
    SHAPES = [("1_2 unary1", 127812, 32, 64, 0), ("1_2 unary2", 127812, 128, 32, 0), ("1_2 short", 127812, 128, 64, 0), ("2_1 unary1", 127812, 32, 128, 0),
              ("2_1 unary2", 51547, 128, 32, 0), ("2_2 unary1", 51547, 64, 128, 0), ("2_2 unary2", 51547, 256, 64, 0), ("2_2 short", 51547, 256, 128, 0),
              ("2_3 unary1", 51547, 64, 256, 0), ("3_1 unary2", 19061, 256, 64, 0), ("3_2 unary1", 19061, 128, 256, 0), ("3_2 unary2", 19061, 512, 128, 0),
              ("4_1 unary2", 6479, 512, 128, 0), ("4_2 unary2", 6479, 1024, 256, 0), ("tiny", 700, 256, 64, 0)]
DEEP_MODE = 2 if "--short" in sys.argv else 1


def main():
    dev = torch.device("cuda")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    res = {}
    tensors = []
    for tag, M, N, K, rd in SHAPES:
        a = torch.randn(M, K, device=dev)
        b = torch.randn(N, K, device=dev)
        bias = torch.randn(N, device=dev)
        div = torch.rand(M, device=dev) + 1 if rd else None
        seg = torch.tensor([M // 8 - 3] * 7 + [M - 7 * (M // 8 - 3)], dtype=torch.int64, device=dev)
        g = 32 if N % 32 == 0 and ((N // 32) & (N // 32 - 1)) == 0 else 0
        kw = dict(trans_b=True, bias=bias, rowdiv=div, seg_len=seg if g else None, groups=g)
        lib.lcr_gemm_debug_deep(0)
        c0, s0 = F.gemm(a, b, **kw)
        lib.lcr_gemm_debug_deep(DEEP_MODE)
        c1, s1 = F.gemm(a, b, **kw)
        torch.cuda.synchronize()
        same = torch.equal(c0, c1)
        err = (c0 - c1).abs().max().item()
        serr = 0.0 if s0 is None else ((s0.sum(0) - s1.sum(0)).abs() / (s0.sum(0).abs() + 1e-30)).max().item()
        ref = (a.double() @ b.double().t())
        if div is not None:
            ref = ref / div.double()[:, None]
        ref = ref + bias.double()
        e64 = ((c1.double() - ref).abs().max() / ref.abs().max()).item()
        res[tag] = (same, err, serr, e64)
        tensors.append((tag, M, N, K, a, b, kw))
    times = {(t[0], m): [] for t in tensors for m in (0, 1)}
    for rnd in range(5):
        for mode in (0, 1):
            lib.lcr_gemm_debug_deep(DEEP_MODE if mode else 0)
            timer = F.KernelTimer({"gemm"})
            F.set_timer(timer)
            for tag, M, N, K, a, b, kw in tensors:
                for _ in range(4):
                    F.gemm(a, b, **kw)
            torch.cuda.synchronize()
            F.set_timer(None)
            recs = timer.records()["gemm"]
            i = 0
            for tag, M, N, K, a, b, kw in tensors:
                ks = [recs[i + j][1] for j in range(4)]
                i += 4
                times[(tag, mode)].append(min(ks[1:]))
    lib.lcr_gemm_debug_deep(-1)
    tot0 = tot1 = 0.0
    for tag, M, N, K, a, b, kw in tensors:
        t0, t1 = sorted(times[(tag, 0)])[2], sorted(times[(tag, 1)])[2]
        tot0 += t0
        tot1 += t1
        same, err, serr, e64 = res[tag]
        print(f"{tag:12s} M={M:7d} N={N:5d} K={K:5d}  old {t0*1e6:7.1f} us {2.0*M*N*K/t0/1e12:6.1f} TF | deep {t1*1e6:7.1f} us {2.0*M*N*K/t1/1e12:6.1f} TF  x{t0/t1:4.2f}"
              f"   identical={same} maxdiff={err:.1e} stats_rel={serr:.1e} rel_err_vs_fp64={e64:.1e}")
    print(f"sum old {tot0*1e6:.1f} us  deep {tot1*1e6:.1f} us")


if __name__ == "__main__":
    main()
