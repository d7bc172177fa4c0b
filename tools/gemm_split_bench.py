"""fp32-MFMA K-deep GEMM (lcr_gemm_f32, pre-transposed B) vs the split-bf16 form (lcr_gemm_f32_bsplit: fp32 operands as three bf16 terms, six
cross products on the bf16 matrix cores) on the encoder's deep shapes: kernel begin-to-end times (interleaved rounds) and the error of BOTH
against an fp64 product of the same fp32 inputs — max |c - ref| / max |ref| and the rms ratio.  One JSON line per shape + a summary
(--table: one short text row per shape instead)."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcrnet_amd import _lib, functional as F  # noqa: E402

SHAPES = [("1_2 kpconv", 127812, 32, 480, 1), ("2_1 kpconv", 51547, 32, 480, 1), ("2_2 kpconv", 51547, 64, 960, 1), ("3_1 kpconv", 19061, 64, 960, 1),
          ("3_2 kpconv", 19061, 128, 1920, 1), ("4_1 kpconv", 6479, 128, 1920, 1), ("4_2 kpconv", 6479, 256, 3840, 1),
          ("3_x unary", 19061, 128, 512, 0), ("4_x unary1", 6479, 256, 1024, 0), ("4_2 short", 6479, 1024, 512, 0),
          ("4_2 unary1", 6479, 256, 512, 0), ("square", 8192, 1024, 1024, 0), ("ragged", 5000, 96, 352, 1)]


if "--splitk-proxy" in sys.argv:       # the same work as a split-K of the stage-3/4 contractions would launch: S x the rows, K / S deep (tiles x S, steps / S)
    SHAPES = [("4_2", 6479, 256, 3840, 1), ("4_2 K/2", 12958, 256, 1920, 1), ("4_2 K/3", 19437, 256, 1280, 1), ("4_2 K/4", 25916, 256, 960, 1),
              ("3_2", 19061, 128, 1920, 1), ("3_2 K/2", 38122, 128, 960, 1), ("3_2 K/3", 57183, 128, 640, 1),
              ("4_1", 6479, 128, 1920, 1), ("4_1 K/2", 12958, 128, 960, 1), ("4_1 K/4", 25916, 128, 480, 1),
              ("4_x unary1", 6479, 256, 1024, 0), ("4_x K/2", 12958, 256, 512, 0), ("3_1", 19061, 64, 960, 1), ("3_1 K/2", 38122, 64, 480, 1)]


def main():
    dev = torch.device("cuda")
    L = _lib.lib()
    sp = lambda t: _lib.stream_ptr(t.device)
    items, out = [], []
    for tag, M, N, K, rd in SHAPES:
        g = torch.Generator(device=dev).manual_seed(hash(tag) % 1000)
        a = torch.randn(M, K, device=dev, generator=g) * torch.rand(M, 1, device=dev, generator=g) * 4      # rows of different scales
        if rd:
            a = torch.where(torch.rand(M, K, device=dev, generator=g) < 0.35, torch.zeros_like(a), a.abs())     # the aggregate: non-negative, 35 % zeros
        b = torch.randn(N, K, device=dev, generator=g) * 0.05
        bias = torch.randn(N, device=dev, generator=g)
        div = (torch.rand(M, device=dev, generator=g) * 40 + 1).floor() if rd else None
        seg = torch.tensor([M // 8 - 3] * 7 + [M - 7 * (M // 8 - 3)], dtype=torch.int64, device=dev)
        gr = 32 if N % 32 == 0 and ((N // 32) & (N // 32 - 1)) == 0 else 0
        planes = F.split_bf16x3(b)
        # the planes really are an exact three-term split
        pl = F.unsplit_bf16x3(planes)
        assert torch.equal(pl[0] + pl[1] + pl[2], b) and torch.equal(pl[0], b.to(torch.bfloat16).float())
        c0 = torch.empty(M, N, device=dev)
        c1 = torch.empty(M, N, device=dev)
        st0 = torch.zeros(8, 8, max(gr, 1), 2, dtype=torch.float64, device=dev)
        st1 = torch.zeros_like(st0)

        def run0(c=c0, st=st0, a=a, b=b, bias=bias, div=div, seg=seg, gr=gr, M=M, N=N, K=K, rd=rd):
            _lib.check(L.lcr_gemm_f32(_lib.ptr(a), _lib.ptr(b), _lib.ptr(c), M, N, K, 0, 1, _lib.ptr(bias), _lib.ptr(div) if rd else None,
                                      _lib.ptr(seg) if gr else None, 8, gr, _lib.ptr(st) if gr else None, sp(a)), "gemm")

        def run1(c=c1, st=st1, a=a, planes=planes, bias=bias, div=div, seg=seg, gr=gr, M=M, N=N, K=K, rd=rd):
            _lib.check(L.lcr_gemm_f32_bsplit(_lib.ptr(a), _lib.ptr(planes), _lib.ptr(c), M, N, K, _lib.ptr(bias), _lib.ptr(div) if rd else None,
                                             _lib.ptr(seg) if gr else None, 8, gr, _lib.ptr(st) if gr else None, sp(a)), "bsplit")
        run0()
        run1()
        torch.cuda.synchronize()
        ref = a.double() @ b.double().t()
        if rd:
            ref = ref / div.double()[:, None]
        ref = ref + bias.double()
        scale = ref.abs().max()
        e0, e1 = ((c0.double() - ref).abs().max() / scale).item(), ((c1.double() - ref).abs().max() / scale).item()
        r0, r1 = (c0.double() - ref).pow(2).mean().sqrt().item(), (c1.double() - ref).pow(2).mean().sqrt().item()
        serr = 0.0
        if gr:                                              # sums relative to their Cauchy-Schwarz bound sqrt(n * sumsq), sums of squares relative
            s0, s1 = st0.sum(0), st1.sum(0)
            n_el = seg.double()[:, None] * (N // gr)
            serr = max(((s0[..., 0] - s1[..., 0]).abs() / (n_el * s0[..., 1]).sqrt()).max().item(),
                       ((s0[..., 1] - s1[..., 1]).abs() / s0[..., 1]).max().item())
        items.append((tag, M, N, K, run0, run1))
        out.append({"shape": tag, "M": M, "N": N, "K": K, "max_err_fp32_mfma": e0, "max_err_split": e1, "rms_err_fp32_mfma": r0, "rms_err_split": r1,
                    "max_diff_between": (c0 - c1).abs().max().item() / scale.item(), "stats_rel_diff": serr})
        del ref
    times = {(t[0], m): [] for t in items for m in (0, 1)}
    for rnd in range(5):
        for mode in (0, 1):
            timer = F.KernelTimer({"gemm"})
            F.set_timer(timer)
            for it in items:
                for _ in range(4):
                    it[4 + mode]()
            torch.cuda.synchronize()
            F.set_timer(None)
            recs = timer.records()["gemm"]
            i = 0
            for it in items:
                times[(it[0], mode)].append(min(recs[i + j][1] for j in range(1, 4)))
                i += 4
    tot0 = tot1 = 0.0
    for it, o in zip(items, out):
        tag, M, N, K = it[:4]
        t0, t1 = sorted(times[(tag, 0)])[2], sorted(times[(tag, 1)])[2]
        tot0 += t0
        tot1 += t1
        o.update({"us_fp32_mfma": round(t0 * 1e6, 1), "us_split": round(t1 * 1e6, 1), "speedup": round(t0 / t1, 3),
                  "tflops_fp32_mfma": round(2.0 * M * N * K / t0 / 1e12, 1), "tflops_equiv_split": round(2.0 * M * N * K / t1 / 1e12, 1)})
        if "--table" in sys.argv:
            print(tag, M, N, K, "fp32", o["us_fp32_mfma"], "split", o["us_split"], "x", o["speedup"], "TFeq", o["tflops_equiv_split"],
                  "err %.1e %.1e" % (o["max_err_fp32_mfma"], o["max_err_split"]), "stats %.1e" % o["stats_rel_diff"])
        else:
            print(json.dumps(o))
    print(json.dumps({"sum_us_fp32_mfma": round(tot0 * 1e6, 1), "sum_us_split": round(tot1 * 1e6, 1), "speedup": round(tot0 / tot1, 3)}))


if __name__ == "__main__":
    main()
