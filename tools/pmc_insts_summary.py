"""Instruction budget of a step from one rocprofv3 --pmc pass over tools/enc_profile.py (csv) -> markdown.

    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES \\
              --kernel-trace --output-format csv -d DIR -o i -- python tools/enc_profile.py PASSES
    python tools/pmc_insts_summary.py DIR/.../i_counter_collection.csv PASSES [BATCHES=4] > profiles/<name>.md

Counters are wavefront instructions summed over the chip.  Encoder kernels are divided by PASSES, pre-processing kernels (run once per
distinct batch before the passes) by BATCHES: both columns are "per 8-scan step".  `valu issue us` = VALU instructions x 4 cycles (a
64-lane instruction on a 16-lane SIMD) / (1024 SIMDs x 2.4 GHz): the time the chip's VALU issue ports would need for this kernel alone if
they never idled — the quantity the step is short of (LABNOTES.md §4.4)."""
import collections
import csv
import sys

PRE = ("k_gs_", "k_rx_", "k_scan", "k_grid_", "k_radius_query", "k_pc_", "k_pre", "k_seg", "k_gather_i", "k_fill", "k_iota")


def fam(k):
    k = k.replace("lcr::", "").replace("void ", "")
    for f in ("k_gemm_f32_deep", "k_gemm_f32_bsplit", "k_gemm_f32<64, 64, 2, 2, false, true, true, true, true", "k_gemm_f32<", "k_kpconv_aggregate_vec",
              "k_gn_apply", "k_maxpool", "k_kpconv_cin1", "k_radius_query", "k_gs_hashorder", "k_gs_reduce", "k_rx_", "k_scan_lookback", "k_gs_", "k_grid_"):
        if f in k:
            return {"k_gemm_f32<64, 64, 2, 2, false, true, true, true, true": "k_gemm_f32 light + normalise-on-load", "k_gemm_f32<": "k_gemm_f32 light (others)"}.get(f, f + "*")
    return k.split("(")[0].split("<")[0][:40]


def main():
    path, passes = sys.argv[1], int(sys.argv[2])
    batches = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        f = fam(r["Kernel_Name"])
        agg[f][r["Counter_Name"]] += float(r["Counter_Value"])
        d = r["Dispatch_Id"]
        if d not in seen[f]:
            seen[f].add(d)
            agg[f]["_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    rows, tot = [], collections.defaultdict(float)
    for f, c in agg.items():
        pre = any(f.startswith(p) or p in f for p in PRE)
        div = batches if pre else passes
        row = {"f": f, "pre": pre, "launches": len(seen[f]) / div, "us": c["_ns"] / 1e3 / div}
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_MFMA"):
            row[k] = c.get(k, 0.0) / div
            tot[k] += row[k]
        row["issue_us"] = row["SQ_INSTS_VALU"] * 4 / (1024 * 2400.0)
        tot["us"] += row["us"]
        tot["issue_us"] += row["issue_us"]
        rows.append(row)
    rows.sort(key=lambda r: -r["SQ_INSTS_VALU"])
    print("# Wavefront instructions per 8-scan step, by kernel family (rocprofv3 --pmc over tools/enc_profile.py %d; pre-processing kernels per batch)\n" % passes)
    print("| kernel family | side | launches | kernel us (alone) | VALU M | valu issue us | SALU M | LDS M | VMEM rd M | VMEM wr M | MFMA M |")
    print("|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    for r in rows[:24]:
        print("| `%s` | %s | %.1f | %.1f | %.2f | %.1f | %.2f | %.2f | %.3f | %.3f | %.3f |" % (
            r["f"], "pre" if r["pre"] else "enc", r["launches"], r["us"], r["SQ_INSTS_VALU"] / 1e6, r["issue_us"], r["SQ_INSTS_SALU"] / 1e6,
            r["SQ_INSTS_LDS"] / 1e6, r["SQ_INSTS_VMEM_RD"] / 1e6, r["SQ_INSTS_VMEM_WR"] / 1e6, r["SQ_INSTS_MFMA"] / 1e6))
    print("| **all kernels** | | | %.1f | %.2f | %.1f | %.2f | %.2f | %.3f | %.3f | %.3f |" % (
        tot["us"], tot["SQ_INSTS_VALU"] / 1e6, tot["issue_us"], tot["SQ_INSTS_SALU"] / 1e6, tot["SQ_INSTS_LDS"] / 1e6,
        tot["SQ_INSTS_VMEM_RD"] / 1e6, tot["SQ_INSTS_VMEM_WR"] / 1e6, tot["SQ_INSTS_MFMA"] / 1e6))


if __name__ == "__main__":
    main()
