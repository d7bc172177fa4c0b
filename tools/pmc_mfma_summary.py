"""Per-kernel MFMA / wave-state counters from one rocprofv3 --pmc pass (csv) -> markdown table.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY \\
              SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d DIR -o m -- python tools/enc_profile.py 3
    python tools/pmc_mfma_summary.py DIR/.../m_counter_collection.csv > profiles/<name>.md

mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs * SQ_BUSY_CU_CYCLES): share of CU-busy time in which a SIMD's matrix pipe is
occupied.
wait / issue-stall / active are SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES.
"""
import collections
import csv
import sys


def fam(k):
    k = k.replace("lcr::", "").replace("void ", "")
    for f in ("k_gemm_f32_deep<64, 64", "k_gemm_f32_deep<128, 32", "k_gemm_f32<64, 64, 2, 2, false, true, true, true, true", "k_gemm_f32<64, 64, 2, 2, false, false", "k_gemm_f32<64, 64, 2, 2, false, true", "k_gemm_f32<128, 32", "k_gemm_f32<128, 128",
              "k_kpconv_aggregate_vec<int, 32", "k_kpconv_aggregate_vec<int, 64", "k_kpconv_aggregate_vec<int, 128",
              "k_kpconv_aggregate_vec<int, 256", "k_gn_apply", "k_maxpool", "k_kpconv_cin1", "k_attention", "k_radius_query"):
        if f in k:
            return f + (">" if "<" in f else "")
    return k.split("(")[0][:60]


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(sys.argv[1])):
        f = fam(r["Kernel_Name"])
        agg[f][r["Counter_Name"]] += float(r["Counter_Value"])
        d = r["Dispatch_Id"]
        if d not in seen[f]:
            seen[f].add(d)
            agg[f]["_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print("# MFMA and wave-state counters per kernel family (rocprofv3 --pmc, encoder + NetVLAD alone)\n")
    print("| kernel | launches | avg us | mfma_busy | MFMA F32 Mops/launch | wait | issue-stall | active |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|")
    for f, c in sorted(agg.items(), key=lambda kv: -kv[1]["_ns"])[:16]:
        n = len(seen[f])
        busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(4.0 * c["SQ_BUSY_CU_CYCLES"], 1.0)
        wc = max(c["SQ_WAVE_CYCLES"], 1.0)
        print("| `%s` | %d | %.1f | %.3f | %.1f | %.2f | %.2f | %.2f |" % (
            f, n, c["_ns"] / n / 1e3, busy, c["SQ_INSTS_VALU_MFMA_MOPS_F32"] / n / 1e6,
            c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc))


if __name__ == "__main__":
    main()
