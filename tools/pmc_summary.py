"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output) into per-kernel HBM traffic per launch.

    python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> profiles/<name>

Writes <name>.md (table) and <name>.json ({kernel family: bytes per launch}).  FETCH_SIZE / WRITE_SIZE are reported by
rocprofv3 in KiB.  Correction (MI355X_MICROARCH.md §HBM): on gfx950 FETCH_SIZE counts 128-B fabric requests at 64 B, i.e.
HALF the bytes of wide coalesced reads — the table lists raw and x2-corrected fetch bytes; WRITE_SIZE is taken as is.
"""
import collections
import csv
import json
import sys


def load(path, name):
    d = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        k = r["Kernel_Name"].replace("lcr::", "")
        d[k][0] += 1
        d[k][1] += float(r["Counter_Value"])
        d[k][2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return d


def family(k):
    for f in ("k_gemm_f32", "k_kpconv_aggregate", "k_radius_query_multi", "k_radius_query", "k_gn_apply", "k_maxpool"):
        if f in k:
            return f
    return k.split("(")[0].replace("void ", "")


def main():
    f = load(sys.argv[1], "FETCH_SIZE")
    w = load(sys.argv[2], "WRITE_SIZE")
    out = sys.argv[3]
    fam = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for k, (n, kb, t) in f.items():
        a = fam[family(k)]
        a[0] += n
        a[1] += kb * 1024
        a[3] += t
        a[2] += w.get(k, [0, 0.0, 0.0])[1] * 1024
    rows = sorted(fam.items(), key=lambda kv: -kv[1][3])
    with open(out + ".md", "w") as fh:
        fh.write("# HBM traffic per launch from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE)\n\n")
        fh.write("FETCH corrected = 2 x raw (gfx950 counts 128-B requests at 64 B, MI355X_MICROARCH.md §HBM); WRITE as reported.\n\n")
        fh.write("| kernel family | launches | avg us (profiled) | fetch MB raw | fetch MB corrected | write MB | traffic MB (corr.) |\n|---|---:|---:|---:|---:|---:|---:|\n")
        js = {}
        for k, (n, fb, wb, t) in rows:
            if n == 0:
                continue
            fa, wa = fb / n, wb / n
            js[k] = {"launches": n, "fetch_bytes_raw": fa, "fetch_bytes_corrected": 2 * fa, "write_bytes": wa,
                     "traffic_bytes": 2 * fa + wa, "avg_us_profiled": t / n / 1e3}
            fh.write(f"| `{k}` | {n} | {t/n/1e3:.1f} | {fa/1e6:.2f} | {2*fa/1e6:.2f} | {wa/1e6:.2f} | {(2*fa+wa)/1e6:.2f} |\n")
    json.dump(js, open(out + ".json", "w"), indent=1)
    print(open(out + ".md").read())


if __name__ == "__main__":
    main()
