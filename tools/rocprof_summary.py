"""Summarise a rocprofv3 run (rocpd sqlite .db, `--kernel-trace --stats`) into a per-kernel table (markdown).

    python tools/rocprof_summary.py <results.db> [--note "text"] [--bench-log file] > profiles/<name>.md

--bench-log: the stdout of the profiled command; if its last line is a bench.py JSON line, the header quotes the PROFILED run's own scans/s
(a kernel trace slows the host: when that rate is far below the un-profiled one, the kernels of different streams did not overlap as they
do in the real pipeline, and their durations are the kernel-alone ones) and, per kernel family, launches x avg us next to the line's fields.
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                          "from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    t0, t1 = c.execute("select min(start), max(end) from kernels").fetchone()
    print(f"# rocprofv3 kernel summary — {db.split('/')[-1]}")
    print(f"\ntotal kernel time {tot/1e6:.3f} ms over a {(t1-t0)/1e6:.3f} ms span, {sum(r[1] for r in rows)} dispatches\n")
    if "--note" in sys.argv:
        print(sys.argv[sys.argv.index("--note") + 1] + "\n")
    if "--bench-log" in sys.argv:
        import json
        try:
            line = json.loads([l for l in open(sys.argv[sys.argv.index("--bench-log") + 1]).read().splitlines() if l.startswith("{")][-1])
            r = line.get("roofline", {})
            print(f"profiled run's own line: **{line.get('value')} {line.get('unit', '')}** ({line.get('ms_per_step')} ms per step, {line.get('steps')} steps) — "
                  f"kernel-alone GEMM clock of that run: {r.get('launches_timed')} launches, avg {r.get('avg_launch_us')} us, "
                  f"{r.get('achieved')} TFLOP/s = frac {r.get('frac')}; in-pipeline clock: avg {r.get('in_pipeline', {}).get('avg_launch_us')} us\n")
        except Exception as e:                                           # noqa: BLE001
            print(f"(no bench line in the log: {e})\n")
    fam = {}
    for name, n, s_, avg, mn, mx in rows:
        for key in ("k_gemm_f32", "k_kpconv_aggregate", "k_radius_query", "k_gn_apply", "k_maxpool"):
            if key in name:
                f = fam.setdefault(key, [0, 0])
                f[0] += n
                f[1] += s_
    if fam:
        print("families: " + "; ".join(f"`{k}*` {v[0]} launches x {v[1]/v[0]/1e3:.1f} us = {v[1]/1e6:.2f} ms" for k, v in fam.items()) + "\n")
    print("| kernel | calls | total ms | % | avg us | min us | max us |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, n, s, avg, mn, mx in rows:
        short = name.replace("lcr::", "")
        if len(short) > 110:
            short = short[:107] + "..."
        print(f"| `{short}` | {n} | {s/1e6:.3f} | {100*s/tot:.1f} | {avg/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} |")


if __name__ == "__main__":
    main()
