"""Summarise a rocprofv3 run (rocpd sqlite .db, `--kernel-trace --stats`) into a per-kernel table (markdown).

    python tools/rocprof_summary.py <results.db> [--skip-steps-frac 0.0] > profiles/<name>.md
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
    print("| kernel | calls | total ms | % | avg us | min us | max us |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, n, s, avg, mn, mx in rows:
        short = name.replace("lcr::", "")
        if len(short) > 110:
            short = short[:107] + "..."
        print(f"| `{short}` | {n} | {s/1e6:.3f} | {100*s/tot:.1f} | {avg/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} |")


if __name__ == "__main__":
    main()
