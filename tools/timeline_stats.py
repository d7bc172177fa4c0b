"""GPU occupancy over time from a rocprofv3 kernel trace (rocpd sqlite .db): fraction of the steady-state window in which at least
one kernel is running, time-weighted number of concurrent kernels, and the same per queue/stream.
    python tools/timeline_stats.py <results.db> [--tail-frac 0.5]   (analyses the last tail-frac of the trace = the timed steps)
"""
import sqlite3
import sys


def union_busy(iv):
    iv = sorted(iv)
    busy, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        busy += cur_e - cur_s
    return busy


def main():
    db = sys.argv[1]
    frac = float(sys.argv[sys.argv.index("--tail-frac") + 1]) if "--tail-frac" in sys.argv else 0.5
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = list(c.execute(f"select start, end, {qcol or 0}, name from kernels order by start"))
    marks = [r[0] for r in rows if "k_kpconv_cin1" in r[3]]       # one per encoder pass
    if len(marks) >= 8:
        w0, t1 = marks[int(len(marks) * (1 - frac))], marks[-2]
    else:
        t0, t1 = rows[0][0], max(r[1] for r in rows)
        w0 = t1 - (t1 - t0) * frac
    rows = [r for r in rows if r[0] >= w0 and r[1] <= t1]
    span = t1 - w0
    print(f"window {span/1e6:.2f} ms, {len(rows)} dispatches, {len([r for r in rows if 'k_kpconv_cin1' in r[3]])} encoder passes")
    print(f"any-kernel-running fraction: {union_busy([(r[0], r[1]) for r in rows]) / span:.3f}")
    print(f"sum of kernel durations / window (mean concurrency): {sum(r[1]-r[0] for r in rows) / span:.3f}")
    qs = {}
    for r in rows:
        qs.setdefault(r[2], []).append((r[0], r[1]))
    for q, iv in sorted(qs.items(), key=lambda kv: -len(kv[1])):
        print(f"  queue {q}: {len(iv)} dispatches, busy fraction {union_busy(iv)/span:.3f}")


if __name__ == "__main__":
    main()
