"""Per-kernel duration and the idle gap BEFORE each kernel on its own queue, from a rocprofv3 kernel trace (.db): shows whether a
chain of dependent launches loses its time inside the kernels or between them.
    python tools/queue_gaps.py <results.db> [name-substring-identifying-the-queue, default k_gs_keys]
"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sys.argv[1]
    probe = sys.argv[2] if len(sys.argv) > 2 else "k_gs_keys"
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else "stream_id"
    rows = list(c.execute(f"select start, end, {qcol}, name from kernels order by start"))
    t0, t1 = rows[0][0], rows[-1][1]
    rows = [r for r in rows if r[0] > t0 + (t1 - t0) * 0.5]          # steady state: second half of the trace
    qs = defaultdict(int)
    for r in rows:
        if probe in r[3]:
            qs[r[2]] += 1
    q = max(qs, key=qs.get)
    chain = [r for r in rows if r[2] == q]
    dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
    prev_end = None
    for s, e, _, name in chain:
        nm = name.split("(")[0][-38:]
        dur[nm] += e - s
        cnt[nm] += 1
        if prev_end is not None and s - prev_end < 200e3:                # ignore the long waits between batches
            gap[nm] += max(0, s - prev_end)
        prev_end = e
    nb = sum(v for k, v in cnt.items() if "k_pack_lengths" in k) or 1
    print(f"queue {q}: {len(chain)} dispatches, {nb} batches")
    print("%-40s %6s %10s %10s" % ("kernel", "n/bat", "dur us/bat", "gap us/bat"))
    for nm in sorted(dur, key=lambda k: -(dur[k] + gap[k])):
        print("%-40s %6.1f %10.1f %10.1f" % (nm, cnt[nm] / nb, dur[nm] / nb / 1e3, gap[nm] / nb / 1e3))
    print("%-40s %6.1f %10.1f %10.1f" % ("TOTAL", len(chain) / nb, sum(dur.values()) / nb / 1e3, sum(gap.values()) / nb / 1e3))


if __name__ == "__main__":
    main()
