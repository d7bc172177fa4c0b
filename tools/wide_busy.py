"""How much of the steady-state window has at least one WIDE kernel (>= 256 workgroups) in flight, only narrow ones, or nothing —
from a rocprofv3 kernel trace (rocpd .db).  A pipeline that is bound by CU time shows wide kernels nearly all the time."""
import sqlite3
import sys


def union(iv):
    iv = sorted(iv)
    out, cs, ce = [], None, None
    for s, e in iv:
        if ce is None or s > ce:
            if ce is not None:
                out.append((cs, ce))
            cs, ce = s, e
        else:
            ce = max(ce, e)
    if ce is not None:
        out.append((cs, ce))
    return out


def length(iv):
    return sum(e - s for s, e in iv)


def main():
    c = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    gx = [x for x in ("grid_x", "grid_size_x", "grid_size") if x in cols][0]
    wx = [x for x in ("workgroup_x", "workgroup_size_x", "workgroup_size") if x in cols][0]
    rows = list(c.execute(f"select start, end, name, {gx}, {wx} from kernels order by start"))
    marks = [r[0] for r in rows if "k_kpconv_cin1" in r[2]]
    w0, t1 = marks[len(marks) // 2], marks[-2]
    rows = [r for r in rows if r[0] >= w0 and r[1] <= t1]
    span = t1 - w0
    wide = [(r[0], r[1]) for r in rows if r[3] // max(r[4], 1) >= 256]
    narrow = [(r[0], r[1]) for r in rows if r[3] // max(r[4], 1) < 256]
    uw, ua = union(wide), union(wide + narrow)
    print("window %.2f ms, %d dispatches (%d wide)" % (span / 1e6, len(rows), len(wide)))
    print("a wide kernel in flight : %.3f" % (length(uw) / span))
    print("only narrow kernels     : %.3f" % ((length(ua) - length(uw)) / span))
    print("nothing in flight       : %.3f" % (1 - length(ua) / span))
    print("sum of wide kernel time / window: %.2f   narrow: %.2f" % (sum(e - s for s, e in wide) / span, sum(e - s for s, e in narrow) / span))
    # the longest stretches without a wide kernel
    gaps, prev = [], w0
    for s, e in uw:
        if s > prev:
            gaps.append((s - prev, prev))
        prev = max(prev, e)
    gaps.sort(reverse=True)
    print("longest stretches without a wide kernel (us):", [round(g / 1e3, 1) for g, _ in gaps[:10]], " total %.3f of the window in %d stretches" % (sum(g for g, _ in gaps) / span, len(gaps)))


if __name__ == "__main__":
    main()
