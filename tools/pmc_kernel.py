#!/usr/bin/env python
"""Per-kernel table of arbitrary rocprofv3 --pmc counters (one or more *_counter_collection.csv files) -> markdown.

    python tools/pmc_kernel.py [--match SUBSTR] a_counter_collection.csv [b_counter_collection.csv ...]

Values are summed over the dispatches of a kernel name and divided by the number of dispatches (per-launch averages)."""
import collections
import csv
import sys


def short(k):
    return k.replace("lcr::", "").replace("void ", "").split("(")[0][:70]


def main():
    args = sys.argv[1:]
    match = None
    if args and args[0] == "--match":
        match, args = args[1], args[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(lambda: collections.defaultdict(set))
    names = []
    for path in args:
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if match and match not in k:
                continue
            c = r["Counter_Name"]
            if c not in names:
                names.append(c)
            agg[k][c] += float(r["Counter_Value"])
            disp[k][c].add((path, r["Dispatch_Id"]))
    print("| kernel | " + " | ".join(names) + " |")
    print("|---|" + "---:|" * len(names))
    for k in sorted(agg):
        row = []
        for c in names:
            n = len(disp[k][c])
            row.append("%.4g" % (agg[k][c] / n) if n else "-")
        print("| %s (x%d) | " % (k, max(len(v) for v in disp[k].values())) + " | ".join(row) + " |")


if __name__ == "__main__":
    main()
