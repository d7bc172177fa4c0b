"""rocprofv3 --pmc csv -> per (kernel, grid size) averages of every counter + derived ratios."""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("lcr::", "").replace("void ", "").split("(")[0][:44]
        key = (k, r["Grid_Size"])
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[key][r["Counter_Name"]] += 1
names = sorted({c for v in agg.values() for c in v})
print("| kernel | grid | " + " | ".join(names) + " |")
print("|---|---|" + "---:|" * len(names))
for key in sorted(agg, key=lambda k: (k[1], k[0])):
    print("| %s | %s | " % key + " | ".join("%.4g" % (agg[key][c] / max(cnt[key][c], 1)) for c in names) + " |")
