"""Summarise a rocprofv3 --pmc counter_collection.csv for our kernels (mean per dispatch)."""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "at::native" in k or "k_" not in k:
            continue
        k = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
