"""Median kernel duration per (kernel, grid) from a rocprofv3 --kernel-trace CSV: GPU-side times of small launches, which
back-to-back HIP-event timing from Python cannot see (the host's launch rate is the floor there).
    python tools/trace_by_grid.py <dir with *_kernel_trace.csv> [name filter]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

d = defaultdict(list)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
        if flt and flt not in name:
            continue
        key = (name[:48], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]))
        d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for key in sorted(d, key=lambda k: (k[0], k[1] * k[2])):
    v = sorted(d[key])
    print(f"{key[0]:48s} grid {key[1]:6d} x {key[2]:5d}  n={len(v):5d}  median {v[len(v) // 2]:8.2f} us  min {v[0]:8.2f}")
