"""Per-call timeline of the K4 interval-histogram route from a rocprofv3 kernel trace:
python tools/mse_timeline.py <kernel_trace.csv>  -> for each distinct call shape, the median duration of every kernel and the
span from the first kernel's start to the last kernel's end (launch gaps included)."""
import csv, statistics, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
calls, cur = [], None
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if name.startswith("k_stage1"):
        cur = []
    if cur is not None and name.startswith("k_"):
        cur.append((name, int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Grid_Size_X"]))
        if name.startswith("k_mse_eval"):
            calls.append(cur)
            cur = None
groups = {}
for c in calls:
    if not c or not c[-1][0].startswith("k_mse_eval"):
        continue
    key = tuple((n, g) for n, _, _, g in c)
    groups.setdefault(key, []).append(c)
for key, cs in groups.items():
    print(f"--- {len(cs)} calls, eval grid {key[-1][1]}")
    for i, (n, g) in enumerate(key):
        print(f"  {n:20s} grid {g:>9s}  {statistics.median((c[i][2] - c[i][1]) / 1e3 for c in cs):8.1f} us")
    print(f"  {'SPAN first start -> last end':31s} {statistics.median((c[-1][2] - c[0][1]) / 1e3 for c in cs):8.1f} us")
    print(f"  {'sum of kernel durations':31s} {statistics.median(sum(e - s for _, s, e, _ in c) / 1e3 for c in cs):8.1f} us")
