"""Per-call kernel timeline of tools/mb_calib_shapes.py from a rocprofv3 kernel trace (CSV):
python tools/calib_timeline.py <kernel_trace.csv> [out.txt]: a call starts with the abs-max launch (k_affine_minmax<true> /
k_minmax_partial) and ends with the K1 launch; calls with the same kernel/grid sequence are grouped, medians reported."""
import csv
import statistics
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
calls, cur = [], None
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if not name.startswith("k_"):
        continue
    if name.startswith("k_affine_minmax<true>") or name.startswith("k_minmax_partial"):
        cur = []
    if cur is not None:
        cur.append((name, int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Grid_Size_X"]))
        if name.startswith("k_quant_rows"):
            calls.append(cur)
            cur = None
groups = {}
for c in calls:
    groups.setdefault(tuple((n, g) for n, _, _, g in c), []).append(c)
for key, cs in groups.items():
    if len(cs) < 3:
        continue
    print(f"--- {len(cs)} calls: {len(key)} launches, first grid {key[0][1]}", file=out)
    for i, (n, g) in enumerate(key):
        gap = statistics.median((c[i][1] - c[i - 1][2]) / 1e3 for c in cs) if i else 0.0
        print(f"  {n:28s} grid {g:>9s}  {statistics.median((c[i][2] - c[i][1]) / 1e3 for c in cs):8.1f} us   (gap before: {gap:5.1f})",
              file=out)
    print(f"  {'SPAN first start -> last end':43s} {statistics.median((c[-1][2] - c[0][1]) / 1e3 for c in cs):8.1f} us", file=out)
    print(f"  {'sum of kernel durations':43s} {statistics.median(sum(e - s for _, s, e, _ in c) / 1e3 for c in cs):8.1f} us", file=out)
