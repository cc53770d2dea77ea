"""Turn rocprofv3 output dirs under gpurun_out/ into the small tracked summaries in profiles/.

usage: python tools/summarize_profiles.py <kernel_trace_dir> <pmc_fetch_dir> <pmc_write_dir> <tag> <kernel-substring>
"""
import collections
import csv
import json
import os
import sys

kt, pf, pw, tag, kname = sys.argv[1:6]
os.makedirs("profiles", exist_ok=True)


def short(k):
    if "at::native" in k:
        return "torch:" + k.split("at::native::")[1][:60].replace('"', "").replace(",", ";")
    return k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].replace(", ", ";")


rows = list(csv.DictReader(open(os.path.join(kt, "bench_kernel_stats.csv"))))
with open(f"profiles/{tag}_bench_kernel_stats.csv", "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py --no-extras --no-cpu-baseline"
            f"   ({tag}, MI355X; kernel names shortened)\n")
    f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n")
    for r in rows:
        f.write(",".join([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                          r["MinNs"], r["MaxNs"], r["StdDev"]]) + "\n")
res = {}
for d, ctr in ((pf, "FETCH_SIZE"), (pw, "WRITE_SIZE")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(d, "bench_counter_collection.csv"))):
        if "at::native" in r["Kernel_Name"]:
            continue
        agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        res.setdefault(k, {})[ctr + "_KB_mean"] = sum(v) / len(v)
        res[k]["n_" + ctr] = len(v)
key = [k for k in res if kname in k][0]
q = res[key]
fetch_b = q["FETCH_SIZE_KB_mean"] * 1024 * 2
write_b = q["WRITE_SIZE_KB_mean"] * 1024
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 5 "
                 "--warmup 1 --no-cpu-baseline --no-extras",
       "correction": "FETCH_SIZE x2 on gfx950 for 16-B/lane coalesced streams (MI355X_MICROARCH.md HBM section: "
                     "FETCH_SIZE = TCC_EA0_RDREQ x 64 B, 128-B requests tallied at 64 B); WRITE_SIZE as reported; KB x1024",
       "kernel": key, "k1_fetch_bytes_per_launch": fetch_b, "k1_write_bytes_per_launch": write_b,
       "k1_bytes_per_launch": fetch_b + write_b, "algorithmic_bytes_per_launch": 308281344 * 8, "raw": res}
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
json.dump(out, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
print(open(f"profiles/{tag}_bench_kernel_stats.csv").read())
print(json.dumps({k: v for k, v in out.items() if k != "raw"}, indent=1))
