"""Turn the per-kernel rocprofv3 output of tools/refresh_profiles.sh (gpurun_out/<tag>_{mse,k3,staged}_*) into the small
tracked summaries under profiles/.   usage: python tools/summarize_kernels.py r02"""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
os.makedirs("profiles", exist_ok=True)


def short(k):
    if "at::native" in k or "at::cuda" in k:
        return None
    return k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].replace(", ", ";")


def stats(name, cmd):
    path = f"gpurun_out/{tag}_{name}_kt/{name}_kernel_stats.csv"
    if not os.path.exists(path):
        print("missing", path)
        return
    rows = [r for r in csv.DictReader(open(path)) if short(r["Name"])]
    with open(f"profiles/{tag}_{name}_kernel_stats.csv", "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- {cmd}   ({tag}, MI355X; torch's own kernels dropped, names shortened)\n")
        f.write("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,StdDev\n")
        for r in rows:
            f.write(",".join([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"],
                              r["StdDev"]]) + "\n")
    print(open(f"profiles/{tag}_{name}_kernel_stats.csv").read())


def counters(dirs, prefix):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        path = f"gpurun_out/{tag}_{d}/{prefix}_counter_collection.csv"
        if not os.path.exists(path):
            print("missing", path)
            continue
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in d.items()} for k, d in agg.items()}


stats("mse", "python tools/mb_mse.py")
stats("k3", "python tools/mb_k3.py")
stats("staged", "python tools/mb_staged.py")

mse = counters(["mse_pmc", "mse_pmc2"], "mse")
if mse:
    out = {"source": "rocprofv3 --pmc {SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE} and "
                     "{SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS} (separate passes) -- python "
                     "tools/mb_mse.py: means over all launches of the script ([64,32,112,112] x 111 candidates with 1, 1 and 6 "
                     "mantissa widths for k_mse_row; [512,512,3,3] per channel for k_mse_grid)",
           "how_to_read": "SQ_INSTS_VALU = wave-level VALU instructions per launch (x64 lanes = lane-instructions; a v_pk_* "
                          "counts once and does two elements); VALU issue utilisation = SQ_ACTIVE_INST_VALU * 4 / "
                          "(GRBM_GUI_ACTIVE * 1024 SIMDs) if the counter ticks once per issued wave-instruction",
           "kernels": mse}
    json.dump(out, open(f"profiles/{tag}_mse_pmc.json", "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1)[:3000])

st = counters(["staged_pf", "staged_pw"], "staged")
if st:
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python tools/mb_staged.py; means "
                     "over all launches (mixed shapes; the [2^21,147] launches dominate)",
           "correction": "FETCH_SIZE x2 on gfx950 for 16-B/lane coalesced streams (MI355X_MICROARCH.md HBM section); KB x1024",
           "kernels": st}
    json.dump(out, open(f"profiles/{tag}_staged_pmc.json", "w"), indent=1)
