"""Turn the per-kernel rocprofv3 output of tools/refresh_profiles.sh (gpurun_out/<tag>_{mse,k3,staged}_*) into the small
tracked summaries under profiles/.   usage: python tools/summarize_kernels.py r02"""
import collections
import csv
import json
import os
import sys



def model_trace(tag, name, trace_csv, log, title, n_fwd=20):
    """profiles/<tag>_<name>_kernels.txt from `rocprofv3 --kernel-trace -- python bench.py --only-model-config ...`:
    the run ends with n_fwd identical validation forwards, i.e. the tail of the dispatch sequence is periodic -- find
    the period, average the kernels of one forward over the n_fwd repeats, and put this library's share next to the
    algorithmic bytes the same run counted (bench.py LaunchTimer, the JSON line in `log`)."""
    rows = sorted(csv.DictReader(open(trace_csv)), key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    period = None
    for k in range(8, len(names) // n_fwd + 1):
        if all(names[len(names) - (j + 1) * k: len(names) - j * k] == names[len(names) - k:] for j in range(1, n_fwd)):
            period = k
            break
    assert period, "no periodic tail found"
    info = {}
    for ln in open(log):
        if ln.startswith("{"):
            info = json.loads(ln)
    entry = next(iter(info.values())) if info else {}
    mine = lambda n: "anonymous namespace" in n and ("k_" in n)
    lib = lambda k: short(k) or k
    per = collections.OrderedDict()
    tail0 = len(names) - n_fwd * period
    for i in range(tail0, len(names)):
        d = per.setdefault(names[i], [0, 0])
        d[0] += 1
        d[1] += dur[i]
    tot_ns = sum(v[1] for v in per.values()) / n_fwd
    lib_ns = sum(v[1] for k, v in per.items() if mine(k)) / n_fwd
    lib_n = sum(v[0] for k, v in per.items() if mine(k)) / n_fwd
    vf = entry.get("validation_forward_default", {})
    gb = vf.get("algorithmic_gb")
    out = [f"# rocprofv3 --kernel-trace -- python bench.py --only-model-config {name}   ({tag}, MI355X)",
           f"# {title}",
           f"# run = 1 calibration batch + fix_ranges() + 1 event-timed + {n_fwd} plain validation forwards (weights cached, "
           "BN+ReLU(+residual)+quantizer fused); the kernel sequence of one forward was found as the period of the trace's tail",
           f"# PER VALIDATION FORWARD (mean of {n_fwd}): {period} kernel dispatches, {tot_ns / 1e6:.3f} ms of kernel time; "
           f"THIS LIBRARY: {lib_n:.0f} launches, {lib_ns / 1e3:.1f} us = {100 * lib_ns / tot_ns:.1f} % of it"]
    if gb:
        out.append(f"# algorithmic bytes of those launches (SURVEY 8d accounting, bench.py LaunchTimer): {gb} GB -> "
                   f"{gb * 1e6 / (lib_ns / 1e3):.0f} GB/s = {gb * 1e6 / (lib_ns / 1e3) / 8000:.3f} of 8 TB/s "
                   f"(by HIP events around the same launches in the same run: {vf.get('library_us')} us, {vf.get('gb_s')} GB/s)")
    out.append("Name,CallsPerForward,UsPerForward,AvgUs")
    for k, v in sorted(per.items(), key=lambda kv: -kv[1][1]):
        if mine(k):
            out.append(f"{lib(k)},{v[0] / n_fwd:g},{v[1] / n_fwd / 1e3:.1f},{v[1] / v[0] / 1e3:.2f}")
    others = sum(v[1] for k, v in per.items() if not mine(k)) / n_fwd
    out.append(f"(everything else: MIOpen / rocBLAS / torch),{sum(v[0] for k, v in per.items() if not mine(k)) / n_fwd:g},"
               f"{others / 1e3:.1f},")
    # the calibration phase: this library's kernels before the validation forwards (first MIOpen calls excluded by name)
    cal = collections.OrderedDict()
    for i in range(0, tail0 - period):
        if mine(names[i]):
            d = cal.setdefault(names[i], [0, 0])
            d[0] += 1
            d[1] += dur[i]
    cb = entry.get("calibration_batch") or entry.get("calibration_batch_fixed_mantissa") or entry.get("calibration_batch_mantissa_search_6") or {}
    out.append(f"# CALIBRATION (the batch FIVE times -- bench.py runs a first pass, a steady-state pass under its event timer and three plain ones for the wall time -- "
               f"+ fix_ranges; this library's kernels before the validation forwards.  Round 6: the weight quantizers' launches (k_mse_grid, k_mse_select, "
               f"k_quant_short_rows_dm, k_rows_*) run on a side stream NEXT TO the activations' chains, so durations here overlap and do not add up to the pass): "
               f"{sum(v[0] for v in cal.values())} launches, {sum(v[1] for v in cal.values()) / 1e3:.1f} us"
               + (f"; ONE steady-state pass by HIP events: {cb.get('library_us')} us in {cb.get('launches')} calls, wall {cb.get('wall_ms')} ms"
                  f" (host-side estimator logic and the convolutions included; first pass: {(cb.get('first_pass') or {}).get('wall_ms')} ms)" if cb else ""))
    out.append("Name,Calls,TotalUs,AvgUs")
    for k, v in sorted(cal.items(), key=lambda kv: -kv[1][1]):
        out.append(f"{lib(k)},{v[0]},{v[1] / 1e3:.1f},{v[1] / v[0] / 1e3:.2f}")
    path = f"profiles/{tag}_{name}_kernels.txt"
    open(path, "w").write("\n".join(out) + "\n")
    print(open(path).read())


def short(k):   # (defined again below for the stats summaries; needed here first)
    if "at::native" in k or "at::cuda" in k:
        return None
    if "rocprim" in k:
        return "rocprim::radix_sort_onesweep" if "onesweep" in k else "rocprim::" + k.split("::")[-1][:40]
    return k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].replace(", ", ";")


if len(sys.argv) > 1 and sys.argv[1] == "model":
    # python tools/summarize_kernels.py model <tag> <c3|c4|c4_search> <kernel_trace.csv> <stdout log> "<title>"
    os.makedirs("profiles", exist_ok=True)
    model_trace(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6])
    sys.exit(0)

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
os.makedirs("profiles", exist_ok=True)


def short(k):
    if "at::native" in k or "at::cuda" in k:
        return None
    if "rocprim" in k:
        return "rocprim::radix_sort_onesweep" if "onesweep" in k else "rocprim::" + k.split("::")[-1][:40]
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

mse = counters(["mse_pmc", "mse_pmc2", "mse_pmc3", "mse_pmc4"], "mse")
if mse:
    out = {"source": "rocprofv3 --pmc {SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE}, "
                     "{SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS}, {FETCH_SIZE}, {WRITE_SIZE} (four separate "
                     "passes) -- python tools/mb_mse_one.py 1: the interval-histogram route of K4 on [64,32,112,112] x 111 candidates "
                     "(E4M3), three launches of the whole chain; means per launch of each kernel",
           "traffic": "FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE x 1024 x 2 on gfx950 for 16-B/lane coalesced streams "
                      "(MI355X_MICROARCH.md HBM section), WRITE_SIZE x 1024.  Algorithmic bytes of the chain: 4 B/element read by the key "
                      "histogram (k_stage1), 4 + 4 B/element by the scatter (k_sort_plan_scatter), 4 B/element read by k_moments = 16 B per "
                      "nonzero element, 12 on the wire when the second read of x still sits in the Infinity Cache",
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
