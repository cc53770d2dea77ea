"""N warm MSE calibration passes of MobileNetV2 (BASELINE config 4) for `rocprofv3 --kernel-trace --stats`:

  python tools/calib_passes.py [search|fixed] [N]         # 1 cold pass + N passes on freshly reset estimators
  python tools/calib_passes.py summarize <stats.csv> <passes> [out.txt]

Everything this library launches in the run belongs to a calibration pass (the fp32 warm-up forwards launch MIOpen / torch
kernels only), so kernel totals of the stats file divided by the number of passes are per-pass figures."""
import csv
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fp8-quantization_amd")]


def summarize(path, passes, out):
    rows = list(csv.DictReader(open(path)))
    mine = [r for r in rows if "anonymous namespace" in r["Name"] or "fp8q" in r["Name"]]
    tot = sum(float(r["TotalDurationNs"]) for r in mine)
    oth = sum(float(r["TotalDurationNs"]) for r in rows) - tot
    p = lambda *a: print(*a, file=out)       # noqa: E731
    p(f"# per calibration pass (totals / {passes} passes): this library {tot / passes / 1e3:.1f} us in "
      f"{sum(int(r['Calls']) for r in mine) / passes:.0f} launches; everything else (MIOpen, torch) {oth / passes / 1e3:.1f} us")
    p("Name,CallsPerPass,UsPerPass,AvgUs")
    for r in sorted(mine, key=lambda r: -float(r["TotalDurationNs"])):
        name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0]
        p(f"{name},{int(r['Calls']) / passes:g},{float(r['TotalDurationNs']) / passes / 1e3:.1f},{float(r['AverageNs']) / 1e3:.2f}")


def main():
    if sys.argv[1:2] == ["summarize"]:
        out = open(sys.argv[4], "w") if len(sys.argv) > 4 else sys.stdout
        return summarize(sys.argv[2], int(sys.argv[3]), out)
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from host_profile import build, reset
    mode = sys.argv[1] if len(sys.argv) > 1 else "search"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    m = build(mode == "search")
    g = torch.Generator(device="cuda").manual_seed(4321)
    x = torch.randn(64, 3, 224, 224, device="cuda", generator=g)
    xc = torch.randn(64, 3, 224, 224, device="cuda", generator=g)
    with torch.no_grad():
        for _ in range(2):
            m(x)
        m.set_quant_state(True, True)
        m.estimate_ranges()
        m(xc)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            reset(m)
            m(xc)
        torch.cuda.synchronize()
        print(f"{mode}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per warm calibration pass ({n} passes)")


if __name__ == "__main__":
    main()
