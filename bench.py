#!/usr/bin/env python
"""bench.py -- FP8 quantize+dequantize throughput on MI355X (driver contract).

    python bench.py --gpus N --steps K --warmup W

Headline workload (BASELINE.json config 2, scaled as north_star allows: "synthetic NxCxKxK tensors"):
    a conv1-shaped weight tensor [2^21, 3, 7, 7] fp32 per GPU (308 M elements, 1.23 GB in +
    1.23 GB out -- far beyond the 256 MB Infinity Cache), per-output-channel E5M2
    (n_bits 8, 2 mantissa bits), ranges from current_minmax (computed once, outside the timed
    region: validation runs with fixed ranges, quantization_manager.py:93-98).
One step = one pass of the hot path quantize_to_fp8_ste_MM (fp8_quantizer.py:91-133) over
the tensor = one launch of the HIP kernel k_rows_flat<0> (short rows cut into aligned 16 KiB chunks) through the C ABI (fp8q_quantize_f32).
Inputs are resident in HBM before the timed region.

N > 1 (one process per GPU, torch.distributed/RCCL).  `value` is north_star's own weight flow END TO END: one
[N * 2^21, 3, 7, 7] tensor held by every rank, rank r quantizes channels channel_partition(C, N)[r] (the N = 1 kernel on
the N = 1 shard: weak scaling) and the shards are re-assembled on every rank with one RCCL all-gather over xGMI
(round 6: as 1-byte storage codes -- fp8q.dist.quantize_weight_sharded_codes with the fixed ranges: encode -> all-gather ->
decode, bit-identical to the fp32 form and a quarter of its xGMI bytes; `value_fp32_wire` is the fp32 form of the same flow,
timed right after the judged region) -- all elements / max-over-ranks wall time, collective included.  Top-level keys next
to it: `value_kernel_only` (the same elements / the kernel phases alone), `kernel_us`,
`collective_us`, `xgmi_gb_s` (bytes each rank receives / collective time), and from `north_star_path`, which times the
two exchange steps north_star names at any N (N = 1: the collectives are no-ops and the figures are the kernels' own):
`value_codes_wire` (same flow with 1-byte storage codes on the wire), `value_resnet18_strong` (ResNet-18's 21 weight
tensors, one bucketed all-gather -- 1-byte codes + fp32 ranges on the wire since round 5, `fp32_wire_*` next to it in
`north_star_path` --; strong scaling), `value_c5` (BASELINE config 5 including its all-reduce).
  weights_allgather  one [N * 2^18, 3, 7, 7] weight tensor held by every rank; rank r finds the ranges of and quantizes
                     channels channel_partition(C, N)[r] (fused min/max+quantize), then the shards and their per-channel
                     ranges are re-assembled with RCCL all-gathers -- fp32 on the wire, and 1-byte storage codes
                     (encode -> all-gather -> decode: 4x less xGMI traffic); kernel / collective time split by events
  c5                 BASELINE config 5: per-rank slab [512, 4096, 512], allminmax fold -> all-reduce of the 2-float
                     range -> E4M3 quantize with the global range
  ranks_seen         sum over ranks of 1 through an RCCL all-reduce: proves the collective spanned N processes
`python bench.py --dry-run-ranks 8` (no GPU needed) runs the same call sequence on 8 gloo ranks with CPU tensors, a
compute stub and scaled-down tensors, and prints every collective with its byte count next to the full-size counts.

The JSON line also carries:
  roofline      the dominant kernel's algorithmic bytes (8 B/element) / its HIP-event launch time
  cpu_baseline  value: the reference's eager ATen op chain on the host cores (kind "reference-equivalent"); port_value: the
                fused C port of the oracle (oracle/fp8q_oracle.c, OpenMP); bounded samples, rank 0, N=1
  extras        other kernels of the path on their own shapes (not part of `value`)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "fp8-quantization_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# multi-process GPU work on this pool needs dmabuf IPC; must be in the environment before the HIP runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PREWARM_S = float(os.environ.get("FP8Q_BENCH_PREWARM_S", "0.25"))          # untimed clock warm-up before the W warm-up steps (setup, see main)
HBM_PEAK_GBS = 8000.0       # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_ELEM = 8          # K1: 4 B read + 4 B written
N_CH = 1 << 21              # channels per GPU
ROW = 3 * 7 * 7             # conv1 filter
MBITS, NBITS, SIGN = 2, 8, 1  # E5M2


def ev_time(fn, iters, warm=2):
    """median / mean seconds per call, HIP events on the current stream."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e-3 for a, b in evs)
    return ts[len(ts) // 2], sum(ts) / len(ts)


def measure_traffic(timeout_s=60):
    """HBM bytes per launch of the headline kernel from the PMC counters, measured NOW: two rocprofv3 passes (FETCH_SIZE
    and WRITE_SIZE cannot share a pass: 3 + 2 of the 4 TCC slots) over a 3-step, headline-only run of this same script,
    corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts 64 B per 128-B request
    of a wide coalesced stream: x2; KB -> x1024).  Returns (bytes, detail) or (None, why)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if prof is None:
        return None, "rocprofv3 not found"
    vals = {}
    env = dict(os.environ, TMPDIR="/tmp", FP8Q_BENCH_PREWARM_S="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="fp8q_pmc_", dir="/tmp")
        cmd = [prof, "--pmc", ctr, "--output-format", "csv", "-d", out, "-o", "b", "--", sys.executable,
               os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--no-extras", "--no-cpu-baseline",
               "--no-north-star-path", "--no-traffic"]
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            got = []
            for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(path)):
                    if "k_rows_flat<0" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                        got.append(float(r["Counter_Value"]))
            if not got:
                return None, f"no {ctr} samples of the headline kernel"
            vals[ctr] = (sum(got) / len(got), len(got))
        except Exception as e:   # noqa: BLE001 -- a profiler problem must not cost the bench line
            return None, f"{ctr} pass failed: {e!r}"[:200]
        finally:
            shutil.rmtree(out, ignore_errors=True)
    fetch, write = vals["FETCH_SIZE"][0] * 1024 * 2, vals["WRITE_SIZE"][0] * 1024
    return fetch + write, dict(fetch_bytes=round(fetch), write_bytes=round(write), launches_sampled=vals["FETCH_SIZE"][1],
                               raw_counters=dict(FETCH_SIZE_KB_avg=round(vals["FETCH_SIZE"][0], 1),
                                                 WRITE_SIZE_KB_avg=round(vals["WRITE_SIZE"][0], 1),
                                                 correction="FETCH_SIZE x 1024 x 2 (gfx950: 64 B counted per 128-B request), "
                                                            "WRITE_SIZE x 1024"))


def cpu_baseline(x_cpu, maxval_cpu):
    """CPU oracle on a bounded sample of the same workload (whole channels), all host cores.

    The sample is passed repeatedly until ~5 s of wall time (tens to hundreds of core-seconds)
    have been spent, so the figure is not a cold-cache one-shot."""
    import numpy as np
    import oracle
    threads = oracle.num_threads()
    xs = np.ascontiguousarray(x_cpu)
    mv = np.ascontiguousarray(maxval_cpu)
    ref = oracle.c_quantize(xs, mv, MBITS, NBITS, SIGN)   # warm-up pass, also the parity sample
    reps, t0 = 0, time.perf_counter()
    while True:
        oracle.c_quantize(xs, mv, MBITS, NBITS, SIGN)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= 5.0 or reps >= 200:
            break
    return dict(value=round(xs.size * reps / dt / 1e9, 4), unit="Gelem/s", cores=threads, kind="port",
                sample=f"first {xs.shape[0]} channels ({xs.size} elements) of the bench tensor x {reps} passes, "
                       f"oracle/fp8q_oracle.c (OpenMP, {threads} threads), {dt:.2f} s wall"), ref, xs.shape[0]


def physical_cores():
    """physical cores of the host (not SMT threads): the thread count of the reference-equivalent leg"""
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    try:
        cores = set()
        phys = core = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":")[1].strip()
            elif not ln.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
        if cores:
            return len(cores)
    except Exception:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def torch_eager_cpu(x_cpu, maxval_cpu, n_ch):
    """The reference-equivalent path on the host: the reference's own eager ATen op chain (oracle/torch_eager.py
    restates fp8_quantizer.py:91-133 op for op; the reference files themselves do not travel to the GPU box), one
    thread per physical core, one warm-up pass, then the MEDIAN of >= 3 timed passes (bounded to ~10 s)."""
    from oracle import torch_eager as te
    cores = physical_cores()
    prev = torch.get_num_threads()
    torch.set_num_threads(cores)
    try:
        xs = x_cpu[:n_ch]
        mv = maxval_cpu[:n_ch]
        mb = torch.tensor([float(MBITS)])
        t0 = time.perf_counter()
        te.fake_quant(xs, NBITS, mv, mb, SIGN)                  # warm-up (allocator, thread pool, page faults)
        warm = time.perf_counter() - t0
        passes = max(3, min(9, int(10.0 / max(warm, 1e-3))))
        ts = []
        for _ in range(passes):
            t0 = time.perf_counter()
            te.fake_quant(xs, NBITS, mv, mb, SIGN)
            ts.append(time.perf_counter() - t0)
        ts.sort()
        med = ts[len(ts) // 2]
    finally:
        torch.set_num_threads(prev)
    return dict(value=round(xs.numel() / med / 1e9, 4), unit="Gelem/s", cores=cores, kind="reference-equivalent",
                sample=f"first {n_ch} channels ({xs.numel()} elements) of the bench tensor, torch {torch.__version__} CPU "
                       f"eager op chain of fp8_quantizer.py:91-133, {cores} threads (physical cores), warm-up pass "
                       f"{warm:.2f} s, median of {passes} passes ({med:.3f} s; min {ts[0]:.3f}, max {ts[-1]:.3f})")


def extras(ops, dev):
    """Other kernels of the path, each on the shape that exercises it (rank 0, N=1 only)."""
    out = {"_note": "every entry: 30 ms of the same launch untimed, then the median of HIP-event times"}
    n = 1 << 28
    x = torch.randn(n, device=dev)
    y = torch.empty_like(x)
    mv1 = torch.tensor([3.0], device=dev)

    def warm(fn, seconds=0.03):
        # same reason as the headline's pre-warm: after the host-side pause of setting a case up, the first launches
        # run at lower clocks; 30 ms of the same launch first (not timed)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(5):
                fn()
            torch.cuda.synchronize()

    def rec(name, nelem, bpe, fn, iters=10):
        warm(fn)
        med, _ = ev_time(fn, iters)
        out[name] = dict(us=round(med * 1e6, 1), gelem_s=round(nelem / med / 1e9, 1),
                         gb_s=round(nelem * bpe / med / 1e9, 1), frac_of_8tbs=round(nelem * bpe / med / 8e12, 3))

    rec("copy_ceiling_1GiB", n, 8, lambda: ops.copy(x, out=y))
    rec("k1_per_tensor_e4m3_1GiB", n, 8, lambda: ops.quantize(x, mv1, 3, 8, 1, out=y))
    rec("k3_minmax_per_tensor_1GiB", n, 4, lambda: ops.minmax(x, False))
    a = x[: 64 * 64 * 112 * 112].view(64, 64, 112, 112)
    ya = y[: a.numel()].view_as(a)
    rec("k1_act_64x64x112x112_e5m2", a.numel(), 8, lambda: ops.quantize(a, mv1, 2, 8, 1, out=ya))
    rec("k3_act_64x64x112x112_allminmax", a.numel(), 4, lambda: ops.minmax(a, False))
    # the headline shape in estimate state: current_minmax + quantize fused (k_rows_staged), and the estimator alone
    xc = x[: (1 << 20) * 147].view(1 << 20, 3, 7, 7)
    yc = y[: xc.numel()].view_as(xc)
    rec("fused_minmax_quant_Nx3x7x7_e5m2", xc.numel(), 8, lambda: ops.minmax_quantize(xc, 2, 8, 1, out=yc))
    rec("k2_minmax_per_channel_Nx3x7x7", xc.numel(), 4, lambda: ops.minmax(xc, True))
    # other ResNet-18 filter shapes, scaled up the same way (per-channel E5M2, fixed ranges / fused estimate)
    for name, rows, shape in (("64x3x3", 1 << 18, (64, 3, 3)), ("512x3x3", 58254, (512, 3, 3))):
        inner = shape[0] * shape[1] * shape[2]
        xv = x[: rows * inner].view(rows, *shape)
        yv = y[: rows * inner].view(rows, *shape)
        _, _, mvv = ops.minmax(xv, True, want_maxval=True)
        rec(f"k1_per_channel_Nx{name}_e5m2", xv.numel(), 8, lambda: ops.quantize(xv, mvv, 2, 8, 1, out=yv))
        rec(f"fused_minmax_quant_Nx{name}_e5m2", xv.numel(), 8, lambda: ops.minmax_quantize(xv, 2, 8, 1, out=yv))
    w = torch.randn(64, 3, 7, 7, device=dev) * 0.1
    yw = torch.empty_like(w)
    rec("conv1_64x3x7x7_fused_minmax_quant_e5m2", w.numel(), 8,
        lambda: ops.minmax_quantize(w, 2, 8, 1, out=yw), iters=200)
    # BASELINE config 2 at its literal size is a 37 KB tensor: the launch is all latency.  An event pair around ONE launch also
    # measures the stream's start-up gap and this script's own Python call (~10 us of host per call: the GPU waits for it); the
    # kernel's own cost shows when launches are queued faster than they run -- 200 of them replayed from a HIP graph
    c1 = out["conv1_64x3x7x7_fused_minmax_quant_e5m2"]
    c1["kernel"] = "k_small_rows_fused<3> (one wave per row, row in registers; round 6)"
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        ops.minmax_quantize(w, 2, 8, 1, out=yw)
    e1.record()
    torch.cuda.synchronize()
    c1["back_to_back_python_loop_us"] = round(e0.elapsed_time(e1) * 1e3 / 200, 2)
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ops.minmax_quantize(w, 2, 8, 1, out=yw)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(200):
                ops.minmax_quantize(w, 2, 8, 1, out=yw)
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        c1["hipgraph_200_launches_us_per_launch"] = round(e0.elapsed_time(e1) * 1e3 / 200, 2)
    except Exception as e:   # noqa: BLE001 -- informational entry only
        c1["hipgraph_200_launches_us_per_launch"] = {"error": repr(e)[:200]}
    # all 21 ResNet-18 weight tensors (11.68 M elements), re-quantized as on every forward
    shapes = [(64, 3, 7, 7)] + [(64, 64, 3, 3)] * 4 + [(128, 64, 3, 3), (128, 128, 3, 3), (128, 64, 1, 1)] + \
        [(128, 128, 3, 3)] * 2 + [(256, 128, 3, 3), (256, 256, 3, 3), (256, 128, 1, 1)] + [(256, 256, 3, 3)] * 2 + \
        [(512, 256, 3, 3), (512, 512, 3, 3), (512, 256, 1, 1)] + [(512, 512, 3, 3)] * 2 + [(1000, 512)]
    ws = [torch.randn(*sh, device=dev) * 0.05 for sh in shapes]
    ys = [torch.empty_like(t) for t in ws]
    mvs = [ops.minmax(t, True, want_maxval=True)[2] for t in ws]
    n_w = sum(t.numel() for t in ws)

    def all_weights():
        for t, o, m in zip(ws, ys, mvs):
            ops.quantize(t, m, 2, 8, 1, out=o)
    rec("resnet18_all_21_weight_tensors_k1_e5m2", n_w, 8, all_weights, iters=50)
    items = [(t, m, 2, 8, 1, o) for t, o, m in zip(ws, ys, mvs)]
    rec("resnet18_all_21_weight_tensors_multi_launch_e5m2", n_w, 8, lambda: ops.multi_quantize(items), iters=50)
    if hasattr(ops, "MultiPlan"):
        # prepared plan (descriptors validated and packed once, as QuantizedModel.fix_ranges() does): per call one
        # ctypes call; device time by events, host time = wall time of the enqueue alone
        plan = ops.MultiPlan(items)
        rec("resnet18_all_21_weight_tensors_plan_e5m2", n_w, 8, plan.launch, iters=200)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            plan.launch()
        host = (time.perf_counter() - t0) / 200
        torch.cuda.synchronize()
        out["resnet18_all_21_weight_tensors_plan_e5m2"]["host_enqueue_us"] = round(host * 1e6, 1)
        # an event pair around ONE ~20 us launch also measures the stream's start-up gap (an empty kernel reads 6-10 us
        # this way); 100 launches queued back to back between one pair of events do not
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            plan.launch()
        e1.record()
        torch.cuda.synchronize()
        b2b = e0.elapsed_time(e1) * 1e-3 / 100
        out["resnet18_all_21_weight_tensors_plan_e5m2"].update(
            back_to_back_us=round(b2b * 1e6, 1), back_to_back_frac_of_8tbs=round(n_w * 8 / b2b / 8e12, 3))
        # the same launch replayed from a HIP graph (what a captured quantized forward / QAT step would do)
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                plan.launch()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                plan.launch()
            rec("resnet18_all_21_weight_tensors_plan_hipgraph_e5m2", n_w, 8, graph.replay, iters=200)
        except Exception as e:   # noqa: BLE001 -- informational entry only
            out["resnet18_all_21_weight_tensors_plan_hipgraph_e5m2"] = {"error": repr(e)[:200]}
    # K4: the grid search of FP_MSE_Estimator (range_estimators.py:337-347) on a MobileNetV2 activation, 111 candidate
    # ranges x {1, 6} mantissa widths in one pass over x (the reference: 111 * |m| full quantizer passes).  Round 5: a
    # tensor of this size takes the interval-histogram route (csrc/fp8q_mse_hist.hip: ONE hand-written partition of the
    # keys at the copy rate + integer moments per interval; ~12 B of HBM traffic per element whatever the number of
    # candidates) -- HBM / LDS-atomic bound, no longer the VALU-bound lane-per-element loop (523 us / 3.1 ms in round 4).
    a4 = x[: 64 * 32 * 112 * 112].view(64, 32, 112, 112)
    mx4 = float(a4.abs().max())
    grid = torch.linspace(0.1 * mx4, 1.2 * mx4, 111, device=dev).view(111, 1).contiguous()
    for mb in ([3.0], [1.0, 2.0, 3.0, 4.0, 5.0, 6.0]):
        mses = torch.zeros(len(mb), 111, 1, device=dev)
        warm(lambda: ops.mse_grid(a4, False, grid, mb, 8, 1, mses))
        med, _ = ev_time(lambda: ops.mse_grid(a4, False, grid, mb, 8, 1, mses), 5)
        ce = a4.numel() * 111 * len(mb)
        out[f"k4_mse_111cand_{len(mb)}m_64x32x112x112"] = dict(
            us=round(med * 1e6, 1), t_cand_elem_s=round(ce / med / 1e12, 3), hbm_gb_s=round(a4.numel() * 4 / med / 1e9, 1),
            algorithmic_bytes_per_element=4, route_bytes_per_element=16, frac_of_8tbs_at_16B=round(a4.numel() * 16 / med / 8e12, 3),
            route="interval histogram: k_stage1 (borders + key histogram) -> k_tab_scan -> k_sort_plan_scatter (border sort + "
                  "plan + key scatter in one launch) -> k_moments -> k_iv_scan_super/top -> k_mse_eval; per-kernel times: "
                  "profiles/r06_mse_timeline.txt")
    del x, y
    return out


class LaunchTimer:
    """HIP-event pairs around every call into fp8q.ops (the torch-facing wrappers of the C ABI), on the stream the
    kernels are launched on: per pass, the time the GPU spent in THIS library's launches, their elements and their
    algorithmic bytes (K1 8 B/elem, min/max 4, fused min/max+quantize 8, MSE search 4, producer epilogue
    8 [+4 with a residual]).  The events cost a marker each on the stream; `forward_ms` is taken with the timer off."""
    BPE = {"quantize": 8, "minmax": 4, "minmax_quantize": 8, "mse_grid": 4, "affine_act_quantize": 8,
           "affine_act_minmax": 4, "multi_quantize": 8, "plan_launch": 8, "mse_select": 4, "mse_linspace": 4}
    MSE_CALIBRATE_BPE = 12      # one-call calibration step: the MSE search reads x (4) + K1 reads x and writes y (8)

    def __init__(self, ops):
        self.ops, self.on, self.rec = ops, False, []
        self.saved = {n: getattr(ops, n) for n in self.BPE if hasattr(ops, n)}
        for n, real in self.saved.items():
            setattr(ops, n, self._wrap(n, real))
        timer, base = self, ops.MultiPlan
        self.saved["MultiPlan"] = base

        class TimedPlan(base):
            def __init__(self, items):
                items = [tuple(it) for it in items]
                self._n = sum(it[0].numel() for it in items)
                super().__init__(items)

            def launch(self):
                if not timer.on:
                    return super().launch()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = super().launch()
                e1.record()
                timer.rec.append(("plan_launch", e0, e1, self._n, self._n * 8))
                return out
        ops.MultiPlan = TimedPlan
        cal = getattr(ops, "MseCalibration", None)
        if cal is not None:
            self.saved_step = real_step = cal.step

            def step(self_cal, x, quantize=True, pre=None):
                if not timer.on:
                    return real_step(self_cal, x, quantize, pre)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = real_step(self_cal, x, quantize, pre)
                e1.record()
                extra = 4 * x.numel() if (pre is not None and pre[1] is not None) else 0       # the residual
                timer.rec.append(("mse_calibrate_fused_epilogue" if pre is not None else "mse_calibrate", e0, e1, x.numel(),
                                  x.numel() * self.MSE_CALIBRATE_BPE + extra))
                return out
            cal.step = step

    def _wrap(self, name, real):
        bpe = self.BPE[name]

        def f(*a, **k):
            if not self.on:
                return real(*a, **k)
            if name == "multi_quantize":
                n = sum(it[0].numel() for it in a[0])
                extra = 0
            else:
                n = a[0].numel()
                extra = 4 * n if k.get("residual") is not None else 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = real(*a, **k)
            e1.record()
            self.rec.append((name, e0, e1, n, n * bpe + extra))
            return out
        return f

    def restore(self):
        for n, real in self.saved.items():
            setattr(self.ops, n, real)
        if getattr(self, "saved_step", None) is not None:
            self.ops.MseCalibration.step = self.saved_step

    def measure(self, fn):
        """run fn() once with the timer on -> {library_us, launches, elements, algorithmic_gb, by_entry}"""
        torch.cuda.synchronize()
        self.rec, self.on = [], True
        try:
            fn()
        finally:
            self.on = False
        torch.cuda.synchronize()
        by = {}
        for name, e0, e1, n, b in self.rec:
            d = by.setdefault(name, dict(calls=0, us=0.0, elements=0, bytes=0))
            d["calls"] += 1
            d["us"] += e0.elapsed_time(e1) * 1e3
            d["elements"] += n
            d["bytes"] += b
        tot_us = sum(d["us"] for d in by.values())
        tot_b = sum(d["bytes"] for d in by.values())
        for d in by.values():
            d["us"] = round(d["us"], 1)
            d["gb_s"] = round(d["bytes"] / max(d["us"], 1e-3) / 1e3, 1)
        return dict(library_us=round(tot_us, 1), launches=sum(d["calls"] for d in by.values()),
                    elements=sum(d["elements"] for d in by.values()), algorithmic_gb=round(tot_b / 1e9, 4),
                    gb_s=round(tot_b / max(tot_us, 1e-3) / 1e3, 1), frac_of_8tbs=round(tot_b / max(tot_us, 1e-3) / 1e3 / HBM_PEAK_GBS, 3),
                    by_entry=by)


def _median_pass(timer, fn, reps=5):
    """fn() `reps` times under the timer; the pass with the median library time"""
    runs = sorted((timer.measure(fn) for _ in range(reps)), key=lambda r: r["library_us"])
    return runs[len(runs) // 2]


def _wall_ms(fn, reps=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / reps * 1e3, 3)


def model_configs(ops, dev, only=None):
    """BASELINE configs 3 and 4 at their full size (batch 64 x 3 x 224 x 224, synthetic images, random-init weights with
    BN statistics re-estimated on synthetic batches, as `image_net.py validate-quantized --synthetic-batches` does):
    one calibration batch (estimate_ranges) -> fix_ranges -> validation forwards.

      c3  ResNet-18, fp_quantizer E5M2, per-channel current_minmax weights, per-tensor allminmax activations
          reference work per forward: 21 weight + 30 activation quantizer calls, 218.9 M elements,
          1.75 GB algorithmic (hijacker.py:88-108 re-quantizes every weight on every forward)
      c4  MobileNetV2, E4M3, MSE range estimator for weights and activations (range_estimators.py:318-369), with the
          mantissa search of the reference CLI's default (--fp8-mse-include-mantissa-bits: 6 widths) and without
          reference work per forward: 53 weight + 64 activation calls, 444.9 M elements, 3.56 GB

    Per pass: wall time, and -- by HIP events around every call into this library -- the GPU time of its launches,
    their elements and algorithmic bytes -> effective GB/s.  Validation forwards in three launch patterns:
      reference_pattern   FP8Q_CACHE_WEIGHTS=0 FP8Q_FUSE_EPILOGUE=0: one quantizer launch wherever the reference runs
                          its 13-op chain (all weights every forward, every activation after BN/ReLU as torch ops)
      cache0_fused        weights still re-quantized every forward, BN+ReLU(+residual)+quantizer in one kernel (N2)
      default             weights quantized once at fix_ranges() (cached), N2 fused
    """
    import image_net
    from models import QuantArchitectures
    from quantization.quantization_manager import QMethods
    from quantization.range_estimators import RangeEstimators

    timer = LaunchTimer(ops)
    out = {}
    g = torch.Generator(device=dev).manual_seed(4321)
    x = torch.randn(64, 3, 224, 224, device=dev, generator=g)
    xc = torch.randn(64, 3, 224, 224, device=dev, generator=g)         # the calibration batch

    def build(arch, mbits, w_est, a_est, search):
        torch.manual_seed(0)
        m = QuantArchitectures[arch](
            pretrained=False, load_type="fp32", method=QMethods.fp_quantizer.cls, n_bits=8, per_channel_weights=True,
            weight_range_method=RangeEstimators[w_est].cls, act_range_method=RangeEstimators[a_est].cls,
            fp8_kwargs=dict(maxval=None, mantissa_bits=mbits, set_maxval=True, learn_maxval=False,
                            learn_mantissa_bits=False, mse_include_mantissa_bits=search, allow_unsigned=False)).to(dev).eval()
        with torch.no_grad():
            m.full_precision()
            image_net.reestimate_bn_stats(m, image_net.SyntheticLoader(2, 64, 224, 1234), 2)
            for _ in range(2):
                m(x)                                  # MIOpen algorithm selection etc., outside every timed pass
        return m

    def calibrate(m):
        with torch.no_grad():
            m.set_quant_state(True, True)
            m.estimate_ranges()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            first = timer.measure(lambda: m(xc))
            first_ms = round((time.perf_counter() - t0) * 1e3, 3)
            # the same batch again from scratch (estimators reset): the first pass grows every workspace and result
            # buffer through the caching allocator (hipMalloc), which a calibration of more than one batch pays once
            from quantization.quantization_manager import QuantizationManager
            for mod in m.modules():
                if isinstance(mod, QuantizationManager) and mod.range_estimator is not None:
                    mod.range_estimator.reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rec = timer.measure(lambda: m(xc))
            rec["wall_ms_with_event_timer"] = round((time.perf_counter() - t0) * 1e3, 3)
            rec["first_pass"] = dict(wall_ms=first_ms, library_us=first["library_us"])
            # the pass as a user runs it: nothing attached (the event timer above costs two event records per library call
            # on the host -- ~8 ms on a pass of 350 calls in round 5, which made a GPU-bound pass look host-bound)
            walls, enq = [], []
            for _ in range(3):
                for mod in m.modules():
                    if isinstance(mod, QuantizationManager) and mod.range_estimator is not None:
                        mod.range_estimator.reset()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                m(xc)
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                walls.append((time.perf_counter() - t0) * 1e3)
                enq.append((t1 - t0) * 1e3)
            rec["wall_ms"] = round(sorted(walls)[1], 3)
            rec["host_enqueue_ms"] = round(sorted(enq)[1], 3)
            # batches >= 3 of a shape: the calibration forward replayed from a HIP graph (quantization/model.py:
            # GraphedCalibration -- the pass enqueues only, its decisions live in device state): no Python, no launches
            try:
                from quantization.base_quantized_model import GraphedCalibration
                for mod in m.modules():
                    if isinstance(mod, QuantizationManager) and mod.range_estimator is not None:
                        mod.range_estimator.reset()
                gc = GraphedCalibration(m)
                for _ in range(3):
                    gc(xc)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    gc(xc)
                torch.cuda.synchronize()
                rec["wall_ms_hipgraph_replay_batch4plus"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
                del gc
            except Exception as e:      # noqa: BLE001 -- informational entry only
                rec["wall_ms_hipgraph_replay_batch4plus"] = {"error": repr(e)[:200]}
            rec["wall_over_library"] = round(rec["wall_ms"] * 1e3 / max(rec["library_us"], 1e-3), 2)
            t0 = time.perf_counter()
            m.fix_ranges()
            torch.cuda.synchronize()
            rec["fix_ranges_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
        return rec

    def parity(m):
        """BASELINE.json's metric, second half ("ResNet-18 PTQ top-1 delta") on a box without ImageNet -- PART OF THE
        cpu_baseline LEG (rank 0, N = 1): the CPU oracle, as the checker, recomputes every quantizer launch of one validation
        batch inside the GPU model (oracle/in_the_loop.py; same convolutions on the GPU in both passes).  Bit-identical logits
        mean top-1 / top-5 of this engine ARE those of the arithmetic the oracle restates: the delta attributable to the
        kernels is exactly zero.  Plus the label-free proxy of the PTQ delta itself: arg-max agreement with the fp32 network."""
        from oracle.in_the_loop import validation_parity
        det = torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark
        torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = True, False
        try:
            with torch.no_grad():
                r = validation_parity(m, x, ops)
                yq = m(x)
                m.full_precision()
                yf = m(x)
                m.set_quant_state(True, True)
        finally:
            torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = det
        ok = r["logits_bit_identical"] and r["weights_bit_identical"] and r["gpu_pass_reproducible"]
        return dict(batch="64 x 3 x 224 x 224 synthetic, ranges fixed after one calibration batch",
                    argmax_agreement_with_fp32=round(float((yq.argmax(1) == yf.argmax(1)).float().mean()), 4),
                    top5_overlap_with_fp32=round(float(sum(len(set(a.tolist()) & set(b.tolist())) for a, b in
                                                            zip(yq.topk(5).indices, yf.topk(5).indices))) / (5 * yq.shape[0]), 4),
                    logits_bit_identical_with_oracle_quantizers=r["logits_bit_identical"],
                    quantized_weights_bit_identical_with_oracle=r["weights_bit_identical"], weights_checked=r["weights_checked"],
                    max_abs_logit_diff_vs_oracle_quantizers=r["max_abs_diff"],
                    top1_delta_attributable_to_kernels=0 if ok else None,
                    note="random-init weights (no checkpoints / ImageNet on the box): agreement with fp32 is a label-free proxy")

    def validation(m, ref_gb):
        res = {}
        with torch.no_grad():
            for name, cache, fuse in (("reference_pattern", "0", "0"), ("cache0_fused", "0", "1"), ("default", "1", "1")):
                os.environ["FP8Q_CACHE_WEIGHTS"], os.environ["FP8Q_FUSE_EPILOGUE"] = cache, fuse
                try:
                    if cache == "1":
                        m.requantize_weights()
                    for _ in range(3):
                        m(x)
                    r = _median_pass(timer, lambda: m(x))
                    r["forward_ms"] = _wall_ms(lambda: m(x))
                    r["env"] = f"FP8Q_CACHE_WEIGHTS={cache} FP8Q_FUSE_EPILOGUE={fuse}"
                    # the reference's quantizer work for this forward, done in the time this library's launches took
                    r["reference_work_gb_s"] = round(ref_gb * 1e6 / max(r["library_us"], 1e-3), 1)
                    res[name] = r
                finally:
                    os.environ.pop("FP8Q_CACHE_WEIGHTS", None)
                    os.environ.pop("FP8Q_FUSE_EPILOGUE", None)
        return res

    import contextlib
    import io
    quiet = contextlib.redirect_stdout(io.StringIO())       # the CLI helpers print progress lines
    try:
        with quiet:
            if only in (None, "c3"):
                m = build("resnet18_quantized", 2, "current_minmax", "allminmax", False)
                fp32_ms = _wall_ms(lambda: m(x))
                cal = calibrate(m)
                if only == "c3":
                    with torch.no_grad():
                        one = timer.measure(lambda: m(x))
                        for _ in range(20):
                            m(x)
                    torch.cuda.synchronize()
                    return {"c3_resnet18_b64": {"profiled": "1 calibration batch + fix_ranges + 1 event-timed + 20 plain default "
                                                            "validation forwards", "calibration_batch": cal,
                                                "validation_forward_default": one}}
                out["c3_resnet18_b64"] = dict(
                    workload="ResNet-18, batch 64 x 3 x 224 x 224 synthetic, fp_quantizer E5M2 (8 bit, 2 mantissa bits), "
                             "per-channel current_minmax weights, per-tensor allminmax activations",
                    reference_work_per_forward=dict(quantizer_calls=51, elements=218.9e6, algorithmic_gb=1.751),
                    fp32_forward_ms=fp32_ms, calibration_batch=cal, validation_forward=validation(m, 1.751))
                out["c3_resnet18_b64"].update(parity(m))
                del m
                torch.cuda.empty_cache()
            if only in (None, "c4", "c4_search"):
                entry = dict(
                    workload="MobileNetV2, batch 64 x 3 x 224 x 224 synthetic, fp_quantizer E4M3 (3 mantissa bits), MSE range "
                             "estimator (111-candidate grid search, K4) for per-channel weights and per-tensor activations",
                    reference_work_per_forward=dict(quantizer_calls=117, elements=444.9e6, algorithmic_gb=3.559))
                for key, search in (("calibration_batch_fixed_mantissa", False), ("calibration_batch_mantissa_search_6", True)):
                    if only == "c4" and search or only == "c4_search" and not search:
                        continue
                    m = build("mobilenet_v2_quantized", 3, "MSE", "MSE", search)
                    if "fp32_forward_ms" not in entry:
                        entry["fp32_forward_ms"] = _wall_ms(lambda: m(x))
                    cal = calibrate(m)
                    k4 = cal["by_entry"].get("mse_grid")
                    if k4 is None:      # the one-call step: search + selection + quantization of every MSE quantizer
                        parts = [cal["by_entry"].get(k) for k in ("mse_calibrate", "mse_calibrate_fused_epilogue")]
                        parts = [p_ for p_ in parts if p_]
                        k4 = dict(us=sum(p_["us"] for p_ in parts), elements=sum(p_["elements"] for p_ in parts)) if parts else None
                    if k4:
                        n_m = 6 if search else 1
                        cal["k4_share_of_library_time"] = round(k4["us"] / max(cal["library_us"], 1e-3), 3)
                        cal["k4_t_cand_elem_s"] = round(k4["elements"] * 111 * n_m / max(k4["us"], 1e-3) / 1e6, 3)
                    entry[key] = cal
                    if only is not None:
                        with torch.no_grad():
                            entry["validation_forward_default"] = timer.measure(lambda: m(x))
                            for _ in range(20):
                                m(x)
                        torch.cuda.synchronize()
                        entry["profiled"] = "1 calibration batch + fix_ranges + 1 event-timed + 20 plain default validation forwards"
                        return {"c4_mobilenetv2_b64": entry}
                    if not search:
                        entry["validation_forward"] = validation(m, 3.559)
                        entry.update(parity(m))
                    del m
                    torch.cuda.empty_cache()
                out["c4_mobilenetv2_b64"] = entry
    finally:
        timer.restore()
    return out


def _phase_us(timing, names, reps):
    """mean microseconds of every phase between consecutive events recorded by fp8q.dist (_mark), over `reps` calls."""
    evs = timing["events"]
    per = len(evs) // reps
    acc = [0.0] * (per - 1)
    for r in range(reps):
        e = evs[r * per:(r + 1) * per]
        for i in range(per - 1):
            acc[i] += e[i].elapsed_time(e[i + 1]) * 1e3
    return {n: round(v / reps, 1) for n, v in zip(names, acc)}


def north_star_path(args, ops, dev, rank, world, backend):
    """The two exchange steps of the path (see the module docstring); every rank runs this, rank 0 reports."""
    from fp8q import dist as fd
    res = {}
    sync_dev = dev if backend == "nccl" else "cpu"

    def wall(fn, reps):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=sync_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt / reps

    if world > 1:
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)
        res["ranks_seen"] = int(one.item())
    else:
        res["ranks_seen"] = 1
    res["backend"] = {"nccl": "RCCL", "gloo": "gloo (smoke test only)"}[backend] if world > 1 else "none (1 rank)"

    # ---- weights: channel shard -> fused min/max + quantize -> all-gather (fp32, and 1-byte codes) ----
    C = world * (1 << 18)
    g = torch.Generator(device=dev).manual_seed(99)          # the same full tensor on every rank
    w = torch.empty(C, 3, 7, 7, device=dev)
    for i in range(0, C, 1 << 18):
        w[i:i + (1 << 18)].normal_(generator=g)
    w *= 0.1
    reps = max(3, min(args.steps, 10))
    n_w = w.numel()
    wa = {"tensor": f"[{C},3,7,7] fp32 on every rank, rank r quantizes channels channel_partition({C},{world})[r] "
                    f"(E5M2, current_minmax)", "elements": n_w}
    for name, fn, names, wire in (
            ("fp32", lambda t: fd.quantize_weight_sharded(w, MBITS, NBITS, SIGN, timing=t), ("kernel_us", "collective_us"), 4),
            ("codes_u8", lambda t: fd.quantize_weight_sharded_codes(w, MBITS, NBITS, SIGN, timing=t),
             ("kernel_us", "collective_us", "decode_us"), 1)):
        for _ in range(2):
            fn(None)
        timing = {}
        for _ in range(reps):
            fn(timing)
        torch.cuda.synchronize()
        ph = _phase_us(timing, names, reps)
        dt = wall(lambda: fn(None), reps)
        ph.update(step_us=round(dt * 1e6, 1), gelem_s=round(n_w / dt / 1e9, 1),
                  xgmi_bytes_received_per_rank=int((n_w // world) * (world - 1) * wire + (C // world) * (world - 1) * 4))
        wa[name] = ph
    res["weights_allgather"] = wa
    del w
    torch.cuda.empty_cache()

    # ---- the real thing, strong-scaled: ResNet-18's 21 weight tensors (11.68 M elements, 46.7 MB), every tensor
    # channel-sharded, ONE packed all-gather for the whole model (fp8q.dist.quantize_weights_sharded_bucketed) ----
    shapes = [(64, 3, 7, 7)] + [(64, 64, 3, 3)] * 4 + [(128, 64, 3, 3), (128, 128, 3, 3), (128, 64, 1, 1)] + \
        [(128, 128, 3, 3)] * 2 + [(256, 128, 3, 3), (256, 256, 3, 3), (256, 128, 1, 1)] + [(256, 256, 3, 3)] * 2 + \
        [(512, 256, 3, 3), (512, 512, 3, 3), (512, 256, 1, 1)] + [(512, 512, 3, 3)] * 2 + [(1000, 512)]
    g = torch.Generator(device=dev).manual_seed(7)
    ws = [torch.randn(*sh, device=dev, generator=g) * 0.05 for sh in shapes]
    for _ in range(2):
        fd.quantize_weights_sharded_bucketed(ws, MBITS, NBITS, SIGN)
    dt = wall(lambda: fd.quantize_weights_sharded_bucketed(ws, MBITS, NBITS, SIGN), reps)
    n_r18 = sum(t.numel() for t in ws)
    # the default wire form at N > 1: 1-byte storage codes + fp32 ranges (a quarter of the fp32 bytes); fp32 for comparison
    dt32 = wall(lambda: fd.quantize_weights_sharded_bucketed(ws, MBITS, NBITS, SIGN, wire="fp32"), reps) if world > 1 else dt
    wire_b = 1 if world > 1 else 4
    res["resnet18_weights_one_allgather"] = dict(
        tensors=len(ws), elements=n_r18, step_us=round(dt * 1e6, 1), gelem_s=round(n_r18 / dt / 1e9, 2),
        wire="codes_u8 + fp32 ranges" if world > 1 else "fp32 (one rank: nothing is shipped)",
        fp32_wire_step_us=round(dt32 * 1e6, 1), fp32_wire_gelem_s=round(n_r18 / dt32 / 1e9, 2),
        scaling="strong (the model is fixed; every rank quantizes 1/N of every tensor's channels)",
        xgmi_bytes_received_per_rank=int(n_r18 * wire_b * (world - 1) / world))
    del ws

    # ---- config 5: allminmax fold -> range all-reduce -> E4M3 quantize of the per-rank slab ----
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.empty(512, 4096, 512, device=dev)
    for i in range(0, 512, 64):
        x[i:i + 64].normal_(generator=g)
    y = torch.empty_like(x)
    state = [None]

    def c5(t=None):
        _, state[0] = fd.calibrate_quantize_sharded(x, 3, 8, 1, state=state[0], out=y, timing=t)

    for _ in range(2):
        c5()
    timing = {}
    for _ in range(reps):
        c5(timing)
    torch.cuda.synchronize()
    ph = _phase_us(timing, ("minmax_us", "collective_us", "quantize_us"), reps)
    dt = wall(c5, reps)
    n = x.numel()
    ph.update(step_ms=round(dt * 1e3, 4), gelem_s=round(n * world / dt / 1e9, 1), elements_per_gpu=n,
              hbm_gb_s_per_gpu=round(n * 12 / dt / 1e9, 1), frac_of_8tbs=round(n * 12 / dt / 1e9 / HBM_PEAK_GBS, 4),
              flow="BASELINE config 5: per-rank slab [512,4096,512] fp32, running min/max fold (4 B/elem) -> one "
                   "all-reduce of the 2-float range -> E4M3 quantize with the global range (8 B/elem)")
    res["c5"] = ph
    del x, y
    torch.cuda.empty_cache()
    return res


class _DryOps:
    """Compute stub of `--dry-run-ranks`: the shapes, dtypes and buffers of fp8q.ops with torch CPU ops and NO FP8
    arithmetic (quantize = copy).  It exists so that the collective call sequence of fp8q.dist can run without a GPU."""

    @staticmethod
    def quantize(x, maxval, mbits, n_bits=8, sign_bits=1, out=None):
        if out is None:
            return x.clone()
        out.copy_(x.reshape(out.shape))
        return out

    @staticmethod
    def new_packed(C, device):
        return torch.empty((C, 4), dtype=torch.float32, device=device)

    @staticmethod
    def minmax(x, per_channel, cur_min=None, cur_max=None, mode=0, momentum=0.9, want_maxval=False, packed=None):
        f = x.reshape(x.shape[0], -1) if per_channel else x.reshape(1, -1)
        mn, mx = f.min(1)[0], f.max(1)[0]
        if cur_min is not None and mode == 1:
            mn, mx = torch.min(cur_min, mn), torch.max(cur_max, mx)
        if packed is not None:
            packed.copy_(torch.stack([-mn, mx, torch.zeros_like(mn), torch.zeros_like(mn)], 1))
        mv = torch.max(mn.abs(), mx).abs()
        return (mn, mx, mv) if want_maxval else (mn, mx)

    @staticmethod
    def ranges_unpack(packed, cur_min=None, cur_max=None, maxval=None):
        p = packed.reshape(-1, 4)
        for dst, v in ((cur_min, -p[:, 0]), (cur_max, p[:, 1]), (maxval, torch.max(p[:, 0].abs(), p[:, 1]).abs())):
            if dst is not None:
                dst.copy_(v)
        return cur_min, cur_max, maxval

    @staticmethod
    def minmax_quantize(x, mbits, n_bits=8, sign_bits=1, out=None):
        mn, mx, mv = _DryOps.minmax(x, True, want_maxval=True)
        return _DryOps.quantize(x, mv, mbits, out=out), mn, mx, mv

    @staticmethod
    def encode(x, maxval, mbits, n_bits=8, sign_bits=1, out=None):
        return torch.zeros(x.shape, dtype=torch.uint8)

    @staticmethod
    def decode(codes, maxval, mbits, n_bits=8, sign_bits=1, out=None):
        return torch.zeros(codes.shape, dtype=torch.float32)


def dry_run(args, world, rank):
    """`--dry-run-ranks R`: the N-rank call sequence of this script without a GPU.  R gloo ranks, CPU tensors, _DryOps:
    the headline flow (quantize_weight_sharded), the 1-byte-codes variant, ResNet-18's 21 tensors through ONE bucketed
    all-gather (real shapes) and config 5's fold -> all-reduce -> quantize, tensors scaled down to `scale` channels /
    slab rows per rank.  Every collective is logged with its operand bytes; rank 0 prints the sequence next to the
    byte counts of the full-size run, so the first real multi-GPU run cannot fail on plumbing or be surprised by sizes."""
    R = args.dry_run_ranks
    if "WORLD_SIZE" not in os.environ:
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
               f"--nproc-per-node={R}", os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    if world != R:
        sys.exit(f"bench.py: --dry-run-ranks {R} but the launcher started WORLD_SIZE={world} ranks")
    dist.init_process_group("gloo")
    from fp8q import dist as fd
    log = []
    real = {n: getattr(dist, n) for n in ("all_gather_into_tensor", "all_reduce", "all_gather", "barrier")}

    def spy(name):
        def f(*a, **k):
            ts = [t for t in a if isinstance(t, torch.Tensor)] + [t for arg in a if isinstance(arg, (list, tuple)) for t in arg
                                                                   if isinstance(t, torch.Tensor)]
            send = ts[-1] if name == "all_gather_into_tensor" else (ts[0] if ts else None)
            log.append(dict(op=name, dtype=str(send.dtype).replace("torch.", "") if send is not None else None,
                            send_bytes=int(send.numel() * send.element_size()) if send is not None else 0,
                            reduce=str(k.get("op", "")).split(".")[-1] or None))
            return real[name](*a, **k)
        return f
    for n in real:
        setattr(dist, n, spy(n))
    try:
        scale = 1 << 10
        ops = _DryOps()
        seq = {}

        def section(name, fn):
            start = len(log)
            fn()
            seq[name] = log[start:]
        g = torch.Generator().manual_seed(1234)
        w = torch.randn(world * scale, 3, 7, 7, generator=g) * 0.1
        mv = ops.minmax(w, True, want_maxval=True)[2]
        section("headline: quantize_weight_sharded_codes(fixed ranges)",
                lambda: fd.quantize_weight_sharded_codes(w, MBITS, NBITS, SIGN, maxval=mv, ops=ops))
        section("headline, fp32 wire (value_fp32_wire): quantize_weight_sharded(fixed ranges)",
                lambda: fd.quantize_weight_sharded(w, MBITS, NBITS, SIGN, maxval=mv, ops=ops))
        section("weights_allgather.fp32: quantize_weight_sharded(current_minmax)", lambda: fd.quantize_weight_sharded(w, MBITS, NBITS, SIGN, ops=ops))
        section("weights_allgather.codes_u8: quantize_weight_sharded_codes", lambda: fd.quantize_weight_sharded_codes(w, MBITS, NBITS, SIGN, ops=ops))
        shapes = [(64, 3, 7, 7)] + [(64, 64, 3, 3)] * 4 + [(128, 64, 3, 3), (128, 128, 3, 3), (128, 64, 1, 1)] + \
            [(128, 128, 3, 3)] * 2 + [(256, 128, 3, 3), (256, 256, 3, 3), (256, 128, 1, 1)] + [(256, 256, 3, 3)] * 2 + \
            [(512, 256, 3, 3), (512, 512, 3, 3), (512, 256, 1, 1)] + [(512, 512, 3, 3)] * 2 + [(1000, 512)]
        ws = [torch.randn(*sh, generator=g) * 0.05 for sh in shapes]
        section("resnet18_weights_one_allgather: quantize_weights_sharded_bucketed (REAL shapes)",
                lambda: fd.quantize_weights_sharded_bucketed(ws, MBITS, NBITS, SIGN, ops=ops))          # default wire: 1-byte codes
        section("resnet18_weights_one_allgather, fp32 wire (round 4's form)",
                lambda: fd.quantize_weights_sharded_bucketed(ws, MBITS, NBITS, SIGN, ops=ops, wire="fp32"))
        xs = torch.randn(8, 64, 64, generator=torch.Generator().manual_seed(1234 + rank))
        section("c5: calibrate_quantize_sharded", lambda: fd.calibrate_quantize_sharded(xs, 3, 8, 1, ops=ops))
        one = torch.ones(1)
        dist.all_reduce(one)
    finally:
        for n, f in real.items():
            setattr(dist, n, f)
    if rank == 0:
        ne, nc = N_CH * ROW, N_CH
        full = {
            "headline (value at --gpus N)": {"all_gather_into_tensor codes": {"send_bytes_per_rank": ne, "received_per_rank": (R - 1) * ne},
                                             "bound": "received_per_rank <= (R - 1) / R * R * 2^21 * 147 B (1 byte per element of the other ranks' shards; the ranges are fixed and not shipped)"},
            "headline, fp32 wire (value_fp32_wire)": {"all_gather_into_tensor values": {"send_bytes_per_rank": ne * 4, "received_per_rank": (R - 1) * ne * 4},
                                                      "all_gather_into_tensor ranges": {"send_bytes_per_rank": nc * 4, "received_per_rank": (R - 1) * nc * 4}},
            "weights_allgather [N*2^18,3,7,7] codes_u8": {"all_gather_into_tensor codes": {"send_bytes_per_rank": (1 << 18) * ROW},
                                                          "all_gather_into_tensor ranges": {"send_bytes_per_rank": (1 << 18) * 4}},
            "resnet18_weights_one_allgather": {"all_gather_into_tensor bucket": {
                "send_bytes_per_rank": seq["resnet18_weights_one_allgather: quantize_weights_sharded_bucketed (REAL shapes)"][0]["send_bytes"],
                "fp32_wire_send_bytes_per_rank": seq["resnet18_weights_one_allgather, fp32 wire (round 4's form)"][0]["send_bytes"],
                "note": "real shapes: this IS the full-size count (1-byte codes + fp32 ranges of every tensor's shard, padded to 16 B)"}},
            "c5 [512,4096,512] per rank": {"all_reduce MAX": {"send_bytes_per_rank": 16, "note": "{-min, max, nan flags}: 4 floats"}},
        }
        # the headline's wire budget: one collective, 1 byte per element of this rank's shard, so a rank
        # receives (R - 1) / R of the R * 2^21 * 147 bytes of the full tensor and nothing else
        head = seq["headline: quantize_weight_sharded_codes(fixed ranges)"]
        assert len(head) == 1 and head[0]["op"] == "all_gather_into_tensor" and head[0]["dtype"] == "uint8", head
        assert head[0]["send_bytes"] == scale * ROW, head
        assert (R - 1) * ne <= (R - 1) * N_CH * ROW and (R - 1) * ne * R <= (R - 1) * (R * N_CH * ROW)
        print(json.dumps({"dry_run": True, "ranks": R, "ranks_seen": int(one.item()), "backend": "gloo (CPU tensors, compute stub)",
                          "scaled_down_to": {"channels_per_rank": scale, "c5_slab": [8, 64, 64]},
                          "call_sequence_as_executed": seq, "full_size_bytes": full}), flush=True)
    dist.destroy_process_group()
    if int(one.item()) != R:
        sys.exit(f"bench.py: dry run spanned {int(one.item())} ranks, expected {R}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)      # 0.2 s timed region by default
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-traffic", action="store_true",
                    help="do not measure roofline.traffic live (two short rocprofv3 --pmc passes of this script, N = 1 only); "
                         "the figure recorded under profiles/ is reported instead")
    ap.add_argument("--no-model-configs", action="store_true",
                    help="skip extras.c3_resnet18_b64 / extras.c4_mobilenetv2_b64 (BASELINE configs 3 and 4)")
    ap.add_argument("--only-model-config", choices=["c3", "c4", "c4_search"], default=None,
                    help="profiling aid: run ONLY that model configuration's passes (for rocprofv3 --kernel-trace) "
                         "and print its entry")
    ap.add_argument("--no-north-star-path", action="store_true",
                    help="skip the sharded-weights / config-5 section (profiling runs of the headline kernel)")
    ap.add_argument("--channels-per-gpu", type=int, default=N_CH,
                    help="rows of the headline tensor per GPU (default 2^21 = the judged workload; smaller only for smoke "
                         "tests of the multi-process path over gloo, disclosed in config.workload)")
    ap.add_argument("--dry-run-ranks", type=int, default=0,
                    help="no GPU: run the N-rank call sequence (headline flow + north_star_path) on that many gloo ranks "
                         "with CPU tensors and a compute stub, print every collective and its byte count")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for --gpus N > 1 (nccl = RCCL; gloo only for smoke-testing the "
                         "multi-process path on a box with fewer GPUs than ranks)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run_ranks:
        return dry_run(args, world, rank)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU) through the same launcher
        # the driver uses; its rank 0 prints the JSON line, this process only relays the exit code
        import subprocess
        # --standalone: the launcher picks its own free rendezvous port (no bind-then-close race between benches)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
               f"--nproc-per-node={args.gpus}", os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the FP8 engine has no CPU path)")
    if args.backend == "nccl" and world > torch.cuda.device_count():
        sys.exit(f"bench.py: {world} RCCL ranks need {world} GPUs, this node shows {torch.cuda.device_count()} "
                 "(--backend gloo lets several ranks share a GPU: smoke test of the multi-process path only)")
    dev_index = local_rank if args.backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    import fp8q
    ops = fp8q.ops
    fp8q.lib()  # fail loudly if the HIP library is missing
    if args.only_model_config:
        print(json.dumps(model_configs(ops, dev, only=args.only_model_config)), flush=True)
        return

    n_ch = int(args.channels_per_gpu)
    from fp8q import dist as fd
    if world == 1:
        # synthetic weights: this GPU's output channels
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        x = torch.randn(n_ch, 3, 7, 7, device=dev, generator=g) * 0.1
        y = torch.empty_like(x)
        _, _, maxval = ops.minmax(x, True, want_maxval=True)   # current_minmax + set_quant_range, once
        shard = x

        def step(t=None):
            ops.quantize(x, maxval, MBITS, NBITS, SIGN, out=y)
    else:
        # north_star's weight flow: the SAME full tensor on every rank (same seed), rank r owns n_ch of its channels
        g = torch.Generator(device=dev).manual_seed(1234)
        x = torch.empty(world * n_ch, 3, 7, 7, device=dev)
        for i in range(0, world * n_ch, 1 << 18):
            x[i:i + (1 << 18)].normal_(generator=g)
        x *= 0.1
        _, _, maxval = ops.minmax(x, True, want_maxval=True)
        lo, hi = fd.channel_partition(world * n_ch, world)[rank]
        shard, y = x[lo:hi], torch.empty(hi - lo, 3, 7, 7, device=dev)
        full = [None]

        # quantize this rank's channels (fixed ranges, as at N = 1), all-gather, re-assemble.  Round 6: the shards travel as
        # 1-byte storage codes (fp8q_encode_u8 -> all-gather -> fp8q_decode_u8): bit-identical tensors, a quarter of the xGMI
        # bytes of the fp32 form, which is timed right after the judged region and reported as value_fp32_wire
        def step(t=None):
            full[0] = fd.quantize_weight_sharded_codes(x, MBITS, NBITS, SIGN, maxval=maxval, timing=t)

        def step_fp32(t=None):
            full[0] = fd.quantize_weight_sharded(x, MBITS, NBITS, SIGN, maxval=maxval, timing=t)
    torch.cuda.synchronize()

    # Setup, before the W warm-up steps of the contract: the GPU's clocks need tens of milliseconds of
    # sustained work to settle (the first ~40 launches after an idle period run 8-10 % slower), so the
    # kernel is run for PREWARM_S first; disclosed in the JSON line as config.prewarm_ms.
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < PREWARM_S:
        for _ in range(10):
            ops.quantize(shard, maxval[:shard.shape[0]] if world == 1 else maxval[lo:hi], MBITS, NBITS, SIGN, out=y)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs, timing = [], {}
    for _ in range(args.steps):
        if world == 1:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            step()
            b.record()
            evs.append((a, b))
        else:
            step(timing)        # fp8q.dist records an event before the kernel, after it and after the collective
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    collective_s = 0.0
    if world > 1:
        t = torch.tensor([elapsed], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        e = timing["events"]       # per step: before | encode done | all-gather done | decode done
        step_s = sorted((e[4 * i].elapsed_time(e[4 * i + 1]) + e[4 * i + 2].elapsed_time(e[4 * i + 3])) * 1e-3 for i in range(args.steps))
        collective_s = sum(e[4 * i + 1].elapsed_time(e[4 * i + 2]) * 1e-3 for i in range(args.steps)) / max(args.steps, 1)
        # the fp32 wire form of the same flow (round 5's headline), same steps, outside the judged region
        for _ in range(min(args.warmup, 2)):
            step_fp32()
        torch.cuda.synchronize()
        dist.barrier()
        t32 = time.perf_counter()
        for _ in range(args.steps):
            step_fp32()
        torch.cuda.synchronize()
        dist.barrier()
        t32 = torch.tensor([time.perf_counter() - t32], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t32, op=dist.ReduceOp.MAX)
        elapsed_fp32 = float(t32.item())
    else:
        step_s = sorted(a.elapsed_time(b) * 1e-3 for a, b in evs)
    kern_s = sum(step_s) / max(len(step_s), 1)

    n_elem = shard.numel()
    total_elem = n_elem * world
    value = total_elem * args.steps / elapsed / 1e9
    achieved = n_elem * BYTES_PER_ELEM / kern_s / 1e9
    wall_gbs = n_elem * BYTES_PER_ELEM / (elapsed / args.steps) / 1e9
    rx_bytes = (world - 1) * n_elem                              # what the all-gather brings to a rank: 1-byte codes (the ranges are fixed: everywhere already)
    rx_bytes_fp32 = (world - 1) * (n_elem + shard.shape[0]) * 4  # the fp32 form: values + ranges

    line = None
    if rank == 0:
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if world == 1 and not args.no_traffic:
            traffic, detail = measure_traffic()
            if traffic is not None:
                traffic_source = ("measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over "
                                  "`bench.py --steps 3 --warmup 1` of the same kernel; FETCH x 1024 x 2 (gfx950 correction) + "
                                  f"WRITE x 1024; {detail}")
            else:
                traffic_source = f"live PMC passes unavailable ({detail}); "
        if traffic is None and os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("k1_bytes_per_launch")
                traffic_source = (traffic_source or "") + (
                    "profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of "
                    "this command, recorded when the profile was taken -- NOT measured in this run")
            except Exception:
                traffic = None
        line = {
            "metric": "FP8 quant+dequant Gelems/sec", "value": round(value, 2), "unit": "Gelem/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "value_kernel_only": round(total_elem / kern_s / 1e9, 2),
            "config": {"workload": f"conv1-shaped weights [{n_ch},3,7,7] fp32 per GPU, per-channel E5M2 "
                                   "quantize+dequantize, fixed ranges from current_minmax (BASELINE config 2, "
                                   "synthetic NxCxKxK scale-up)" + ("" if n_ch == N_CH else " -- REDUCED by --channels-per-gpu: smoke test, not the judged workload"),
                       "elements_per_gpu": n_elem,
                       "parallelism": "1 GPU" if world == 1 else (
                           f"one [{world * n_ch},3,7,7] tensor on every rank, channel-sharded x{world}: each rank quantizes its "
                           f"{n_ch} channels to 1-byte storage codes, one all-gather re-assembles the codes on every rank, every rank "
                           "decodes (fixed ranges: every rank holds them already); `value` INCLUDES the collective, "
                           "`value_kernel_only` (encode + decode launches) does not; `value_fp32_wire`: the same flow shipping fp32 values"),
                       "prewarm_ms": int(PREWARM_S * 1e3)},
            "roofline": {"bound": "hbm", "kernel": "k_rows_flat<0,NT>", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "frac_wall": round(wall_gbs / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": n_elem * BYTES_PER_ELEM,
                         "avg_launch_us": round(kern_s * 1e6, 1),
                         "median_launch_us": round(step_s[len(step_s) // 2] * 1e6, 1),
                         "min_launch_us": round(step_s[0] * 1e6, 1), "max_launch_us": round(step_s[-1] * 1e6, 1)},
        }
        if world > 1:
            line.update(kernel_us=round(kern_s * 1e6, 1), collective_us=round(collective_s * 1e6, 1),
                        wire="codes_u8 (1 byte per element; ranges fixed, not shipped)",
                        xgmi_bytes_received_per_rank=int(rx_bytes),
                        xgmi_gb_s=round(rx_bytes / max(collective_s, 1e-9) / 1e9, 2) if args.backend == "nccl" else None,
                        value_fp32_wire=round(total_elem * args.steps / elapsed_fp32 / 1e9, 2),
                        fp32_wire_ms_per_step=round(elapsed_fp32 / args.steps * 1e3, 4),
                        fp32_wire_xgmi_bytes_received_per_rank=int(rx_bytes_fp32))
        if world == 1 and not args.no_cpu_baseline:
            # bounded sample: generate only what the CPU leg needs
            sample_ch = 1 << 18
            xc = x[:sample_ch].cpu()
            mvc = maxval[:sample_ch].cpu()
            port, ref, n_ch = cpu_baseline(xc.numpy(), mvc.numpy())
            # `value` = the reference's CPU path: its own eager ATen op chain on the host cores (oracle/torch_eager.py restates
            # fp8_quantizer.py:91-133 op for op; the reference files do not travel to the GPU box).  The fused C port of the
            # oracle (OpenMP, all threads) rides along as port_*: the "fair fused CPU" figure of SURVEY.md 8(d).
            cb = torch_eager_cpu(xc, mvc, min(n_ch, 1 << 16))
            cb.update(port_value=port["value"], port_unit=port["unit"], port_cores=port["cores"], port_kind=port["kind"],
                      port_sample=port["sample"])
            line["cpu_baseline"] = cb
            # the port's sample doubles as a parity check of the timed kernel's output
            import numpy as np
            got = y[:n_ch].cpu().numpy()
            line["cpu_baseline"]["gpu_output_bit_exact_on_sample"] = bool(
                np.array_equal(got.view(np.int32), ref.view(np.int32)))
    del x, y, shard
    if world > 1:
        del full
    torch.cuda.empty_cache()
    if not args.no_north_star_path:
        nsp = north_star_path(args, ops, dev, rank, world, args.backend)     # every rank takes part
        if rank == 0:
            line["north_star_path"] = nsp
            # the other end-to-end rates of the path, as top-level keys (collectives included at N > 1)
            line["value_codes_wire"] = nsp["weights_allgather"]["codes_u8"]["gelem_s"]
            line["value_resnet18_strong"] = nsp["resnet18_weights_one_allgather"]["gelem_s"]
            line["value_c5"] = nsp["c5"]["gelem_s"]
    # every rank must have been seen by a collective: an all-reduce of ones before anything is reported
    seen = world
    if world > 1:
        one = torch.ones(1, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(one)
        seen = int(one.item())
    if rank == 0:
        line["ranks_seen"] = seen
        if world == 1 and not args.no_extras:
            line["extras"] = extras(ops, dev)
            if not args.no_model_configs:
                line["extras"].update(model_configs(ops, dev))
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if seen != args.gpus:
        sys.exit(f"bench.py: the collectives spanned {seen} ranks, expected {args.gpus}")


if __name__ == "__main__":
    main()
