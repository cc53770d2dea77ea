"""Host timeline of one MobileNetV2 MSE calibration pass and of fix_ranges().

  python tools/host_profile.py [search|fixed] [out.txt]

Prints (a) wall time of the calibration forward with nothing attached (no event timer, no profiler), (b) the same under
cProfile: the top functions by own time and by cumulative time, (c) fix_ranges() split into its steps, each bracketed by a
device synchronisation, (d) the count of torch dispatcher ops and library calls of the pass.
"""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fp8-quantization_amd")]
import torch  # noqa: E402

import fp8q  # noqa: E402
import image_net  # noqa: E402
from models import QuantArchitectures  # noqa: E402
from quantization import model as qmodel  # noqa: E402
from quantization.quantization_manager import QMethods, QuantizationManager  # noqa: E402
from quantization.range_estimators import RangeEstimators  # noqa: E402


def build(search, arch="mobilenet_v2_quantized", mbits=3, w_est="MSE", a_est="MSE"):
    torch.manual_seed(0)
    m = QuantArchitectures[arch](
        pretrained=False, load_type="fp32", method=QMethods.fp_quantizer.cls, n_bits=8, per_channel_weights=True,
        weight_range_method=RangeEstimators[w_est].cls, act_range_method=RangeEstimators[a_est].cls,
        fp8_kwargs=dict(maxval=None, mantissa_bits=mbits, set_maxval=True, learn_maxval=False,
                        learn_mantissa_bits=False, mse_include_mantissa_bits=search, allow_unsigned=False)).cuda().eval()
    with torch.no_grad():
        m.full_precision()
        image_net.reestimate_bn_stats(m, image_net.SyntheticLoader(2, 64, 224, 1234), 2)
    return m


def reset(m):
    for mod in m.modules():
        if isinstance(mod, QuantizationManager) and mod.range_estimator is not None:
            mod.range_estimator.reset()


def wall(fn, reps=1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / reps * 1e3, (t2 - t0) / reps * 1e3       # enqueue-only, with the drain


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "search"
    if len(sys.argv) > 2:
        os.makedirs(os.path.dirname(os.path.abspath(sys.argv[2])), exist_ok=True)
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    p = lambda *a: print(*a, file=out, flush=True)       # noqa: E731
    search = mode == "search"
    arch = "resnet18_quantized" if mode == "c3" else "mobilenet_v2_quantized"
    if mode == "c3":
        m = build(False, arch, 2, "current_minmax", "allminmax")
    else:
        m = build(search)
    g = torch.Generator(device="cuda").manual_seed(4321)
    x = torch.randn(64, 3, 224, 224, device="cuda", generator=g)
    xc = torch.randn(64, 3, 224, 224, device="cuda", generator=g)
    with torch.no_grad():
        for _ in range(3):
            m(x)
        e, d = wall(lambda: m(x), 10)
        p(f"[{mode}] fp32 forward: enqueue {e:.2f} ms, with drain {d:.2f} ms")
        m.set_quant_state(True, True)
        m.estimate_ranges()
        e, d = wall(lambda: m(xc))
        p(f"calibration pass 1 (cold: allocations): enqueue {e:.2f} ms, with drain {d:.2f} ms")
        for i in range(3):
            reset(m)
            e, d = wall(lambda: m(xc))
            p(f"calibration pass (first batch again, warm) #{i}: enqueue {e:.2f} ms, with drain {d:.2f} ms")
        e, d = wall(lambda: m(xc))
        p(f"calibration pass, SECOND batch of the same estimators: enqueue {e:.2f} ms, with drain {d:.2f} ms")
        # --- cProfile of one warm first-batch pass
        reset(m)
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        pr.enable()
        m(xc)
        pr.disable()
        torch.cuda.synchronize()
        for key in ("tottime", "cumulative"):
            s = io.StringIO()
            pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(45)
            p(f"---- cProfile by {key} ----")
            p(s.getvalue())
        # --- torch dispatcher ops of the pass
        reset(m)
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU]) as prof:
            m(xc)
        torch.cuda.synchronize()
        p("---- torch ops (CPU side) ----")
        p(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=30))
        # --- fix_ranges in steps
        reset(m)
        m(xc)
        torch.cuda.synchronize()

        def step(name, fn):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            p(f"fix_ranges step {name}: host {1e3 * (t1 - t0):.3f} ms, with drain {1e3 * (t2 - t0):.3f} ms -> {r}")
            return r
        from quantization.layers import _for_managers
        step("managers.fix_ranges", lambda: _for_managers(m, lambda mm: mm.fix_ranges(), need_init=True))
        step("materialize_mantissa_bits", lambda: qmodel.materialize_mantissa_bits(m))
        step("check_workspaces", lambda: fp8q.ops.check_workspaces())
        step("release_workspaces", lambda: fp8q.ops.release_workspaces())
        found = step("_plan_layers", lambda: len(qmodel._plan_layers(m)))
        pr = cProfile.Profile()
        pr.enable()
        step("prequantize_weights", lambda: m.prequantize_weights())
        pr.disable()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).strip_dirs().sort_stats("tottime").print_stats(25)
        p("---- cProfile of prequantize_weights ----")
        p(s.getvalue())
        held = qmodel._PLANS.get(m)
        if held is not None:
            p("plan launches:", held[0].launches, "layers:", found)
        # whole fix_ranges once more on a fresh calibration
        m.estimate_ranges()
        reset(m)
        m(xc)
        e, d = wall(lambda: m.fix_ranges())
        p(f"fix_ranges() whole: host {e:.3f} ms, with drain {d:.3f} ms")
        e, d = wall(lambda: m(x), 10)
        p(f"validation forward: enqueue {e:.2f} ms, with drain {d:.2f} ms")


if __name__ == "__main__":
    main()
