"""The one-call MSE calibration step (fp8q_mse_calibrate_f32: [epilogue +] abs-max + grid, search, selection, quantization) on
MobileNetV2's activation shapes at batch 64, first batch of a fresh estimator, GPU time per call by HIP events around 8 calls.

  python tools/mb_calib_shapes.py [fixed|search] [pre|plain] [C,hw ...]      # e.g. 32,112 96,14
  FP8Q_MSE_SLICE / FP8Q_MSE_HIST / ... act as usual (A/B runs: one setting per process)

With rocprofv3 --kernel-trace around it, tools/calib_timeline.py turns the trace into a per-shape kernel timeline."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fp8-quantization_amd")]
import torch  # noqa: E402

import fp8q  # noqa: E402

SHAPES = [(32, 112), (16, 112), (96, 112), (96, 56), (24, 56), (144, 56), (144, 28), (32, 28), (192, 28), (192, 14), (64, 14),
          (384, 14), (96, 14), (576, 14), (576, 7), (160, 7), (960, 7), (320, 7), (1280, 7)]


def main():
    ops = fp8q.ops
    args = sys.argv[1:]
    search = "search" in args
    pre = "pre" in args
    shapes = [tuple(int(v) for v in a.split(",")) for a in args if "," in a] or SHAPES
    mb = [float(m) for m in range(1, 7)] if search else [3.0]
    torch.manual_seed(0)
    tot = 0.0
    reps = 8
    for C, hw in shapes:
        x = torch.randn(64, C, hw, hw, device="cuda") * 2.5
        ab = None
        if pre:
            bn = (torch.randn(C, device="cuda") * 0.1, torch.rand(C, device="cuda") + 0.7, torch.rand(C, device="cuda") + 0.5,
                  torch.randn(C, device="cuda") * 0.1)
            ab = ops.bn_fold(bn)
        else:
            x = torch.clamp(x, 0, 6)
        cals = [ops.MseCalibration(1, x.device, mb, 8, 1) for _ in range(reps + 2)]
        for c in cals[:2]:
            c.step(x, pre=(ab, None, 2) if pre else None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for c in cals[2:]:
            c.step(x, pre=(ab, None, 2) if pre else None)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        tot += us
        print(f"[64,{C},{hw},{hw}] n={x.numel():9d} {'search' if search else 'fixed '} {'pre  ' if pre else 'plain'}: {us:8.1f} us per step",
              flush=True)
    print(f"TOTAL {tot / 1e3:.3f} ms over {len(shapes)} shapes")


if __name__ == "__main__":
    main()
