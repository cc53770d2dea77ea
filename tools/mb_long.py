import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fp8-quantization_amd"), os.path.join(ROOT, "tools")]
import torch, fp8q
from microbench import timeit, report
ops = fp8q.ops
dev = "cuda"
tag = os.environ.get("FP8Q_DIRECT_MAX_INNER", "default")
x = torch.randn(1 << 28, device=dev) * 0.1
y = torch.empty_like(x)
for rows, inner in ((1 << 16, 4096), (58254, 4608), (1 << 17, 2304), (1 << 14, 16384), (4096, 65536)):
    xv = x[: rows * inner].view(rows, inner); yv = y[: rows * inner].view(rows, inner)
    mv = torch.rand(rows, device=dev) * 0.3 + 0.1
    report(f"[{tag}] K1 [{rows},{inner}] E5M2", rows * inner, 8, timeit(lambda: ops.quantize(xv, mv, 2, 8, 1, out=yv)))
    if inner <= 16384:
        report(f"[{tag}] fused [{rows},{inner}] E5M2", rows * inner, 8, timeit(lambda: ops.minmax_quantize(xv, 2, 8, 1, out=yv)))
    report(f"[{tag}] K2 [{rows},{inner}]", rows * inner, 4, timeit(lambda: ops.minmax(xv, True)))
