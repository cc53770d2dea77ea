"""short-row kernels only (quick A/B)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fp8-quantization_amd")]
import torch, fp8q
from microbench import timeit, report
ops = fp8q.ops
dev = "cuda"
N = 1 << 21
xw = (torch.randn(N * 147, device=dev) * 0.1).view(N, 3, 7, 7)
yw = torch.empty_like(xw)
mn, mx, mvw = ops.minmax(xw, True, want_maxval=True)
tag = os.environ.get("TAG", "default")
timeit(lambda: ops.quantize(xw, mvw, 2, 8, 1, out=yw), iters=40)   # clocks / first-touch warm-up: discard
report(f"[{tag}] K1 [2^21,3,7,7] E5M2", N * 147, 8, timeit(lambda: ops.quantize(xw, mvw, 2, 8, 1, out=yw)))
report(f"[{tag}] K1 [2^21,3,7,7] E4M3", N * 147, 8, timeit(lambda: ops.quantize(xw, mvw, 3, 8, 1, out=yw)))
report(f"[{tag}] K2 [2^21,3,7,7]", N * 147, 4, timeit(lambda: ops.minmax(xw, True)))
report(f"[{tag}] fused [2^21,3,7,7] E5M2", N * 147, 8, timeit(lambda: ops.minmax_quantize(xw, 2, 8, 1, out=yw)))
x6 = xw.view(-1)[: (1 << 19) * 576].view(1 << 19, 64, 3, 3)
y6 = yw.view(-1)[: x6.numel()].view_as(x6)
mv6 = torch.rand(x6.shape[0], device=dev) + 0.5
report(f"[{tag}] K1 [2^19,64,3,3] E5M2", x6.numel(), 8, timeit(lambda: ops.quantize(x6, mv6, 2, 8, 1, out=y6)))
report(f"[{tag}] fused [2^19,64,3,3] E5M2", x6.numel(), 8, timeit(lambda: ops.minmax_quantize(x6, 2, 8, 1, out=y6)))

for rows, inner in ((1 << 21, 128), (1 << 20, 192), (1 << 20, 256), (1 << 19, 384), (1 << 19, 288), (1 << 19, 512), (1 << 18, 1024), (1 << 18, 1152), (58254, 4608), (1 << 17, 2048), (32768, 8192)):
    xk = xw.view(-1)[: rows * inner].view(rows, inner)
    yk = yw.view(-1)[: rows * inner].view(rows, inner)
    mvk = torch.rand(rows, device=dev) + 0.5
    report(f"[{tag}] K1 [{rows},{inner}] E5M2", xk.numel(), 8, timeit(lambda: ops.quantize(xk, mvk, 2, 8, 1, out=yk)))
    report(f"[{tag}] fused [{rows},{inner}] E5M2", xk.numel(), 8, timeit(lambda: ops.minmax_quantize(xk, 2, 8, 1, out=yk)))
    report(f"[{tag}] K2min/max [{rows},{inner}]", xk.numel(), 4, timeit(lambda: ops.minmax(xk, True)))
