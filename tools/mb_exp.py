"""experiment: K1 / fused / K2 on [2^21,3,7,7] (library named by FP8Q_SO, knobs from the environment),
optionally next to per-tensor K1 and copy on the same buffer (REF=1)"""
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
tag = os.environ.get("TAG", os.path.basename(os.environ.get("FP8Q_SO", "default")))
report(f"[{tag}] K1 [2^21,3,7,7] E5M2", N * 147, 8, timeit(lambda: ops.quantize(xw, mvw, 2, 8, 1, out=yw)))
report(f"[{tag}] K1 [2^21,3,7,7] E4M3", N * 147, 8, timeit(lambda: ops.quantize(xw, mvw, 3, 8, 1, out=yw)))
report(f"[{tag}] K1 [2^21,3,7,7] E5M2 (again)", N * 147, 8, timeit(lambda: ops.quantize(xw, mvw, 2, 8, 1, out=yw)))
report(f"[{tag}] K1 [2^21,3,7,7] E3M4", N * 147, 8, timeit(lambda: ops.quantize(xw, mvw, 4, 8, 1, out=yw)))
if not os.environ.get("K1ONLY"):
    report(f"[{tag}] fused [2^21,3,7,7] E5M2", N * 147, 8, timeit(lambda: ops.minmax_quantize(xw, 2, 8, 1, out=yw)))
if os.environ.get("REF"):
    one = mvw.max().reshape(1)
    report(f"[{tag}] K1 per-tensor same buffer", N * 147, 8, timeit(lambda: ops.quantize(xw, one, 2, 8, 1, out=yw)))
    report(f"[{tag}] copy same buffer", N * 147, 8, timeit(lambda: ops.copy(xw, yw)))
    report(f"[{tag}] K2 per-channel minmax", N * 147, 4, timeit(lambda: ops.minmax(xw, True)))
