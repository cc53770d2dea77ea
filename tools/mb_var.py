import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fp8-quantization_amd"), os.path.join(ROOT, "tools")]
import torch, fp8q
from microbench import timeit, report
ops = fp8q.ops
dev = "cuda"
tag = os.path.basename(os.environ.get("FP8Q_SO", "default"))
N = 1 << 21
xw = (torch.randn(N * 147, device=dev) * 0.1).view(N, 3, 7, 7)
yw = torch.empty_like(xw)
mn, mx, mvw = ops.minmax(xw, True, want_maxval=True)
report(f"[{tag}] K1 [2^21,3,7,7] E5M2", N * 147, 8, timeit(lambda: ops.quantize(xw, mvw, 2, 8, 1, out=yw)))
report(f"[{tag}] K1 [2^21,3,7,7] E4M3", N * 147, 8, timeit(lambda: ops.quantize(xw, mvw, 3, 8, 1, out=yw)))
report(f"[{tag}] fused [2^21,3,7,7] E5M2", N * 147, 8, timeit(lambda: ops.minmax_quantize(xw, 2, 8, 1, out=yw)))
x = torch.randn(1 << 28, device=dev); y = torch.empty_like(x); mv1 = torch.tensor([3.0], device=dev)
report(f"[{tag}] K1 per-tensor E4M3 1GiB", 1 << 28, 8, timeit(lambda: ops.quantize(x, mv1, 3, 8, 1, out=y)))
report(f"[{tag}] copy 1GiB", 1 << 28, 8, timeit(lambda: ops.copy(x, out=y)))
