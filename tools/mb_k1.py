"""per-tensor K1 / long-row K1 at several sizes (grid-size knob FP8Q_K1_BLOCKS)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fp8-quantization_amd")]
import torch, fp8q
from microbench import timeit, report
ops = fp8q.ops
dev = "cuda"
tag = os.environ.get("TAG", "")
x = torch.randn(1 << 28, device=dev)
y = torch.empty_like(x)
mv = torch.tensor([3.0], device=dev)
timeit(lambda: ops.copy(x, out=y), iters=40)
for n in (1 << 28, 64 * 64 * 112 * 112, 64 * 64 * 56 * 56, 64 * 512 * 7 * 7):
    xs, ys = x[:n], y[:n]
    report(f"[{tag}] copy n={n}", n, 8, timeit(lambda: ops.copy(xs, out=ys)))
    report(f"[{tag}] K1 per-tensor E4M3 n={n}", n, 8, timeit(lambda: ops.quantize(xs, mv, 3, 8, 1, out=ys)))
nk = (1 << 28) // 4608
xk, yk = x[: nk * 4608].view(nk, 4608), y[: nk * 4608].view(nk, 4608)
mvk = torch.rand(nk, device=dev) + 0.5
report(f"[{tag}] K1 per-channel [{nk},4608] E5M2", xk.numel(), 8, timeit(lambda: ops.quantize(xk, mvk, 2, 8, 1, out=yk)))
xr, yr = x.view(4096, -1), y.view(4096, -1)
mvr = torch.rand(4096, device=dev) + 0.5
report(f"[{tag}] K1 per-channel [4096,65536] E4M3", x.numel(), 8, timeit(lambda: ops.quantize(xr, mvr, 3, 8, 1, out=yr)))
