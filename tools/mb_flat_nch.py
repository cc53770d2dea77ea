"""K1 per channel through k_rows_flat on a few row lengths; env FP8Q_FLAT_NCH = chunks per tile (A/B)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fp8-quantization_amd"), os.path.dirname(os.path.abspath(__file__))]
import torch, fp8q
from microbench import timeit
ops = fp8q.ops
x = torch.randn(1 << 28, device="cuda"); y = torch.empty_like(x)
for rows, inner in ((1 << 20, 147), (1 << 18, 576), (1 << 17, 1152), (1 << 19, 288), (1 << 21, 64), (1 << 16, 2304 - 257)):
    xv = x[: rows * inner].view(rows, inner); yv = y[: rows * inner].view(rows, inner)
    mv = ops.minmax(xv, True, want_maxval=True)[2]
    timeit(lambda: ops.quantize(xv, mv, 2, 8, 1, out=yv), iters=30, warm=10)
    r = timeit(lambda: ops.quantize(xv, mv, 2, 8, 1, out=yv), iters=30, warm=5)
    print(f"nch={os.environ.get('FP8Q_FLAT_NCH','dflt'):5s} K1 [{rows},{inner}]: {r[0]*1e6:7.1f} us  {xv.numel()*8/r[0]/1e12:.3f} TB/s", flush=True)
