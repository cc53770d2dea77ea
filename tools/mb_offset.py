"""Does the distance between the input and the output buffer matter?  K1 (k_rows_flat) on [2^21,3,7,7] with y placed at
different offsets behind x inside one allocation (HBM channel / bank interleaving of the read and the write stream)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fp8-quantization_amd")]
import torch, fp8q
from microbench import timeit, report
ops = fp8q.ops
N = 1 << 21
n = N * 147
slack = 64 << 20
big = torch.empty(2 * n + slack // 4 + 4096, device="cuda")
x = big[:n].view(N, 147)
x.normal_()
x.mul_(0.1)
_, _, mv = ops.minmax(x, True, want_maxval=True)
y0 = torch.empty_like(x)
timeit(lambda: ops.quantize(x, mv, 2, 8, 1, out=y0), iters=40)
report("separate allocations", n, 8, timeit(lambda: ops.quantize(x, mv, 2, 8, 1, out=y0)))
print("delta(y0 - x) mod 2^k:", [(k, (y0.data_ptr() - x.data_ptr()) % (1 << k)) for k in (12, 16, 20, 24, 30)])
for off_bytes in (0, 256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20, (1 << 20) + 4096, (1 << 22) + 16384, 12345 * 16, (32 << 20) + 8192):
    o = off_bytes // 4
    y = big[n + o: n + o + n].view(N, 147)
    report(f"y = x + tensor + {off_bytes} B", n, 8, timeit(lambda: ops.quantize(x, mv, 2, 8, 1, out=y)))
