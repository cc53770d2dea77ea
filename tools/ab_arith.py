import os, sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fp8-quantization_amd"), os.path.dirname(os.path.abspath(__file__))]
import torch, fp8q
from microbench import timeit
ops = fp8q.ops
tag = os.environ.get("TAG", "?")
x = torch.randn(1 << 28, device="cuda"); y = torch.empty_like(x)
def t(name, n, bpe, fn):
    timeit(fn, iters=30, warm=10)
    r = timeit(fn, iters=30, warm=5)
    print(f"[{tag}] {name:44s} {r[0]*1e6:8.1f} us  {n*bpe/r[0]/1e12:.3f} TB/s", flush=True)
mv1 = torch.tensor([3.0], device="cuda")
t("k1 per tensor 1GiB e4m3", x.numel(), 8, lambda: ops.quantize(x, mv1, 3, 8, 1, out=y))
xc = x[: (1 << 20) * 147].view(1 << 20, 147); yc = y[: xc.numel()].view_as(xc)
mvc = ops.minmax(xc, True, want_maxval=True)[2]
t("k1 per channel [2^20,147] e5m2 (headline)", xc.numel(), 8, lambda: ops.quantize(xc, mvc, 2, 8, 1, out=yc))
t("fused [2^20,147] e5m2 (staged)", xc.numel(), 8, lambda: ops.minmax_quantize(xc, 2, 8, 1, out=yc))
x3 = x[: (1 << 18) * 576].view(1 << 18, 576); y3 = y[: x3.numel()].view_as(x3)
t("fused [2^18,576] e5m2 (reg)", x3.numel(), 8, lambda: ops.minmax_quantize(x3, 2, 8, 1, out=y3))
x4 = x[: 58254 * 4608].view(58254, 4608); y4 = y[: x4.numel()].view_as(x4)
t("fused [58254,4608] e5m2 (reg)", x4.numel(), 8, lambda: ops.minmax_quantize(x4, 2, 8, 1, out=y4))
a = x[: 64 * 64 * 112 * 112].view(64, 64, 112, 112); ya = y[: a.numel()].view_as(a)
C = 64
bn = tuple(torch.rand(C, device="cuda") + 0.5 for _ in range(4))
t("bn+relu+quant [64,64,112,112]", a.numel(), 8, lambda: ops.affine_act_quantize(a, mv1, 3, 8, 1, bn=bn, act=1, out=ya))
c = torch.empty(xc.shape, dtype=torch.uint8, device="cuda")
t("encode [2^20,147]", xc.numel(), 5, lambda: ops.encode(xc, mvc, 2, 8, 1, out=c))
