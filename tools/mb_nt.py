"""Kernels on a 205 MB activation [64,64,112,112] (fits the 256 MB Infinity Cache) under FP8Q_NT_MB (nontemporal threshold)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fp8-quantization_amd"), os.path.dirname(os.path.abspath(__file__))]
import torch, fp8q
from microbench import timeit
ops = fp8q.ops
tag = os.environ.get("FP8Q_NT_MB", "64")
a = torch.randn(64, 64, 112, 112, device="cuda"); y = torch.empty_like(a); res = torch.randn_like(a)
mv = torch.tensor([3.0], device="cuda")
bn = tuple(torch.rand(64, device="cuda") + 0.5 for _ in range(4))
def t(name, bpe, fn):
    timeit(fn, iters=40, warm=10)
    r = timeit(fn, iters=40, warm=5)
    print(f"[nt>={tag}MiB] {name:34s} {r[0]*1e6:7.1f} us  {a.numel()*bpe/r[0]/1e12:.3f} TB/s", flush=True)
t("K1 e5m2", 8, lambda: ops.quantize(a, mv, 2, 8, 1, out=y))
t("K3 allminmax", 4, lambda: ops.minmax(a, False))
t("bn+relu+quant", 8, lambda: ops.affine_act_quantize(a, mv, 3, 8, 1, bn=bn, act=1, out=y))
t("bn+res+relu+quant", 12, lambda: ops.affine_act_quantize(a, mv, 3, 8, 1, bn=bn, residual=res, act=1, out=y))
t("K1 then K1 of the result (chain)", 16, lambda: ops.quantize(ops.quantize(a, mv, 2, 8, 1, out=y), mv, 2, 8, 1, out=a))
