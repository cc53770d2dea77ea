"""N2 (fused epilogue) and N3 (codes) kernels at activation sizes, next to K1 / copy on the same buffers"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fp8-quantization_amd")]
import torch, fp8q
from microbench import timeit, report
ops = fp8q.ops
dev = "cuda"
torch.manual_seed(0)
shapes = ((64, 64, 112, 112), (256, 64, 112, 112), (64, 64, 56, 56), (64, 512, 7, 7), (64, 1280, 7, 7))
for shape in shapes[: int(os.environ.get("NSHAPES", "5"))]:
    x = torch.randn(*shape, device=dev)
    r = torch.randn(*shape, device=dev)
    y = torch.empty_like(x)
    C = shape[1]
    bn = (torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev))
    mv = torch.tensor([4.0], device=dev)
    n = x.numel()
    timeit(lambda: ops.copy(x, out=y), iters=40)
    tag = os.environ.get("TAG", "") + "x".join(map(str, shape))
    report(f"[{tag}] copy", n, 8, timeit(lambda: ops.copy(x, out=y)))
    report(f"[{tag}] K1 per-tensor E4M3", n, 8, timeit(lambda: ops.quantize(x, mv, 3, 8, 1, out=y)))
    report(f"[{tag}] N2 bn+relu+quant", n, 8, timeit(lambda: ops.affine_act_quantize(x, mv, 3, 8, 1, bn=bn, act=1, out=y)))
    report(f"[{tag}] N2 bn+res+relu+quant (12 B/elem)", n, 12, timeit(lambda: ops.affine_act_quantize(x, mv, 3, 8, 1, bn=bn, residual=r, act=1, out=y)))
    report(f"[{tag}] N2 relu+quant (no bn)", n, 8, timeit(lambda: ops.affine_act_quantize(x, mv, 3, 8, 1, act=1, out=y)))
    report(f"[{tag}] N2 bn+relu minmax (4 B/elem)", n, 4, timeit(lambda: ops.affine_act_minmax(x, bn=bn, act=1)))
    report(f"[{tag}] K3 minmax", n, 4, timeit(lambda: ops.minmax(x, False)))
    codes = torch.empty(shape, dtype=torch.uint8, device=dev)
    report(f"[{tag}] N3 encode (5 B/elem)", n, 5, timeit(lambda: ops.encode(x, mv, 3, 8, 1, out=codes)))
    report(f"[{tag}] N3 decode (5 B/elem)", n, 5, timeit(lambda: ops.decode(codes, mv, 3, 8, 1, out=y)))
