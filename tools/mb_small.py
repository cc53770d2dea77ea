import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fp8-quantization_amd")]
import torch, fp8q
from microbench import timeit, report
ops = fp8q.ops
tag = os.environ.get("TAG", "x")
big = torch.randn(1 << 26, device="cuda")
timeit(lambda: ops.quantize(big, torch.tensor([3.0], device="cuda"), 3, 8, 1), iters=30)
for C in (64, 256, 1024, 4096, 16384):
    w = torch.randn(C, 3, 7, 7, device="cuda") * 0.1
    yw = torch.empty_like(w)
    for rep in range(2):
        report(f"[{tag}] fused [{C},147]", w.numel(), 8, timeit(lambda: ops.minmax_quantize(w, 2, 8, 1, out=yw), iters=300))
    report(f"[{tag}] K2 [{C},147]", w.numel(), 4, timeit(lambda: ops.minmax(w, True), iters=300))
    # device time only: a HIP graph of 20 calls
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ops.minmax_quantize(w, 2, 8, 1, out=yw)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20):
                ops.minmax_quantize(w, 2, 8, 1, out=yw)
    t = timeit(lambda: g.replay(), iters=50)
    print(f"[{tag}] fused [{C},147] in a graph: {t[0] * 1e6 / 20:.2f} us per kernel", flush=True)
