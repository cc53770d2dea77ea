"""K4 (MSE grid) timings"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fp8-quantization_amd")]
import torch, fp8q
from microbench import timeit
ops = fp8q.ops
dev = "cuda"
torch.manual_seed(0)
am = torch.randn(64, 32, 112, 112, device=dev)
grid1 = torch.linspace(0.5, 6.0, 111, device=dev)[:, None].contiguous()
ar = torch.relu(am)
for name, t_ in (("randn", am), ("relu ", ar)):
    for ms in ([3], [1, 2, 3, 4, 5, 6]):
        mses = torch.zeros(len(ms), 111, 1, device=dev)
        t = timeit(lambda: ops.mse_grid(t_, False, grid1, ms, 8, 1, mses), iters=5, warm=2)
        print(f"MSE grid act {name} [64,32,112,112] x111 m={ms}: {t[0]*1e3:.3f} ms  = {am.numel()*111*len(ms)/t[0]/1e12:.2f} T cand-elem/s", flush=True)
w = torch.randn(512, 512, 3, 3, device=dev) * 0.05
gw = (torch.linspace(0.1, 1.2, 111, device=dev)[:, None] * w.view(512, -1).abs().amax(1)[None, :]).contiguous()
mw = torch.zeros(1, 111, 512, device=dev)
t = timeit(lambda: ops.mse_grid(w, True, gw, [3], 8, 1, mw), iters=10, warm=2)
print(f"MSE grid weights [512,512,3,3] per-channel x111: {t[0]*1e6:.1f} us = {w.numel()*111/t[0]/1e12:.2f} T cand-elem/s")
