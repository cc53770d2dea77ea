"""K4 on per-tensor rows of growing size, both routes (run with FP8Q_MSE_HIST=0 and =1): calibrates mse_use_hist_shape()"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fp8-quantization_amd"), os.path.join(ROOT, "tools")]
import torch, fp8q
from microbench import timeit
ops = fp8q.ops
torch.manual_seed(0)
for lg in (17, 18, 19, 20, 21, 22, 23, 24):
    for relu in (0, 1):
        x = torch.randn(1 << lg, device="cuda")
        if relu:
            x = torch.relu(x)
        grid1 = ops.mse_linspace(x.abs().max().reshape(1), 111)
        for ms in ([3], [1, 2, 3, 4, 5, 6]):
            mses = torch.zeros(len(ms), 111, 1, device="cuda")
            t = timeit(lambda: ops.mse_grid(x, False, grid1, ms, 8, 1, mses), iters=10, warm=3)
            print(f"HIST={os.environ.get('FP8Q_MSE_HIST', '1')} n=2^{lg} relu={relu} n_m={len(ms)}: {t[0]*1e6:8.1f} us", flush=True)
