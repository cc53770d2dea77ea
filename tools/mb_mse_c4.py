"""K4 (111 candidates, E4M3) on MobileNetV2's activation shapes at batch 64, ReLU6-like and signed data, one route per process:
FP8Q_MSE_HIST=0 (lane-per-element kernel) / =3 (interval histogram for everything) / unset (the routing model)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fp8-quantization_amd"), os.path.join(ROOT, "tools")]
import torch, fp8q
from microbench import timeit
ops = fp8q.ops
torch.manual_seed(0)
shapes = [(32, 112), (16, 112), (96, 112), (96, 56), (24, 56), (144, 56), (144, 28), (32, 28), (192, 28), (192, 14), (64, 14),
          (384, 14), (96, 14), (576, 14), (576, 7), (160, 7), (960, 7), (320, 7), (1280, 7)]
tot = {"relu6": 0.0, "signed": 0.0}
for C, hw in shapes:
    for kind in ("relu6", "signed"):
        x = torch.randn(64, C, hw, hw, device="cuda") * 2.5
        if kind == "relu6":
            x = torch.clamp(x, 0, 6)
        grid = ops.mse_linspace(x.abs().max().reshape(1), 111)
        mses = torch.zeros(1, 111, 1, device="cuda")
        t = timeit(lambda: ops.mse_grid(x, False, grid, [3.0], 8, 1, mses), iters=10, warm=3)
        tot[kind] += t[0]
        print(f"HIST={os.environ.get('FP8Q_MSE_HIST', 'model')} [64,{C},{hw},{hw}] n={x.numel():9d} {kind:6s}: {t[0]*1e6:8.1f} us", flush=True)
print("TOTAL", {k: round(v * 1e3, 3) for k, v in tot.items()}, "ms")
