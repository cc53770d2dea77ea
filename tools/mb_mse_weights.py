"""K4 on MobileNetV2's per-channel weight shapes (k_mse_grid / k_mse_row), 111 candidates, 1 and 6 mantissa widths.
   python tools/mb_mse_weights.py            (A/B: FP8Q_MSE_GRID_TILE=2048 = whole rows per workgroup)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fp8-quantization_amd"))
import fp8q  # noqa: E402
from models.mobilenet_v2 import MobileNetV2  # noqa: E402

ops = fp8q.ops
dev = torch.device("cuda")
shapes = [tuple(m.weight.shape) for m in MobileNetV2().modules() if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear))]


def ev(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for mb in ([3.0], [1.0, 2.0, 3.0, 4.0, 5.0, 6.0]):
    tot = 0.0
    rows = {}
    for shp in shapes:
        w = torch.randn(*shp, device=dev) * 0.05
        mx = w.reshape(shp[0], -1).abs().amax(1)
        grid = ops.mse_linspace(mx, 111)
        mses = torch.zeros(len(mb), 111, shp[0], device=dev)
        t = ev(lambda: ops.mse_grid(w, True, grid, mb, 8, 1, mses))
        tot += t
        key = (shp[0], int(w[0].numel()))
        rows.setdefault(key, []).append(t)
    print(f"m={mb}: {len(shapes)} weight tensors, total {tot:.1f} us (tile env {os.environ.get('FP8Q_MSE_GRID_TILE', 'auto')})")
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[:14]:
        print(f"   [C={k[0]:5d}, inner={k[1]:5d}] x{len(v)}: {sum(v) / len(v):7.1f} us each")
