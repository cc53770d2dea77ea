"""Run one kernel configuration a few times (for rocprofv3 --pmc / --kernel-trace)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fp8-quantization_amd")]
import torch  # noqa: E402
import fp8q  # noqa: E402

ops = fp8q.ops
which = sys.argv[1] if len(sys.argv) > 1 else "multi"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = "cuda"
torch.manual_seed(0)
if which in ("multi", "fused", "waverow"):
    N = 1 << 21
    x = (torch.randn(N * 147, device=dev) * 0.1).view(N, 3, 7, 7)
    y = torch.empty_like(x)
    _, _, mv = ops.minmax(x, True, want_maxval=True)
    for _ in range(reps):
        if which == "multi":
            ops.quantize(x, mv, 2, 8, 1, out=y)
        elif which == "fused":
            ops.minmax_quantize(x, 2, 8, 1, out=y)
        else:
            ops.minmax(x, True)
elif which == "fusedlong":
    x = (torch.randn(58254, 4608, device=dev) * 0.1)
    y = torch.empty_like(x)
    for _ in range(reps):
        ops.minmax_quantize(x, 2, 8, 1, out=y)
elif which == "tensor":
    x = torch.randn(1 << 28, device=dev)
    y = torch.empty_like(x)
    mv = torch.tensor([3.0], device=dev)
    for _ in range(reps):
        ops.quantize(x, mv, 3, 8, 1, out=y)
elif which == "k3":
    x = torch.randn(64, 64, 112, 112, device=dev)
    cur = ops.minmax(x, False)
    for _ in range(reps):
        cur = ops.minmax(x, False, cur[0], cur[1], mode=1)
torch.cuda.synchronize()
