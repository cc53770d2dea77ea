"""k_rows_staged alone on [2^21,147] E5M2 (for rocprofv3 --pmc passes): 3 warm-up + 10 launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fp8-quantization_amd")]
import torch, fp8q
ops = fp8q.ops
x = (torch.randn((1 << 21) * 147, device="cuda") * 0.1).view(1 << 21, 147)
y = torch.empty_like(x)
for _ in range(13):
    ops.minmax_quantize(x, 2, 8, 1, out=y)
torch.cuda.synchronize()
