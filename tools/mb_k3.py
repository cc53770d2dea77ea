"""K3 (per-tensor min/max, single launch with reducer block) timings; env FP8Q_K3_BLOCKS = streaming-block cap."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fp8-quantization_amd")]
import torch, fp8q
from microbench import timeit
ops = fp8q.ops
x = torch.randn(1 << 28, device="cuda")
for name, t in (("act[64,64,112,112]", x[: 64 * 64 * 112 * 112]), ("1GiB", x), ("act[64,256,56,56]", x[: 64 * 256 * 56 * 56]),
                ("act[64,512,7,7]", x[: 64 * 512 * 49])):
    cur = ops.minmax(t, False)
    r = timeit(lambda: ops.minmax(t, False, cur[0], cur[1], mode=1), iters=30, warm=5)
    print(f"K3 {name:22s} cap={os.environ.get('FP8Q_K3_BLOCKS', 'default')}: median {r[0]*1e6:7.1f} us  {t.numel()*4/r[0]/1e12:.2f} TB/s", flush=True)
