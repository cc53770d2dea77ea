import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fp8-quantization_amd")]
import torch, fp8q
ops = fp8q.ops
dev = "cuda"
def ev(fn, iters=200, warm=50):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    es = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in es:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in es)
    return ts[len(ts) // 2], ts[0]
shapes = [(64, 3, 7, 7)] + [(64, 64, 3, 3)] * 4 + [(128, 64, 3, 3), (128, 128, 3, 3), (128, 64, 1, 1)] + \
    [(128, 128, 3, 3)] * 2 + [(256, 128, 3, 3), (256, 256, 3, 3), (256, 128, 1, 1)] + [(256, 256, 3, 3)] * 2 + \
    [(512, 256, 3, 3), (512, 512, 3, 3), (512, 256, 1, 1)] + [(512, 512, 3, 3)] * 2 + [(1000, 512)]
ws = [torch.randn(*sh, device=dev) * 0.05 for sh in shapes]
ys = [torch.empty_like(t) for t in ws]
mvs = [ops.minmax(t, True, want_maxval=True)[2] for t in ws]
def plan_of(idx):
    return ops.MultiPlan([(ws[i], mvs[i], 2, 8, 1, ys[i]) for i in idx])
allp = plan_of(range(21))
print("all 21:", ev(allp.launch), sum(w.numel() for w in ws))
big = [i for i in range(21) if ws[i].numel() >= 1 << 20]
p = plan_of(big); print("big only", len(big), sum(ws[i].numel() for i in big), ev(p.launch))
small = [i for i in range(21) if ws[i].numel() < 1 << 20]
p = plan_of(small); print("small only", len(small), sum(ws[i].numel() for i in small), ev(p.launch))
n = sum(w.numel() for w in ws)
for inner in (576, 4608):
    x = torch.randn(n // inner, inner, device=dev) * 0.05; y = torch.empty_like(x)
    mv = ops.minmax(x, True, want_maxval=True)[2]
    print(f"single K1 [{n // inner},{inner}]", ev(lambda: ops.quantize(x, mv, 2, 8, 1, out=y)))
    pl = ops.MultiPlan([(x, mv, 2, 8, 1, y)]); print("  as a plan", ev(pl.launch))
x = torch.randn(n, device=dev); y = torch.empty_like(x)
print("copy 46.7 MB", ev(lambda: ops.copy(x[: n // 4 * 4], out=y[: n // 4 * 4])))
mv1 = torch.tensor([3.0], device=dev)
print("K1 per tensor 46.7 MB", ev(lambda: ops.quantize(x, mv1, 2, 8, 1, out=y)))
e = torch.empty(16, device=dev)
print("tiny kernel (event floor)", ev(lambda: ops.quantize(e, mv1, 2, 8, 1, out=e)))
