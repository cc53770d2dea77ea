"""k_rows_staged (fused min/max + quantize of short rows, one fetch per element): check against the two-step
path (fp8q_minmax_f32 + fp8q_quantize_f32, different kernels) and time it.  Env: FP8Q_STAGED=0|1,
FP8Q_STAGED_GRID=<cap, 0 = one chunk per block>; TAG labels the output lines."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fp8-quantization_amd")]
import torch, fp8q
from microbench import timeit, report
ops = fp8q.ops
dev = "cuda"
tag = os.environ.get("TAG", "default")
CHECK = os.environ.get("CHECK", "1") == "1"


def check(C, inner, M, seed, nan_row=None):
    g = torch.Generator(device=dev).manual_seed(seed)
    scale = torch.rand(C, 1, device=dev, generator=g) * 3 + 0.01
    x = torch.randn(C, inner, device=dev, generator=g) * scale
    if nan_row is not None:
        x[nan_row % C, inner // 2] = float("nan")
    mn, mx, mv = ops.minmax(x, True, want_maxval=True)
    ref = ops.quantize(x, mv, M, 8, 1)
    y, fmn, fmx, fmv = ops.minmax_quantize(x, M, 8, 1)
    ok = (torch.equal(y.view(torch.int32), ref.view(torch.int32)) and torch.equal(fmn.view(torch.int32), mn.view(torch.int32))
          and torch.equal(fmx.view(torch.int32), mx.view(torch.int32)) and torch.equal(fmv.view(torch.int32), mv.view(torch.int32)))
    if not ok:
        bad = (y.view(torch.int32) != ref.view(torch.int32)).nonzero()
        print(f"[{tag}] MISMATCH C={C} inner={inner} M={M}: {bad.shape[0]} elements, first {bad[:4].tolist()}; "
              f"ranges equal: {torch.equal(fmv.view(torch.int32), mv.view(torch.int32))}", flush=True)
    return ok


if CHECK:
    n_ok = n = 0
    cases = [(1, 147, 2), (2, 147, 2), (27, 147, 2), (28, 147, 3), (29, 147, 2), (64, 147, 2), (1000, 147, 2), (4099, 147, 3),
             (65537, 147, 2), (333, 255, 2), (333, 256, 3), (5000, 253, 2), (7001, 99, 3), (7001, 101, 2), (9000, 68, 2),
             (9000, 67, 2), (9001, 70, 2), (12345, 131, 2), (40000, 201, 3), (30011, 75, 3), (50021, 41, 3), (100003, 39, 3),
             (17, 241, 4), (5, 250, 1), (2049, 130, 2), (8192, 134, 3), (3, 67, 2), (1 << 16, 98, 2)]
    for i, (C, inner, M) in enumerate(cases):
        n += 1
        n_ok += check(C, inner, M, 100 + i, nan_row=(C // 2 if i % 5 == 0 else None))
    print(f"[{tag}] check: {n_ok}/{n} geometries bit-equal to the two-step path", flush=True)

N = 1 << 21
xw = (torch.randn(N * 147, device=dev) * 0.1).view(N, 147)
yw = torch.empty_like(xw)
timeit(lambda: ops.minmax_quantize(xw, 2, 8, 1, out=yw), iters=40)   # clock warm-up: discard
report(f"[{tag}] fused [2^21,147] E5M2", N * 147, 8, timeit(lambda: ops.minmax_quantize(xw, 2, 8, 1, out=yw)))
report(f"[{tag}] fused [2^21,147] E4M3", N * 147, 8, timeit(lambda: ops.minmax_quantize(xw, 3, 8, 1, out=yw)))
for rows, inner, M in (((1 << 21) + 77, 99, 3), (1 << 20, 201, 2), (1 << 20, 255, 2), (1 << 22, 70, 2), (1 << 17, 147, 2), (4096, 147, 2)):
    xk = xw.view(-1)[: rows * inner].view(rows, inner)
    yk = yw.view(-1)[: rows * inner].view(rows, inner)
    report(f"[{tag}] fused [{rows},{inner}] M={M}", xk.numel(), 8, timeit(lambda: ops.minmax_quantize(xk, M, 8, 1, out=yk)))

# K2 twin (k_rows_staged_mm): row ranges against torch, then timings
if CHECK:
    n_ok = n = 0
    for i, (C, inner) in enumerate([(2, 147), (29, 147), (1000, 147), (65537, 147), (333, 255), (7001, 99), (9000, 67), (50021, 41),
                                    (100003, 9), (300000, 27), (40000, 5), (77777, 4), (12345, 131), (5, 250), (1 << 16, 98)]):
        g = torch.Generator(device=dev).manual_seed(500 + i)
        x = torch.randn(C, inner, device=dev, generator=g)
        if i % 3 == 0:
            x[C // 2, inner // 2] = float("nan")
        mn, mx, mv = ops.minmax(x, True, want_maxval=True)
        rmn, rmx = x.amin(1), x.amax(1)
        ok = (torch.equal(mn.view(torch.int32), rmn.view(torch.int32)) and torch.equal(mx.view(torch.int32), rmx.view(torch.int32))
              and torch.equal(mv.view(torch.int32), torch.maximum(rmn.abs(), rmx).abs().view(torch.int32)))
        n += 1
        n_ok += ok
        if not ok:
            print(f"[{tag}] K2 MISMATCH C={C} inner={inner}: {(mn != rmn).sum().item()} mins, {(mx != rmx).sum().item()} maxs", flush=True)
    print(f"[{tag}] K2 check: {n_ok}/{n} geometries equal to torch amin/amax", flush=True)
for rows, inner in ((1 << 21, 147), ((1 << 21) + 77, 99), (1 << 20, 201), (1 << 22, 70), (1 << 24, 9), (1 << 23, 27), (1 << 17, 147)):
    xk = xw.view(-1)[: rows * inner].view(rows, inner)
    report(f"[{tag}] K2 [{rows},{inner}]", xk.numel(), 4, timeit(lambda: ops.minmax(xk, True)))
