"""Quick per-kernel timing on one GPU (development aid; bench.py is the contract)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fp8-quantization_amd")]
import torch  # noqa: E402
import fp8q  # noqa: E402

ops = fp8q.ops


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e-3, ts[0] * 1e-3


def report(name, n_elem, bytes_per_elem, t):
    med, mn = t
    print(f"{name:58s} {med*1e6:10.1f} us  {n_elem/med/1e9:8.1f} Gelem/s  "
          f"{n_elem*bytes_per_elem/med/1e12:6.3f} TB/s (min {n_elem*bytes_per_elem/mn/1e12:6.3f})", flush=True)


def main():
    torch.manual_seed(0)
    dev = "cuda"
    print(torch.cuda.get_device_name(0))
    n = 1 << 28  # 1 GiB fp32
    x = torch.randn(n, device=dev)
    y = torch.empty_like(x)
    report("copy k_copy 1GiB", n, 8, timeit(lambda: ops.copy(x, out=y)))
    report("copy torch y.copy_(x) 1GiB", n, 8, timeit(lambda: y.copy_(x)))
    mv1 = torch.tensor([3.0], device=dev)
    report("K1 per-tensor E4M3 1GiB", n, 8, timeit(lambda: ops.quantize(x, mv1, 3, 8, 1, out=y)))
    report("K1 per-tensor E5M2 1GiB", n, 8, timeit(lambda: ops.quantize(x, mv1, 2, 8, 1, out=y)))
    report("K3 minmax per-tensor 1GiB", n, 4, timeit(lambda: ops.minmax(x, False)))
    # conv1-shaped synthetic [N,3,7,7]
    N = 1 << 21
    xw = (torch.randn(N * 147, device=dev) * 0.1).view(N, 3, 7, 7)
    yw = torch.empty_like(xw)
    mn, mx, mvw = ops.minmax(xw, True, want_maxval=True)
    report("K1 per-channel [2^21,3,7,7] E5M2 (multi, LUT)", N * 147, 8,
           timeit(lambda: ops.quantize(xw, mvw, 2, 8, 1, out=yw)))
    report("K2 minmax per-channel [2^21,3,7,7]", N * 147, 4, timeit(lambda: ops.minmax(xw, True)))
    report("fused minmax+quant [2^21,3,7,7] E5M2", N * 147, 8,
           timeit(lambda: ops.minmax_quantize(xw, 2, 8, 1, out=yw)))
    # long rows
    xr = x.view(4096, -1)
    yr = y.view(4096, -1)
    mvr = torch.rand(4096, device=dev) + 0.5
    report("K1 per-channel [4096,65536] E4M3 (rows)", n, 8, timeit(lambda: ops.quantize(xr, mvr, 3, 8, 1, out=yr)))
    nk = (1 << 27) // 4608
    xk = x[: nk * 4608].view(nk, 4608)
    yk = y[: nk * 4608].view(nk, 4608)
    report("fused minmax+quant [29127,4608] E5M2", xk.numel(), 8, timeit(lambda: ops.minmax_quantize(xk, 2, 8, 1, out=yk)))
    xd = x[: 9 << 22].view(-1, 1, 3, 3)
    yd = y[: 9 << 22].view(-1, 1, 3, 3)
    mvd = torch.rand(xd.shape[0], device=dev) + 0.5
    report("K1 per-channel [4M,1,3,3] E4M3 (multi, direct)", xd.numel(), 8,
           timeit(lambda: ops.quantize(xd, mvd, 3, 8, 1, out=yd)))
    # real layer shapes: launch-bound
    w = torch.randn(64, 3, 7, 7, device=dev) * 0.1
    yw1 = torch.empty_like(w)
    report("conv1 [64,3,7,7] fused minmax+quant E5M2", w.numel(), 8, timeit(lambda: ops.minmax_quantize(w, 2, 8, 1, out=yw1), iters=200))
    a = torch.randn(64, 64, 112, 112, device=dev)
    ya = torch.empty_like(a)
    report("act [64,64,112,112] K1 per-tensor E5M2", a.numel(), 8, timeit(lambda: ops.quantize(a, mv1, 2, 8, 1, out=ya)))
    report("act [64,64,112,112] K3 allminmax", a.numel(), 4, timeit(lambda: ops.minmax(a, False)))
    # MSE grid
    ws = torch.randn(32, 3, 3, 3, device=dev)
    grid = (torch.linspace(0.1, 1.2, 111, device=dev)[:, None] * ws.view(32, -1).abs().amax(1)[None, :]).contiguous()
    mses = torch.zeros(6, 111, 32, device=dev)
    t = timeit(lambda: ops.mse_grid(ws, True, grid, [1, 2, 3, 4, 5, 6], 8, 1, mses))
    print(f"MSE grid weights [32,3,3,3] x111x6: {t[0]*1e6:.1f} us")
    am = torch.randn(64, 32, 112, 112, device=dev)
    grid1 = torch.linspace(0.5, 6.0, 111, device=dev)[:, None].contiguous()
    mses1 = torch.zeros(1, 111, 1, device=dev)
    t = timeit(lambda: ops.mse_grid(am, False, grid1, [3], 8, 1, mses1), iters=5, warm=1)
    print(f"MSE grid act [64,32,112,112] x111x1: {t[0]*1e3:.2f} ms  = {am.numel()*111/t[0]/1e12:.2f} T cand-elem/s")
    mses6 = torch.zeros(6, 111, 1, device=dev)
    t = timeit(lambda: ops.mse_grid(am, False, grid1, [1, 2, 3, 4, 5, 6], 8, 1, mses6), iters=3, warm=1)
    print(f"MSE grid act [64,32,112,112] x111x6: {t[0]*1e3:.2f} ms  = {am.numel()*666/t[0]/1e12:.2f} T cand-elem/s")


if __name__ == "__main__":
    main()
