"""A/B timing of two builds of libfp8q_hip.so in ONE process on one GPU box (boxes of the pool differ by a few percent,
so an old-vs-new comparison has to alternate the two libraries on the same box).

    python tools/ab.py <set> [path/to/other/libfp8q_hip.so]        sets: epi  multi  mse  enc  k1  f64

Without a second library only the current build is timed.  Times are the median of back-to-back launches by HIP events
(small kernels: includes the launch gap, like the model's forward does)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fp8-quantization_amd")]
import torch  # noqa: E402
import fp8q  # noqa: E402
from fp8q import _lib  # noqa: E402

ops = fp8q.ops
dev = "cuda"


def use(path):
    """switch the process to another build of the library (fp8q._lib caches one handle)"""
    import ctypes
    _lib._lib = None
    if path:
        os.environ["FP8Q_SO"] = path
        L = ctypes.CDLL(os.path.abspath(path))     # an older build may lack newer entry points: bind what it has
        for name, (res, args) in _lib.SIGNATURES.items():
            fn = getattr(L, name, None)
            if fn is not None:
                fn.restype, fn.argtypes = res, args
        _lib._lib = L
    else:
        os.environ.pop("FP8Q_SO", None)
        fp8q.lib()


def bench(fn, iters=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / iters)
    return sorted(ts)[len(ts) // 2]


def cases_epi():
    """fused epilogue on the activation shapes of MobileNetV2 / ResNet-18 at batch 64 (bn + relu6 / relu [+ residual])"""
    shapes = [(32, 112), (16, 112), (96, 112), (96, 56), (24, 56), (144, 56), (144, 28), (32, 28), (192, 28), (192, 14), (64, 14),
              (384, 14), (96, 14), (576, 14), (576, 7), (160, 7), (960, 7), (320, 7), (1280, 7), (64, 56), (128, 28), (256, 14), (512, 7)]
    out = []
    mv = torch.tensor([4.0], device=dev)
    for C, hw in shapes:
        x = torch.randn(64, C, hw, hw, device=dev)
        y = torch.empty_like(x)
        bn = tuple(torch.rand(C, device=dev) + 0.5 for _ in range(4))
        kw = dict(bn=bn, bn_ab=ops.bn_fold(bn)) if os.environ.get("FOLD") == "1" else dict(bn=bn)     # FOLD=1: folded BN constants
        out.append((f"bn+relu6+q [64,{C},{hw},{hw}]", x.numel() * 8, lambda x=x, y=y, kw=kw: ops.affine_act_quantize(x, mv, 3, 8, 1, act=2, out=y, **kw)))
        if (C, hw) in ((64, 56), (128, 28), (256, 14), (512, 7), (24, 56), (32, 28), (64, 14), (96, 14), (160, 7)):
            r = torch.randn_like(x)
            out.append((f"bn+res+relu+q [64,{C},{hw},{hw}]", x.numel() * 12, lambda x=x, y=y, kw=kw, r=r: ops.affine_act_quantize(x, mv, 3, 8, 1, residual=r, act=1, out=y, **kw)))
        out.append((f"   plain K1 [64,{C},{hw},{hw}]", x.numel() * 8, lambda x=x, y=y: ops.quantize(x, mv, 3, 8, 1, out=y)))
    return out


def cases_multi():
    """ResNet-18's 21 weight tensors through the prepared multi-tensor plan"""
    shapes = [(64, 3, 7, 7)] + [(64, 64, 3, 3)] * 4 + [(128, 64, 3, 3), (128, 128, 3, 3), (128, 64, 1, 1)] + \
        [(128, 128, 3, 3)] * 2 + [(256, 128, 3, 3), (256, 256, 3, 3), (256, 128, 1, 1)] + [(256, 256, 3, 3)] * 2 + \
        [(512, 256, 3, 3), (512, 512, 3, 3), (512, 256, 1, 1)] + [(512, 512, 3, 3)] * 2 + [(1000, 512)]
    g = torch.Generator(device=dev).manual_seed(7)
    ws = [torch.randn(*sh, device=dev, generator=g) * 0.05 for sh in shapes]
    mvs = [ops.minmax(w, True, want_maxval=True)[2] for w in ws]
    n = sum(w.numel() for w in ws)
    plans = {}

    def run():
        key = os.environ.get("FP8Q_SO", "")
        if key not in plans:
            plans[key] = ops.MultiPlan([(w, mv, 2, 8, 1) for w, mv in zip(ws, mvs)])
        plans[key].launch()
    flat = torch.randn(n, device=dev)
    yflat = torch.empty_like(flat)
    return [("resnet18 21 tensors, plan", n * 8, run), ("   copy of the same bytes", n * 8, lambda: ops.copy(flat, out=yflat))]


def cases_mse():
    am = torch.randn(64, 32, 112, 112, device=dev)
    grid1 = torch.linspace(0.5, 6.0, 111, device=dev)[:, None].contiguous()
    out = []
    for ms in ([3.0], [2.0], [4.0], [5.0], [1.0, 2.0, 3.0, 4.0, 5.0, 6.0]):
        mses = torch.zeros(len(ms), 111, 1, device=dev)
        out.append((f"K4 [64,32,112,112] x111 m={ms}", am.numel() * 111 * len(ms), lambda ms=ms, mses=mses: ops.mse_grid(am, False, grid1, ms, 8, 1, mses)))
    relu = torch.relu(am)
    mses6 = torch.zeros(6, 111, 1, device=dev)
    out.append(("K4 relu([64,32,112,112]) x111 x6", am.numel() * 666, lambda: ops.mse_grid(relu, False, grid1, [1.0, 2.0, 3.0, 4.0, 5.0, 6.0], 8, 1, mses6)))
    return out


def cases_enc():
    out = []
    for C, inner, M in ((1 << 18, 147, 2), (1 << 20, 147, 2), (1 << 20, 99, 3), (1 << 18, 576, 2), (1 << 17, 1152, 2)):
        x = torch.randn(C, inner, device=dev)
        mv = ops.minmax(x, True, want_maxval=True)[2]
        codes = torch.empty(C, inner, dtype=torch.uint8, device=dev)
        y = torch.empty_like(x)
        out.append((f"encode [{C},{inner}] M={M}", x.numel() * 5, lambda x=x, mv=mv, codes=codes, M=M: ops.encode(x, mv, M, 8, 1, out=codes)))
        out.append((f"   decode [{C},{inner}]", x.numel() * 5, lambda x=x, mv=mv, codes=codes, M=M, y=y: ops.decode(codes, mv, M, 8, 1, out=y)))
        out.append((f"   K1 [{C},{inner}]", x.numel() * 8, lambda x=x, mv=mv, y=y, M=M: ops.quantize(x, mv, M, 8, 1, out=y)))
    for shape, pc in (((64, 64, 112, 112), False), ((16384, 4608), True), ((65536, 2304), True)):
        x = torch.randn(*shape, device=dev)
        mv = ops.minmax(x, True, want_maxval=True)[2] if pc else torch.tensor([3.0], device=dev)
        codes = torch.empty(shape, dtype=torch.uint8, device=dev)
        out.append((f"encode {list(shape)} {'per channel' if pc else 'per tensor'}", x.numel() * 5, lambda x=x, mv=mv, codes=codes: ops.encode(x, mv, 3, 8, 1, out=codes)))
    return out


def cases_f64():
    """the float64 lane: K1 at 16 B/element, row min/max at 8, the 1000-candidate line search of config 1 (5 M samples)"""
    x = torch.randn(1 << 26, dtype=torch.float64, device=dev)          # 512 MiB
    y = torch.empty_like(x)
    mv = torch.tensor([2.7361], device=dev)
    xs = torch.randn(5_000_000, dtype=torch.float64, device=dev)
    thr = (torch.arange(1, 1001, device=dev, dtype=torch.float32) * 0.0105).view(1000, 1).contiguous()
    sse = torch.zeros(1, 1000, 1, dtype=torch.float64, device=dev)
    return [("K1 f64 [2^26] E4M3 per tensor", x.numel() * 16, lambda: ops.quantize(x, mv, 3, 8, 1, out=y)),
            ("min/max f64 [2^26]", x.numel() * 8, lambda: ops.minmax_f64(x, False)),
            ("K4 f64 5M samples x 1000 candidates (E4M3)", xs.numel() * 1000, lambda: ops.mse_grid_f64(xs, False, thr, [3.0], 8, 1, sse, reduce="sum"))]


def cases_k1():
    x = torch.randn(1 << 21, 147, device=dev)
    y = torch.empty_like(x)
    mv = ops.minmax(x, True, want_maxval=True)[2]
    return [("K1 [2^21,147] e5m2 (headline)", x.numel() * 8, lambda: ops.quantize(x, mv, 2, 8, 1, out=y))]


def main():
    which = sys.argv[1]
    other = sys.argv[2] if len(sys.argv) > 2 else None
    iters = int(os.environ.get("ITERS", "200"))
    cases = globals()["cases_" + which]()
    libs = [("new", None)] + ([("old", other)] if other else [])
    res = {}
    for rep in range(2 if other else 1):
        for tag, path in libs:
            use(path)
            for name, nbytes, fn in cases:
                t = bench(fn, iters=iters if nbytes < (1 << 28) else max(iters // 10, 5))
                res.setdefault(name, {}).setdefault(tag, []).append(t)
    for name, nbytes, _ in cases:
        r = res[name]
        new = min(r["new"])
        line = f"{name:44s} new {new:9.1f} us {nbytes / new / 1e6:7.3f} TB/s"
        if "old" in r:
            old = min(r["old"])
            line += f" | old {old:9.1f} us {nbytes / old / 1e6:7.3f} TB/s | {100 * (old - new) / old:+5.1f} %"
        print(line, flush=True)


if __name__ == "__main__":
    main()
