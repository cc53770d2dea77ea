"""Soak of the LDS-staged short-row kernels (k_rows_staged, k_rows_staged_mm) against the oracle: rows of 4..256
elements, all formats, small tensors (one chunk per block) and tensors of more chunks than the persistent grid
(software-pipelined loop).  usage: soak4.py <seed> <cases>"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fp8-quantization_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch, oracle, fp8q
ops = fp8q.ops
def bits(a): return np.ascontiguousarray(a, dtype=np.float32).view(np.int32)
def same(y, ref, what):
    y, ref = np.asarray(y, np.float32), np.asarray(ref, np.float32)
    na, nb = np.isnan(y), np.isnan(ref)
    assert np.array_equal(na, nb), what + " NaN pattern"
    bad = (bits(y) != bits(ref)) & ~na
    assert not bad.any(), f"{what}: {bad.sum()} differ, first {np.argwhere(bad)[:3].tolist()}"
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
seed = int(sys.argv[1]); ncase = int(sys.argv[2])
rng = np.random.RandomState(seed)
for case in range(ncase):
    inner = int(rng.randint(4, 257))
    big = rng.rand() < 0.3
    total = rng.randint(8_500_000, 12_000_000) if big else rng.randint(8, 700_000)
    C = max(2, int(total // inner))
    M = int(rng.randint(1, 7)); sb = int(rng.rand() < 0.85)
    x = (rng.randn(C, inner) * np.exp(rng.uniform(-5, 5, (C, 1)))).astype(np.float32)
    if sb == 0: x = np.abs(x)
    if rng.rand() < 0.3: x[rng.randint(C)] = 0.0
    if rng.rand() < 0.3: x.reshape(-1)[rng.randint(x.size)] = np.nan
    if rng.rand() < 0.2: x[rng.randint(C), rng.randint(inner)] = np.inf
    xd = dev(x)
    mn, mx = oracle.c_minmax(x, True); mv = oracle.c_absmax(mn, mx)
    what = f"seed {seed} case {case}: C={C} inner={inner} M={M} sb={sb}"
    ref = oracle.c_quantize(x, mv, M, 8, sb)
    yf, gmn, gmx, gmv = ops.minmax_quantize(xd, M, 8, sb)
    same(gmn.cpu().numpy(), mn, "fused min " + what); same(gmx.cpu().numpy(), mx, "fused max " + what)
    same(gmv.cpu().numpy(), mv, "fused maxval " + what)
    same(yf.cpu().numpy(), ref, "fused " + what)
    kmn, kmx, kmv = ops.minmax(xd, True, want_maxval=True)
    same(kmn.cpu().numpy(), mn, "K2 min " + what); same(kmx.cpu().numpy(), mx, "K2 max " + what); same(kmv.cpu().numpy(), mv, "K2 maxval " + what)
    mode = int(rng.randint(1, 3))
    x2 = (x * np.float32(rng.uniform(0.5, 1.5))).astype(np.float32)
    r2 = oracle.c_minmax(x2, True)
    cur = ops.minmax(dev(x2), True, kmn.clone(), kmx.clone(), mode=mode, momentum=0.9)
    emn, emx = oracle.c_fold(mn, mx, r2[0], r2[1], mode, 0.9)
    same(cur[0].cpu().numpy(), emn, f"K2 fold {mode} min " + what); same(cur[1].cpu().numpy(), emx, f"K2 fold {mode} max " + what)
print(f"soak4 seed {seed}: {ncase} cases clean")
