"""soak 2: per-tensor K1 / K3 (+ fold), N2 epilogue and MSE grid on random geometries, against the oracle"""
import sys
sys.path[:0] = ["/root/repo", "/root/repo/fp8-quantization_amd", "/root/repo/tests"]
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
    # ---- per-tensor K1 / K3 with sizes around step multiples
    k = int(rng.choice([1, 2, 3, 7, 8, 31, 64, 100, 255, 256, 257, 511, 2047, 2048, 2049]))
    n = max(1, k * int(rng.choice([1024, 4096, 8192])) + int(rng.randint(-5, 6)))
    M = int(rng.randint(1, 7)); sb = int(rng.rand() < 0.8); off = int(rng.choice([0, 0, 0, 1, 3]))
    x = (rng.randn(n) * np.exp(rng.uniform(-3, 3))).astype(np.float32)
    if sb == 0: x = np.abs(x)
    base = torch.empty(n + 4, device="cuda"); xd = base[off: off + n]; xd.copy_(torch.from_numpy(x))
    what = f"seed {seed} case {case}: n={n} M={M} sb={sb} off={off}"
    mn, mx = oracle.c_minmax(x, False); mv = oracle.c_absmax(mn, mx)
    gmn, gmx, gmv = ops.minmax(xd, False, want_maxval=True)
    same(gmn.cpu().numpy(), mn, "K3 min " + what); same(gmx.cpu().numpy(), mx, "K3 max " + what); same(gmv.cpu().numpy(), mv, "K5 " + what)
    for mode in (1, 2):
        fmn, fmx = ops.minmax(xd * 0.5, False, gmn.clone(), gmx.clone(), mode=mode)
        rmn, rmx = oracle.c_fold(mn, mx, *oracle.c_minmax(x * np.float32(0.5), False), mode, 0.9)
        same(fmn.cpu().numpy(), rmn, f"fold{mode} min " + what); same(fmx.cpu().numpy(), rmx, f"fold{mode} max " + what)
    same(ops.quantize(xd, dev(mv), M, 8, sb).cpu().numpy(), oracle.c_quantize(x, mv, M, 8, sb), "K1 per-tensor " + what)
    # ---- N2 epilogue
    N = int(rng.randint(1, 9)); C = int(rng.choice([1, 3, 8, 16, 24, 64, 100])); HW = int(rng.choice([1, 4, 9, 49, 64, 196, 225, 1024, 3136, 5000]))
    if (C * HW) % 4 == 0:
        a = (rng.randn(N, C, HW) * 2).astype(np.float32); r = rng.randn(N, C, HW).astype(np.float32)
        var = (rng.rand(C) + 0.5).astype(np.float32)
        invstd = (np.float32(1) / np.sqrt(var + np.float32(1e-5))).astype(np.float32)
        mean, gamma, beta = rng.randn(C).astype(np.float32), (rng.rand(C) + 0.5).astype(np.float32), rng.randn(C).astype(np.float32)
        use_bn, use_res, act = bool(rng.rand() < 0.7), bool(rng.rand() < 0.5), int(rng.randint(0, 3))
        t = a.copy()
        if use_bn:
            alpha = (invstd * gamma).astype(np.float32)
            bp = np.array([np.float32(np.float64(beta[c]) - np.float64(mean[c]) * np.float64(alpha[c])) for c in range(C)], np.float32)
            # fma(x, alpha, beta'): exact product + sum in float64 then one rounding (53 bits hold a 24x24-bit product + addend)
            bpf = np.array([np.float32(np.float64(-mean[c]) * np.float64(alpha[c]) + np.float64(beta[c])) for c in range(C)], np.float32)
            t = (t.astype(np.float64) * alpha.reshape(1, -1, 1).astype(np.float64) + bpf.reshape(1, -1, 1).astype(np.float64)).astype(np.float32)
        if use_res: t = (t + r).astype(np.float32)
        if act >= 1: t = np.where(t < 0, np.float32(0), t)
        if act == 2: t = np.where(t > 6, np.float32(6), t)
        mvt = np.array([2.5], np.float32)
        bn = tuple(dev(b) for b in (mean, invstd, gamma, beta)) if use_bn else None
        y = ops.affine_act_quantize(dev(a), dev(mvt), 3, 8, 1, bn=bn, residual=dev(r) if use_res else None, act=act)
        same(y.cpu().numpy(), oracle.c_quantize(t, mvt, 3, 8, 1), f"N2 quant seed {seed} case {case} N={N} C={C} HW={HW} bn={use_bn} res={use_res} act={act}")
        amn, amx, _ = ops.affine_act_minmax(dev(a), bn=bn, residual=dev(r) if use_res else None, act=act)
        tmn, tmx = oracle.c_minmax(t, False)
        same(amn.cpu().numpy(), tmn, f"N2 min case {case}"); same(amx.cpu().numpy(), tmx, f"N2 max case {case}")
    # ---- MSE grid
    if case % 4 == 0:
        Cm = int(rng.choice([1, 3, 16])); im = int(rng.choice([27, 147, 1000, 5000]))
        xm = rng.randn(Cm, im).astype(np.float32)
        pc = Cm > 1
        mxm = np.abs(xm).max(1) if pc else np.array([np.abs(xm).max()])
        grid = np.stack([np.linspace(0.1 * m, 1.2 * m, 111, dtype=np.float32) for m in mxm], 1).astype(np.float32)
        ms = [1.0, 2.0, 3.0, 4.0, 5.0, 6.0]
        got = torch.zeros(6, 111, len(mxm), device="cuda")
        ops.mse_grid(dev(xm), pc, dev(grid), ms, 8, 1, got)
        ref = oracle.c_mse_grid(xm, pc, grid, ms, 8, 1)
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=2e-4, atol=1e-12, err_msg=f"MSE case {case}")
print("soak2 ok", seed, ncase)
