"""Soak of the round-2 kernels against the oracle: the single-launch two-stage min/max (tagged granules + reducer block:
per tensor and long rows, every fold mode, NaN, two streams at once, back-to-back calls of different shapes on one
workspace), k_mse_row (random ranges over all formats: one / two scale mantissas / exact path, ragged tiles, several rows),
the short-row codec modes of k_rows_flat, the multi-tensor plan.  usage: soak5.py <seed> <cases>"""
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
side = torch.cuda.Stream()
for case in range(ncase):
    what = f"seed {seed} case {case}"
    # ---- min/max, per tensor or a few long rows, folded twice; a second stream works on another tensor meanwhile ----
    if rng.rand() < 0.5:
        C, inner = 1, int(np.exp(rng.uniform(np.log(5), np.log(3e7))))
    else:
        C, inner = int(rng.randint(2, 40)), int(rng.randint(8200, 400000))
    x = (rng.randn(C, inner) * np.exp(rng.uniform(-3, 3))).astype(np.float32)
    if rng.rand() < 0.25: x.reshape(-1)[rng.randint(x.size)] = np.nan
    x2 = (rng.randn(C, inner) * np.exp(rng.uniform(-3, 3))).astype(np.float32)
    z = (rng.randn(int(rng.randint(1 << 16, 1 << 22)))).astype(np.float32)
    xd, x2d, zd = dev(x), dev(x2), dev(z)
    pc = C > 1
    mode = int(rng.randint(0, 3))
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        zs = [ops.minmax(zd, False) for _ in range(3)]
    mn, mx, mv = ops.minmax(xd, pc, want_maxval=True)
    cur = ops.minmax(x2d, pc, mn.clone(), mx.clone(), mode=mode, momentum=0.9, want_maxval=True)
    torch.cuda.current_stream().wait_stream(side)
    rmn, rmx = oracle.c_minmax(x if pc else x.reshape(-1), pc)
    same(mn.cpu().numpy(), rmn, "min " + what); same(mx.cpu().numpy(), rmx, "max " + what)
    same(mv.cpu().numpy(), oracle.c_absmax(rmn, rmx), "maxval " + what)
    r2 = oracle.c_minmax(x2 if pc else x2.reshape(-1), pc)
    emn, emx = oracle.c_fold(rmn, rmx, r2[0], r2[1], mode, 0.9)
    same(cur[0].cpu().numpy(), emn, f"fold {mode} min " + what); same(cur[1].cpu().numpy(), emx, f"fold {mode} max " + what)
    zmn, zmx = oracle.c_minmax(z, False)
    for a, b in zs:
        same(a.cpu().numpy(), zmn, "side-stream min " + what); same(b.cpu().numpy(), zmx, "side-stream max " + what)
    # ---- k_mse_row ----
    C, inner = (1, int(rng.randint(2048, 300000))) if rng.rand() < 0.6 else (int(rng.randint(2, 6)), int(rng.randint(2048, 9000)))
    sb = int(rng.rand() < 0.8)
    x = (rng.randn(C, inner) * np.exp(rng.uniform(-2, 2, (C, 1)))).astype(np.float32)
    if sb == 0: x = np.abs(x)
    ncand = int(rng.randint(1, 40))
    grid = np.exp(rng.uniform(np.log(1e-3), np.log(1e3), (ncand, C))).astype(np.float32)
    nm = int(rng.randint(1, 4))
    mb = sorted(set(float(v) for v in rng.randint(1, 8 - sb + 1, nm)))
    mses = torch.zeros(len(mb), ncand, C, device="cuda")
    ops.mse_grid(dev(x), C > 1, dev(grid), mb, 8, sb, mses)
    ref = oracle.c_mse_grid(x if C > 1 else x.reshape(-1), C > 1, grid, mb, 8, sb)
    got = mses.cpu().numpy()
    assert np.array_equal(np.isfinite(got), np.isfinite(ref)), "mse finite pattern " + what
    ok = np.isfinite(ref)
    np.testing.assert_allclose(got[ok], ref[ok], rtol=2e-5, atol=1e-37, err_msg="mse " + what + f" C={C} inner={inner} mb={mb} sb={sb}")
    # ---- short-row codec + plan ----
    inner = int(rng.randint(4, 2047)); C = max(2, int(rng.randint(8, 600000) // inner))
    M = int(rng.randint(1, 7)); sb = int(rng.rand() < 0.85)
    mvv = (np.abs(rng.randn(C)) * np.exp(rng.uniform(-3, 3)) + 1e-3).astype(np.float32)
    x = (rng.randn(C, inner) * (mvv[:, None] / 2)).astype(np.float32)
    if sb == 0: x = np.abs(x)
    xd, mvd = dev(x), dev(mvv)
    codes = ops.encode(xd, mvd, M, 8, sb)
    assert np.array_equal(codes.cpu().numpy(), oracle.c_encode(x, mvv, M, 8, sb)), f"encode {what} C={C} inner={inner} M={M}"
    yd = ops.decode(codes, mvd, M, 8, sb).cpu().numpy()
    same(yd, oracle.c_decode(codes.cpu().numpy(), mvv, M, 8, sb), f"decode {what} C={C} inner={inner} M={M}")
    plan = ops.MultiPlan([(xd, mvd, M, 8, sb), (zd, dev([1.5]), 3, 8, 1)])
    outs = plan.launch()
    same(outs[0].cpu().numpy(), oracle.c_quantize(x, mvv, M, 8, sb), "plan tensor 0 " + what)
    same(outs[1].cpu().numpy(), oracle.c_quantize(z, [1.5], 3, 8, 1), "plan tensor 1 " + what)
print(f"soak5 seed {seed}: {ncase} cases clean")
