import sys, os
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
    inner = int(rng.choice([rng.randint(1, 300), rng.randint(1, 2100), rng.randint(2040, 9000), 4 * rng.randint(16, 2100)]))
    C = int(rng.randint(1, max(2, min(900, 600000 // inner))))
    M = int(rng.randint(1, 7)); sb = int(rng.rand() < 0.85); off = int(rng.choice([0, 0, 0, 0, 1, 2, 3]))
    x = (rng.randn(C, inner) * np.exp(rng.uniform(-5, 5, (C, 1)))).astype(np.float32)
    if sb == 0: x = np.abs(x)
    if rng.rand() < 0.2 and C > 2: x[rng.randint(C)] = 0.0
    if rng.rand() < 0.1: x.reshape(-1)[rng.randint(x.size)] = np.nan
    base = torch.empty(x.size + 4, device="cuda"); xd = base[off: off + x.size].view(C, inner); xd.copy_(torch.from_numpy(x))
    mn, mx = oracle.c_minmax(x, True); mv = oracle.c_absmax(mn, mx)
    what = f"seed {seed} case {case}: C={C} inner={inner} M={M} sb={sb} off={off}"
    ref = oracle.c_quantize(x, mv, M, 8, sb)
    same(ops.quantize(xd, dev(mv), M, 8, sb).cpu().numpy(), ref, "K1 " + what)
    yf, gmn, gmx, gmv = ops.minmax_quantize(xd, M, 8, sb)
    assert np.array_equal(bits(gmn.cpu().numpy()), bits(mn)) or np.array_equal(np.isnan(gmn.cpu().numpy()), np.isnan(mn)), what
    np.testing.assert_array_equal(gmn.cpu().numpy(), mn, err_msg=what); np.testing.assert_array_equal(gmx.cpu().numpy(), mx, err_msg=what)
    same(yf.cpu().numpy(), ref, "fused " + what)
    kmn, kmx = ops.minmax(xd, True)
    np.testing.assert_array_equal(kmn.cpu().numpy(), mn, err_msg=what); np.testing.assert_array_equal(kmx.cpu().numpy(), mx, err_msg=what)
    codes = ops.encode(xd.contiguous(), dev(mv), M, 8, sb)
    assert np.array_equal(codes.cpu().numpy(), oracle.c_encode(x, mv, M, 8, sb)), "codes " + what
    dec = ops.decode(codes, dev(mv), M, 8, sb).cpu().numpy()
    same(dec, oracle.c_decode(codes.cpu().numpy(), mv, M, 8, sb), "decode vs oracle " + what)
    ok = ~np.isnan(ref)
    ulp = np.abs(bits(dec)[ok].astype(np.int64) - bits(ref)[ok].astype(np.int64))
    assert ulp.size == 0 or ulp.max() <= 64, f"codec round trip {what}: {ulp.max()} ulp"
    outs = ops.multi_quantize([(xd, dev(mv), M, 8, sb), (xd.contiguous(), dev(mv), M, 8, sb)])
    same(outs[0].cpu().numpy(), ref, "multi0 " + what); same(outs[1].cpu().numpy(), ref, "multi1 " + what)
print("soak ok", seed, ncase)
