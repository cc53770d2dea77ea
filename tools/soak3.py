"""soak 3: odd formats (n_bits 3..8, fractional mantissa widths, unsigned) and special values (inf, NaN, denormals,
huge / tiny / degenerate ranges) through K1, fused, codec and multi, per-channel and per-tensor, against the oracle"""
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
    assert not bad.any(), f"{what}: {bad.sum()} differ, first {np.argwhere(bad)[:3].tolist()} y={y[bad][:3]} ref={ref[bad][:3]}"
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
seed = int(sys.argv[1]); ncase = int(sys.argv[2])
rng = np.random.RandomState(seed)
specials = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1.17549435e-38, 3.4028235e38, -3.4028235e38, 1.0, -1.0,
                     2.0 ** -126, 2.0 ** 127, 6e-8, 65504.0], np.float32)
for case in range(ncase):
    n_bits = int(rng.choice([3, 4, 5, 6, 7, 8])); sb = int(rng.rand() < 0.8)
    mbits = float(rng.choice([rng.randint(0, n_bits + 1), rng.randint(1, max(2, n_bits - sb)) + rng.choice([0.0, 0.5, 0.25, 0.75, 0.49999])]))
    if n_bits - sb - max(1, min(round(mbits), n_bits - sb)) > 7: continue
    C = int(rng.randint(1, 40)); inner = int(rng.choice([1, 5, 17, 64, 147, 256, 300, 576, 1024, 2048, 4100]))
    scale = np.exp(rng.uniform(-60, 60, (C, 1))) if rng.rand() < 0.3 else np.exp(rng.uniform(-4, 4, (C, 1)))
    x = (rng.randn(C, inner) * scale).astype(np.float32)
    k = rng.randint(0, x.size, size=min(x.size, 24)); x.reshape(-1)[k] = specials[rng.randint(len(specials), size=len(k))]
    if sb == 0 and rng.rand() < 0.7: x = np.abs(x)
    mv = np.abs(x).max(1).astype(np.float32) if rng.rand() < 0.6 else np.exp(rng.uniform(-70, 70, C)).astype(np.float32)
    mv[np.isnan(mv)] = 1.0
    if rng.rand() < 0.2: mv[rng.randint(C)] = rng.choice([0.0, np.inf, 1e-45, 3.4e38, np.nan])
    what = f"seed {seed} case {case}: C={C} inner={inner} n_bits={n_bits} mbits={mbits} sb={sb}"
    with np.errstate(all="ignore"):
        ref = oracle.c_quantize(x, mv, mbits, n_bits, sb)
        same(ops.quantize(dev(x), dev(mv), mbits, n_bits, sb).cpu().numpy(), ref, "K1 " + what)
        one = mv[:1]
        same(ops.quantize(dev(x), dev(one), mbits, n_bits, sb).cpu().numpy(), oracle.c_quantize(x, one, mbits, n_bits, sb), "K1 per-tensor " + what)
        if inner <= ops.fused_max_inner():
            mn, mx = oracle.c_minmax(x, True); fmv = oracle.c_absmax(mn, mx)
            yf, gmn, gmx, gmv = ops.minmax_quantize(dev(x), mbits, n_bits, sb)
            np.testing.assert_array_equal(gmn.cpu().numpy(), mn, err_msg="fused min " + what)   # by value: the sign of a zero
            np.testing.assert_array_equal(gmx.cpu().numpy(), mx, err_msg="fused max " + what)   # min / max is unspecified
            same(yf.cpu().numpy(), oracle.c_quantize(x, fmv, mbits, n_bits, sb), "fused " + what)
        Mr = max(1, min(int(np.round(np.float32(mbits))), n_bits - sb))
        if n_bits - sb - Mr < 1: continue   # no exponent bit: not encodable
        codes = ops.encode(dev(x), dev(mv), mbits, n_bits, sb).cpu().numpy()
        assert np.array_equal(codes, oracle.c_encode(x, mv, mbits, n_bits, sb)), "encode " + what
        dec = ops.decode(torch.from_numpy(codes).cuda(), dev(mv), mbits, n_bits, sb).cpu().numpy()
        same(dec, oracle.c_decode(codes, mv, mbits, n_bits, sb), "decode " + what)
        outs = ops.multi_quantize([(dev(x), dev(mv), mbits, n_bits, sb), (dev(x), dev(one), mbits, n_bits, sb)])
        same(outs[0].cpu().numpy(), ref, "multi " + what)
print("soak3 ok", seed, ncase)
