"""Seeded random-geometry checks of the round-4 entry points against the CPU oracle: float64 K1 / min-max / search,
K1 with a device-resident mantissa width, the device-side selection, and the sort-once K4 path at awkward sizes."""
import numpy as np
import pytest
import torch

import oracle
from oracle import torch_eager as te

pytestmark = pytest.mark.gpu


def _bits64(a):
    a = np.asarray(a, np.float64)
    return np.where(np.isnan(a), np.int64(-1), a.view(np.int64))


def test_fuzz_f64_lane():
    import fp8q
    rng = np.random.RandomState(2024)
    for it in range(40):
        per_channel = bool(rng.randint(2))
        C = int(rng.choice([1, 2, 3, 17, 64, 300]))
        inner = int(rng.choice([1, 2, 5, 147, 2047, 2048, 2049, 4097, 10000]))
        x = rng.standard_normal((C, inner)) * 10 ** rng.uniform(-3, 3)
        if it % 7 == 0:
            x.reshape(-1)[rng.randint(x.size, size=3)] = [np.nan, np.inf, -0.0]
        if it % 5 == 0:
            x = np.ldexp(1.0, rng.randint(-30, 10, size=x.shape)) * rng.choice([-1.0, 1.0], size=x.shape)
        mv = (np.abs(rng.standard_normal(C if per_channel else 1)) * np.abs(x[np.isfinite(x)]).max() + 1e-3).astype(np.float32)
        M = float(rng.choice([1, 2, 3, 4, 5, 6, 2.5]))
        sb = int(rng.choice([1, 1, 0]))
        xd = torch.from_numpy(x).cuda()
        y = fp8q.ops.quantize(xd, torch.from_numpy(mv).cuda(), M, 8, sb).cpu().numpy()
        ref = oracle.c_quantize_f64(x, mv, M, 8, sb)
        assert np.array_equal(_bits64(y), _bits64(ref)), (it, C, inner, M, sb, per_channel)
        if not np.isnan(x).any():
            mn, mx = fp8q.ops.minmax_f64(xd, per_channel)
            rmn, rmx = oracle.c_minmax_f64(x, per_channel)
            assert np.array_equal(mn.cpu().numpy(), rmn) and np.array_equal(mx.cpu().numpy(), rmx), it
        if it % 4 == 0 and np.isfinite(x).all():
            Cg = C if per_channel else 1
            grid = (np.abs(rng.standard_normal((9, Cg))) * np.abs(x).max() + 1e-6).astype(np.float32)
            out = torch.zeros(2, 9, Cg, dtype=torch.float64, device="cuda")
            fp8q.ops.mse_grid_f64(xd, per_channel, torch.from_numpy(grid).cuda(), [2.0, 4.0], 8, 1, out, reduce="sum")
            ref = oracle.c_sse_grid_f64(x, per_channel, grid, [2.0, 4.0], 8, 1, reduce="sum")
            np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-13, atol=1e-300, err_msg=str(it))


def test_fuzz_device_mantissa_bits_and_select():
    import fp8q
    rng = np.random.RandomState(7)
    for it in range(30):
        per_channel = bool(rng.randint(2))
        C = int(rng.choice([1, 3, 64, 1000]))
        inner = int(rng.choice([1, 9, 147, 576, 4099]))
        x = torch.from_numpy((rng.standard_normal((C, inner)) * 10 ** rng.uniform(-2, 2)).astype(np.float32)).cuda()
        mv = torch.from_numpy((np.abs(rng.standard_normal(C if per_channel else 1)) + 0.01).astype(np.float32)).cuda()
        m = float(rng.uniform(0.0, 8.5))
        sb = int(rng.choice([1, 0]))
        want = fp8q.ops.quantize(x, mv, m, 8, sb)
        got = fp8q.ops.quantize(x, mv, torch.tensor([m], device="cuda"), 8, sb)
        assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (it, m, sb)
        n_m, n_cand = int(rng.randint(1, 7)), int(rng.choice([1, 5, 111, 200]))
        mses = rng.uniform(0, 1, (n_m, n_cand, C)).astype(np.float32)
        mses = np.round(mses * 8) / 8 if it % 3 == 0 else mses        # ties
        if it % 5 == 0:
            mses[rng.randint(n_m), rng.randint(n_cand), rng.randint(C)] = np.nan
        grid = rng.uniform(0.1, 9, (n_cand, C)).astype(np.float32)
        mb = [float(k + 1) for k in range(n_m)]
        g_mb, g_vote, g_mv, g_xm = fp8q.ops.mse_select(torch.from_numpy(mses).cuda(), torch.from_numpy(grid).cuda(), mb, sb)
        r_mb, r_mv, _ = te.mse_select(torch.from_numpy(mses), torch.from_numpy(grid), mb)
        assert float(g_mb) == r_mb and torch.equal(g_mv.cpu(), r_mv), it
        assert torch.equal(g_xm.cpu().view(torch.int32), (sb * -1.0 * r_mv).view(torch.int32)), it


@pytest.mark.parametrize("n,kind", [(1 << 20, "gauss"), ((1 << 20) + 256, "gauss"), ((1 << 20) + 255, "relu6"), (1 << 21, "neg"),
                                    ((1 << 20) + 1, "zeros"), ((1 << 20) + 1024 * 256, "levels"), (3 * (1 << 20) + 13, "heavy")])
def test_sorted_mse_awkward_sizes_vs_oracle(n, kind):
    """the sort-once K4 path (>= 2^20 elements, >= 256 candidates) on sizes at prefix-block / superblock borders and on
    data with massive duplicates; candidates far below and far above the data; a subset of the table against the oracle"""
    import fp8q
    g = torch.Generator(device="cuda").manual_seed(n % 1000)
    x = torch.randn(n, device="cuda", generator=g)
    if kind == "relu6":
        x = torch.clamp(x * 3, 0, 6)
    elif kind == "neg":
        x = -x.abs() * 0.01
    elif kind == "zeros":
        x = torch.zeros(n, device="cuda")
        x[5] = 1.0
    elif kind == "levels":
        x = torch.round(x * 20) / 20                        # ~200 distinct values: millions of equal keys
    elif kind == "heavy":
        x = x * torch.exp(torch.randn(n, device="cuda", generator=g) * 2)
    mx = x.abs().max().reshape(1)
    grid = torch.cat([fp8q.ops.mse_linspace(mx, 111), fp8q.ops.mse_linspace(mx * 50, 8), fp8q.ops.mse_linspace(mx * 1e-3, 8)])
    mb = [1.0, 3.0, 6.0]
    mses = torch.zeros(3, grid.shape[0], 1, device="cuda")
    fp8q.ops.mse_grid(x, False, grid, mb, 8, 1, mses)
    idx = [0, 3, 57, 110, 111, 118, 119, 126]
    ref = oracle.c_mse_grid(x.cpu().numpy(), False, grid.cpu().numpy()[idx], mb, 8, 1)
    got = mses.cpu().numpy()[:, idx, :]
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    ok = ~np.isnan(ref)
    np.testing.assert_allclose(got[ok], ref[ok], rtol=1e-5, atol=1e-24 * float(np.nanmax(got)))


@pytest.mark.parametrize("kind", ["relu6", "mixed", "negative", "nan"])
def test_hist_mse_unsigned_formats_vs_oracle(kind):
    """unsigned formats (allow_unsigned: sign_bits = 0) on the interval-histogram route: a negative element is clipped to 0,
    so its squared error is x^2 for every candidate -- summed apart from the partition (exact squares in double, fixed
    order) and added to every candidate; the mantissa search then covers M = 1 ... 7 (up to 386 cells per candidate)"""
    import fp8q
    n = (1 << 20) + 13
    g = torch.Generator(device="cuda").manual_seed(17)
    x = torch.randn(n, device="cuda", generator=g) * 2.0
    if kind == "relu6":
        x = torch.clamp(x * 2, 0, 6)
    elif kind == "negative":
        x = -x.abs()
        x[7] = 0.5
    elif kind == "nan":
        x[1234] = float("nan")
    mx = x[torch.isfinite(x)].abs().max().reshape(1)
    grid = fp8q.ops.mse_linspace(mx, 111)
    mb = [1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0]
    mses = torch.zeros(7, 111, 1, device="cuda")
    fp8q.ops.mse_grid(x, False, grid, mb, 8, 0, mses)
    idx = [0, 9, 55, 110]
    ref = oracle.c_mse_grid(x.cpu().numpy(), False, grid.cpu().numpy()[idx], mb, 8, 0)
    got = mses.cpu().numpy()[:, idx, :]
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    ok = ~np.isnan(ref)
    np.testing.assert_allclose(got[ok], ref[ok], rtol=1e-5)


@pytest.mark.parametrize("n_cand,M", [(1000, 3.0), (2000, 6.0), (3000, 6.0)])
def test_hist_mse_many_candidates_vs_oracle(n_cand, M):
    """LineSearchEstimator-sized candidate sets on the interval-histogram route: 136 K ... 580 K intervals, i.e. the
    superblock totals scanned inside k_mse_eval in one trip (<= 256 superblocks), in two (<= 512), and by the separate
    k_iv_scan_top launch (more); a sample of the table against the oracle"""
    import fp8q
    g = torch.Generator(device="cuda").manual_seed(n_cand)
    x = torch.randn(1 << 20, device="cuda", generator=g) * 1.7
    grid = torch.linspace(0.02, 9.0, n_cand, device="cuda").reshape(n_cand, 1).contiguous()
    mses = torch.zeros(1, n_cand, 1, device="cuda")
    fp8q.ops.mse_grid(x, False, grid, [M], 8, 1, mses)
    idx = sorted(set(np.linspace(0, n_cand - 1, 24).astype(int).tolist()))
    ref = oracle.c_mse_grid(x.cpu().numpy(), False, grid.cpu().numpy()[idx], [M], 8, 1)
    np.testing.assert_allclose(mses.cpu().numpy()[:, idx, :], ref, rtol=1e-5)


@pytest.mark.parametrize("inner", [1, 31, 2047, 2048, 5000])
def test_mse_degenerate_candidates_divide_like_the_reference(inner):
    """found by tests/soak.py: an E7M1 (8 bits, unsigned) candidate below ~2^-21 underflows its first scale to 0, so the
    reference's 0 / 0 turns every ZERO element into NaN -- but not the zero padding of a tile -- while a denormal element
    in a higher binade is quantized normally; bias > 128 also puts binade 1 below the exponent-field range the fast
    kernels read p from.  Both lane-per-candidate (short rows) and lane-per-element (>= 2048) kernels, against the oracle."""
    import fp8q
    rows = {"zero": np.zeros(inner, np.float32), "denormal": np.full(inner, 1e-42, np.float32),
            "mixed": np.where(np.arange(inner) % 3 == 0, 0.0, 3e-8).astype(np.float32)}
    grid = np.array([[1e-7], [3e-7], [1e-3]], np.float32)
    for name, x in rows.items():
        for mb, sb in (([1.0], 0), ([1.0, 7.0], 0), ([1.0, 2.0], 1)):
            out = torch.zeros(len(mb), 3, 1, device="cuda")
            fp8q.ops.mse_grid(torch.from_numpy(x).cuda().reshape(1, -1), False, torch.from_numpy(grid).cuda(), mb, 8, sb, out)
            ref = oracle.c_mse_grid(x.reshape(1, -1), False, grid, mb, 8, sb)
            got = out.cpu().numpy()
            assert np.array_equal(np.isnan(got), np.isnan(ref)), (name, inner, mb, sb, got.ravel(), ref.ravel())
            ok = ~np.isnan(ref)
            np.testing.assert_allclose(got[ok], ref[ok], rtol=1e-5, atol=0, err_msg=str((name, inner, mb, sb)))
    if inner == 1:
        assert np.isnan(oracle.c_mse_grid(np.zeros((1, 1), np.float32), False, grid[:1], [1.0], 8, 0)).all()   # the case itself


def test_minmax_signed_zero_contract():
    """found by tests/soak.py: a zero minimum is -0.0 when the row holds one, a zero maximum +0.0 (IEEE 754-2019
    minimum / maximum; include/fp8q.h) -- in every min/max route and in the oracle, whatever the element order"""
    import fp8q
    rng = np.random.RandomState(3)
    for C, inner in ((1000, 27), (64, 147), (3, 5000), (1, 1 << 21), (8, 2049)):
        x = np.maximum(rng.standard_normal((C, inner)), 0.0).astype(np.float32)
        x[:, rng.randint(inner)] = -0.0
        x[0] = -np.abs(x[0])                                    # a row whose maximum is a zero
        x[0, : 2] = [0.0, -0.0]
        xd = torch.from_numpy(x).cuda()
        rmn, rmx = oracle.c_minmax(x, True)
        assert np.signbit(rmn[1:]).all() and not np.signbit(rmx[0])
        mn, mx = fp8q.ops.minmax(xd, True)
        assert np.array_equal(mn.cpu().numpy().view(np.int32), rmn.view(np.int32)), (C, inner)
        assert np.array_equal(mx.cpu().numpy().view(np.int32), rmx.view(np.int32)), (C, inner)
        if inner <= fp8q.ops.fused_max_inner():
            _, fmn, fmx, _ = fp8q.ops.minmax_quantize(xd, 2.0)
            assert np.array_equal(fmn.cpu().numpy().view(np.int32), rmn.view(np.int32)), (C, inner)
            assert np.array_equal(fmx.cpu().numpy().view(np.int32), rmx.view(np.int32)), (C, inner)
        tmn, tmx = fp8q.ops.minmax(xd, False)
        omn, omx = oracle.c_minmax(x, False)
        assert tmn.cpu().numpy().view(np.int32)[0] == omn.view(np.int32)[0] and tmx.cpu().numpy().view(np.int32)[0] == omx.view(np.int32)[0]
