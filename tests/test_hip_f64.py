"""The float64 lane on the GPU (fp8q_quantize_f64 / fp8q_minmax_f64 / fp8q_mse_grid_f64) against the CPU oracle
(bit for bit per element) and against the reference's own float64 line search (g5 loss arrays)."""
import os

import numpy as np
import pytest
import torch

import oracle
from test_oracle_f64 import cases_f64, f64_parity, line_search_sample, line_search_thresholds

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g1c(golden_dir):
    return np.load(os.path.join(golden_dir, "g1c_quantize_f64.npz"))


@pytest.fixture(scope="module")
def g5(golden_dir):
    return np.load(os.path.join(golden_dir, "g5_quant_error.npz"))


def same_bits_f64(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    nan = np.isnan(b)
    return np.array_equal(np.isnan(a), nan) and np.array_equal(a[~nan].view(np.int64), b[~nan].view(np.int64))


def hip_quantize(x, maxval, mbits, sb):
    import fp8q
    y = fp8q.ops.quantize(torch.from_numpy(np.ascontiguousarray(x)).cuda(), torch.from_numpy(np.asarray(maxval, np.float32)).cuda(),
                          mbits, 8, sb)
    assert y.dtype == torch.float64
    return y.cpu().numpy()


def test_quantize_f64_bit_exact_vs_oracle_and_reference(g1c):
    """Every fixture case: bit-identical to the oracle (ties, binade borders +-1 ulp, denormals, +-0, +-inf, NaN,
    degenerate channels), hence inside the reference parity metric the oracle is pinned with."""
    tot = 0
    for cid, mbits, maxval, sb, x, y_ref in cases_f64(g1c):
        y = hip_quantize(x, maxval, mbits, sb)
        assert same_bits_f64(y, oracle.c_quantize_f64(x, maxval, mbits, 8, sb)), f"case {cid}"
        f64_parity(y, y_ref, x, sb, f"case {cid} vs reference")
        tot += y.size
    assert tot > 40000


@pytest.mark.parametrize("shape,per_channel", [((1 << 22,), False), ((3, 5, 7), False), ((64, 3, 7, 7), True),
                                               ((513, 2049), True), ((1, 4097), True), ((2, 2048), False)])
def test_quantize_f64_shapes(shape, per_channel):
    rng = np.random.RandomState(hash(shape) % 1000)
    x = rng.standard_normal(int(np.prod(shape))).reshape(shape) * 1.7
    mv = (np.abs(rng.standard_normal(shape[0])) * 2 + 0.05).astype(np.float32) if per_channel else np.float32([2.7361])
    for M in (2, 3, 5):
        y = hip_quantize(x, mv, M, 1)
        assert same_bits_f64(y, oracle.c_quantize_f64(x, mv, M, 8, 1)), (shape, M)
    # binade borders in bulk: powers of two and their neighbours under an integer-bias format (c1 == 1)
    if not per_channel:
        e = rng.randint(-20, 8, size=x.size)
        xb = np.ldexp(1.0, e) * rng.choice([1.0, 1 - 2.0 ** -53, 1 + 2.0 ** -52, -1.0], size=x.size)
        xb = xb.reshape(shape)
        for M, mvd in ((3, 448.0), (2, 57344.0)):
            assert same_bits_f64(hip_quantize(xb, [mvd], M, 1), oracle.c_quantize_f64(xb, [mvd], M, 8, 1))


def test_quantize_f64_in_place_and_errors():
    import fp8q
    x = torch.randn(1000, dtype=torch.float64, device="cuda")
    mv = torch.tensor([1.5], device="cuda")
    ref = fp8q.ops.quantize(x, mv, 3)
    out = x.clone()
    fp8q.ops.quantize(out, mv, 3, out=out)
    assert torch.equal(out, ref)
    with pytest.raises(fp8q.Fp8qError):
        fp8q.ops.quantize(x, mv, 3, out=torch.empty(1000, dtype=torch.float32, device="cuda"))
    with pytest.raises(fp8q.Fp8qError):
        fp8q.ops.quantize(x, mv.double(), 3)
    assert fp8q.ops.quantize(x[:0], mv, 3).numel() == 0


def test_minmax_f64():
    import fp8q
    rng = np.random.RandomState(5)
    for shape, pc in (((1 << 21) + 3,), False), ((7, 1001), True), ((3, 4, 5), False):
        x = rng.standard_normal(int(np.prod(shape))).reshape(shape)
        mn, mx = fp8q.ops.minmax_f64(torch.from_numpy(x).cuda(), pc)
        rmn, rmx = oracle.c_minmax_f64(x, pc)
        assert np.array_equal(mn.cpu().numpy(), rmn) and np.array_equal(mx.cpu().numpy(), rmx)
    x = rng.standard_normal((4, 5000))
    x[1, 77] = np.nan
    mn, mx = fp8q.ops.minmax_f64(torch.from_numpy(x).cuda(), True)
    assert same_bits_f64(np.where(np.isnan(mn.cpu().numpy()), np.nan, mn.cpu().numpy()), oracle.c_minmax_f64(x, True)[0])
    assert torch.isnan(mx[1]) and not torch.isnan(mx[0])


@pytest.mark.parametrize("per_channel", [False, True])
def test_mse_grid_f64_vs_oracle(per_channel):
    import fp8q
    rng = np.random.RandomState(11)
    x = rng.standard_normal((6, 3000)) * np.array([0.1, 1, 3, 0.5, 2, 10])[:, None]
    C = 6 if per_channel else 1
    grid = (np.linspace(0.05, 4.0, 300)[:, None] * np.ones((1, C))).astype(np.float32)
    grid[7, 0] = 0.0                                     # a degenerate candidate: NaN, like the reference's
    for reduce in ("sum", "mean"):
        out = torch.zeros(2, 300, C, dtype=torch.float64, device="cuda")
        fp8q.ops.mse_grid_f64(torch.from_numpy(x).cuda(), per_channel, torch.from_numpy(grid).cuda(), [3.0, 2.0], 8, 1, out,
                              reduce=reduce)
        fp8q.ops.mse_grid_f64(torch.from_numpy(x).cuda(), per_channel, torch.from_numpy(grid).cuda(), [3.0, 2.0], 8, 1, out,
                              reduce=reduce)         # accumulates
        ref = oracle.c_sse_grid_f64(x, per_channel, grid, [3.0, 2.0], 8, 1, reduce=reduce)
        got = out.cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.isnan(ref[:, 7, 0]).all()
        np.testing.assert_allclose(got[~np.isnan(ref)], 2 * ref[~np.isnan(ref)], rtol=1e-13)


@pytest.mark.parametrize("name", ["uniform", "gauss", "student"])
def test_line_search_losses_equal_the_references(g5, name):
    """All 1000 per-candidate float64 sums of the reference's LineSearchEstimator on its own 200 k-sample draw, the
    same argmin, the same returned range -- for every FP format of compute_quant_error.py."""
    import fp8q
    x = torch.from_numpy(line_search_sample(name)).cuda()
    for exp_bits in (5, 4, 3, 2):
        loss = g5[f"{name}_loss_{exp_bits}"][0]
        thr, step = line_search_thresholds(g5, name, exp_bits)
        out = torch.zeros(1, 1000, 1, dtype=torch.float64, device="cuda")
        fp8q.ops.mse_grid_f64(x, False, torch.from_numpy(thr.reshape(-1, 1)).cuda(), [7.0 - exp_bits], 8, 1, out, reduce="sum")
        got = out.cpu().numpy()[0, :, 0]
        np.testing.assert_allclose(got, loss[1:], rtol=1e-12)
        assert int(np.argmin(got)) + 1 == int(np.argmin(loss))


@pytest.mark.gpu
@pytest.mark.parametrize("name,pc,incl", [("pt", False, True), ("pc", True, False)])
def test_fp_mse_estimator_on_float64_data_vs_reference(golden_dir, name, pc, incl):
    """g4c: FP_MSE_Estimator fed float64 tensors.  The search grid is the reference's bit for bit (products of the FLOAT64
    maximum, narrowed by torch.linspace), the accumulated table follows to 1e-6 (float64 means added into the float32 table
    in float64, as ATen's in-place float32 += float64 does), the same mantissa width and maxval come out."""
    import os
    from quantization.quantizers.fp8_quantizer import FPQuantizer
    from quantization.range_estimators import FP_MSE_Estimator
    g = np.load(os.path.join(golden_dir, "g4c_mse_f64.npz"))
    q = FPQuantizer(n_bits=8, per_channel=pc, mantissa_bits=3, maxval=None, set_maxval=True, mse_include_mantissa_bits=incl)
    est = FP_MSE_Estimator(per_channel=pc, quantizer=q)
    for b in range(2):
        x = torch.from_numpy(g[f"{name}_x{b}"]).cuda()
        assert x.dtype == torch.float64
        mn, mx = est(x)
        if b == 0:
            assert np.array_equal(est.search_grid.cpu().numpy().view(np.int32), g[f"{name}_grid"].view(np.int32))
        np.testing.assert_allclose(est.mses.cpu().numpy(), g[f"{name}_mses{b}"], rtol=1e-6)
        assert float(q.mantissa_bits) == float(g[f"{name}_mbits{b}"])
        np.testing.assert_array_equal(mx.cpu().numpy().reshape(-1), g[f"{name}_max{b}"].reshape(-1))
