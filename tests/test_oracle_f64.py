"""The float64 lane of the CPU oracle (BASELINE config 1's dtype) pinned against the reference (CPU-only tests).

Fixtures: g1c_quantize_f64.npz (quantize_to_fp8_ste_MM on float64 tensors) and g5_quant_error.npz
(LineSearchEstimator.loss_array: the 1000 per-candidate sums of squares the reference's float64 search produced),
both written by tests/golden/make_golden.py from the imported reference.
"""
import math
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import torch_eager as te


@pytest.fixture(scope="module")
def g1c(golden_dir):
    return np.load(os.path.join(golden_dir, "g1c_quantize_f64.npz"))


@pytest.fixture(scope="module")
def g5(golden_dir):
    return np.load(os.path.join(golden_dir, "g5_quant_error.npz"))


def cases_f64(g):
    for cid, mbits, mv, sb, nmv in g["cases"]:
        cid = int(cid)
        maxval = g[f"c{cid}_maxval"] if mv < 0 else np.array([mv], np.float32)
        yield cid, float(mbits), maxval, int(sb), g[f"c{cid}_x"], g[f"c{cid}_y"]


def f64_parity(y, y_ref, x, sign_bits, what=""):
    """float64 parity metric.  NaN pattern and sign of zero identical; every element either within 2 ulp(double) of the
    reference's, or a TIE FLIP: the two results are neighbouring grid points and x sits within 8 ulp of their midpoint
    (the hand-built ties of the fixture: with a non-integer bias the scale is irrational, two 1-ulp `pow` routines
    disagree in its last bit and an input built to the last bit of one of them rounds the other way in the other --
    both neighbours are then equally far from x to ~1e-16 relative).  Returns (n, n_flips, worst ulp of the rest)."""
    y, y_ref = np.asarray(y, np.float64), np.asarray(y_ref, np.float64)
    assert np.array_equal(np.isnan(y), np.isnan(y_ref)), f"{what}: NaN pattern"
    ok = ~np.isnan(y_ref)
    yy, rr = y[ok], y_ref[ok]
    xx = np.broadcast_to(np.asarray(x, np.float64), y_ref.shape)[ok]
    assert np.array_equal(np.signbit(yy), np.signbit(rr)), f"{what}: sign pattern"
    with np.errstate(all="ignore"):
        ulps = np.abs(yy - rr) / np.spacing(np.maximum(np.abs(yy), np.abs(rr)))
    far = ulps > 2
    mid = (yy + rr) / 2
    xa = np.abs(xx) if sign_bits else np.abs(np.maximum(xx, 0.0))
    tie = np.abs(xa - np.abs(mid)) <= 8 * np.spacing(np.abs(mid))
    assert not (far & ~tie).any(), f"{what}: {int((far & ~tie).sum())} elements differ by more than 2 ulp off a tie"
    return int(ok.sum()), int((far & tie).sum()), float(ulps[~far].max()) if (~far).any() else 0.0


def test_table_log2_exp2_against_libm():
    """orc_log2_d / orc_exp2_d are the f64 lane's DEFINITION of log2 / 2^x (shared op sequence with the kernels);
    they must be what a 1-ulp double library computes."""
    rng = np.random.RandomState(0)
    xs = np.concatenate([np.exp(rng.uniform(-700, 700, 4000)), 1 + rng.uniform(-1e-3, 1e-3, 500),
                         [5e-324, 1e-310, 2.2250738585072014e-308, 2.0, 1.0, 0.5, 3.0]])
    for a in xs:
        v, r = oracle.c_log2_f64(a), math.log2(a)
        assert abs(v - r) <= 2.0 ** -45 * (1 + abs(r)), (a, v, r)
    assert oracle.c_log2_f64(0.0) == -math.inf and oracle.c_log2_f64(math.inf) == math.inf
    assert math.isnan(oracle.c_log2_f64(-1.0)) and math.isnan(oracle.c_log2_f64(math.nan))
    for e in np.concatenate([rng.uniform(-300, 300, 4000), np.float64(np.float32(rng.uniform(-150, 150, 2000))),
                             np.arange(-40, 40, dtype=np.float64)]):
        v, r = oracle.c_exp2_f64(e), 2.0 ** float(e)
        assert abs(v - r) <= 2 * np.spacing(r), (e, v, r)
    assert oracle.c_exp2_f64(math.inf) == math.inf and oracle.c_exp2_f64(-math.inf) == 0.0
    assert math.isnan(oracle.c_exp2_f64(math.nan)) and oracle.c_exp2_f64(-1074.0) == 5e-324


def test_c_oracle_f64_vs_reference(g1c):
    tot = flips = 0
    worst = 0.0
    for cid, mbits, maxval, sb, x, y_ref in cases_f64(g1c):
        y = oracle.c_quantize_f64(x, maxval, mbits, 8, sb)
        n, f, w = f64_parity(y, y_ref, x, sb, f"case {cid}")
        tot, flips, worst = tot + n, flips + f, max(worst, w)
    print(f"\nC oracle (f64) vs reference: {tot} elements, worst {worst} ulp off ties, {flips} flips on hand-built ties")
    assert worst <= 2 and flips <= 0.005 * tot


def test_torch_eager_f64_vs_reference(g1c):
    """The eager chain on a float64 tensor IS the reference's (same ATen ops, same promotion): bit-identical on the torch
    build that wrote the fixture, the parity metric elsewhere."""
    same = True
    for cid, mbits, maxval, sb, x, y_ref in cases_f64(g1c):
        y = te.fake_quant(torch.from_numpy(x.copy()), 8, torch.from_numpy(np.asarray(maxval, np.float32)),
                          torch.tensor([mbits]), sb)
        assert y.dtype == torch.float64
        y = y.numpy()
        ok = ~np.isnan(y_ref)
        if not (np.array_equal(np.isnan(y), np.isnan(y_ref)) and np.array_equal(y[ok].view(np.int64), y_ref[ok].view(np.int64))):
            same = False
            f64_parity(y, y_ref, x, sb, f"eager case {cid}")
    print("\ntorch eager f64 vs reference: bit-identical" if same else "\ntorch eager f64: within the parity metric")


def test_minmax_f64():
    rng = np.random.RandomState(3)
    x = rng.randn(5, 1000)
    x[2, 17] = np.nan
    mn, mx = oracle.c_minmax_f64(x, True)
    with np.errstate(all="ignore"):
        np.testing.assert_array_equal(mn, x.min(1))
        np.testing.assert_array_equal(mx, x.max(1))
    mn, mx = oracle.c_minmax_f64(x[:2], False)
    assert mn[0] == x[:2].min() and mx[0] == x[:2].max()


def distributions():
    from quantization.distributions import ClippedGaussDistr, UniformDistr, ClippedStudentTDistr
    return {"uniform": UniformDistr(range_min=-1.0, range_max=1.0, params_dict={}),
            "gauss": ClippedGaussDistr(params_dict={"mu": 0.0, "sigma": 1.0}, range_min=-10.0, range_max=10.0),
            "student": ClippedStudentTDistr(params_dict={"nu": 8.0}, range_min=-100.0, range_max=100.0)}


def line_search_sample(name, n=200000):
    np.random.seed(10)                      # seed_all(10) of the reference (compute_quant_error.py:18)
    return distributions()[name].sample((n,))


def line_search_thresholds(g5, name, exp_bits):
    """float32 candidate thresholds: what torch.Tensor([step_size * k]) holds (range_estimators.py:243-244, fp8_quantizer.py:230)."""
    max_pos_thr, max_search_range, step_size, one_sided = g5[f"{name}_search_{exp_bits}"]
    return np.float32(step_size * np.arange(1, 1001)), step_size


@pytest.mark.parametrize("name", ["uniform", "gauss", "student"])
def test_c_oracle_line_search_losses_vs_reference(g5, name):
    """Per-candidate float64 sums of squares of the reference's line search, on a subset of the 1000 candidates (every
    16th, plus the neighbourhood of the reference's optimum): equal to 1e-12 relative, and the reference's optimum is
    the oracle's optimum among its neighbours."""
    x = line_search_sample(name)
    for exp_bits in (5, 4, 3, 2):
        loss = g5[f"{name}_loss_{exp_bits}"][0]
        assert np.isinf(loss[0])
        thr, step = line_search_thresholds(g5, name, exp_bits)
        best = int(np.argmin(loss))
        idx = sorted(set(range(16, 1001, 16)) | set(range(max(best - 4, 1), min(best + 5, 1001))) | {1, 2, 1000})
        got = oracle.c_sse_grid_f64(x, False, thr[np.array(idx) - 1].reshape(-1, 1), [7 - exp_bits], 8, 1)[0, :, 0]
        np.testing.assert_allclose(got, loss[idx], rtol=1e-12)
        near = [i for i in idx if abs(i - best) <= 4]
        assert idx[int(np.argmin(got))] == best or near[int(np.argmin(got[[idx.index(i) for i in near]]))] == best
        rows = {int(r[0]): r for r in g5[f"{name}_rows"]}
        assert float(np.float32(step * best)) == rows[exp_bits][2]          # the range the reference returned
