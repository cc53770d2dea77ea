"""Parity metric shared by the oracle-vs-golden and HIP-vs-oracle tests.

`steps` = |y - y_ref| in units of the local step of the emulated FP8 grid;
north_star tolerance: <= 1 step everywhere.  We additionally bound how often a non-zero
step happens and the fp32-ULP distance of everything else.
"""
import numpy as np


def elem_step(x, maxval, mbits, n_bits=8, sign_bits=1):
    """float64 local grid step s(x) = 2^(p - M - bias) of every element (exact formula)."""
    x = np.asarray(x, np.float64)
    mv = np.asarray(maxval, np.float64).reshape(-1)
    if mv.size != 1:
        mv = mv.reshape([-1] + [1] * (x.ndim - 1))
    M = float(np.clip(np.round(mbits), 1, n_bits - sign_bits))
    E = n_bits - sign_bits - M
    with np.errstate(all="ignore"):
        bias = 2.0 ** E - np.log2(mv) + np.log2(2 - 2.0 ** (-M)) - 1
        lo = -mv if sign_bits == 1 else 0 * mv
        xc = np.minimum(np.maximum(x, lo), mv)
        p = np.maximum(np.floor(np.log2(np.abs(xc)) + bias), 1.0)
        return 2.0 ** (p - M - bias) * np.ones_like(x)


def ulp_dist(a, b):
    """distance in fp32 ULPs (monotone integer mapping); NaNs must be handled by the caller."""
    a = np.asarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.asarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, np.int64(-2147483648) - a, a)
    b = np.where(b < 0, np.int64(-2147483648) - b, b)
    return np.abs(a - b)


def compare(y, y_ref, step):
    y = np.asarray(y, np.float32)
    y_ref = np.asarray(y_ref, np.float32)
    assert y.shape == y_ref.shape
    nan_a, nan_b = np.isnan(y), np.isnan(y_ref)
    ok = ~(nan_a | nan_b)
    with np.errstate(all="ignore"):
        steps = np.where(ok, np.abs(y.astype(np.float64) - y_ref.astype(np.float64)) / step, 0.0)
    steps = np.nan_to_num(steps, nan=0.0, posinf=np.inf)
    ulps = np.where(ok, ulp_dist(np.where(ok, y, 0), np.where(ok, y_ref, 0)), 0)
    flips = steps >= 0.5
    return dict(
        n=int(y.size),
        nan_equal=bool(np.array_equal(nan_a, nan_b)),
        zero_sign_equal=bool(np.array_equal(np.signbit(y[ok & (y_ref == 0)]),
                                            np.signbit(y_ref[ok & (y_ref == 0)]))),
        exact_frac=float(np.mean((y.view(np.int32) == y_ref.view(np.int32)) | (nan_a & nan_b))),
        max_steps=float(steps.max()) if y.size else 0.0,
        flip_frac=float(flips.mean()) if y.size else 0.0,
        n_flips=int(flips.sum()),
        max_ulp_nonflip=int(ulps[~flips].max()) if (~flips).any() else 0,
    )


def assert_parity(y, y_ref, step, max_flip_frac=1e-4, max_ulp=2, min_flips_allowed=0, what=""):
    r = compare(y, y_ref, step)
    assert r["nan_equal"], f"{what}: NaN pattern differs {r}"
    assert r["zero_sign_equal"], f"{what}: sign of zero differs {r}"
    assert r["max_steps"] <= 1.0 + 1e-6, f"{what}: more than one grid step {r}"
    assert r["n_flips"] <= max(min_flips_allowed, max_flip_frac * r["n"]), f"{what}: too many flips {r}"
    assert r["max_ulp_nonflip"] <= max_ulp, f"{what}: ulp distance {r}"
    return r
