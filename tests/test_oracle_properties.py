"""Size-independent properties of K1 (fp8_quantizer.py:91-133) checked on the CPU oracle with random formats and
ranges: the same properties the GPU tests assert at BASELINE sizes (tests/test_hip_parity.py)."""
import numpy as np
from hypothesis import assume, given, settings, strategies as st

import oracle


def _case(seed, M, sign_bits, log_mv, n=4096):
    rng = np.random.RandomState(seed)
    mv = np.float32(np.exp(log_mv))
    x = (rng.randn(n) * mv * rng.choice([0.01, 0.3, 1.0, 3.0])).astype(np.float32)
    x[:4] = [0.0, mv, -mv, mv * 7]
    if sign_bits == 0:
        x = np.abs(x)
    return x, mv


@settings(max_examples=150, deadline=None, derandomize=True)
@given(seed=st.integers(0, 2 ** 31 - 1), M=st.integers(1, 6), sign_bits=st.integers(0, 1),
       log_mv=st.floats(-20.0, 20.0))
def test_quantizer_properties(seed, M, sign_bits, log_mv):
    # formats whose smallest scale 2^(1 - M - bias) is a normal fp32 number; beyond that the reference's x / s is 0/0
    # (E = 7 with a tiny maxval: NaN outputs, reproduced bit for bit by the kernels, tests/test_hip_parity.py)
    assume(2.0 ** (8 - sign_bits - M) - log_mv / np.log(2.0) + M < 120)
    x, mv = _case(seed, M, sign_bits, log_mv)
    y = oracle.c_quantize(x, [mv], M, 8, sign_bits)
    # idempotent up to the reference's own rounding: its scale table 2^(fl32((p - M) - bias)) is not exactly geometric
    # (fp32 subtraction per binade), so a grid point at a binade top can be re-rendered a few ULP away
    # (tests/test_codes.py::test_roundtrip_on_non_geometric_scale_tables) -- never on another grid point
    y2 = oracle.c_quantize(y, [mv], M, 8, sign_bits)
    ulp = np.abs(y2.view(np.int32).astype(np.int64) - y.view(np.int32).astype(np.int64))
    assert ulp.max() <= 64 and np.array_equal(np.signbit(y2), np.signbit(y))
    # clamped to the range: the top grid value is maxval up to the fp32 rounding of the bias chain (fp8_quantizer.py:110-113)
    top = np.abs(y).max()
    # (|bias| up to ~130 for E = 7: ulp(bias) ln 2 ~ 1e-5 relative)
    assert top <= mv * (1 + 2.0 ** -15) and y[1] == top and y[3] == top     # maxval and 7 maxval land on the top value
    # at most 2^8 distinct grid points (a binade top can appear in its two few-ULP-apart renderings: count clusters)
    u = np.unique(y.astype(np.float64))
    distinct = 1 + int(np.sum(np.diff(u) > 1e-5 * np.maximum(np.abs(u[1:]), np.abs(u[:-1]))))
    assert distinct <= 256
    # monotone -- up to the same few-ULP effect at binade tops (2^(M+1) s_p vs 2^M s_(p+1))
    xs = np.sort(x)
    ys = oracle.c_quantize(xs, [mv], M, 8, sign_bits).astype(np.float64)
    assert np.all(np.diff(ys) >= -1e-5 * np.abs(ys[1:]))
    if sign_bits == 1:                                       # odd symmetry
        np.testing.assert_array_equal(oracle.c_quantize(-x, [mv], M, 8, 1).view(np.int32), (-y).view(np.int32))
    # error: at most half a step; a step is 2^-M of its binade's lower end, so |y - x| <= 2^-(M+1) |y| above the
    # subnormal range (whose fixed step s_1 is the smallest positive output)
    ladder = (mv * 2.0 ** -np.arange(0, 80, 0.25)).astype(np.float32)
    lo = oracle.c_quantize(ladder, [mv], M, 8, sign_bits)
    s1 = lo[lo > 0].min()
    inside = np.abs(x) <= mv
    err = np.abs(y.astype(np.float64) - x.astype(np.float64))[inside]
    bound = np.maximum(np.abs(y[inside]).astype(np.float64) * 2.0 ** -(M + 1), 0.5 * float(s1)) * (1 + 1e-5)
    assert np.all(err <= bound)


@settings(max_examples=25, deadline=None, derandomize=True)
@given(seed=st.integers(0, 2 ** 31 - 1), M=st.integers(1, 6), C=st.integers(1, 9))
def test_per_channel_equals_per_tensor_on_every_channel(seed, M, C):
    rng = np.random.RandomState(seed)
    mv = np.exp(rng.uniform(-6, 6, C)).astype(np.float32)
    x = (rng.randn(C, 257) * mv[:, None]).astype(np.float32)
    y = oracle.c_quantize(x, mv, M, 8, 1)
    for c in range(C):
        np.testing.assert_array_equal(y[c].view(np.int32), oracle.c_quantize(x[c], [mv[c]], M, 8, 1).view(np.int32))


@settings(max_examples=25, deadline=None, derandomize=True)
@given(seed=st.integers(0, 2 ** 31 - 1), splits=st.integers(2, 6))
def test_minmax_fold_is_associative(seed, splits):
    """allminmax over batches == current_minmax over the concatenation (the multi-GPU all-reduce relies on it)."""
    rng = np.random.RandomState(seed)
    x = rng.randn(splits, 5, 33).astype(np.float32)
    whole = oracle.c_minmax(np.ascontiguousarray(x.transpose(1, 0, 2)).reshape(5, -1), True)
    cur = oracle.c_minmax(x[0], True)
    for b in range(1, splits):
        mn, mx = oracle.c_minmax(x[b], True)
        cur = oracle.c_fold(cur[0], cur[1], mn, mx, 1)
    np.testing.assert_array_equal(cur[0], whole[0])
    np.testing.assert_array_equal(cur[1], whole[1])


def test_minmax_signed_zero_is_order_independent():
    """IEEE 754-2019 minimum / maximum: a zero minimum is -0.0 iff the row holds a -0.0, a zero maximum +0.0 iff it holds a
    +0.0 -- whatever the order of the elements (ATen returns whichever zero its reduction met first; include/fp8q.h)."""
    import oracle
    rng = np.random.RandomState(0)
    base = np.array([0.0, -0.0, 0.0, 0.0, -0.0, 0.5, 0.25], np.float32)
    for _ in range(20):
        row = rng.permutation(base)
        mn, mx = oracle.c_minmax(row.reshape(1, -1), True)
        assert mn[0] == 0 and np.signbit(mn[0]) and mx[0] == 0.5
        mn, mx = oracle.c_minmax(-row.reshape(1, -1), True)
        assert mx[0] == 0 and not np.signbit(mx[0]) and mn[0] == -0.5
        mn64, mx64 = oracle.c_minmax_f64(-row.astype(np.float64).reshape(1, -1), True)
        assert mx64[0] == 0 and not np.signbit(mx64[0])
    only_pos = np.zeros((1, 5), np.float32)
    mn, mx = oracle.c_minmax(only_pos, True)
    assert not np.signbit(mn[0]) and not np.signbit(mx[0])
    mn, mx = oracle.c_minmax(-only_pos, True)
    assert np.signbit(mn[0]) and np.signbit(mx[0])
    long_row = np.zeros((1, (1 << 20) + 5), np.float32)      # the thread-parallel branch
    long_row[0, 12345] = -0.0
    mn, mx = oracle.c_minmax(long_row, True)
    assert np.signbit(mn[0]) and not np.signbit(mx[0])
    # the allminmax fold of two estimates
    cmn, cmx = oracle.c_fold(np.float32([0.0]), np.float32([-0.0]), np.float32([-0.0]), np.float32([0.0]), 1)
    assert np.signbit(cmn[0]) and not np.signbit(cmx[0])
    cmn, cmx = oracle.c_fold(np.float32([-0.0]), np.float32([0.0]), np.float32([0.0]), np.float32([-0.0]), 1)
    assert np.signbit(cmn[0]) and not np.signbit(cmx[0])
