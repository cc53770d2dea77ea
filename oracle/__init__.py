"""CPU oracle for the FP8 fake-quantization hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package, and only as the checker.  The product (fp8-quantization_amd/) never does.

Two restatements of the reference's algorithm live here:

* ``oracle.c_*``  -- ctypes binding of oracle/fp8q_oracle.c (plain C, correctly rounded
  log2/pow, OpenMP).  This is the arithmetic contract the HIP kernels reproduce bit for bit.
* ``oracle.torch_eager`` -- the same op chain written with torch CPU ops; on the torch build
  that produced the golden fixtures it is bit-identical to the reference, so it pins the
  formula; it is also the "reference-equivalent eager CPU path" timed by bench.py.

Parity pin: tests/test_oracle_golden.py checks both against tests/golden/*.npz, which were
produced by importing the reference itself (tests/golden/make_golden.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libfp8q_oracle.so")
_lib = None


def build(force=False):
    """Compile oracle/fp8q_oracle.c -> oracle/_ref/libfp8q_oracle.so (gcc, seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ("fp8q_oracle.c", "fp8q_oracle_tables.h")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        f32p = ctypes.POINTER(ctypes.c_float)
        i64 = ctypes.c_int64
        L.orc_quantize_f32.argtypes = [f32p, f32p, i64, i64, f32p, i64, ctypes.c_float,
                                       ctypes.c_int, ctypes.c_int]
        L.orc_quantize_flat_f32.argtypes = [f32p, f32p, i64, ctypes.c_float, ctypes.c_float,
                                            ctypes.c_int, ctypes.c_int]
        L.orc_minmax_f32.argtypes = [f32p, i64, i64, f32p, f32p]
        L.orc_fold_f32.argtypes = [f32p, f32p, f32p, f32p, i64, ctypes.c_int, ctypes.c_double,
                                   ctypes.c_int]
        L.orc_absmax_f32.argtypes = [f32p, f32p, i64, f32p]
        L.orc_mse_grid_f32.argtypes = [f32p, i64, i64, f32p, i64, f32p, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_int, f32p]
        L.orc_affine_act_f32.argtypes = [f32p, f32p, f32p, i64, i64, i64, f32p, f32p, f32p, f32p, ctypes.c_int]
        u8p = ctypes.POINTER(ctypes.c_uint8)
        L.orc_encode_u8.argtypes = [f32p, u8p, i64, i64, f32p, i64, ctypes.c_float, ctypes.c_int, ctypes.c_int]
        L.orc_decode_u8.argtypes = [u8p, f32p, i64, i64, f32p, i64, ctypes.c_float, ctypes.c_int, ctypes.c_int]
        L.orc_fp_grid.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                  ctypes.POINTER(ctypes.c_double)]
        f64p = ctypes.POINTER(ctypes.c_double)
        L.orc_quantize_f64.argtypes = [f64p, f64p, i64, i64, f32p, i64, ctypes.c_float, ctypes.c_int, ctypes.c_int]
        L.orc_minmax_f64.argtypes = [f64p, i64, i64, f64p, f64p]
        L.orc_sse_grid_f64.argtypes = [f64p, i64, i64, f32p, i64, f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, f64p,
                                       ctypes.c_int]
        L.orc_log2_f64.argtypes = [ctypes.c_double]
        L.orc_log2_f64.restype = ctypes.c_double
        L.orc_exp2_f64.argtypes = [ctypes.c_double]
        L.orc_exp2_f64.restype = ctypes.c_double
        L.orc_num_threads.restype = ctypes.c_int
        L.orc_set_num_threads.argtypes = [ctypes.c_int]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _as_2d(x, per_channel):
    x = _f32(x)
    if per_channel:
        return x.reshape(x.shape[0], -1) if x.ndim > 0 else x.reshape(1, 1)
    return x.reshape(1, -1)


def c_quantize(x, maxval, mbits, n_bits=8, sign_bits=1):
    """quantize_to_fp8_ste_MM (fp8_quantizer.py:91-133).  maxval: scalar/[1] or [C] (dim 0)."""
    x = _f32(x)
    mv = _f32(np.atleast_1d(maxval)).reshape(-1)
    if mv.size == 1 and x.size >= (1 << 16):   # one long row: the element-parallel twin (same orc_quant1 per element)
        return c_quantize_flat(x, mv[0], mbits, n_bits, sign_bits).reshape(x.shape)
    x2 = _as_2d(x, mv.size != 1)
    assert mv.size in (1, x2.shape[0])
    y = np.empty_like(x2)
    rc = lib().orc_quantize_f32(_p(x2), _p(y), x2.shape[0], x2.shape[1], _p(mv), mv.size,
                                float(mbits), int(n_bits), int(sign_bits))
    assert rc == 0
    return y.reshape(x.shape)


def c_quantize_flat(x, maxval, mbits, n_bits=8, sign_bits=1, out=None):
    x = _f32(x).reshape(-1)
    y = np.empty_like(x) if out is None else out
    lib().orc_quantize_flat_f32(_p(x), _p(y), x.size, float(maxval), float(mbits), int(n_bits),
                                int(sign_bits))
    return y


def c_minmax(x, per_channel):
    """min/max over all dims (per tensor) or over dims >= 1 (per channel), NaN-propagating."""
    x2 = _as_2d(x, per_channel)
    mn = np.empty(x2.shape[0], np.float32)
    mx = np.empty(x2.shape[0], np.float32)
    lib().orc_minmax_f32(_p(x2), x2.shape[0], x2.shape[1], _p(mn), _p(mx))
    return mn, mx


def c_fold(cur_mn, cur_mx, mn, mx, mode, momentum=0.9, first=False):
    cur_mn, cur_mx = _f32(cur_mn).copy(), _f32(cur_mx).copy()
    mn, mx = _f32(mn), _f32(mx)
    lib().orc_fold_f32(_p(cur_mn), _p(cur_mx), _p(mn), _p(mx), mn.size, int(mode),
                       float(momentum), int(first))
    return cur_mn, cur_mx


def c_absmax(mn, mx):
    mn, mx = _f32(np.atleast_1d(mn)), _f32(np.atleast_1d(mx))
    out = np.empty_like(mn)
    lib().orc_absmax_f32(_p(mn), _p(mx), mn.size, _p(out))
    return out


def c_affine_act(x, bn=None, residual=None, act=0):
    """act(batch_norm_eval(x) + residual) on [N, C, ...] (quantized_folded_bn.py:39-55); bn = (mean, invstd, gamma,
    beta) or None; act 0 none / 1 ReLU / 2 ReLU6."""
    x = _f32(x)
    N, C = x.shape[0], x.shape[1]
    HW = x.size // max(N * C, 1)
    y = np.empty_like(x)
    null = ctypes.POINTER(ctypes.c_float)()
    b = [_f32(t) for t in bn] if bn is not None else None
    r = _f32(residual) if residual is not None else None
    rc = lib().orc_affine_act_f32(_p(x), _p(r) if r is not None else null, _p(y), N, C, HW,
                                  *([_p(t) for t in b] if b else [null] * 4), int(act))
    assert rc == 0
    return y


def c_mse_grid(x, per_channel, grid, mbits_list, n_bits=8, sign_bits=1, mses=None):
    """Accumulate mses[n_m, n_cand, C] += mean_c((x - q(x; m, grid[i, c]))^2)."""
    x2 = _as_2d(x, per_channel)
    grid = _f32(grid)
    mb = _f32(np.atleast_1d(mbits_list))
    C = x2.shape[0]
    assert grid.ndim == 2 and grid.shape[1] == C
    if mses is None:
        mses = np.zeros((mb.size, grid.shape[0], C), np.float32)
    mses = _f32(mses)
    lib().orc_mse_grid_f32(_p(x2), C, x2.shape[1], _p(grid), grid.shape[0], _p(mb), mb.size,
                           int(n_bits), int(sign_bits), _p(mses))
    return mses


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _pd(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _as_2d_f64(x, per_channel):
    x = _f64(x)
    if per_channel:
        return x.reshape(x.shape[0], -1) if x.ndim > 0 else x.reshape(1, 1)
    return x.reshape(1, -1)


def c_quantize_f64(x, maxval, mbits, n_bits=8, sign_bits=1):
    """quantize_to_fp8_ste_MM on a float64 tensor (fp8_quantizer.py:91-133 under ATen's type promotion: bias in
    float32, everything downstream of x in float64).  maxval: float32 scalar/[1] or [C]."""
    x = _f64(x)
    mv = _f32(np.atleast_1d(maxval)).reshape(-1)
    x2 = _as_2d_f64(x, mv.size != 1)
    assert mv.size in (1, x2.shape[0])
    y = np.empty_like(x2)
    rc = lib().orc_quantize_f64(_pd(x2), _pd(y), x2.shape[0], x2.shape[1], _p(mv), mv.size, float(mbits), int(n_bits),
                                int(sign_bits))
    assert rc == 0
    return y.reshape(x.shape)


def c_minmax_f64(x, per_channel):
    x2 = _as_2d_f64(x, per_channel)
    mn, mx = np.empty(x2.shape[0], np.float64), np.empty(x2.shape[0], np.float64)
    lib().orc_minmax_f64(_pd(x2), x2.shape[0], x2.shape[1], _pd(mn), _pd(mx))
    return mn, mx


def c_sse_grid_f64(x, per_channel, grid, mbits_list, n_bits=8, sign_bits=1, out=None, reduce="sum"):
    """out[n_m, n_cand, C] (float64) += sum (LineSearchEstimator.loss_fx, range_estimators.py:161-169) or mean
    (FP_MSE_Estimator, :337-347) over each row of (x - q(x; m, grid[i, c]))^2, x float64."""
    x2 = _as_2d_f64(x, per_channel)
    grid = _f32(grid)
    mb = _f32(np.atleast_1d(mbits_list))
    C = x2.shape[0]
    assert grid.ndim == 2 and grid.shape[1] == C
    out = np.zeros((mb.size, grid.shape[0], C), np.float64) if out is None else _f64(out)
    lib().orc_sse_grid_f64(_pd(x2), C, x2.shape[1], _p(grid), grid.shape[0], _p(mb), mb.size, int(n_bits),
                           int(sign_bits), _pd(out), int(reduce == "sum"))
    return out


def c_log2_f64(a):
    return lib().orc_log2_f64(float(a))


def c_exp2_f64(e):
    return lib().orc_exp2_f64(float(e))


def c_encode(x, maxval, mbits, n_bits=8, sign_bits=1):
    """uint8 storage codes of c_quantize(x) (layout of fp8_quantizer.py:13-41)."""
    x = _f32(x)
    mv = _f32(np.atleast_1d(maxval)).reshape(-1)
    x2 = _as_2d(x, mv.size != 1)
    codes = np.empty(x2.shape, np.uint8)
    rc = lib().orc_encode_u8(_p(x2), codes.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), x2.shape[0],
                             x2.shape[1], _p(mv), mv.size, float(mbits), int(n_bits), int(sign_bits))
    assert rc == 0
    return codes.reshape(x.shape)


def c_decode(codes, maxval, mbits, n_bits=8, sign_bits=1):
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    mv = _f32(np.atleast_1d(maxval)).reshape(-1)
    c2 = codes.reshape(codes.shape[0], -1) if mv.size != 1 else codes.reshape(1, -1)
    y = np.empty(c2.shape, np.float32)
    rc = lib().orc_decode_u8(c2.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), _p(y), c2.shape[0], c2.shape[1],
                             _p(mv), mv.size, float(mbits), int(n_bits), int(sign_bits))
    assert rc == 0
    return y.reshape(codes.shape)


def c_fp_grid(n_bits, ebits, bias):
    out = np.empty(2 ** n_bits, np.float64)
    rc = lib().orc_fp_grid(n_bits, ebits, bias, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    assert rc == 0
    return out


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))
