"""ctypes binding of libfp8q_hip.so (include/fp8q.h).  There is NO fallback: if the library is
missing or fails to load, importing an op raises -- the product never runs on a CPU path."""
import ctypes
import os

from . import build as _build

_lib = None

_i64 = ctypes.c_int64
_vp = ctypes.c_void_p
_f = ctypes.c_float
_i = ctypes.c_int



class MseState(ctypes.Structure):
    """fp8q_mse_state (include/fp8q.h): device pointers of one MSE estimator's persistent state"""
    _fields_ = [("cur_min", _vp), ("cur_max", _vp), ("absmax", _vp), ("grid", _vp), ("mses", _vp), ("maxval", _vp),
                ("xmin", _vp), ("mbits", _vp), ("vote", _vp)]


class AffinePre(ctypes.Structure):
    """fp8q_affine_pre (include/fp8q.h): the BN + activation (+ residual) in front of a per-tensor MSE-calibrated quantizer"""
    _fields_ = [("x", _vp), ("residual", _vp), ("alpha_beta", _vp), ("N", _i64), ("C", _i64), ("HW", _i64), ("act", _i)]


class TensorDesc(ctypes.Structure):
    """fp8q_tensor_desc (include/fp8q.h)"""
    _fields_ = [("x", _vp), ("y", _vp), ("maxval", _vp), ("C", _i64), ("inner", _i64), ("n_maxval", _i64),
                ("mbits", _f), ("n_bits", _i), ("sign_bits", _i)]


SIGNATURES = {
    "fp8q_version": (_i, []),
    "fp8q_strerror": (ctypes.c_char_p, [_i]),
    "fp8q_quantize_f32": (_i, [_vp, _vp, _i64, _i64, _vp, _i64, _f, _i, _i, _vp]),
    "fp8q_minmax_workspace_bytes": (ctypes.c_size_t, [_i64, _i64]),
    "fp8q_minmax_f32": (_i, [_vp, _i64, _i64, _vp, _vp, _vp, _i, ctypes.c_double, _i, _vp,
                             ctypes.c_size_t, _vp]),
    "fp8q_minmax_packed_f32": (_i, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _i, ctypes.c_double, _i, _vp,
                                    ctypes.c_size_t, _vp]),
    "fp8q_ranges_unpack_f32": (_i, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "fp8q_minmax_workspace_check": (_i, [_vp, ctypes.c_size_t, _i, _vp]),
    "fp8q_fused_max_inner": (_i64, []),
    "fp8q_minmax_quantize_f32": (_i, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _f, _i, _i, _vp]),
    "fp8q_mse_workspace_bytes": (ctypes.c_size_t, [_i64, _i64, _i64, _i]),
    "fp8q_mse_grid_f32": (_i, [_vp, _i64, _i64, _vp, _i64, ctypes.POINTER(_f), _i, _i, _i, _vp, _vp,
                               ctypes.c_size_t, _vp]),
    "fp8q_mse_linspace_f32": (_i, [_vp, _i64, _i, ctypes.c_double, ctypes.c_double, _vp, _vp]),
    "fp8q_minmax_linspace_f32": (_i, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _i, ctypes.c_double, ctypes.c_double, _vp,
                                      ctypes.c_size_t, _vp]),
    "fp8q_mse_select_workspace_bytes": (ctypes.c_size_t, [_i64, _i]),
    "fp8q_mse_select_f32": (_i, [_vp, _vp, _i64, _i64, ctypes.POINTER(_f), _i, _i, _vp, _vp, _vp, _vp, _vp,
                                 ctypes.c_size_t, _vp]),
    "fp8q_quantize_dm_f32": (_i, [_vp, _vp, _i64, _i64, _vp, _i64, _vp, _i, _i, _vp]),
    "fp8q_quantize_ds_f32": (_i, [_vp, _vp, _i64, _i64, _vp, _i64, _f, _i, _vp, _vp]),
    "fp8q_sign_fold_u8": (_i, [_vp, _i64, _vp, _vp]),
    "fp8q_quantize_dms_f32": (_i, [_vp, _vp, _i64, _i64, _vp, _i64, _vp, _i, _vp, _vp]),
    "fp8q_mse_calibrate_workspace_bytes": (ctypes.c_size_t, [_i64, _i64, _i64, _i, ctypes.POINTER(ctypes.c_size_t),
                                                             ctypes.POINTER(ctypes.c_size_t)]),
    "fp8q_mse_calibrate_f32": (_i, [_vp, _vp, _i64, _i64, ctypes.POINTER(MseState), _i, _i, ctypes.POINTER(_f), _i, _i, _i,
                                    ctypes.POINTER(AffinePre), _vp, ctypes.c_size_t, _vp, ctypes.c_size_t, _vp, ctypes.c_size_t, _vp]),
    "fp8q_quantize_f64": (_i, [_vp, _vp, _i64, _i64, _vp, _i64, _f, _i, _i, _vp]),
    "fp8q_minmax_f64_workspace_bytes": (ctypes.c_size_t, [_i64, _i64]),
    "fp8q_minmax_f64": (_i, [_vp, _i64, _i64, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    "fp8q_mse_f64_workspace_bytes": (ctypes.c_size_t, [_i64, _i64, _i64, _i]),
    "fp8q_mse_grid_f64": (_i, [_vp, _i64, _i64, _vp, _i64, ctypes.POINTER(_f), _i, _i, _i, _vp, _i, _vp,
                               ctypes.c_size_t, _vp]),
    "fp8q_affine_act_quantize_f32": (_i, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i, _vp, _f, _i, _i,
                                          _vp]),
    "fp8q_bn_fold_f32": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "fp8q_quantizer_prepare_f32": (_i, [_vp, _f, _i, _i, _vp, _vp]),
    "fp8q_affine_act_quantize_ab_f32": (_i, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _i, _vp, _vp, _f, _i, _i, _vp]),
    "fp8q_affine_act_f32": (_i, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _i, _vp]),
    "fp8q_affine_act_minmax_linspace_f32": (_i, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _i, _vp, _vp, _vp, _vp, _i, ctypes.c_double,
                                                 ctypes.c_double, _vp, ctypes.c_size_t, _vp]),
    "fp8q_affine_act_minmax_workspace_bytes": (ctypes.c_size_t, [_i64, _i64, _i64]),
    "fp8q_affine_act_minmax_f32": (_i, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i,
                                        ctypes.c_double, _i, _vp, ctypes.c_size_t, _vp]),
    "fp8q_affine_act_minmax_packed_f32": (_i, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp,
                                               _i, ctypes.c_double, _i, _vp, ctypes.c_size_t, _vp]),
    "fp8q_encode_u8": (_i, [_vp, _vp, _i64, _i64, _vp, _i64, _f, _i, _i, _vp]),
    "fp8q_decode_u8": (_i, [_vp, _vp, _i64, _i64, _vp, _i64, _f, _i, _i, _vp]),
    "fp8q_multi_quantize_f32": (_i, [ctypes.POINTER(TensorDesc), _i, _vp]),
    "fp8q_multi_minmax_quantize_f32": (_i, [ctypes.POINTER(TensorDesc), ctypes.POINTER(_vp), _i, _vp]),
    "fp8q_multi_encode_u8": (_i, [ctypes.POINTER(TensorDesc), _i, _vp]),
    "fp8q_multi_minmax_encode_u8": (_i, [ctypes.POINTER(TensorDesc), ctypes.POINTER(_vp), _i, _vp]),
    "fp8q_multi_decode_u8": (_i, [ctypes.POINTER(TensorDesc), _i, _vp]),
    "fp8q_multi_plan_create": (_i, [ctypes.POINTER(TensorDesc), _i, ctypes.POINTER(_vp)]),
    "fp8q_multi_plan_launch": (_i, [_vp, _vp]),
    "fp8q_multi_plan_launches": (_i, [_vp]),
    "fp8q_multi_plan_destroy": (None, [_vp]),
    "fp8q_copy_f32": (_i, [_vp, _vp, _i64, _vp]),
}


class Fp8qError(RuntimeError):
    pass


def so_path():
    # FP8Q_SO: load an alternative build of the same library (kernel-tuning experiments only)
    return os.environ.get("FP8Q_SO") or _build.SO


def lib():
    """Load (once) the HIP library.  Raises if it has not been built: no silent fallback."""
    global _lib
    if _lib is None:
        path = so_path()
        if not os.path.exists(path):
            raise Fp8qError(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  The FP8 engine has no CPU/eager fallback.")
        L = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the .so is stale
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().fp8q_strerror(rc).decode()
        raise Fp8qError(f"{what} failed: {msg} (code {rc})")
