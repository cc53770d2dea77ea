"""Oracle-backed stand-in for fp8q.ops on CPU tensors -- TEST USE ONLY.

Lets the host-side mirror of the reference API (quantization/, models/) run end to end on a box
without a GPU, with the CPU oracle doing the arithmetic, so that host logic can be checked against
the reference's goldens independently of MIOpen/rocBLAS rounding.  The product never imports this."""
import contextlib

import numpy as np
import torch

import oracle


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def quantize(x, maxval, mbits, n_bits=8, sign_bits=1, out=None):
    mbits = float(mbits)        # (a 1-element tensor on the GPU path: the device-resident vote)
    if x.dtype == torch.float64:
        y = _t(oracle.c_quantize_f64(x.detach().numpy(), maxval.detach().numpy(), mbits, n_bits, sign_bits))
    else:
        y = _t(oracle.c_quantize(x.detach().numpy(), maxval.detach().numpy(), mbits, n_bits, sign_bits))
    if out is not None:
        out.copy_(y)
        return out
    return y


def new_packed(C, device):
    return torch.empty((C, 4), dtype=torch.float32, device=device)


def pack_ranges(mn, mx, packed):
    """what fold_store (csrc/fp8q_common.h) writes: {-min, max, isnan(min), isnan(max)}, a NaN travelling as -inf + flag"""
    nmn, nmx = np.isnan(mn), np.isnan(mx)
    with np.errstate(invalid="ignore"):
        p = np.stack([np.where(nmn, -np.inf, -mn), np.where(nmx, -np.inf, mx), nmn.astype(np.float32),
                      nmx.astype(np.float32)], 1).astype(np.float32)
    packed.copy_(_t(p))


def ranges_unpack(packed, cur_min=None, cur_max=None, maxval=None):
    p = packed.numpy().reshape(-1, 4)
    mn = np.where(p[:, 2] > 0, np.float32("nan"), -p[:, 0]).astype(np.float32)
    mx = np.where(p[:, 3] > 0, np.float32("nan"), p[:, 1]).astype(np.float32)
    res = [_t(mn), _t(mx), _t(oracle.c_absmax(mn, mx))]
    for i, dst in enumerate((cur_min, cur_max, maxval)):
        if dst is not None:
            dst.copy_(res[i].view_as(dst))
            res[i] = dst
    return tuple(res)


def minmax(x, per_channel, cur_min=None, cur_max=None, mode=0, momentum=0.9, want_maxval=False, packed=None):
    mn, mx = oracle.c_minmax(x.detach().numpy(), per_channel)
    if cur_min is not None and cur_max is not None:
        mn, mx = oracle.c_fold(cur_min.numpy(), cur_max.numpy(), mn, mx, mode, momentum)
    if packed is not None:
        pack_ranges(mn, mx, packed)
    out = (_t(mn), _t(mx))
    return out + (_t(oracle.c_absmax(mn, mx)),) if want_maxval else out


def minmax_quantize(x, mbits, n_bits=8, sign_bits=1, out=None):
    mn, mx = oracle.c_minmax(x.detach().numpy(), True)
    mv = oracle.c_absmax(mn, mx)
    return _t(oracle.c_quantize(x.detach().numpy(), mv, mbits, n_bits, sign_bits)), _t(mn), _t(mx), _t(mv)


def mse_grid(x, per_channel, grid, mbits_list, n_bits, sign_bits, mses):
    res = oracle.c_mse_grid(x.detach().numpy(), per_channel, grid.numpy(), list(mbits_list), n_bits, sign_bits,
                            mses.numpy().copy())
    mses.copy_(_t(res))
    return mses


def mse_linspace(mx, steps=111, lo_frac=0.1, hi_frac=1.2):
    """range_estimators.py:296-305: torch.linspace per channel, python-float products"""
    return torch.stack([torch.linspace(lo_frac * v, hi_frac * v, steps) for v in mx.reshape(-1).tolist()], 1)


def minmax_linspace(x, per_channel, steps=111, lo_frac=0.1, hi_frac=1.2):
    mn, mx, mv = minmax(x, per_channel, want_maxval=True)
    return mn, mx, mv, mse_linspace(mv, steps, lo_frac, hi_frac)


def mse_select(mses, grid, mbits_list, sign_bits=1):
    """range_estimators.py:350-369 with the reference's own torch calls"""
    best_m_per_ch = mses.min(1)[0].argmin(0)
    vote = int(torch.mode(best_m_per_ch).values.item())
    arg = mses[vote].argmin(0)
    maxval = grid.gather(0, arg.unsqueeze(0)).squeeze(0)
    return (torch.tensor([float(mbits_list[vote])]), torch.tensor([vote], dtype=torch.int32), maxval,
            sign_bits * -1.0 * maxval)


def minmax_f64(x, per_channel):
    mn, mx = oracle.c_minmax_f64(x.detach().numpy(), per_channel)
    return _t(mn), _t(mx)


def mse_grid_f64(x, per_channel, grid, mbits_list, n_bits, sign_bits, out, reduce="sum"):
    res = oracle.c_sse_grid_f64(x.detach().numpy(), per_channel, grid.numpy(), list(mbits_list), n_bits, sign_bits,
                                out.numpy().copy(), reduce=reduce)
    out.copy_(_t(res))
    return out


def fused_max_inner():
    return 16384


@contextlib.contextmanager
def patched():
    """with oracle_ops.patched(): ... -> fp8q.ops.* run on the CPU oracle inside the block."""
    import fp8q
    names = ("quantize", "minmax", "minmax_quantize", "mse_grid", "fused_max_inner", "new_packed", "ranges_unpack",
             "mse_linspace", "minmax_linspace", "mse_select", "minmax_f64", "mse_grid_f64")
    saved = {n: getattr(fp8q.ops, n) for n in names}
    try:
        for n in names:
            setattr(fp8q.ops, n, globals()[n])
        yield
    finally:
        for n, f in saved.items():
            setattr(fp8q.ops, n, f)
