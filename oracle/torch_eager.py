"""Torch-eager restatement of the hot path -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Each function issues the same ATen ops, in the same order and dtype, as the reference
function it cites, so on the torch build that generated tests/golden it is bit-identical to
the reference (tests/test_oracle_golden.py asserts exact equality there and falls back to the
ULP metric on another CPU/torch build).  bench.py times `fake_quant` on the host cores as the
"reference-equivalent eager CPU path".
"""
import torch


def fake_quant(x, n_bits, maxval, mbits, sign_bits):
    """quantize_to_fp8_ste_MM, /root/reference/quantization/quantizers/fp8_quantizer.py:91-133.

    x: fp32 tensor; maxval: tensor [1] or [C] (C = x.shape[0]); mbits: tensor [1] / 0-dim.
    """
    m = torch.clamp(torch.round(mbits), 1, n_bits - sign_bits)          # :105
    e = n_bits - sign_bits - m                                          # :106
    if maxval.shape[0] != 1 and maxval.dim() != x.dim():                # :108-109
        maxval = maxval.view([-1] + [1] * (x.dim() - 1))
    bias = 2 ** e - torch.log2(maxval) + torch.log2(2 - 2 ** (-m)) - 1  # :110
    lo = -maxval if sign_bits == 1 else torch.zeros_like(maxval)        # :112
    xc = torch.min(torch.max(x, lo), maxval)                            # :113
    p = torch.clamp(torch.floor(torch.log2(torch.abs(xc)) + bias), 1.0)  # :128
    s = 2.0 ** (p - m - bias)                                           # :130
    return torch.round(xc / s) * s                                      # :132


def minmax(x, per_channel):
    """range_estimators.py:62-74 / :84-91: batch min and max."""
    if per_channel:
        f = x.view(x.shape[0], -1)
        return f.min(-1)[0], f.max(-1)[0]
    return torch.min(x), torch.max(x)


def fold(cur, new, mode, momentum=0.9):
    """cur/new: (min, max) pairs; mode: 'current' | 'all' | 'running' (range_estimators.py:72,97,122)."""
    if cur is None or mode == "current":
        return new
    if mode == "all":
        return torch.min(cur[0], new[0]), torch.max(cur[1], new[1])
    return ((1 - momentum) * new[0] + momentum * cur[0],
            (1 - momentum) * new[1] + momentum * cur[1])


def absmax(xmin, xmax):
    """fp8_quantizer.py:236."""
    return torch.abs(torch.max(torch.abs(xmin), xmax))


def mse_grid(x, per_channel, grid, mbits_list, n_bits, sign_bits, mses):
    """The double loop of FP_MSE_Estimator.forward, range_estimators.py:337-347 (mses +=)."""
    dims = list(range(x.dim()))
    if per_channel:
        dims = dims[1:]
    for mi, mb in enumerate(mbits_list):
        mbt = torch.tensor([float(mb)])
        for i in range(grid.shape[0]):
            mv = torch.abs(torch.max(torch.abs(-grid[i]), grid[i]))     # set_quant_range
            if mv.dim() == 0:
                mv = mv.view(1)
            xq = fake_quant(x, n_bits, mv, mbt, sign_bits)
            mses[mi, i, :] += ((x - xq) ** 2).mean(dims)
    return mses


def mse_search_grid(x, per_channel):
    """range_estimators.py:295-309: 111 candidates from 0.1*mx to 1.2*mx per channel -> [111, C]."""
    f = x.view(x.shape[0], -1) if per_channel else x.view(1, -1)
    cols = []
    for row in f:
        mx = torch.max(torch.abs(row.min()), torch.abs(row.max()))
        cols.append(torch.linspace(0.1 * mx.item(), 1.2 * mx.item(), 111))
    return torch.stack(cols).transpose(0, 1)


def mse_select(mses, grid, mbits_list):
    """range_estimators.py:350-362: plurality vote on mantissa bits, per-channel argmin maxval."""
    best_m_per_ch = mses.min(1)[0].argmin(0)
    mi = int(torch.mode(best_m_per_ch).values.item())
    idx = mses[mi].argmin(0)
    maxval = torch.stack([grid[idx[c], c] for c in range(grid.shape[1])])
    return float(mbits_list[mi]), maxval, idx
