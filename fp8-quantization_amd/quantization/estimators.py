"""Range estimators on the MI355X engine.

Same registry names, constructor arguments, buffers (`current_xmin`, `current_xmax`) and
`forward(x) -> (xmin, xmax)` protocol as /root/reference/quantization/range_estimators.py:
  CurrentMinMaxEstimator :56-76   AllMinMaxEstimator :79-100   RunningMinMaxEstimator :103-125
  FP_MSE_Estimator :285-369 (CLI name "MSE")                   RangeEstimators :389-393
Each forward is one two-output reduction kernel (fp8q_minmax_f32) that also folds the batch
estimate into the running one on the device -- no host synchronisation.
"""
import numpy as np
import torch
from torch import nn

from fp8q import ops as _ops
from .registry import ClassEnumOptions, MethodMap


class NoDataPassedError(Exception):
    def __init__(self):
        super().__init__("Data must be pass through the range estimator to be initialized")


class RangeEstimatorBase(nn.Module):
    def __init__(self, per_channel=False, quantizer=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.register_buffer("current_xmin", None)
        self.register_buffer("current_xmax", None)
        self.per_channel = per_channel
        self.quantizer = quantizer
        self.last_maxval = None   # |max(|xmin|, xmax)| of the latest estimate (device tensor)

    def forward(self, x):
        raise NotImplementedError()

    def reset(self):
        self.current_xmin = None
        self.current_xmax = None
        self.last_maxval = None

    def __repr__(self):
        # the attached quantizer is deliberately left out of the printout
        lines = self.extra_repr().split("\n")
        extra = lines[0] if len(lines) == 1 else "\n  " + "\n  ".join(lines) + "\n"
        return f"{self._get_name()}({extra})"

    # shared implementation: one kernel launch, fold mode chosen by the subclass
    _fold_mode = _ops.FOLD_CURRENT
    momentum = 0.9

    def _update(self, x):
        cur_min, cur_max = self.current_xmin, self.current_xmax
        shaped = cur_min is not None and cur_min.dim() == 0
        if cur_min is not None:
            cur_min, cur_max = cur_min.reshape(-1), cur_max.reshape(-1)
            if self._fold_mode == _ops.FOLD_CURRENT:
                cur_min = cur_max = None   # overwritten anyway; keeps buffers of old shapes out
        mn, mx, mv = _ops.minmax(x, self.per_channel, cur_min, cur_max, mode=self._fold_mode,
                                 momentum=self.momentum, want_maxval=True)
        if not self.per_channel:      # reference returns 0-dim tensors for per-tensor ranges
            mn, mx = mn.reshape(()), mx.reshape(())
        self.current_xmin, self.current_xmax, self.last_maxval = mn, mx, mv
        return self.current_xmin, self.current_xmax


class CurrentMinMaxEstimator(RangeEstimatorBase):
    _fold_mode = _ops.FOLD_CURRENT

    def __init__(self, percentile=None, *args, **kwargs):
        self.percentile = percentile
        super().__init__(*args, **kwargs)

    def forward(self, x):
        if self.percentile:
            # unreachable from the reference CLI (hijacker.py:57 compares a class with an enum
            # member); kept for API completeness through torch.quantile on the device
            f = x.reshape(x.shape[0], -1) if self.per_channel else x.reshape(-1)
            q = torch.tensor([self.percentile / 100.0, 1 - self.percentile / 100.0], device=x.device,
                             dtype=torch.float32)
            lo, hi = torch.quantile(f.float(), q, dim=-1)
            self.current_xmin, self.current_xmax = lo, hi
            self.last_maxval = None
            return lo, hi
        return self._update(x)


class AllMinMaxEstimator(RangeEstimatorBase):
    _fold_mode = _ops.FOLD_ALL

    def forward(self, x):
        return self._update(x)


class RunningMinMaxEstimator(RangeEstimatorBase):
    _fold_mode = _ops.FOLD_RUNNING

    def __init__(self, momentum=0.9, *args, **kwargs):
        self.momentum = momentum
        super().__init__(*args, **kwargs)

    def forward(self, x):
        return self._update(x)


class FP_MSE_Estimator(RangeEstimatorBase):
    """Grid search over 111 clipping values (x 1..n mantissa widths) minimising the MSE.

    The double Python loop of the reference (range_estimators.py:337-347: 111*|m| full quantizer
    passes, each ~16 kernel launches) is ONE pass over x (fp8q_mse_grid_f32).  Host round trips:
    one on the first batch to lay out the search grid exactly as the reference does (python
    floats -> torch.linspace), one per call for the plurality vote on mantissa bits (the reference
    synchronises there too, :353).
    """
    N_GRID = 111   # hard-coded in the reference (:305); `num_candidates` is accepted and ignored

    def __init__(self, num_candidates=100, opt_method=None, range_margin=0.5, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.num_candidates = num_candidates
        self.mses = self.search_grid = None

    def reset(self):
        super().reset()
        self.mses = self.search_grid = None

    def _define_search_range(self, x, n_m):
        if self.search_grid is None:
            assert self.mses is None
            _, _, mx = _ops.minmax(x, self.per_channel, want_maxval=True)
            mx_host = mx.detach().cpu().tolist()            # one sync, first batch only
            cols = [torch.linspace(0.1 * m, 1.2 * m, self.N_GRID) for m in mx_host]
            self.search_grid = torch.stack(cols).to(x.device).transpose(0, 1).contiguous()  # [111, C]
            self.mses = torch.zeros(n_m, self.N_GRID, len(cols), device=x.device)
        return self.search_grid, self.mses

    def forward(self, x):
        q = self.quantizer
        mbit_list = [float(q.mantissa_bits)]
        if q.mse_include_mantissa_bits:
            mbit_list = [float(m) for m in range(1, q.n_bits - q.sign_bits)]
        grid, mses = self._define_search_range(x, len(mbit_list))
        assert mses.shape[1:] == grid.shape, f"{mses.shape}, {grid.shape}"

        sign_bits = int(torch.any(x < 0)) if q.allow_unsigned else 1
        # the reference calls set_quant_range(-sign*g, g) per candidate: with allow_unsigned and
        # one-sided data that flips the live quantizer to unsigned before the first evaluation
        if q.allow_unsigned and sign_bits == 0:
            q.sign_bits = 0
        if q.set_maxval:
            _ops.mse_grid(x, self.per_channel, grid, mbit_list, q.n_bits, q.sign_bits, mses)
        else:
            # set_maxval=False: set_quant_range is a no-op, every candidate scores the same
            cur = q.maxval.to(x.device).reshape(1, -1).expand(1, grid.shape[1]).contiguous()
            one = torch.zeros(len(mbit_list), 1, grid.shape[1], device=x.device)
            _ops.mse_grid(x, self.per_channel, cur, mbit_list, q.n_bits, q.sign_bits, one)
            mses += one

        best_m_per_ch = mses.min(1)[0].argmin(0)
        best_idx = int(torch.mode(best_m_per_ch).values.item())
        best_mbits = float(mbit_list[best_idx])
        arg = mses[best_idx].argmin(0)                                  # [C]
        maxval = grid.gather(0, arg.unsqueeze(0)).squeeze(0)            # grid[arg[c], c]
        q.mantissa_bits = torch.tensor(best_mbits)
        if q.set_maxval:
            q.maxval = grid[-1].clone()      # the reference leaves the last candidate in place
        self.last_maxval = None
        return sign_bits * -1.0 * maxval, maxval


def estimate_range_line_search(W, quant, num_candidates=None):
    raise NotImplementedError(
        "LineSearchEstimator (range_estimators.py:133-282, compute_quant_error.py only) is a "
        "'next' row of SURVEY.md 8(f) and is not part of the GPU hot path yet")


class RangeEstimators(ClassEnumOptions):
    current_minmax = MethodMap(CurrentMinMaxEstimator)
    allminmax = MethodMap(AllMinMaxEstimator)
    running_minmax = MethodMap(RunningMinMaxEstimator)
    MSE = MethodMap(FP_MSE_Estimator)
