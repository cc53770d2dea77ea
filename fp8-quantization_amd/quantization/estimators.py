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
from enum import auto

from .registry import BaseEnumOptions, ClassEnumOptions, MethodMap


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
        self.dist_group = None    # set by fp8q.dist.enable_distributed_calibration: data-parallel calibration

    def forward(self, x):
        raise NotImplementedError()

    def _dist_active(self):
        """Data-parallel calibration is on (fp8q.dist.enable_distributed_calibration) and there is more than one rank."""
        if self.dist_group is None or self.per_channel:
            return False
        from fp8q import dist as _fd
        return _fd._multi(self._group())

    def _group(self):
        return None if self.dist_group is True else self.dist_group

    def _packed(self, device):
        """The operand of the range all-reduce, or None.  Data-parallel calibration (batch sharded over the ranks of
        `dist_group`): the min/max kernel writes {-min, max, nan flags} of the folded estimate next to the estimate
        itself, the ranks all-reduce(MAX) those 16 bytes, one tiny kernel unpacks them (+ K5) -- every rank then holds
        what ONE process would have computed on the concatenated batch.  Exact for all three folds: min / max commute
        with the union, and the EMA (1-m)*new + m*cur is monotone in `new` with the same `cur` on every rank, so the
        max over ranks of the folded value equals the fold of the max over ranks."""
        return _ops.new_packed(1, device) if self._dist_active() else None

    def _exchange(self, packed, mn, mx, mv):
        """all-reduce(MAX) of `packed` and unpack into mn / mx / mv (in place; [1] tensors)."""
        import torch.distributed as dist
        dist.all_reduce(packed, op=dist.ReduceOp.MAX, group=self._group())
        _ops.ranges_unpack(packed, mn, mx, mv)

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
        packed = self._packed(x.device)
        if packed is None:
            mn, mx, mv = _ops.minmax(x, self.per_channel, cur_min, cur_max, mode=self._fold_mode,
                                     momentum=self.momentum, want_maxval=True)
        else:
            mn, mx, mv = _ops.minmax(x, self.per_channel, cur_min, cur_max, mode=self._fold_mode,
                                     momentum=self.momentum, want_maxval=True, packed=packed)
            self._exchange(packed, mn, mx, mv)      # global range before the batch is quantized with it
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
            # (in float64, as numpy's percentile interpolates: with a float32 `q` the position 0.999 * (n - 1) is already off by
            # 1e-5 of a step; the result is narrowed to float32 -- the reference keeps numpy's float64, which would drag the whole
            # quantizer into float64)
            f = x.reshape(x.shape[0], -1) if self.per_channel else x.reshape(-1)
            q = torch.tensor([self.percentile / 100.0, 1 - self.percentile / 100.0], device=x.device, dtype=torch.float64)
            lo, hi = torch.quantile(f.double(), q, dim=-1).float()
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


def _fma_f32(a, b, c):
    """fl32(a * b + c) with ONE rounding, for float32 arrays.  The product of two floats is exact in double; the double
    sum t is rounded once more when it is narrowed to float32, which goes wrong only if t sits exactly on a float32
    rounding midpoint (low 29 fraction bits == 0x10000000) while the exact sum does not -- those (rare) elements are
    redone with the sum rounded to ODD (exact residual by TwoSum), after which the narrowing rounds correctly."""
    p = a.astype(np.float64) * b.astype(np.float64)
    c = np.broadcast_to(c.astype(np.float64), p.shape)
    t = p + c
    risky = (t.view(np.int64) & 0x1FFFFFFF) == 0x10000000
    if risky.any():
        pr, cr, tr = p[risky], c[risky], t[risky]
        bb = tr - pr
        err = (pr - (tr - bb)) + (cr - bb)                  # exact: p + c == t + err
        other = np.nextafter(tr, np.where(err > 0, np.inf, -np.inf))
        t = t.copy()
        t[risky] = np.where(err == 0, tr, other)            # t is even here (midpoint pattern): the odd neighbour on err's side
    return t.astype(np.float32)


def linspace_columns(mx_host, steps):
    """[C, steps] float32: row c == torch.linspace(0.1 * m_c, 1.2 * m_c, steps) (python-float products, as the reference
    computes its MSE search grid per channel, range_estimators.py:296-305) WITHOUT one torch call per channel (15 us
    each: 0.3 s of host time for MobileNetV2's 18 119 channels).  ATen's CPU kernel for fewer steps than its parallel
    grain evaluates element i as fl32(start + step * i) for i < steps // 2 and fl32(end - step * (steps - 1 - i)) after,
    each with a fused multiply-add; that formula is reproduced here and CHECKED against torch.linspace itself on a few
    channels of every call -- on any difference (another ATen build, a CPU without FMA) the per-channel loop is used."""
    m = np.asarray(mx_host, np.float64).reshape(-1)
    if m.size == 0:
        return torch.empty((0, steps), dtype=torch.float32)
    with np.errstate(all="ignore"):
        start, end = (0.1 * m).astype(np.float32), (1.2 * m).astype(np.float32)
        step = ((end - start) / np.float32(steps - 1)).astype(np.float32)
        i = np.arange(steps, dtype=np.float32)
        half = steps // 2
        lo = _fma_f32(step[:, None], i[None, :half], start[:, None])
        hi = _fma_f32(-step[:, None], (np.float32(steps - 1) - i[None, half:]), end[:, None])
    cols = np.concatenate([lo, hi], 1)
    probe = sorted({0, len(m) // 2, len(m) - 1, int(np.argmax(m)), int(np.argmin(m))})
    try:
        ok = np.isfinite(m).all() and np.isfinite(cols).all() and steps >= 2 and all(
            np.array_equal(torch.linspace(0.1 * float(m[c]), 1.2 * float(m[c]), steps).numpy().view(np.int32),
                           cols[c].view(np.int32)) for c in probe)
    except Exception:       # e.g. a range that overflows float32: let the per-channel loop raise what the reference raises
        ok = False
    if not ok:
        return torch.stack([torch.linspace(0.1 * v, 1.2 * v, steps) for v in m.tolist()])
    return torch.from_numpy(cols)


class FP_MSE_Estimator(RangeEstimatorBase):
    """Grid search over 111 clipping values (x 1..n mantissa widths) minimising the MSE.

    The double Python loop of the reference (range_estimators.py:337-347: 111*|m| full quantizer
    passes, each ~16 kernel launches) is ONE pass over x (fp8q_mse_grid_f32), and the three host round trips of the
    reference (:305 mx.item() for the search grid, :353 the mantissa vote's .item(), :360 one gather per channel)
    are device kernels (fp8q_mse_linspace_f32, fp8q_mse_select_f32): a call enqueues work and returns.  The voted
    mantissa width reaches the quantizer as a device scalar (FPQuantizer.mantissa_bits keeps it pending; the batch
    that follows in the same forward is quantized by fp8q_quantize_dm_f32) and comes to the host when somebody reads
    it -- QuantizedModel.fix_ranges() collects all of a model's in one copy.
    allow_unsigned=True: the sign is a device flag as well (_forward_sign_on_device; float64 data and multi-process
    calibration keep the reference's host decision).
    """
    N_GRID = 111   # hard-coded in the reference (:305); `num_candidates` is accepted and ignored

    def __init__(self, num_candidates=100, opt_method=None, range_margin=0.5, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.num_candidates = num_candidates
        self.mses = self.search_grid = None
        self._mbit_list = None
        self.__dict__["_cal"] = None

    def reset(self):
        super().reset()
        self.mses = self.search_grid = None
        self._mbit_list = None
        self.__dict__["_cal"] = None

    # ---- the one-call calibration step (fp8q_mse_calibrate_f32) ---------------------------------------------------------
    def one_call_ok(self, x):
        """True when estimate + set_quant_range + quantize of this batch can run as ONE library call with the same
        results as the protocol calls: plain FP8 quantizer that takes its range from us (set_maxval), signed-ness fixed,
        float32 data in the layout the kernels read, single process, nothing learned."""
        q = self.quantizer
        return (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.numel() > 0 and x.dim() > 0
                and q.set_maxval and not q.allow_unsigned and self.dist_group is None
                and not q.__dict__["_parameters"]                      # maxval / mantissa_bits being learned
                and (not self.per_channel or x.shape[0] <= 65535)
                and not (x.requires_grad and torch.is_grad_enabled()))

    def calibrate_quantize(self, x, pre=None):
        """QuantizationManager.forward for this estimator (quantization_manager.py:114-122): update the MSE tables with x,
        choose (mantissa width, maxval), hand both to the quantizer and quantize x with them -- one ctypes call, four to
        eleven kernel launches, no host round trip.  The estimator's observable state (`search_grid`, `mses`,
        `last_maxval`) and the quantizer's (`maxval`, `mantissa_bits`, `_range_epoch`) end up as after
        `xmin, xmax = est(x); q.set_quant_range(xmin, xmax); q(x)`; tensors are views of one per-estimator block that
        the next batch updates in place."""
        q = self.quantizer
        qd = q.__dict__
        C = x.shape[0] if self.per_channel else 1
        cal = self.__dict__.get("_cal")
        if cal is None or cal.C != C or cal.sign_bits != q.sign_bits or cal.n_bits != q.n_bits or cal.mses is not self.mses:
            if q.mse_include_mantissa_bits:
                mbit_list = [float(m) for m in range(1, q.n_bits - q.sign_bits)]
            else:
                mbit_list = [float(q.mantissa_bits)]
            if self.mses is not None and len(mbit_list) != self.mses.shape[0]:
                mbit_list = self._mbit_list
            self._mbit_list = mbit_list
            if self.mses is not None and (self.mses.dtype != torch.float32 or self.mses.device != x.device
                                          or self.mses.shape[2] != C):
                return None                         # tables of another kind (float64 data, another device): protocol calls
            cal = _ops.MseCalibration(C, x.device, mbit_list, q.n_bits, q.sign_bits, self.N_GRID, self.search_grid, self.mses)
            self.__dict__["_cal"] = cal
            self.search_grid, self.mses = cal.grid, cal.mses
        y = cal.step(x, pre=pre)
        # what the protocol calls would leave behind (estimators.forward + set_quant_range), without nn.Module.__setattr__:
        qd["maxval"] = cal.maxval
        qd["_range_epoch"] = qd.get("_range_epoch", 0) + 1
        if cal.n_m > 1:
            qd["_mbits_dev"], qd["_mbits_host"] = cal.mbits, None       # the vote stays on the device until somebody reads it
        self.__dict__["last_maxval"] = cal.maxval
        return y

    def _dist_batch(self):
        return self._dist_active()

    def _define_search_range(self, x, n_m):
        if self.search_grid is None:
            assert self.mses is None
            if x.dtype == torch.float64:
                # the reference multiplies the FLOAT64 maximum (mx.item(), a python float) by 0.1 / 1.2 and lets
                # torch.linspace narrow the products: fl32(0.1 * mx64), which fl32(0.1 * fl32(mx64)) misses by an ulp half
                # of the time.  float64 data is config 1's side road: build the grid the reference's way, on the host
                mn, hi = _ops.minmax_f64(x, self.per_channel)
                mx64 = torch.max(mn.abs(), hi.abs())
                if self._dist_batch():
                    import torch.distributed as dist
                    dist.all_reduce(mx64, op=dist.ReduceOp.MAX, group=self._group())
                self.search_grid = torch.stack([torch.linspace(0.1 * v, 1.2 * v, self.N_GRID) for v in mx64.cpu().tolist()],
                                               1).contiguous().to(x.device)
                mx = None
            elif not self._dist_batch():
                # max|x| and the grid [111, C] (== torch.linspace per channel) in the abs-max launch itself
                mn_rows, _, _, self.search_grid = _ops.minmax_linspace(x, self.per_channel, self.N_GRID)
                self.__dict__["_batch_min"] = mn_rows        # (allow_unsigned decides the sign from it: forward)
                mx = None
            else:
                _, _, mx = _ops.minmax(x, self.per_channel, want_maxval=True)
            if mx is not None and self._dist_batch():       # batch-sharded: the grid comes from the global max
                import torch.distributed as dist
                dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=self._group())
            if mx is not None:
                self.search_grid = _ops.mse_linspace(mx, self.N_GRID)          # [111, C], == torch.linspace per channel
            self.mses = torch.zeros(n_m, self.N_GRID, self.search_grid.shape[1], device=x.device)
        return self.search_grid, self.mses

    def _accumulate(self, x, grid, mbit_list, q, mses, sign_bits=None):
        """mses += this batch's per-channel mean squared error of every (width, candidate)"""
        sign_bits = q.sign_bits if sign_bits is None else sign_bits
        if x.dtype == torch.float64:        # the reference adds a float64 mean into its float32 table (:346-347)
            inc = torch.zeros(mses.shape, dtype=torch.float64, device=mses.device)
            _ops.mse_grid_f64(x, self.per_channel, grid, mbit_list, q.n_bits, sign_bits, inc, reduce="mean")
            mses.copy_((mses.double() + inc).float())       # float32 += float64: ATen adds in float64, then rounds once
        else:
            _ops.mse_grid(x, self.per_channel, grid, mbit_list, q.n_bits, sign_bits, mses)

    def _sign_on_device(self, x):
        """allow_unsigned without the reference's `int(torch.any(x < 0))` round trip (:333): possible for float32 CUDA data in
        one process when the quantizer takes its range from us"""
        q = self.quantizer
        return (q.allow_unsigned and q.set_maxval and x.is_cuda and x.dtype == torch.float32 and hasattr(_ops, "sign_fold")
                and not self._dist_batch() and type(q).__name__ == "FPQuantizer" and q.n_bits <= 8)

    def _forward_sign_on_device(self, x, q):
        """forward() for allow_unsigned with the sign decided on the GPU.  The reference's flow per batch: sign = any(x < 0);
        one-sided data switches the quantizer to unsigned for good BEFORE the candidates are evaluated (set_quant_range inside
        the loop, fp8_quantizer.py:216-225); the batch's errors are then those of the formats the quantizer has now.  Here the
        decision is a device flag (fp8q_sign_fold_u8 on the batch's row minima: `all(min >= 0)` for `not any(x < 0)`; they
        differ only on NaN data, whose tables are NaN either way).  While the host has not seen the flag, the batch is
        searched with BOTH format sets and the flag picks the increment (twice the K4 time for such a quantizer instead of a
        host round trip per quantizer and batch); once the host knows the quantizer is unsigned only that set runs."""
        d = q.__dict__
        pend = d.get("_sign_host") is None
        if pend and self._mbit_list is not None:
            mbit_list = self._mbit_list
        elif q.mse_include_mantissa_bits:
            mbit_list = [float(m) for m in range(1, q.n_bits - q.sign_bits)]
        else:
            mbit_list = [float(q.mantissa_bits)]
        if self.mses is not None and len(mbit_list) != self.mses.shape[0]:
            mbit_list = self._mbit_list                  # (see forward: the candidate set of the first batch stays)
        self._mbit_list = mbit_list
        self.__dict__["_batch_min"] = None
        grid, mses = self._define_search_range(x, len(mbit_list))
        assert mses.shape[1:] == grid.shape, f"{mses.shape}, {grid.shape}"
        mn_rows = self.__dict__.pop("_batch_min", None)
        if mn_rows is None:
            mn_rows = _ops.minmax(x, self.per_channel)[0]
        neg = _ops.sign_fold(mn_rows.reshape(-1))                      # 1: this batch has an element that is not >= 0
        if d.get("_sign_host") == 0:
            self._accumulate(x, grid, mbit_list, q, mses, sign_bits=0)
        else:
            flag = d.get("_sign_dev")
            if flag is None or flag.device != x.device:
                flag = torch.ones(1, dtype=torch.uint8, device=x.device)
            _ops.sign_fold(mn_rows.reshape(-1), flag)                  # sticky: cleared once a batch is one-sided
            inc_s, inc_u = torch.zeros_like(mses), torch.zeros_like(mses)
            self._accumulate(x, grid, mbit_list, q, inc_s, sign_bits=1)
            self._accumulate(x, grid, mbit_list, q, inc_u, sign_bits=0)
            mses += torch.where(flag.bool(), inc_s, inc_u)
            d["_sign_dev"], d["_sign_host"] = flag, None
            object.__setattr__(q, "_range_epoch", getattr(q, "_range_epoch", 0) + 1)
        mbits_dev, _vote, maxval, xmin = _ops.mse_select(mses, grid, mbit_list, 1)
        xmin = xmin * neg.to(xmin.dtype)                               # sign_bits * -1.0 * maxval (:369); -0.0 for one-sided data
        q.mantissa_bits = mbits_dev if len(mbit_list) > 1 else torch.tensor([float(mbit_list[0])])
        q.maxval = grid[-1].clone()
        self.last_maxval = maxval
        return xmin, maxval

    def forward(self, x):
        q = self.quantizer
        if self._sign_on_device(x):
            return self._forward_sign_on_device(x, q)
        if q.mse_include_mantissa_bits:
            mbit_list = [float(m) for m in range(1, q.n_bits - q.sign_bits)]
        else:
            mbit_list = [float(q.mantissa_bits)]
        if self.mses is not None and len(mbit_list) != self.mses.shape[0]:
            # allow_unsigned flipped sign_bits after the first batch (one-sided data), which changes the number of
            # candidate mantissa widths; the reference indexes past its accumulated [|m|, 111, C] table here
            # (range_estimators.py:337-347: IndexError).  Keep the candidate set of the first batch.
            mbit_list = self._mbit_list
        self._mbit_list = mbit_list
        grid, mses = self._define_search_range(x, len(mbit_list))
        assert mses.shape[1:] == grid.shape, f"{mses.shape}, {grid.shape}"

        sign_bits = int(torch.any(x < 0)) if q.allow_unsigned else 1
        if q.allow_unsigned and self._dist_batch():          # any negative value on any rank
            import torch.distributed as dist
            flag = torch.tensor([sign_bits], dtype=torch.int32, device=x.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self._group())
            sign_bits = int(flag.item())
        # the reference calls set_quant_range(-sign*g, g) per candidate: with allow_unsigned and
        # one-sided data that flips the live quantizer to unsigned before the first evaluation
        if q.allow_unsigned and sign_bits == 0:
            q.sign_bits = 0
        if q.set_maxval and self._dist_batch():
            # data-parallel calibration: this rank's mean squared errors, weighted by its element count, summed
            # over the ranks in float64 (<= 2.7 KB) -> the mean over the concatenated batch
            import torch.distributed as dist
            inc = torch.zeros_like(mses)
            self._accumulate(x, grid, mbit_list, q, inc)
            n_local = float(x.numel())
            packed = torch.cat([inc.double().reshape(-1) * n_local,
                                torch.tensor([n_local], dtype=torch.float64, device=inc.device)])
            dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=self._group())
            mses += (packed[:-1] / packed[-1]).to(mses.dtype).view_as(mses)
        elif q.set_maxval:
            self._accumulate(x, grid, mbit_list, q, mses)
        else:
            # set_maxval=False: set_quant_range is a no-op, every candidate scores the same
            cur = q.maxval.to(x.device).reshape(1, -1).expand(1, grid.shape[1]).contiguous()
            one = torch.zeros(len(mbit_list), 1, grid.shape[1], device=x.device)
            self._accumulate(x, cur, mbit_list, q, one)
            mses += one

        mbits_dev, _vote, maxval, xmin = _ops.mse_select(mses, grid, mbit_list, sign_bits)
        # a single candidate width needs no vote: keep the host value (no pending device scalar, the tuned K1 routes)
        q.mantissa_bits = mbits_dev if len(mbit_list) > 1 else torch.tensor([float(mbit_list[0])])
        if q.set_maxval:
            q.maxval = grid[-1].clone()      # the reference leaves the last candidate in place
        self.last_maxval = maxval            # == |max(|xmin|, maxval)|: what set_quant_range(xmin, maxval) will store
        return xmin, maxval


class OptMethod(BaseEnumOptions):
    grid = auto()
    golden_section = auto()


class LineSearchEstimator(RangeEstimatorBase):
    """1-D grid search of the clipping threshold that minimises the squared quantization error
    (reference range_estimators.py:133-282; used by compute_quant_error.py through
    estimate_range_line_search).  Candidate k of `num_candidates` clips at
    k * (max|x| + range_margin) * expand_range / num_candidates; losses accumulate over calls.

    FP8 quantizers evaluate ALL candidates in one pass over x with the MSE-grid kernel
    (fp8q_mse_grid_f32 with n_cand = num_candidates); INT quantizers, the comparison baseline,
    loop over candidates with elementwise torch ops.  Only the symmetric 1-D search exists: the
    reference takes it for every quantizer that can reach this estimator (`quantizer.symmetric`
    is used without being called there, so it is truthy for FPQuantizer too -- SURVEY.md 3.4).
    Precision follows the data, as in the reference: a float64 sample (compute_quant_error.py:19-20) is searched in
    float64 -- fp8q_minmax_f64 for the search range, fp8q_mse_grid_f64 for the candidates: the reference's arithmetic
    under ATen's type promotion (bias in float32, everything downstream of x in float64), per-candidate sums of squares
    in float64 -- so the chosen candidate is the reference's; float32 data runs the float32 kernels.
    """

    def __init__(self, num_candidates=1000, opt_method=OptMethod.grid, range_margin=0.5, expand_range=10.0,
                 *args, **kwargs):
        super().__init__(*args, **kwargs)
        assert opt_method in OptMethod
        if opt_method != OptMethod.grid:
            # range_estimators.py:186-196 names _golden_section_symmetric / _golden_section_asymmetric for this option,
            # but the reference defines neither (nor _perform_2D_search): it ends in AttributeError on the first batch
            raise NotImplementedError("only the grid search exists (the reference names golden-section methods it never defines)")
        if self.quantizer is None:
            raise NotImplementedError("A Quantizer must be given as an argument to the MSE RangeEstimator")
        self.opt_method = opt_method
        self.num_candidates = num_candidates
        self.expand_range = expand_range
        self.range_margin = range_margin
        self.loss_array = None
        self.max_pos_thr = self.max_neg_thr = self.max_search_range = None
        self.one_sided_dist = None

    @property
    def step_size(self):
        if self.one_sided_dist is None:
            raise NoDataPassedError()
        return self.max_search_range / self.num_candidates

    def reset(self):
        super().reset()
        self.loss_array = None

    def _define_search_range(self, data):
        self.channel_groups = len(data) if self.per_channel else 1
        self.loss_array = np.zeros((self.channel_groups, self.num_candidates + 1))
        self.loss_array[:, 0] = np.inf            # candidate 0 would be an empty range
        mn, mx = _ops.minmax_f64(data, False) if data.dtype == torch.float64 else _ops.minmax(data, False)
        lo, hi = float(mn), float(mx)              # (the reference synchronises here too: float(data.min()))
        if self.one_sided_dist is None:
            self.one_sided_dist = lo >= 0
        self.max_pos_thr = max(abs(lo), hi) + self.range_margin
        self.max_neg_thr = -self.max_pos_thr * self.expand_range
        self.max_search_range = self.max_pos_thr * self.expand_range

    def _candidate_losses(self, data):
        """[channel_groups, num_candidates] sums of squared errors for candidates 1..N."""
        from .fp8 import FPQuantizer
        q = self.quantizer
        n = self.num_candidates
        thr = np.float32(self.step_size * np.arange(1, n + 1))            # what Tensor([x_max]) holds
        C = self.channel_groups
        inner = data.numel() // C
        if isinstance(q, FPQuantizer):
            if not q.set_maxval:
                raise NotImplementedError("line search needs set_maxval=True to change the range")
            grid = torch.from_numpy(thr).to(data.device).view(n, 1).expand(n, C).contiguous()
            sign_bits = 0 if (q.allow_unsigned and self.one_sided_dist) else q.sign_bits
            if data.dtype == torch.float64:      # loss_fx: torch.sum((data - y) ** 2) in float64, per candidate
                sse = torch.zeros(1, n, C, dtype=torch.float64, device=data.device)
                _ops.mse_grid_f64(data, self.per_channel, grid, [float(q.mantissa_bits)], q.n_bits, sign_bits, sse,
                                  reduce="sum")
                return sse[0].transpose(0, 1).cpu().numpy()
            mses = torch.zeros(1, n, C, device=data.device)
            _ops.mse_grid(data, self.per_channel, grid, [float(q.mantissa_bits)], q.n_bits, sign_bits, mses)
            return (mses[0].double() * inner).transpose(0, 1).cpu().numpy()
        import copy
        out = np.zeros((C, n))
        flat = data.view(C, -1)
        for k in range(n):
            tq = copy.deepcopy(q)
            tq.per_channel = False
            tq.set_quant_range(0.0 if self.one_sided_dist else -float(thr[k]), float(thr[k]))
            out[:, k] = ((flat - tq(flat)) ** 2).sum(1).double().cpu().numpy()
        return out

    def forward(self, data):
        data = data.detach()
        if data.dtype not in (torch.float32, torch.float64):
            data = data.float()
        if self.loss_array is None:
            self._define_search_range(data)
        self.loss_array[:, 1:] += self._candidate_losses(data.contiguous())
        best = self.loss_array.argmin(axis=1)
        xmax = (self.step_size * best).astype(np.single)
        xmin = np.zeros(self.channel_groups, np.single) if self.one_sided_dist else (-self.step_size * best).astype(np.single)
        self.current_xmax = torch.tensor(xmax).to(device=data.device)
        self.current_xmin = torch.tensor(xmin).to(device=data.device)
        return self.current_xmin, self.current_xmax

    def extra_repr(self):
        return f"opt_method={self.opt_method.name} ,num_candidates={self.num_candidates}"


def estimate_range_line_search(W, quant, num_candidates=None):
    kw = {} if num_candidates is None else dict(num_candidates=num_candidates)
    return LineSearchEstimator(quantizer=quant, **kw).forward(W)


class RangeEstimators(ClassEnumOptions):
    current_minmax = MethodMap(CurrentMinMaxEstimator)
    allminmax = MethodMap(AllMinMaxEstimator)
    running_minmax = MethodMap(RunningMinMaxEstimator)
    MSE = MethodMap(FP_MSE_Estimator)
