"""QuantizationManager: range-estimation state machine in front of a quantizer.

API of /root/reference/quantization/quantization_manager.py:28-135 (constructor arguments,
`state`, `estimate_ranges/fix_ranges/learn_ranges/reset_ranges`, `forward`).  `forward` keeps
the reference order -- update the range from the CURRENT batch, then quantize that batch with
it (:114-122) -- but picks the cheapest kernel sequence for it:

  estimate, per-channel current_minmax, set_maxval  -> 1 launch  (fp8q_minmax_quantize_f32)
  estimate, any min/max estimator                    -> 2-3 launches (minmax [+final], quantize)
  fixed ranges                                       -> 1 launch  (fp8q_quantize_f32)
  anything else (custom estimator / quantizer)       -> the generic protocol calls
"""
from enum import auto

import torch
from torch import nn

from fp8q import ops as _ops
from .registry import BaseEnumOptions, ClassEnumOptions, MethodMap
from .fp8 import FPQuantizer, QuantizerBase, QuantizerNotInitializedError
from .uniform import SymmetricUniformQuantizer, AsymmetricUniformQuantizer
from .estimators import (RangeEstimators, RangeEstimatorBase, CurrentMinMaxEstimator,
                         AllMinMaxEstimator, RunningMinMaxEstimator, FP_MSE_Estimator)


class QMethods(ClassEnumOptions):
    symmetric_uniform = MethodMap(SymmetricUniformQuantizer)
    asymmetric_uniform = MethodMap(AsymmetricUniformQuantizer)
    fp_quantizer = MethodMap(FPQuantizer)


class Qstates(BaseEnumOptions):
    estimate_ranges = auto()         # ranges follow the data in train and eval mode
    fix_ranges = auto()              # ranges frozen
    learn_ranges = auto()            # range parameters are nn.Parameters
    estimate_ranges_train = auto()   # follow the data in train mode, frozen in eval mode


_MINMAX = (CurrentMinMaxEstimator, AllMinMaxEstimator, RunningMinMaxEstimator)


class QuantizationManager(nn.Module):
    def __init__(self, qmethod=QMethods.fp_quantizer.cls, init=RangeEstimators.current_minmax.cls,
                 per_channel=False, x_min=None, x_max=None, qparams=None, range_estim_params=None):
        super().__init__()
        self.state = Qstates.estimate_ranges
        self.qmethod = qmethod
        self.init = init
        self.per_channel = per_channel
        self.qparams = qparams if qparams else {}
        self.range_estim_params = range_estim_params if range_estim_params else {}
        self.range_estimator = None

        self.quantizer = self.qmethod(per_channel=self.per_channel, **self.qparams)
        self.quantizer.state = self.state
        if x_min is not None and x_max is not None:
            self.set_quant_range(x_min, x_max)
            self.fix_ranges()
        else:
            self.range_estimator = self.init(per_channel=self.per_channel, quantizer=self.quantizer,
                                             **self.range_estim_params)

    @property
    def n_bits(self):
        return self.quantizer.n_bits

    def _set_state(self, state):
        self.state = state
        self.quantizer.state = state

    def estimate_ranges(self):
        self._set_state(Qstates.estimate_ranges)

    def fix_ranges(self):
        if not self.quantizer.is_initialized:
            raise QuantizerNotInitializedError()
        self._set_state(Qstates.fix_ranges)

    def learn_ranges(self):
        self.quantizer.make_range_trainable()
        self._set_state(Qstates.learn_ranges)

    def estimate_ranges_train(self):
        self._set_state(Qstates.estimate_ranges_train)

    def reset_ranges(self):
        self.range_estimator.reset()
        self.quantizer.reset()
        self.estimate_ranges()

    def set_quant_range(self, x_min, x_max):
        self.quantizer.set_quant_range(x_min, x_max)

    def _estimating(self):
        return self.state == Qstates.estimate_ranges or (
            self.state == Qstates.estimate_ranges_train and self.training)

    def forward(self, x):
        q, est = self.quantizer, self.range_estimator
        if not self._estimating():
            return q(x)
        fast = (type(q) is FPQuantizer and type(est) in _MINMAX and not q.allow_unsigned
                and not getattr(est, "percentile", None) and x.is_cuda
                and not (x.requires_grad and torch.is_grad_enabled()))   # weights are Parameters: fine under no_grad
        if type(est) is FP_MSE_Estimator and type(q) is FPQuantizer and est.one_call_ok(x):
            y = est.calibrate_quantize(x)            # estimate + set_quant_range + quantize: one library call
            if y is not None:
                return y
        if not fast:
            xmin, xmax = est(x)                      # generic protocol, reference order
            if (type(q) is FPQuantizer and type(est) is FP_MSE_Estimator and q.set_maxval
                    and est.last_maxval is not None):
                # set_quant_range(xmin, xmax) would store |max(|xmin|, xmax)| == xmax (three tiny launches): the
                # estimator's select kernel already wrote it (allow_unsigned: the estimator has set sign_bits itself --
                # xmin = -sign_bits * maxval says nothing new)
                q._set_maxval_tensor(est.last_maxval)
            else:
                self.set_quant_range(xmin, xmax)
            return q(x)
        if not q.set_maxval:
            est(x)                                   # estimate is tracked, the format's maxval stays
            return q(x)
        inner = x.numel() // max(x.shape[0], 1) if x.dim() > 0 else 1
        if (type(est) is CurrentMinMaxEstimator and self.per_channel and x.dim() > 0
                and 0 < inner <= _ops.fused_max_inner()):
            y, mn, mx, mv = _ops.minmax_quantize(x, float(q.mantissa_bits), q.n_bits, q.sign_bits)
            est.current_xmin, est.current_xmax, est.last_maxval = mn, mx, mv
            q._set_maxval_tensor(mv)
            return y
        est(x)
        q._set_maxval_tensor(est.last_maxval)        # == |max(|xmin|, xmax)| (fp8_quantizer.py:236)
        return q(x)

    # ---- N2: producer epilogue fused into the quantizer ----------------------------
    def can_fuse(self, x):
        """True when quantize(act(bn(x) + residual)) can run as ONE kernel with the same result as the
        unfused chain: FP8 per-tensor quantizer on a contiguous CUDA fp32 [N, C, ...] tensor, ranges
        fixed or followed by a plain min/max estimator."""
        import os
        q, est = self.quantizer, self.range_estimator
        if os.environ.get("FP8Q_FUSE_EPILOGUE", "1") == "0" or type(q) is not FPQuantizer or self.per_channel:
            return False
        if not (x.is_cuda and x.dtype.is_floating_point and x.dim() >= 2 and not x.requires_grad):
            return False
        if x.dtype != __import__("torch").float32 or not x.is_contiguous() or not _ops.affine_act_supported(x):
            return False
        if self._estimating():
            if type(est) is FP_MSE_Estimator:      # act(bn(x) + residual) in one pass, then the one-call search on it
                return est.one_call_ok(x)
            return (type(est) in _MINMAX and not q.allow_unsigned and not getattr(est, "percentile", None))
        return True

    def forward_fused(self, x, bn=None, residual=None, act=0, bn_ab=None):
        """quantize(act(bn(x) + residual)); range update first when estimating (reference order)."""
        q, est = self.quantizer, self.range_estimator
        if self._estimating() and type(est) is FP_MSE_Estimator:
            # the search needs the tensor the quantizer will see: the epilogue with the quantizer switched off (8 B / element
            # instead of torch's batch_norm + activation passes; the first batch's abs-max and search grid come out of the same
            # launch), then search + set_quant_range + quantize -- all inside ONE library call
            ab = bn_ab if bn is not None else None
            if bn is not None and ab is None:
                ab = _ops.bn_fold(bn)
            if ab is None and residual is None and not act:
                return self.forward(x)
            y = est.calibrate_quantize(x, pre=(ab, residual, act)) if est.one_call_ok(x) else None
            if y is not None:
                return y
            return self.forward(_ops.affine_act(x, ab, residual, act))
        if self._estimating():
            cur_min, cur_max = est.current_xmin, est.current_xmax
            if cur_min is not None and est._fold_mode != _ops.FOLD_CURRENT:
                cur_min, cur_max = cur_min.reshape(-1), cur_max.reshape(-1)
            else:
                cur_min = cur_max = None
            packed = est._packed(x.device)
            if packed is None:
                mn, mx, mv = _ops.affine_act_minmax(x, cur_min, cur_max, mode=est._fold_mode, momentum=est.momentum,
                                                    bn=bn, residual=residual, act=act)
            else:                                    # data-parallel calibration: global range before quantizing
                mn, mx, mv = _ops.affine_act_minmax(x, cur_min, cur_max, mode=est._fold_mode, momentum=est.momentum,
                                                    bn=bn, residual=residual, act=act, packed=packed)
                est._exchange(packed, mn, mx, mv)
            est.current_xmin, est.current_xmax, est.last_maxval = mn.reshape(()), mx.reshape(()), mv
            if q.set_maxval:
                q._set_maxval_tensor(est.last_maxval)
        if q.maxval.device != x.device:
            q.maxval = q.maxval.to(x.device)
        # fixed ranges: the quantizer's constants and table are prepared once (small activations are latency-bound)
        prep = q._prepared() if not self._estimating() and hasattr(_ops, "quantizer_prepare") else None
        return _ops.affine_act_quantize(x, q.maxval, float(q.mantissa_bits), q.n_bits, q.sign_bits, bn=bn,
                                        residual=residual, act=act, bn_ab=bn_ab if bn is not None else None, prep=prep)

    def extra_repr(self):
        return f"state={self.state.name}"
