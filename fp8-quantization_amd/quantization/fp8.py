"""FP8 quantizer on the MI355X engine.

`quantize_to_fp8_ste_MM` and `FPQuantizer` keep the names, arguments and observable state of
/root/reference/quantization/quantizers/fp8_quantizer.py:91-133 and :151-272; the arithmetic is
one HIP kernel launch (fp8q.ops.quantize -> fp8q_quantize_f32) instead of 13 eager ATen ops.
There is no CPU path: tensors must live on the GPU (fp8q raises otherwise).
"""
import numpy as np
import torch
from torch import nn

from fp8q import ops as _ops


class QuantizerNotInitializedError(Exception):
    """Raised when fix_ranges() is requested for a quantizer that has no range yet
    (reference: quantization/quantizers/utils.py:6-12)."""

    def __init__(self):
        super().__init__("Quantizer has  not been initialized yet")


class _RoundSTE(torch.autograd.Function):
    """round-half-even forward, identity backward (reference rounding_utils.py:12-19)."""

    @staticmethod
    def forward(ctx, x):
        return torch.round(x)

    @staticmethod
    def backward(ctx, grad):
        return grad


round_ste_func = _RoundSTE.apply


class _FakeQuantSTE(torch.autograd.Function):
    """HIP forward; the backward the reference's autograd chain yields (fp8_quantizer.py:105-133):

      d/dx       straight-through inside the clamp range, 0 outside (torch.min / torch.max route the gradient to the
                 bound there; an element exactly ON a bound splits it half / half, as ATen does for ties);
      d/dmaxval  +-1 per element clipped at +-maxval (minval = -maxval is tied to it for signed formats), plus the
                 scale term: y = round_ste(xc / s) * s gives dy/ds = round(xc / s) - xc / s, and s = 2^(p - M - bias)
                 with bias = 2^E - log2(maxval) + const (the floor(log2|xc| + bias) part is detached) gives
                 ds/dmaxval = s / maxval, i.e. (y - xc) / maxval per element; reduced to maxval's shape.

      d/dmbits   (learn_mantissa_bits: the width as an nn.Parameter, :105-110)  M = clamp(round_ste(mbits), 1, n_bits -
                 sign_bits) enters through E = n_bits - sign_bits - M in the bias and through the scale exponent:
                 s = 2^(p - M - bias(M)), bias'(M) = -ln2 2^E + 2^-M / (2 - 2^-M), so dy/dM = (y - xc) ln2 (-1 - bias'(M))
                 per element, summed; round_ste passes the gradient, the clamp cuts it off outside [1, n_bits - sign_bits].

    The backward runs as a handful of torch ops (PTQ, the path this engine accelerates, never calls it).  The kernel takes
    the mantissa width by value (float(mbits): a host round trip per forward when the Parameter lives on the GPU -- QAT
    territory, outside the accelerated path)."""

    @staticmethod
    def forward(ctx, x, maxval, mbits, n_bits, sign_bits):
        mb_val = _host_float(mbits)
        y = _ops.quantize(x.detach(), maxval.detach(), mb_val, n_bits, sign_bits)
        ctx.save_for_backward(x, maxval, y)
        ctx.sign_bits, ctx.n_bits, ctx.mb_val = sign_bits, n_bits, mb_val
        ctx.mbits_like = mbits if isinstance(mbits, torch.Tensor) else None
        return y

    @staticmethod
    def backward(ctx, grad):
        x, maxval, y = ctx.saved_tensors
        mv = maxval.view([-1] + [1] * (x.dim() - 1)) if maxval.numel() != 1 else maxval
        lo = -mv if ctx.sign_bits == 1 else torch.zeros_like(mv)
        at_hi, at_lo = (x == mv), (x == lo)
        w_x = ((x > lo) & (x < mv)).to(grad.dtype) + 0.5 * (at_hi | at_lo).to(grad.dtype)
        grad_x = grad * w_x if ctx.needs_input_grad[0] else None
        grad_mv = None
        if ctx.needs_input_grad[1]:
            xc = torch.min(torch.max(x, lo), mv)
            w = (y - xc) / mv + (x > mv).to(grad.dtype) + 0.5 * at_hi.to(grad.dtype)
            if ctx.sign_bits == 1:
                w = w - (x < lo).to(grad.dtype) - 0.5 * at_lo.to(grad.dtype)
            g = grad * w
            grad_mv = g.sum().reshape(maxval.shape) if maxval.numel() == 1 else \
                g.reshape(maxval.numel(), -1).sum(1).reshape(maxval.shape)
        grad_mb = None
        if ctx.mbits_like is not None and ctx.needs_input_grad[2]:
            hi = ctx.n_bits - ctx.sign_bits
            r = float(np.float32(ctx.mb_val).round())          # torch.round: half to even, as np.round
            if 1.0 <= r <= hi:
                M = r
                E = hi - M
                dbias = -np.log(2.0) * 2.0 ** E + 2.0 ** -M / (2.0 - 2.0 ** -M)
                xc = torch.min(torch.max(x, lo), mv)
                grad_mb = ((grad * (y - xc)).sum() * (np.log(2.0) * (-1.0 - dbias))).reshape(ctx.mbits_like.shape).to(device=ctx.mbits_like.device, dtype=ctx.mbits_like.dtype)
            else:
                grad_mb = torch.zeros_like(ctx.mbits_like)
        return grad_x, grad_mv, grad_mb, None, None


def _host_float(t):
    """python float of a 1-element tensor / number (mantissa bits are host-side scalars)."""
    if isinstance(t, torch.Tensor):
        return float(t.detach().reshape(-1)[0].item()) if t.numel() == 1 else float(t)
    return float(t)


def quantize_to_fp8_ste_MM(x_float, n_bits, maxval, num_mantissa_bits, sign_bits):
    """Same call signature as fp8_quantizer.py:91-97.  maxval: tensor [1] or [C] on x's device.  num_mantissa_bits: a
    number / host tensor (passed to the kernel by value) or a 1-element tensor on x's GPU (read by the kernel).
    sign_bits: 0 / 1, or a 1-element uint8 tensor on x's GPU (FPQuantizer's pending flag: read by the kernel)."""
    mb_grad = isinstance(num_mantissa_bits, torch.Tensor) and num_mantissa_bits.requires_grad and torch.is_grad_enabled()
    on_device = (isinstance(num_mantissa_bits, torch.Tensor) and num_mantissa_bits.is_cuda and num_mantissa_bits.numel() == 1
                 and x_float.dtype == torch.float32 and not mb_grad
                 and not (torch.is_grad_enabled() and x_float.requires_grad))
    mbits = num_mantissa_bits.detach().reshape(1).float() if on_device else _host_float(num_mantissa_bits)
    if not isinstance(maxval, torch.Tensor):
        maxval = torch.tensor([float(maxval)], dtype=torch.float32)
    maxval = maxval.to(device=x_float.device, dtype=torch.float32).reshape(-1)
    if torch.is_grad_enabled() and (x_float.requires_grad or maxval.requires_grad or mb_grad):
        return _FakeQuantSTE.apply(x_float, maxval, num_mantissa_bits if mb_grad else mbits, int(n_bits), int(sign_bits))
    if isinstance(sign_bits, torch.Tensor):
        # FPQuantizer's pending device flag (allow_unsigned, not yet read by the host): the kernel reads it
        if not (sign_bits.is_cuda and sign_bits.dtype == torch.uint8 and x_float.dtype == torch.float32 and x_float.is_cuda):
            sign_bits = int(sign_bits)
        return _ops.quantize(x_float, maxval.detach(), mbits, int(n_bits), sign_bits)
    return _ops.quantize(x_float, maxval.detach(), mbits, int(n_bits), int(sign_bits))


# ---- format enumeration (independent definition of the grid, fp8_quantizer.py:13-50) ------------
def generate_all_values_fp(num_total_bits=8, num_exponent_bits=4, bias=8):
    """Every value representable with 1 sign bit, `num_exponent_bits` exponent bits and the rest
    fraction bits; exponent code 0 is subnormal, the top exponent code is an ordinary binade."""
    fbits = num_total_bits - 1 - num_exponent_bits
    e = np.arange(2 ** num_exponent_bits, dtype=np.float64)[:, None]
    f = np.arange(2 ** fbits, dtype=np.float64)[None, :] / 2.0 ** fbits
    sub = (e == 0).astype(np.float64)
    mag = 2.0 ** (e - bias + sub) * (f + 1.0 - sub)
    return np.sort(np.concatenate([-mag.ravel(), mag.ravel()]))


def generate_all_float_values_scaled(num_total_bits, num_exp_bits, exp_bias, range_limit_fp):
    grid = generate_all_values_fp(num_total_bits, num_exp_bits, exp_bias)
    return grid / (np.max(np.abs(grid)) / range_limit_fp)


def get_max_value(num_exponent_bits=4, bias=8):
    fbits = 7 - num_exponent_bits
    return 2.0 ** (2 ** num_exponent_bits - 1 - bias) * (2.0 - 2.0 ** -fbits)


class QuantizerBase(nn.Module):
    """Protocol of every quantizer (reference base_quantizers.py:8-47)."""

    def __init__(self, n_bits, per_channel=False, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.n_bits = n_bits
        self.per_channel = per_channel
        self.state = None
        self.x_min_fp32 = self.x_max_fp32 = None

    @property
    def is_initialized(self):
        raise NotImplementedError()

    @property
    def symmetric(self):
        raise NotImplementedError()

    def forward(self, x_float):
        raise NotImplementedError()

    def set_quant_range(self, x_min, x_max):
        raise NotImplementedError()

    def extra_repr(self):
        return f"n_bits={self.n_bits}, per_channel={self.per_channel}, is_initalized={self.is_initialized}"

    def reset(self):
        self._delta = None


class FPQuantizer(QuantizerBase):
    """8-bit floating point fake-quantizer with an ExMy split and a real-valued exponent bias.

    Constructor kwargs, attributes (`maxval`, `mantissa_bits`, `sign_bits`, `set_maxval`,
    `allow_unsigned`, `mse_include_mantissa_bits`) and methods follow the reference class
    (fp8_quantizer.py:151-272).  `maxval` lives on the GPU ([1] or [C]); `mantissa_bits` stays a
    host tensor because the kernel takes it by value.
    """

    def __init__(self, *args, scale_domain=None, mantissa_bits=4, maxval=3, set_maxval=False,
                 learn_maxval=False, learn_mantissa_bits=False, mse_include_mantissa_bits=True,
                 allow_unsigned=False, **kwargs):
        super().__init__(*args, **kwargs)
        m = mantissa_bits
        self.ebits = self.n_bits - m - 1
        self.default_bias = 2 ** (self.ebits - 1)
        # largest value of the format with the IEEE-like default bias (:177-179)
        default_maxval = (2 - 2 ** (-m)) * 2 ** (2 ** self.ebits - 1 - self.default_bias)
        self.maxval = torch.Tensor([maxval if maxval is not None else default_maxval])
        self.mantissa_bits = torch.Tensor([float(m)])
        self.set_maxval = set_maxval
        self.learning_maxval = learn_maxval
        self.learning_mantissa_bits = learn_mantissa_bits
        self.mse_include_mantissa_bits = mse_include_mantissa_bits
        self.allow_unsigned = allow_unsigned
        self.sign_bits = 1

    _RANGE_ATTRS = ("maxval", "mantissa_bits", "sign_bits")

    # `mantissa_bits` is a [1] HOST tensor, as everywhere in the reference's host logic (float(q.mantissa_bits), state
    # dicts, printouts).  One producer writes it on the DEVICE: the MSE estimator's plurality vote
    # (fp8q_mse_select_f32).  Such a value stays pending on the GPU -- forward() hands it to the kernel as a device
    # scalar -- until somebody reads the attribute (one .cpu(), then cached) or QuantizedModel.fix_ranges() brings all
    # pending widths of a model over in one copy.
    @property
    def mantissa_bits(self):
        p = self.__dict__["_parameters"].get("mantissa_bits") if "_parameters" in self.__dict__ else None
        if p is not None:                       # learn_mantissa_bits(): the width is an nn.Parameter (:253-255)
            return p
        host = self.__dict__.get("_mbits_host")
        if host is None:
            host = self.__dict__["_mbits_dev"].detach().reshape(1).float().cpu()     # synchronises
            self.__dict__["_mbits_host"] = host
        return host

    @mantissa_bits.setter
    def mantissa_bits(self, value):
        # (reached through __setattr__ below, which keeps nn.Module.__setattr__ away from this name: it would hand a
        # Parameter to register_parameter -- refused, the class defines `mantissa_bits` -- and refuse a plain tensor while
        # the width is a Parameter)
        params = self.__dict__.get("_parameters")
        if isinstance(value, nn.Parameter):
            params["mantissa_bits"] = value
            self.__dict__["_mbits_dev"], self.__dict__["_mbits_host"] = None, None
            return
        cur = params.get("mantissa_bits") if params is not None else None
        if cur is not None and value is not None:
            # the width is being learned (learn_mantissa_bits) and somebody assigns a value -- the MSE estimator's vote in
            # estimate_ranges_train: keep the Parameter, replace its value
            with torch.no_grad():
                cur.copy_(torch.as_tensor(value, dtype=cur.dtype).reshape(cur.shape).to(cur.device))
            return
        if params is not None and "mantissa_bits" in params:
            del params["mantissa_bits"]
        if isinstance(value, torch.Tensor) and value.is_cuda:
            self.__dict__["_mbits_dev"], self.__dict__["_mbits_host"] = value, None
        else:
            self.__dict__["_mbits_dev"], self.__dict__["_mbits_host"] = None, value

    # `sign_bits` is a host int, as in the reference.  One producer decides it on the DEVICE: set_quant_range() with
    # allow_unsigned and a CUDA range minimum (fp8q_sign_fold_u8 instead of the reference's `if torch.all(x_min >= 0)`,
    # a host round trip per call).  Such a flag stays pending -- forward() hands it to the kernel (fp8q_quantize_ds_f32) --
    # until somebody reads the attribute (one copy, then cached) or QuantizedModel.fix_ranges() collects a model's.
    @property
    def sign_bits(self):
        host = self.__dict__.get("_sign_host")
        if host is None:
            host = int(self.__dict__["_sign_dev"].item())                             # synchronises
            self.__dict__["_sign_host"] = host
            if host == 0:
                self.__dict__["_sign_dev"] = None
        return host

    @sign_bits.setter
    def sign_bits(self, value):
        self.__dict__["_sign_host"], self.__dict__["_sign_dev"] = int(value), None

    def _pending_sign_bits(self):
        """the device flag (uint8 [1], 1 = signed) not yet seen by the host, or None"""
        return self.__dict__.get("_sign_dev") if self.__dict__.get("_sign_host") is None else None

    def _sign_bits_arg(self):
        """what forward() passes to the kernel: the host int when it is known, else the pending device flag (next to a
        pending mantissa width: fp8q_quantize_dms_f32 reads both)"""
        host = self.__dict__.get("_sign_host")
        return host if host is not None else self.__dict__["_sign_dev"]

    def _mantissa_bits_arg(self):
        """what forward() passes to the kernel: the Parameter when the width is being learned, else the host value when
        it is known, else the pending device scalar"""
        p = self.__dict__["_parameters"].get("mantissa_bits")
        if p is not None:
            return p
        host = self.__dict__.get("_mbits_host")
        return host if host is not None else self.__dict__["_mbits_dev"]

    def _pending_mantissa_bits(self):
        """the device scalar not yet seen by the host, or None"""
        if self.__dict__["_parameters"].get("mantissa_bits") is not None:
            return None
        return self.__dict__.get("_mbits_dev") if self.__dict__.get("_mbits_host") is None else None

    def __setattr__(self, name, value):
        # every assignment of a range attribute starts a new range epoch: consumers that cache results computed with
        # these ranges (the layers' quantized-weight cache) key on it instead of on tensor addresses
        if name in FPQuantizer._RANGE_ATTRS:
            object.__setattr__(self, "_range_epoch", getattr(self, "_range_epoch", 0) + 1)
        if name == "mantissa_bits":
            FPQuantizer.mantissa_bits.fset(self, value)
            return
        if name == "sign_bits":
            FPQuantizer.sign_bits.fset(self, value)
            return
        super().__setattr__(name, value)

    # -- hot path ---------------------------------------------------------------------------
    def forward(self, x_float):
        if self.maxval.device != x_float.device:
            self.maxval = self.maxval.to(x_float.device)
        return quantize_to_fp8_ste_MM(x_float, self.n_bits, self.maxval, self._mantissa_bits_arg(),
                                      self._sign_bits_arg())

    # NB: plain methods, as in the reference (:207-211): truthy when used without a call
    def is_initialized(self):
        return True

    def symmetric(self):
        return False

    def effective_bit_width(self):
        return None

    def _make_unsigned(self, x_min):
        if not self.allow_unsigned:
            return False
        if isinstance(x_min, torch.Tensor):
            return bool(torch.all(x_min >= 0))
        return x_min >= 0

    def _fold_sign(self, x_min):
        """:216-225 without the host round trip: `if allow_unsigned and torch.all(x_min >= 0): sign_bits = 0` as one tiny
        launch on a device flag (sticky, like the reference's attribute); the host value is pending until it is read."""
        d = self.__dict__
        if d.get("_sign_host") == 0:
            return                                   # unsigned already: stays
        flag = d.get("_sign_dev")
        if flag is None or flag.device != x_min.device:
            flag = None
        d["_sign_dev"] = _ops.sign_fold(x_min.detach().float(), flag)
        d["_sign_host"] = None
        object.__setattr__(self, "_range_epoch", getattr(self, "_range_epoch", 0) + 1)

    def set_quant_range(self, x_min, x_max):
        """:222-240.  Only acts when set_maxval=True: maxval = |max(|x_min|, x_max)|."""
        if self.allow_unsigned and isinstance(x_min, torch.Tensor) and x_min.is_cuda and hasattr(_ops, "sign_fold") \
                and x_min.dtype in (torch.float32, torch.float64, torch.float16, torch.bfloat16):
            self._fold_sign(x_min)
        elif self._make_unsigned(x_min):
            self.sign_bits = 0
        if not self.set_maxval:
            return
        dev = self.maxval.device
        if not isinstance(x_max, torch.Tensor):
            x_max = torch.tensor([float(x_max)], dtype=torch.float32, device=dev)
            x_min = torch.tensor([float(x_min)], dtype=torch.float32, device=dev)
        mv = torch.abs(torch.max(torch.abs(x_min), x_max)).detach().to(torch.float32)
        self.maxval = mv.reshape(1) if mv.dim() == 0 else mv

    def _prepared(self):
        """The [264] fp32 block of fp8q.ops.quantizer_prepare for this (per-tensor, fixed-range) quantizer, or None.
        Cached until the range changes: every assignment of maxval / mantissa_bits / sign_bits bumps `_range_epoch`, an
        in-place edit of the maxval tensor bumps its version."""
        mv = self.maxval
        if not (isinstance(mv, torch.Tensor) and mv.is_cuda and mv.numel() == 1) or isinstance(mv, nn.Parameter) \
                or self._pending_mantissa_bits() is not None or isinstance(self.mantissa_bits, nn.Parameter):
            return None
        key = (getattr(self, "_range_epoch", None), mv.data_ptr(), mv._version, self.n_bits)
        if self.__dict__.get("_prep_key") != key:
            self.__dict__["_prep"] = _ops.quantizer_prepare(mv, float(self.mantissa_bits), self.n_bits, self.sign_bits)
            self.__dict__["_prep_key"] = key
        return self.__dict__["_prep"]

    def _set_maxval_tensor(self, mv):
        """engine fast path: maxval already computed on the device by the range kernel."""
        self.maxval = mv

    def make_range_trainable(self):
        if self.learning_maxval:
            self.learn_maxval()
        if self.learning_mantissa_bits:
            self.learn_mantissa_bits()

    def learn_maxval(self):
        self.learning_maxval = True
        self.maxval = nn.Parameter(self.maxval)

    def learn_mantissa_bits(self):
        """:253-255: the mantissa width becomes an nn.Parameter; quantize_to_fp8_ste_MM's backward then yields
        d/dmbits (see _FakeQuantSTE).  The forward still hands the kernel the width by value."""
        self.learning_mantissa_bits = True
        # on the quantizer's device (maxval's): a host Parameter inside a CUDA model would stay behind until the next .to()
        self.mantissa_bits = nn.Parameter(self.mantissa_bits.detach().clone().float().to(self.maxval.device))

    def fix_ranges(self):
        for name in ("maxval", "mantissa_bits"):
            p = getattr(self, name)
            if isinstance(p, nn.Parameter):
                delattr(self, name)
                setattr(self, name, p.data.clone())

    def exponent_bias(self):
        """the real-valued bias implied by maxval (what the reference prints in extra_repr)."""
        m = float(np.clip(np.round(_host_float(self.mantissa_bits)), 1, 7))
        e = 7 - m
        return 2 ** e - torch.log2(self.maxval.float().cpu()) + float(np.log2(2 - 2 ** (-m))) - 1, e

    def extra_repr(self):
        bias, e = self.exponent_bias()
        b = "[per_channel]" if bias.numel() > 1 else f"{bias.item()}"
        return f"Exponent: {e} bits; mode: ; bias: {b}"
