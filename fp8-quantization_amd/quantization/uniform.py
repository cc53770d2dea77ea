"""Uniform (INT) fake-quantizers -- the comparison baseline of compute_quant_error.py (config 1).

Reference: quantization/quantizers/uniform_quantizers.py:13-331.  These are NOT part of the FP8 hot
path: they are a handful of elementwise torch ops on whatever device the
tensor lives on, kept so that the `symmetric_uniform` / `asymmetric_uniform` registry entries and
the INT8 row of the SQNR study work.
"""
import torch

from .fp8 import QuantizerBase, QuantizerNotInitializedError, round_ste_func


class AsymmetricUniformQuantizer(QuantizerBase):
    def __init__(self, n_bits, scale_domain="linear", discretizer=round_ste_func, discretizer_args=tuple(),
                 grad_scaling=False, eps=1e-8, **kwargs):
        super().__init__(n_bits=n_bits, **kwargs)
        assert scale_domain in ("linear", "log")
        self.register_buffer("_delta", None)
        self.register_buffer("_zero_float", None)
        self.discretizer = discretizer(*discretizer_args) if isinstance(discretizer, type) else discretizer
        self.scale_domain = scale_domain
        self.grad_scaling = grad_scaling
        self.eps = eps

    @property
    def delta(self):
        if self._delta is None:
            raise QuantizerNotInitializedError()
        return self._delta

    @property
    def zero_float(self):
        if self._zero_float is None:
            raise QuantizerNotInitializedError()
        return self._zero_float

    @property
    def is_initialized(self):
        return self._delta is not None

    @property
    def symmetric(self):
        return False

    @property
    def int_min(self):
        return 0.0

    @property
    def int_max(self):
        return 2.0 ** self.n_bits - 1

    @property
    def scale(self):
        return torch.clamp(self.delta, min=self.eps) if self.scale_domain == "linear" else torch.exp(self.delta)

    @property
    def zero_point(self):
        return torch.clamp(self.discretizer(self.zero_float), self.int_min, self.int_max)

    @property
    def x_max(self):
        return self.scale * (self.int_max - self.zero_point)

    @property
    def x_min(self):
        return self.scale * (self.int_min - self.zero_point)

    def _params_like(self, x):
        scale, zp = self.scale, self.zero_point
        if torch.is_tensor(scale) and scale.device != x.device:
            scale = scale.to(x.device)
        if torch.is_tensor(zp) and zp.device != x.device:
            zp = zp.to(x.device)
        if self.per_channel and torch.is_tensor(scale) and scale.dim() == 1 and x.dim() > 1:
            shape = [-1] + [1] * (x.dim() - 1)
            scale = scale.view(shape)
            zp = zp.view(shape) if torch.is_tensor(zp) and zp.dim() == 1 else zp
        return scale, zp

    def to_integer_forward(self, x_float, *args, **kwargs):
        scale, zp = self._params_like(x_float)
        return torch.clamp(self.discretizer(x_float / scale) + zp, self.int_min, self.int_max)

    def forward(self, x_float, *args, **kwargs):
        scale, zp = self._params_like(x_float)
        return scale * (self.to_integer_forward(x_float) - zp)

    def _tensorize_min_max(self, x_min, x_max):
        if not torch.is_tensor(x_min):
            x_min, x_max = torch.tensor(x_min).float(), torch.tensor(x_max).float()
        if x_min.dim() > 0 and len(x_min) > 1 and not self.per_channel:
            raise ValueError("x_min and x_max must be a float or 1-D Tensor for per-tensor quantization "
                             "(per_channel=False)")
        # the range always contains zero; a positive upper end avoids a zero scale
        return torch.min(x_min, torch.zeros_like(x_min)), torch.max(x_max, torch.ones_like(x_max) * self.eps)

    def set_quant_range(self, x_min, x_max):
        self.x_min_fp32, self.x_max_fp32 = x_min, x_max
        x_min, x_max = self._tensorize_min_max(x_min, x_max)
        delta = (x_max - x_min) / self.int_max
        self._zero_float = (-x_min / delta).detach()
        self._delta = (torch.log(delta) if self.scale_domain == "log" else delta).detach()

    def make_range_trainable(self):
        if not isinstance(self._delta, torch.nn.Parameter):
            self._delta = torch.nn.Parameter(self._delta)
            if self._zero_float is not None:
                self._zero_float = torch.nn.Parameter(self._zero_float)

    def fix_ranges(self):
        for name in ("_delta", "_zero_float"):
            p = getattr(self, name, None)
            if isinstance(p, torch.nn.Parameter):
                delattr(self, name)
                self.register_buffer(name, p.data)

    def generate_grid(self):
        return self.scale * (torch.arange(self.int_min, self.int_max + 1, device=self.delta.device) - self.zero_point)


class SymmetricUniformQuantizer(AsymmetricUniformQuantizer):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.register_buffer("_signed", None)

    @property
    def signed(self):
        if self._signed is None:
            raise QuantizerNotInitializedError()
        return self._signed.item()

    @property
    def symmetric(self):
        return True

    @property
    def int_min(self):
        return -(2.0 ** (self.n_bits - 1)) if self.signed else 0

    @property
    def int_max(self):
        return 2.0 ** (self.n_bits - self.signed) - 1

    @property
    def zero_point(self):
        return 0.0

    def set_quant_range(self, x_min, x_max):
        self.x_min_fp32, self.x_max_fp32 = x_min, x_max
        x_min, x_max = self._tensorize_min_max(x_min, x_max)
        self._signed = x_min.min() < 0
        delta = torch.max(x_min.abs(), x_max) / self.int_max
        self._delta = (torch.log(delta) if self.scale_domain == "log" else delta).detach()
