"""Alias of quantization.uniform under the reference's module name."""
from ..uniform import AsymmetricUniformQuantizer, SymmetricUniformQuantizer  # noqa: F401
