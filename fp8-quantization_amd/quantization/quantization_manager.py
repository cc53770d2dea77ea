"""Alias of quantization.manager under the reference's module name."""
from .manager import (QuantizationManager, Qstates, QMethods, SymmetricUniformQuantizer,  # noqa: F401
                      AsymmetricUniformQuantizer)
