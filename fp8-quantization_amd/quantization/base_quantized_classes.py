"""Alias: QuantizedModule / QuantizedActivation / FP32Acts live in quantization.layers."""
from .layers import QuantizedModule, QuantizedActivation, FP32Acts  # noqa: F401
