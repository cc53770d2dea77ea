"""Alias: QuantizationHijacker lives in quantization.layers."""
from .layers import QuantizationHijacker, activations_set  # noqa: F401
