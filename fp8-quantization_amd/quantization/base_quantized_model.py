"""Alias: QuantizedModel lives in quantization.model."""
from .model import QuantizedModel, quantizer_ranges, load_quantizer_ranges, prequantize_weights, GraphedForward, RANGES_KEY  # noqa: F401
