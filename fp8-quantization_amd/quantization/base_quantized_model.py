"""Alias: QuantizedModel lives in quantization.model."""
from .model import QuantizedModel, quantizer_ranges, load_quantizer_ranges, RANGES_KEY  # noqa: F401
