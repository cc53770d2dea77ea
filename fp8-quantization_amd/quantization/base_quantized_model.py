"""Alias: QuantizedModel lives in quantization.model."""
from .model import QuantizedModel, quantizer_ranges, load_quantizer_ranges, prequantize_weights, GraphedForward, GraphedCalibration, export_fp8_weights, decode_fp8_weights, RANGES_KEY  # noqa: F401
