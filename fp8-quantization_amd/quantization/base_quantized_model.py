"""Alias: QuantizedModel lives in quantization.model."""
from .model import QuantizedModel  # noqa: F401
