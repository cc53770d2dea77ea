"""Alias: BNFusedHijacker lives in quantization.layers."""
from .layers import BNFusedHijacker  # noqa: F401
