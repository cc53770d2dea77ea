"""Alias of quantization.fp8 under the reference's module name."""
from ..fp8 import (FPQuantizer, quantize_to_fp8_ste_MM, generate_all_values_fp,  # noqa: F401
                   generate_all_float_values_scaled, get_max_value)
