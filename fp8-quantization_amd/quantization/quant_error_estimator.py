"""Alias of quantization.quant_error under the reference's module name."""
from .quant_error import *  # noqa: F401,F403
from .quant_error import (compute_expected_quant_mse, compute_expected_dot_prod_mse,  # noqa: F401
                          estimate_rounding_error_analyt, estimate_dot_prod_error_analyt,
                          estimate_rounding_error_empirical, integrate_pdf_grid_func_analyt)
