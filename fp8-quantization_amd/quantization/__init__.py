"""Host-side mirror of the reference's `quantization` package on the MI355X FP8 engine.

Import paths of the reference keep working -- quantization.quantizers.fp8_quantizer, quantization.range_estimators,
quantization.quantization_manager, quantization.hijacker, quantization.autoquant_utils, quantization.quantized_folded_bn,
quantization.base_quantized_classes, quantization.base_quantized_model, quantization.quant_error_estimator --: one table
(`_reference_paths.ALIASES`) maps each to the implementation module that holds its classes.
Implementation modules: registry, fp8, uniform, estimators, manager, layers, model, utils, quant_error, distributions, refprec.
"""
from . import _reference_paths

_reference_paths.install()
from . import registry, fp8, estimators, manager, layers, model, utils  # noqa: E402,F401
