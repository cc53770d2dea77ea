"""Host-side mirror of the reference's `quantization` package on the MI355X FP8 engine.

Import paths of the reference keep working (thin alias modules):
    quantization.quantizers.fp8_quantizer, quantization.range_estimators,
    quantization.quantization_manager, quantization.hijacker, quantization.autoquant_utils,
    quantization.quantized_folded_bn, quantization.base_quantized_classes,
    quantization.base_quantized_model
Implementation modules: registry, fp8, estimators, manager, layers, model.
"""
from . import registry, fp8, estimators, manager, layers, model  # noqa: F401
from . import autoquant_utils, quantized_folded_bn, utils  # noqa: F401
