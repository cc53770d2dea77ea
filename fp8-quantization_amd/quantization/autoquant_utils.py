"""Alias: the wrapped layers and model-rewriting helpers live in quantization.layers."""
from .layers import (QuantConv1d, QuantConv, QuantConvTransposeBase, QuantConvTranspose1d,  # noqa: F401
                     QuantConvTranspose, QuantLinear, BNQConv1d, BNQConv, BNQLinear, QuantLayerNorm,
                     QuantizedActivationWrapper, Flattener, non_bn_module_map, bn_module_map,
                     non_param_modules, quant_conv_modules, next_bn, get_act, get_module_args, fold_bn,
                     quantize_sequential, quantize_model)
