"""Quantized model zoo of the reference's validate-quantized command (models/__init__.py:10-13)."""
from quantization.registry import ClassEnumOptions, MethodMap
from .mobilenet_v2_quantized import mobilenetv2_quantized
from .resnet_quantized import resnet18_quantized, resnet50_quantized


class QuantArchitectures(ClassEnumOptions):
    mobilenet_v2_quantized = MethodMap(mobilenetv2_quantized)
    resnet18_quantized = MethodMap(resnet18_quantized)
    resnet50_quantized = MethodMap(resnet50_quantized)
