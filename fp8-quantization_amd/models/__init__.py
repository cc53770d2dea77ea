"""Quantized model zoo behind `validate-quantized --architecture ...` (reference models/__init__.py:10-13)."""
from quantization.registry import ClassEnumOptions, MethodMap
from . import mobilenet_v2_quantized as _mbv2
from . import resnet_quantized as _resnet

_BUILDERS = (("mobilenet_v2_quantized", _mbv2.mobilenetv2_quantized), ("resnet18_quantized", _resnet.resnet18_quantized),
             ("resnet50_quantized", _resnet.resnet50_quantized))
QuantArchitectures = ClassEnumOptions("QuantArchitectures", {name: MethodMap(fn) for name, fn in _BUILDERS})
