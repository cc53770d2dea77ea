"""The `--quant-setup` presets of the model zoo as data: each preset is a list of (selector, action) pairs applied
to the freshly built quantized network.  Semantics follow the reference's per-architecture if-chains
(models/resnet_quantized.py:73-122, models/mobilenet_v2_quantized.py:49-101); one table-driven implementation serves
both architectures here."""
from quantization.base_quantized_classes import QuantizedActivation, FP32Acts


def weight_bits(n):
    def act(layer):
        layer.weight_quantizer.quantizer.n_bits = n
    return act


def act_bits(n):
    def act(layer):
        layer.activation_quantizer.quantizer.n_bits = n
    return act


def fp32_output(layer):
    layer.activation_quantizer = FP32Acts()


def fp32_all_activations(features):
    for m in features.modules():
        if isinstance(m, QuantizedActivation):
            m.activation_quantizer = FP32Acts()


def apply_preset(net, preset, table, arch):
    """table: {preset name: (message or None, [(selector(net) -> module or iterable of modules, action), ...])}"""
    if preset in (None, "all"):
        return
    if preset not in table:
        raise ValueError(f"Quantization setup '{preset}' not supported for {arch}")
    message, steps = table[preset]
    if message:
        print(message)
    for select, action in steps:
        target = select(net)
        for module in (target if isinstance(target, (list, tuple)) else [target]):
            action(module)
