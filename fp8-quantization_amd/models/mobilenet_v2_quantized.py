"""Quantized MobileNetV2 (reference models/mobilenet_v2_quantized.py:15-113) on the FP8 engine."""
import os

import torch

from quantization.autoquant_utils import quantize_sequential, Flattener, quantize_model, BNQConv
from quantization.base_quantized_classes import QuantizedActivation
from quantization.base_quantized_model import QuantizedModel
from ._setups import apply_preset, weight_bits, act_bits, fp32_output, fp32_all_activations
from .mobilenet_v2 import MobileNetV2, InvertedResidual


class QuantizedInvertedResidual(QuantizedActivation):
    """The residual sum gets its own activation quantizer; blocks without a skip do not."""

    def __init__(self, inv_res_orig, **quant_params):
        super().__init__(**quant_params)
        self.use_res_connect = inv_res_orig.use_res_connect
        self.conv = quantize_sequential(inv_res_orig.conv, **quant_params)

    def forward(self, x):
        branch = self.conv(x)
        if not self.use_res_connect:
            return branch
        aq = self.activation_quantizer
        if self._qa and hasattr(aq, "can_fuse") and aq.can_fuse(branch) and x.shape == branch.shape:
            return aq.forward_fused(branch, residual=x.contiguous(), act=0)        # add + quantize
        return self.quantize_activations(x + branch)


def _stem(net):
    return net.features[0][0]


def _fc(net):
    return net.classifier[1]


def _depthwise(net):
    hits = [(n, m) for n, m in net.named_modules() if isinstance(m, BNQConv) and m.groups == m.in_channels]
    for n, _ in hits:
        print(f"Set layer {n} to 8 bits")
    return [m for _, m in hits]


_PRESETS = {
    "FP_logits": ("Do not quantize output of FC layer", [(_fc, fp32_output)]),
    "fc4": (None, [(_stem, weight_bits(8)), (_fc, weight_bits(4))]),
    "fc4_dw8": (None, [(_stem, weight_bits(8)), (_fc, weight_bits(4)), (_depthwise, weight_bits(8))]),
    "LSQ": ("Set quantization to LSQ (first+last layer in 8 bits)",
            [(_stem, weight_bits(8)), (lambda net: net.features[-2][0], act_bits(8)), (_fc, weight_bits(8)),
             (_fc, fp32_output)]),
    "LSQ_paper": (None, [(_stem, fp32_output), (_stem, weight_bits(8)), (_fc, weight_bits(8)), (_fc, act_bits(8)),
                         (lambda net: net.features, fp32_all_activations)]),
}


class QuantizedMobileNetV2(QuantizedModel):
    def __init__(self, model_fp, input_size=(1, 3, 224, 224), quant_setup=None, **quant_params):
        super().__init__(input_size)
        # LSQ_paper quantizes layer inputs instead of outputs: then the stem's activation quantizers are not tied
        tie = not (bool(quant_setup) and quant_setup == "LSQ_paper")
        self.features = quantize_sequential(model_fp.features, tie_activation_quantizers=tie,
                                            specials={InvertedResidual: QuantizedInvertedResidual}, **quant_params)
        self.flattener = Flattener()
        self.classifier = quantize_model(model_fp.classifier, **quant_params)
        apply_preset(self, quant_setup, _PRESETS, "MobilenetV2")

    def forward(self, x):
        return self.classifier(self.flattener(self.features(x)))


def mobilenetv2_quantized(pretrained=True, model_dir=None, load_type="fp32", **qparams):
    if load_type not in ("fp32", "quantized"):
        raise ValueError("wrong load_type specified")
    fp_model = MobileNetV2()
    if load_type == "fp32" and pretrained:
        assert model_dir and os.path.exists(model_dir), "pretrained MobileNetV2 needs --model-dir"
        print(f"Loading pretrained weights from {model_dir}")
        fp_model.load_state_dict(torch.load(model_dir, map_location="cpu"))
    model = QuantizedMobileNetV2(fp_model, **qparams)          # fp32 without a checkpoint: random init (synthetic runs)
    if load_type == "quantized":
        print(f"Loading pretrained quantized model from {model_dir}")
        model.load_state_dict(torch.load(model_dir, map_location="cpu"), strict=False)
    return model
