"""Quantized MobileNetV2 (reference models/mobilenet_v2_quantized.py:15-113) on the FP8 engine."""
import os

import torch

from quantization.autoquant_utils import quantize_sequential, Flattener, quantize_model, BNQConv
from quantization.base_quantized_classes import QuantizedActivation, FP32Acts
from quantization.base_quantized_model import QuantizedModel
from .mobilenet_v2 import MobileNetV2, InvertedResidual


class QuantizedInvertedResidual(QuantizedActivation):
    """The residual sum gets its own activation quantizer; blocks without a skip do not."""

    def __init__(self, inv_res_orig, **quant_params):
        super().__init__(**quant_params)
        self.use_res_connect = inv_res_orig.use_res_connect
        self.conv = quantize_sequential(inv_res_orig.conv, **quant_params)

    def forward(self, x):
        if not self.use_res_connect:
            return self.conv(x)
        out = self.conv(x)
        aq = self.activation_quantizer
        if self._qa and hasattr(aq, "can_fuse") and aq.can_fuse(out) and x.shape == out.shape:
            return aq.forward_fused(out, residual=x.contiguous(), act=0)        # add + quantize
        return self.quantize_activations(x + out)


class QuantizedMobileNetV2(QuantizedModel):
    def __init__(self, model_fp, input_size=(1, 3, 224, 224), quant_setup=None, **quant_params):
        super().__init__(input_size)
        quantize_input = bool(quant_setup) and quant_setup == "LSQ_paper"
        self.features = quantize_sequential(
            model_fp.features, tie_activation_quantizers=not quantize_input,
            specials={InvertedResidual: QuantizedInvertedResidual}, **quant_params)
        self.flattener = Flattener()
        self.classifier = quantize_model(model_fp.classifier, **quant_params)
        self._apply_setup(quant_setup)

    def _apply_setup(self, setup):
        stem, fc = self.features[0][0], self.classifier[1]
        if setup in (None, "all"):
            return
        if setup == "FP_logits":
            print("Do not quantize output of FC layer")
            fc.activation_quantizer = FP32Acts()
        elif setup in ("fc4", "fc4_dw8"):
            stem.weight_quantizer.quantizer.n_bits = 8
            fc.weight_quantizer.quantizer.n_bits = 4
            if setup == "fc4_dw8":
                for name, m in self.named_modules():
                    if isinstance(m, BNQConv) and m.groups == m.in_channels:
                        m.weight_quantizer.quantizer.n_bits = 8
                        print(f"Set layer {name} to 8 bits")
        elif setup == "LSQ":
            print("Set quantization to LSQ (first+last layer in 8 bits)")
            stem.weight_quantizer.quantizer.n_bits = 8
            self.features[-2][0].activation_quantizer.quantizer.n_bits = 8
            fc.weight_quantizer.quantizer.n_bits = 8
            fc.activation_quantizer = FP32Acts()
        elif setup == "LSQ_paper":
            stem.activation_quantizer = FP32Acts()
            stem.weight_quantizer.quantizer.n_bits = 8
            fc.weight_quantizer.quantizer.n_bits = 8
            fc.activation_quantizer.quantizer.n_bits = 8
            for layer in self.features.modules():
                if isinstance(layer, QuantizedActivation):
                    layer.activation_quantizer = FP32Acts()
        else:
            raise ValueError(f"Quantization setup '{setup}' not supported for MobilenetV2")

    def forward(self, x):
        return self.classifier(self.flattener(self.features(x)))


def mobilenetv2_quantized(pretrained=True, model_dir=None, load_type="fp32", **qparams):
    fp_model = MobileNetV2()
    if pretrained and load_type == "fp32":
        assert model_dir and os.path.exists(model_dir), "pretrained MobileNetV2 needs --model-dir"
        print(f"Loading pretrained weights from {model_dir}")
        fp_model.load_state_dict(torch.load(model_dir, map_location="cpu"))
        return QuantizedMobileNetV2(fp_model, **qparams)
    if load_type == "fp32":
        return QuantizedMobileNetV2(fp_model, **qparams)    # random init (synthetic runs)
    if load_type == "quantized":
        print(f"Loading pretrained quantized model from {model_dir}")
        model = QuantizedMobileNetV2(fp_model, **qparams)
        model.load_state_dict(torch.load(model_dir, map_location="cpu"), strict=False)
        return model
    raise ValueError("wrong load_type specified")
