"""Quantized ResNet (reference models/resnet_quantized.py:14-150) on the MI355X FP8 engine."""
import torch
from torch import nn

from quantization.autoquant_utils import quantize_model, Flattener, QuantizedActivationWrapper
from quantization.base_quantized_classes import QuantizedActivation
from quantization.base_quantized_model import QuantizedModel
from ._setups import apply_preset, weight_bits, act_bits, fp32_output, fp32_all_activations
from .resnet import BasicBlock, Bottleneck, resnet18, resnet50


class QuantizedBlock(QuantizedActivation):
    """Residual block: conv-bn-relu-conv-bn as fused quantized layers, plus one more activation
    quantizer on relu(out + residual) (reference :39-46)."""

    def __init__(self, block, **quant_params):
        super().__init__(**quant_params)
        if isinstance(block, Bottleneck):
            body = nn.Sequential(block.conv1, block.bn1, block.relu, block.conv2, block.bn2, block.relu,
                                 block.conv3, block.bn3)
        elif isinstance(block, BasicBlock):
            body = nn.Sequential(block.conv1, block.bn1, block.relu, block.conv2, block.bn2)
        else:
            raise NotImplementedError(f"unknown residual block {type(block).__name__}")
        self.features = quantize_model(body, **quant_params)
        self.downsample = quantize_model(block.downsample, **quant_params) if block.downsample else None
        self.relu = block.relu

    def forward(self, x):
        residual = x if self.downsample is None else self.downsample(x)
        out = self.features(x)
        aq = self.activation_quantizer
        if self._qa and isinstance(self.relu, nn.ReLU) and hasattr(aq, "can_fuse") and aq.can_fuse(out) \
                and residual.shape == out.shape:
            return aq.forward_fused(out, residual=residual.contiguous(), act=1)   # add + relu + quantize
        out += residual
        return self.quantize_activations(self.relu(out))


def _first(net):
    return net.features[0]


def _last_block(net):
    return net.features[-1][-1]


_PRESETS = {
    "LSQ": ("Set quantization to LSQ (first+last layer in 8 bits)",
            [(_first, weight_bits(8)), (_last_block, act_bits(8)),
             (lambda net: _last_block(net).features[-1], act_bits(8)),
             (lambda net: net.fc, weight_bits(8)), (lambda net: net.fc, fp32_output)]),
    "LSQ_paper": (None, [(_first, fp32_output), (_first, weight_bits(8)), (lambda net: net.fc, act_bits(8)),
                         (lambda net: net.fc, weight_bits(8)), (lambda net: net.features, fp32_all_activations)]),
    "FP_logits": ("Do not quantize output of FC layer", [(lambda net: net.fc, fp32_output)]),
    "fc4": (None, [(_first, weight_bits(8)), (lambda net: net.fc, weight_bits(4))]),
}


class QuantizedResNet(QuantizedModel):
    def __init__(self, resnet, input_size=(1, 3, 224, 224), quant_setup=None, **quant_params):
        super().__init__(input_size)
        specials = {BasicBlock: QuantizedBlock, Bottleneck: QuantizedBlock}
        stem = [resnet.conv1, resnet.bn1, resnet.relu]
        if hasattr(resnet, "maxpool"):          # ImageNet variant; Tiny-ImageNet nets have no maxpool
            stem.append(resnet.maxpool)
        body = nn.Sequential(*stem, resnet.layer1, resnet.layer2, resnet.layer3, resnet.layer4)
        self.features = quantize_model(body, specials=specials, **quant_params)
        if quant_setup == "LSQ_paper":
            self.avgpool = resnet.avgpool       # the input of the last layer is quantized instead
        else:
            # pooled output reuses the last block's quantizer, without a range update
            self.avgpool = QuantizedActivationWrapper(
                resnet.avgpool, tie_activation_quantizers=True,
                input_quantizer=self.features[-1][-1].activation_quantizer, **quant_params)
        self.flattener = Flattener()
        self.fc = quantize_model(resnet.fc, **quant_params)
        apply_preset(self, quant_setup, _PRESETS, "Resnet")

    def forward(self, x):
        x = self.avgpool(self.features(x))
        return self.fc(self.flattener(x))


def _resnet_quantized(builder, what, pretrained, model_dir, load_type, qparams):
    fp_model = builder()
    if load_type == "fp32":
        if pretrained:
            if not model_dir:
                raise RuntimeError(f"pretrained=True needs --model-dir <torchvision {what} state dict>: "
                                   "there is no network access to download weights")
            fp_model.load_state_dict(torch.load(model_dir, map_location="cpu"))
        return QuantizedResNet(fp_model, **qparams)
    if load_type == "quantized":
        print(f"Loading pretrained quantized model from {model_dir}")
        model = QuantizedResNet(fp_model, **qparams)
        model.load_state_dict(torch.load(model_dir, map_location="cpu"))
        return model
    raise ValueError("wrong load_type specified")


def resnet18_quantized(pretrained=True, model_dir=None, load_type="fp32", **qparams):
    return _resnet_quantized(resnet18, "resnet18", pretrained, model_dir, load_type, qparams)


def resnet50_quantized(pretrained=True, model_dir=None, load_type="fp32", **qparams):
    """reference models/resnet_quantized.py:153-170"""
    return _resnet_quantized(resnet50, "resnet50", pretrained, model_dir, load_type, qparams)
