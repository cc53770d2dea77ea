"""Quantized ResNet (reference models/resnet_quantized.py:14-150) on the MI355X FP8 engine."""
import torch
from torch import nn

from quantization.autoquant_utils import quantize_model, Flattener, QuantizedActivationWrapper
from quantization.base_quantized_classes import QuantizedActivation, FP32Acts
from quantization.base_quantized_model import QuantizedModel
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
        self._apply_setup(quant_setup)

    def _apply_setup(self, setup):
        first, last_block = self.features[0], self.features[-1][-1]
        if setup in (None, "all"):
            return
        if setup == "LSQ":
            print("Set quantization to LSQ (first+last layer in 8 bits)")
            first.weight_quantizer.quantizer.n_bits = 8
            last_block.activation_quantizer.quantizer.n_bits = 8
            last_block.features[-1].activation_quantizer.quantizer.n_bits = 8
            self.fc.weight_quantizer.quantizer.n_bits = 8
            self.fc.activation_quantizer = FP32Acts()
        elif setup == "LSQ_paper":
            first.activation_quantizer = FP32Acts()
            first.weight_quantizer.quantizer.n_bits = 8
            self.fc.activation_quantizer.quantizer.n_bits = 8
            self.fc.weight_quantizer.quantizer.n_bits = 8
            for layer in self.features.modules():
                if isinstance(layer, QuantizedActivation):
                    layer.activation_quantizer = FP32Acts()
        elif setup == "FP_logits":
            print("Do not quantize output of FC layer")
            self.fc.activation_quantizer = FP32Acts()
        elif setup == "fc4":
            first.weight_quantizer.quantizer.n_bits = 8
            self.fc.weight_quantizer.quantizer.n_bits = 4
        else:
            raise ValueError(f"Quantization setup '{setup}' not supported for Resnet")

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
