"""Whole-model switches (reference base_quantized_model.py:19-135)."""
import torch
from torch import nn

from .fp8 import QuantizerBase
from .layers import QuantizedModule, _for_managers


class QuantizedModel(nn.Module):
    def __init__(self, input_size=(1, 3, 224, 224)):
        super().__init__()
        self.input_size = input_size

    def load_state_dict(self, state_dict, strict=True):
        """First restore the _quant_w/_quant_a flags, run one dummy forward so that every None
        buffer (estimator ranges) gets its shape, then load everything (:34-62)."""
        flags = {k: v for k, v in state_dict.items() if k.endswith("_quant_a") or k.endswith("_quant_w")}
        if not flags:
            raise ValueError("The quantization states of activations or weights should be "
                             "included in the state dict ")
        super().load_state_dict(flags, strict=False)
        device = next(self.parameters()).device
        with torch.no_grad():
            self.forward(torch.rand(*self.input_size, device=device))
        return super().load_state_dict(state_dict, strict)

    def _each(self, method):
        def visit(layer):
            if isinstance(layer, QuantizedModule):
                getattr(layer, method)()
        self.apply(visit)

    def quantized_weights(self):
        self._each("quantized_weights")

    def full_precision_weights(self):
        self._each("full_precision_weights")

    def quantized_acts(self):
        self._each("quantized_acts")

    def full_precision_acts(self):
        self._each("full_precision_acts")

    def quantized(self):
        self._each("quantized")

    def full_precision(self):
        self._each("full_precision")

    def set_quant_state(self, weight_quant, act_quant):
        (self.quantized_acts if act_quant else self.full_precision_acts)()
        (self.quantized_weights if weight_quant else self.full_precision_weights)()

    def grad_scaling(self, grad_scaling=True):
        def visit(m):
            if isinstance(m, QuantizerBase):
                m.grad_scaling = grad_scaling
        self.apply(visit)

    def estimate_ranges(self):
        _for_managers(self, lambda m: m.estimate_ranges(), need_init=False)

    def estimate_ranges_train(self):
        _for_managers(self, lambda m: m.estimate_ranges_train(), need_init=True)

    def learn_ranges(self):
        _for_managers(self, lambda m: m.learn_ranges(), need_init=True)

    def fix_ranges(self):
        _for_managers(self, lambda m: m.fix_ranges(), need_init=True)
