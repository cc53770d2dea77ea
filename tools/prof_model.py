"""Quantized ResNet-18 forward (batch 64, fixed ranges, fused epilogue) x N for rocprofv3 --kernel-trace --stats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fp8-quantization_amd")]
import torch
from models import QuantArchitectures
from quantization.quantization_manager import QMethods
from quantization.range_estimators import RangeEstimators
torch.manual_seed(0)
m = QuantArchitectures["resnet18_quantized"](pretrained=False, load_type="fp32", method=QMethods.fp_quantizer.cls,
                                             weight_range_method=RangeEstimators.current_minmax.cls,
                                             act_range_method=RangeEstimators.allminmax.cls, n_bits=8, per_channel_weights=True,
                                             fp8_kwargs=dict(maxval=None, mantissa_bits=2, set_maxval=True)).cuda().eval()
x = torch.randn(64, 3, 224, 224, device="cuda")
with torch.no_grad():
    m.set_quant_state(True, True); m(x); m.fix_ranges()
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
        m(x)
torch.cuda.synchronize()
