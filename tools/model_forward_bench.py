"""Quantized ResNet-18 / MobileNetV2 forward time at batch 64 (fixed ranges), fused vs unfused epilogue."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fp8-quantization_amd")]
import torch
from models import QuantArchitectures
from quantization.quantization_manager import QMethods
from quantization.range_estimators import RangeEstimators

def qparams(M):
    return dict(method=QMethods.fp_quantizer.cls, weight_range_method=RangeEstimators.current_minmax.cls,
                act_range_method=RangeEstimators.allminmax.cls, n_bits=8, per_channel_weights=True,
                fp8_kwargs=dict(maxval=None, mantissa_bits=M, set_maxval=True))

def run(arch, M, batch=64):
    torch.manual_seed(0)
    m = QuantArchitectures[arch](pretrained=False, load_type="fp32", **qparams(M)).cuda().eval()
    x = torch.randn(batch, 3, 224, 224, device="cuda")
    out = {}
    with torch.no_grad():
        m.full_precision(); 
        for _ in range(3): m(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): m(x)
        torch.cuda.synchronize(); out["fp32"] = (time.perf_counter() - t0) / 10
        m.set_quant_state(True, True); m(x); m.fix_ranges()
        for fuse in ("1", "0"):
            os.environ["FP8Q_FUSE_EPILOGUE"] = fuse
            for _ in range(3): m(x)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): m(x)
            torch.cuda.synchronize(); out["quant_fused" if fuse == "1" else "quant_unfused"] = (time.perf_counter() - t0) / 10
        # HIP graph of the whole quantized forward (fixed ranges: no host-side decisions in the forward)
        os.environ["FP8Q_FUSE_EPILOGUE"] = "1"
        try:
            ref = m(x).clone()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3): m(x)
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                yg = m(x)
            for _ in range(3): g.replay()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): g.replay()
            torch.cuda.synchronize(); out["quant_fused_hipgraph"] = (time.perf_counter() - t0) / 10
            out["graph_bit_identical"] = float(torch.equal(yg, ref))
        except Exception as e:   # noqa
            print("graph capture failed:", repr(e)[:300])
    print(arch, f"batch {batch}", {k: (f"{v*1e3:.2f} ms" if k != "graph_bit_identical" else bool(v)) for k, v in out.items()}, flush=True)

for b in [int(v) for v in os.environ.get("BATCHES", "64").split(",")]:
    run("resnet18_quantized", 2, b)
    run("mobilenet_v2_quantized", 3, b)
