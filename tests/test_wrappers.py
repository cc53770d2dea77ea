"""The operator wrappers the model-zoo fixtures never reach -- QuantConv1d, QuantConvTranspose1d / QuantConvTranspose (weight
dims 0/1 swapped around the per-channel quantizer), BNQConv1d, BNQLinear, QuantLayerNorm -- against the reference's own
quantize_model on two toy nets (/root/reference/quantization/autoquant_utils.py:20-31, 46-87, 94-105, 120-122, 166-174;
fixture tests/golden/g11_wrappers.npz written by tests/golden/make_golden.py:make_g11).  On the CPU the oracle stands in for
the HIP ops (host logic + arithmetic contract); on the GPU the kernels run, including the multi-tensor weight plan over the
transposed convolutions.
"""
import os

import numpy as np
import pytest
import torch
from torch import nn

CASES = [("e5m2", 2, True), ("e4m3", 3, True), ("e4m3_pt", 3, False)]
EXPECT_CLASSES = {
    "net1d": ["BNQConv1d", "QuantConv1d", "QuantConvTranspose1d", "Flatten", "BNQLinear", "QuantLayerNorm", "QuantLinear"],
    "net2d": ["QuantConv", "QuantConvTranspose", "QuantConvTranspose", "QuantizedActivationWrapper", "Flatten", "QuantLinear"],
}


def _nets(g):
    net1d = nn.Sequential(nn.Conv1d(4, 8, 3, padding=1, bias=False), nn.BatchNorm1d(8), nn.ReLU(),
                          nn.Conv1d(8, 8, 3, padding=1, bias=True), nn.ReLU6(),
                          nn.ConvTranspose1d(8, 6, 4, stride=2, padding=1, bias=True),
                          nn.Flatten(), nn.Linear(6 * 32, 16, bias=False), nn.BatchNorm1d(16), nn.ReLU(),
                          nn.LayerNorm(16), nn.Linear(16, 5))
    net2d = nn.Sequential(nn.Conv2d(3, 6, 3, padding=1), nn.ReLU(),
                          nn.ConvTranspose2d(6, 10, 3, stride=2, padding=1, output_padding=1, bias=False), nn.ReLU(),
                          nn.ConvTranspose2d(10, 4, 2, stride=1, groups=2, bias=True),
                          nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(4, 7))
    nets = dict(net1d=net1d, net2d=net2d)
    for name, net in nets.items():
        net.load_state_dict({k[len(name) + 4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"{name}_sd_")})
        net.eval()
    return nets


def _quantized(net, M, pcw):
    from quantization.autoquant_utils import quantize_model
    from quantization.quantization_manager import QMethods
    from quantization.range_estimators import RangeEstimators
    return quantize_model(net, tie_activation_quantizers=True, method=QMethods.fp_quantizer.cls,
                          weight_range_method=RangeEstimators.current_minmax.cls, act_range_method=RangeEstimators.allminmax.cls,
                          n_bits=8, n_bits_act=8, per_channel_weights=pcw,
                          fp8_kwargs=dict(maxval=None, mantissa_bits=M, set_maxval=True, learn_maxval=False, learn_mantissa_bits=False,
                                          mse_include_mantissa_bits=False, allow_unsigned=False)).eval()


def _procedure(q, calib, val):
    from quantization.base_quantized_classes import QuantizedModule

    def each(fn):
        for m in q.modules():
            if isinstance(m, QuantizedModule):
                fn(m)
    with torch.no_grad():
        fp = q(val)
        each(lambda m: m.quantized())
        cal = q(calib)
        each(lambda m: m.fix_ranges())
        out = q(val)
        wq = {n: m.get_params()[0] for n, m in q.named_modules() if hasattr(m, "weight_quantizer") and hasattr(m, "get_params")}
    return fp, cal, out, wq


def _check(g, name, tag, q, fp, cal, out, wq, conv_tol):
    from quantization.quantization_manager import QuantizationManager
    assert [type(m).__name__ for m in q] == EXPECT_CLASSES[name] == [str(c) for c in g[f"{name}_{tag}_classes"]]
    np.testing.assert_allclose(fp.cpu().numpy(), g[f"{name}_{tag}_fp_logits"], rtol=1e-4, atol=1e-5)
    names = [n for n, m in q.named_modules() if isinstance(m, QuantizationManager)]
    assert names == [str(n) for n in g[f"{name}_{tag}_mgr_names"]]
    for n, m in q.named_modules():
        if not isinstance(m, QuantizationManager):
            continue
        ref, got = g[f"{name}_{tag}_maxval_{n}"], m.quantizer.maxval.cpu().numpy()
        assert got.shape == ref.shape, (n, got.shape, ref.shape)
        if n.endswith("weight_quantizer"):
            np.testing.assert_array_equal(got, ref)                  # weights: bit-equal ranges (per OUTPUT channel for transposed convs)
        else:
            np.testing.assert_allclose(got, ref, rtol=conv_tol)     # activations: the convolution's own rounding
    for n, w in wq.items():
        ref = g[f"{name}_{tag}_wq_{n}"]
        got = w.cpu().numpy()
        assert got.shape == ref.shape
        # the reference's fp32 chain vs the arithmetic contract: <= 2 ulp, never another grid point
        np.testing.assert_allclose(got, ref, rtol=3e-7, atol=0)
    for got, ref in ((cal, g[f"{name}_{tag}_calib_logits"]), (out, g[f"{name}_{tag}_val_logits"])):
        got = got.cpu().numpy()
        scale = np.abs(ref).max()
        assert np.array_equal(got.argmax(1), ref.argmax(1))
        np.testing.assert_allclose(got, ref, rtol=0, atol=0.05 * scale)
        assert np.mean(np.abs(got - ref)) < 0.01 * scale


@pytest.mark.parametrize("tag,M,pcw", CASES)
@pytest.mark.parametrize("name", ["net1d", "net2d"])
def test_wrappers_on_oracle_backend_cpu(golden_dir, name, tag, M, pcw):
    import oracle_ops
    g = np.load(os.path.join(golden_dir, "g11_wrappers.npz"))
    q = _quantized(_nets(g)[name], M, pcw)
    with oracle_ops.patched():
        fp, cal, out, wq = _procedure(q, torch.from_numpy(g[f"{name}_calib"]), torch.from_numpy(g[f"{name}_val"]))
    _check(g, name, tag, q, fp, cal, out, wq, conv_tol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,M,pcw", CASES)
@pytest.mark.parametrize("name", ["net1d", "net2d"])
def test_wrappers_vs_reference_gpu(golden_dir, name, tag, M, pcw):
    g = np.load(os.path.join(golden_dir, "g11_wrappers.npz"))
    q = _quantized(_nets(g)[name], M, pcw).cuda()
    fp, cal, out, wq = _procedure(q, torch.from_numpy(g[f"{name}_calib"]).cuda(), torch.from_numpy(g[f"{name}_val"]).cuda())
    _check(g, name, tag, q, fp, cal, out, wq, conv_tol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("pcw", [True, False])
def test_multi_tensor_plan_covers_transposed_convolutions(golden_dir, pcw):
    """QuantizedModel.fix_ranges() -> prequantize_weights(): every FP8 weight of the model in one multi-tensor launch, the
    transposed convolutions included (quantized on the [out, in, ...] view the reference hands its quantizer,
    autoquant_utils.py:46-58) -- same tensors, bit for bit, as each layer's own quantize_weights()."""
    from quantization.base_quantized_model import QuantizedModel
    from quantization.autoquant_utils import QuantConvTranspose, QuantConvTranspose1d
    from quantization import model as qmodel
    g = np.load(os.path.join(golden_dir, "g11_wrappers.npz"))
    nets = _nets(g)

    class Both(QuantizedModel):
        def __init__(self):
            super().__init__((1, 3, 8, 8))
            self.a = _quantized(nets["net1d"], 3, pcw)
            self.b = _quantized(nets["net2d"], 3, pcw)

        def forward(self, xs):
            return self.a(xs[0]), self.b(xs[1])

    m = Both().cuda().eval()
    xs = (torch.from_numpy(g["net1d_calib"]).cuda(), torch.from_numpy(g["net2d_calib"]).cuda())
    with torch.no_grad():
        m.set_quant_state(True, True)
        m.estimate_ranges()
        m(xs)
        m.fix_ranges()
        plan, mods, _ = qmodel._PLANS[m]
        covered = [type(mod).__name__ for mod, *_ in mods]
        tconvs = [mod for mod in m.modules() if isinstance(mod, (QuantConvTranspose, QuantConvTranspose1d))]
        assert len(tconvs) == 3 and all(any(mod is t for mod, *_ in mods) for t in tconvs), covered
        n_hijack = sum(1 for mod in m.modules() if hasattr(mod, "weight_quantizer"))
        assert len(mods) == n_hijack
        for mod in m.modules():
            if hasattr(mod, "weight_quantizer"):
                cached = mod.get_params()[0]
                fresh = mod.quantize_weights(mod.weight)
                assert cached.shape == mod.weight.shape
                assert torch.equal(cached.contiguous().view(torch.int32), fresh.contiguous().view(torch.int32)), type(mod).__name__
        y1 = m(xs)
        os.environ["FP8Q_CACHE_WEIGHTS"] = "0"
        try:
            y2 = m(xs)
        finally:
            os.environ.pop("FP8Q_CACHE_WEIGHTS")
        assert all(torch.equal(a, b) for a, b in zip(y1, y2))
        # in-place weight update -> one replay of the plan refreshes the transposed layers too
        tconvs[0].weight.mul_(1.5)
        assert m.requantize_weights() == len(mods)
        fresh = tconvs[0].quantize_weights(tconvs[0].weight)
        assert torch.equal(tconvs[0].get_params()[0].contiguous().view(torch.int32), fresh.contiguous().view(torch.int32))
