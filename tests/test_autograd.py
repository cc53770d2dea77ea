"""Backward of the FP8 fake-quantizer (STE + clamp + the scale's dependence on maxval) against the reference's autograd
(g10_autograd.npz, produced by tests/golden/make_golden.py:make_g10 from fp8_quantizer.py:105-133)."""
import os

import numpy as np
import pytest
import torch


def _run(g, cid, M, sb, dev):
    from quantization.quantizers.fp8_quantizer import quantize_to_fp8_ste_MM
    x = torch.from_numpy(g[f"c{cid}_x"]).to(dev).requires_grad_(True)
    mv = torch.from_numpy(g[f"c{cid}_maxval"]).to(dev).requires_grad_(True)
    y = quantize_to_fp8_ste_MM(x, 8, mv, torch.Tensor([float(M)]), sb)
    y.backward(torch.from_numpy(g[f"c{cid}_g"]).to(dev))
    return y.detach().cpu().numpy(), x.grad.cpu().numpy(), mv.grad.cpu().numpy()


def _check(g, dev):
    for cid, M, sb, pc in g["cases"]:
        y, gx, gmv = _run(g, int(cid), int(M), int(sb), dev)
        np.testing.assert_allclose(y, g[f"c{cid}_y"], rtol=1e-6, atol=0)
        # a 0 / 0.5 / 1 mask times the upstream gradient; the reference's chain forms it as (g * s) / s: 1 ULP of noise
        np.testing.assert_array_equal(gx == 0, g[f"c{cid}_gx"] == 0)
        np.testing.assert_allclose(gx, g[f"c{cid}_gx"], rtol=2.5e-7, atol=0)
        # sums of ~300 terms, (y - xc) / maxval formed in another order than autograd's chain: fp32 rounding only
        scale = np.abs(g[f"c{cid}_g"]).sum() / g[f"c{cid}_gmaxval"].size
        np.testing.assert_allclose(gmv, g[f"c{cid}_gmaxval"], rtol=2e-5, atol=2e-6 * scale)


def test_backward_on_oracle_backend_cpu(golden_dir):
    """host-side autograd logic with the CPU oracle substituted for the HIP forward (test-only)"""
    import oracle_ops
    g = np.load(os.path.join(golden_dir, "g10_autograd.npz"))
    with oracle_ops.patched():
        _check(g, "cpu")


@pytest.mark.gpu
def test_backward_hip(golden_dir):
    _check(np.load(os.path.join(golden_dir, "g10_autograd.npz")), "cuda")


def test_learn_maxval_gets_a_gradient_and_mantissa_bits_refuse(golden_dir):
    import oracle_ops
    from quantization.quantizers.fp8_quantizer import FPQuantizer
    q = FPQuantizer(n_bits=8, mantissa_bits=3, maxval=1.5, set_maxval=True)
    q.learn_maxval()
    assert isinstance(q.maxval, torch.nn.Parameter)
    x = torch.randn(4, 16) * 2
    with oracle_ops.patched():
        q(x).sum().backward()
    assert q.maxval.grad is not None and torch.isfinite(q.maxval.grad).all() and float(q.maxval.grad.abs().sum()) > 0
    with pytest.raises(NotImplementedError):
        q.learn_mantissa_bits()
