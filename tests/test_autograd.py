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
    mb = torch.Tensor([float(M)]).requires_grad_(True)        # learn_mantissa_bits: the width as a Parameter
    y = quantize_to_fp8_ste_MM(x, 8, mv, mb, sb)
    y.backward(torch.from_numpy(g[f"c{cid}_g"]).to(dev))
    return y.detach().cpu().numpy(), x.grad.cpu().numpy(), mv.grad.cpu().numpy(), mb.grad.cpu().numpy()


def _check(g, dev):
    for cid, M, sb, pc in g["cases"]:
        y, gx, gmv, gmb = _run(g, int(cid), int(M), int(sb), dev)
        np.testing.assert_allclose(y, g[f"c{cid}_y"], rtol=1e-6, atol=0)
        # a 0 / 0.5 / 1 mask times the upstream gradient; the reference's chain forms it as (g * s) / s: 1 ULP of noise
        np.testing.assert_array_equal(gx == 0, g[f"c{cid}_gx"] == 0)
        np.testing.assert_allclose(gx, g[f"c{cid}_gx"], rtol=2.5e-7, atol=0)
        # sums of ~300 terms, (y - xc) / maxval formed in another order than autograd's chain: fp32 rounding only
        scale = np.abs(g[f"c{cid}_g"]).sum() / g[f"c{cid}_gmaxval"].size
        np.testing.assert_allclose(gmv, g[f"c{cid}_gmaxval"], rtol=2e-5, atol=2e-6 * scale)
        # d/dmbits: one sum over all elements of g (y - xc) ln2 (-1 - bias'(M)) -- the reference's chain goes through
        # 2^E, log2(2 - 2^-M) and the scale exponent in fp32 (fp8_quantizer.py:105-130)
        np.testing.assert_allclose(gmb, g[f"c{cid}_gmbits"], rtol=2e-4, atol=2e-5 * np.abs(g[f"c{cid}_g"]).sum())
    # non-integer / out-of-range widths: round_ste passes the gradient, the clamp cuts it off
    from quantization.quantizers.fp8_quantizer import quantize_to_fp8_ste_MM
    for k, (mbv, want) in enumerate(g["mb_cases"]):
        mb = torch.Tensor([mbv]).requires_grad_(True)
        y = quantize_to_fp8_ste_MM(torch.from_numpy(g["mb_x"]).to(dev), 8, torch.Tensor([1.3]).to(dev), mb, 1)
        y.backward(torch.from_numpy(g["mb_g"]).to(dev))
        if mbv < 6.5:      # (E = 0 formats put every clipped element on an exact tie: maxval / s_1 = 2^M - 1/2 -- which way it
            #  rounds is decided by the last bit of the reference's `pow`; the gradient check stays)
            np.testing.assert_allclose(y.detach().cpu().numpy(), g[f"mb{k}_y"], rtol=1e-6)
        if mbv < 6.5 or want == 0:
            np.testing.assert_allclose(float(mb.grad[0]), want, rtol=2e-4, atol=2e-5 * np.abs(g["mb_g"]).sum() if want else 0)
        else:              # (the tie-decided elements carry a whole grid step each in (y - xc): only finiteness is comparable)
            assert np.isfinite(float(mb.grad[0]))


def test_backward_on_oracle_backend_cpu(golden_dir):
    """host-side autograd logic with the CPU oracle substituted for the HIP forward (test-only)"""
    import oracle_ops
    g = np.load(os.path.join(golden_dir, "g10_autograd.npz"))
    with oracle_ops.patched():
        _check(g, "cpu")


@pytest.mark.gpu
def test_backward_hip(golden_dir):
    _check(np.load(os.path.join(golden_dir, "g10_autograd.npz")), "cuda")


def test_learn_maxval_and_mantissa_bits_get_gradients(golden_dir):
    import oracle_ops
    from quantization.quantizers.fp8_quantizer import FPQuantizer
    q = FPQuantizer(n_bits=8, mantissa_bits=3, maxval=1.5, set_maxval=True)
    q.learn_maxval()
    assert isinstance(q.maxval, torch.nn.Parameter)
    x = torch.randn(4, 16) * 2
    with oracle_ops.patched():
        q(x).sum().backward()
    assert q.maxval.grad is not None and torch.isfinite(q.maxval.grad).all() and float(q.maxval.grad.abs().sum()) > 0
    # learn_mantissa_bits (fp8_quantizer.py:253-255): the width becomes a Parameter of the module and receives a gradient
    q2 = FPQuantizer(n_bits=8, mantissa_bits=3, maxval=1.5, set_maxval=True, learn_mantissa_bits=True)
    q2.make_range_trainable()
    assert isinstance(q2.mantissa_bits, torch.nn.Parameter) and "mantissa_bits" in dict(q2.named_parameters())
    assert "mantissa_bits" in q2.state_dict()
    with oracle_ops.patched():
        q2(x).sum().backward()
    assert q2.mantissa_bits.grad is not None and torch.isfinite(q2.mantissa_bits.grad).all()
    assert float(q2.mantissa_bits.grad.abs().sum()) > 0
    # assigning a value while the width is being learned (the MSE estimator's vote in estimate_ranges_train) keeps the
    # Parameter and replaces its value (nn.Module.__setattr__ alone would raise TypeError here)
    q2.mantissa_bits = torch.Tensor([4.0])
    assert isinstance(q2.mantissa_bits, torch.nn.Parameter) and float(q2.mantissa_bits) == 4.0
    assert q2.mantissa_bits.device == q2.maxval.device
    q2.mantissa_bits = torch.Tensor([3.0])
    q2.fix_ranges()                                    # back to a plain tensor, as parameter_to_fixed does
    assert not isinstance(q2.mantissa_bits, torch.nn.Parameter) and float(q2.mantissa_bits) == 3.0
    assert "mantissa_bits" not in dict(q2.named_parameters())
    import copy
    q3 = copy.deepcopy(q2)
    assert float(q3.mantissa_bits) == 3.0
