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
