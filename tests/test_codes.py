"""N3: FP8 storage codes.  CPU: the oracle's codec against the reference's grid enumerator.
GPU: HIP encode/decode bit-exact vs the oracle, decode(encode(x)) == quantize(x)."""
import os

import numpy as np
import pytest
import torch

import oracle


def _default_maxval(M):
    E = 7 - M
    return float((2 - 2.0 ** (-M)) * 2.0 ** (2 ** E - 1 - 2 ** (E - 1)))


@pytest.mark.parametrize("M", [2, 3, 4, 5])
def test_all_codes_decode_to_the_reference_grid(golden_dir, M):
    """With the default maxval the bias is the integer 2^(E-1): the 256 decoded codes are exactly the
    values the reference's generate_all_values_fp enumerates (g2 golden), in sign-magnitude order."""
    E = 7 - M
    g2 = np.load(os.path.join(golden_dir, "g2_grids.npz"))
    ref = g2[f"e{E}_b{2 ** (E - 1)}"]
    codes = np.arange(256, dtype=np.uint8)
    vals = oracle.c_decode(codes, [_default_maxval(M)], M)
    # the reference forms `bias` with three fp32 roundings: for some formats (E3M4: maxval 15.5) it comes
    # out as 4.0000005 instead of 4, so the emulated grid sits 4e-7 (relative) off the ideal one -- in the
    # reference's quantizer as well.  Hence rtol, not equality.
    np.testing.assert_allclose(np.sort(vals.astype(np.float64)), ref, rtol=1e-6, atol=0)
    mag = vals[:128]
    assert np.all(np.diff(mag) > 0) and vals[0] == 0.0                 # codes 0..127 ascend from +0
    assert np.array_equal(vals[128:], -mag) and np.signbit(vals[128])    # 128..255 mirror them, 128 = -0
    # code -> value -> code is the identity
    np.testing.assert_array_equal(oracle.c_encode(vals, [_default_maxval(M)], M), codes)


def test_oracle_roundtrip_equals_quantize():
    rng = np.random.RandomState(0)
    for M, mv, sb in ((2, 57344.0, 1), (3, 0.7361, 1), (5, 3.0, 1), (3, 2.5, 0), (1, 0.31, 0), (6, 1.0, 1), (7, 1.0, 0)):
        x = (rng.randn(20000) * mv / 2.2).astype(np.float32)
        x[:6] = [0.0, -0.0, mv, -mv, mv * 3, 1e-30]
        q = oracle.c_quantize(x, [mv], M, 8, sb)
        rt = oracle.c_decode(oracle.c_encode(x, [mv], M, 8, sb), [mv], M, 8, sb)
        assert np.array_equal(rt.view(np.int32), q.view(np.int32)), (M, mv, sb)


def test_formats_without_an_exponent_bit_are_not_encodable():
    """E = n_bits - sign - M = 0 (a uniform grid): K1 supports it, but a value that rounds up to 2^M steps has no
    code in M fraction bits -- encode / decode refuse instead of emitting colliding codes."""
    x = np.linspace(-1, 1, 64, dtype=np.float32)
    with pytest.raises(AssertionError):
        oracle.c_encode(x, [1.0], 7, 8, 1)
    assert oracle.c_quantize(x, [1.0], 7, 8, 1).shape == x.shape      # K1 itself is fine


def test_roundtrip_on_non_geometric_scale_tables():
    """Where the reference's fp32 (p - M) - bias rounds differently in neighbouring binades (small |bias|, e.g.
    E3M4 with maxval 207), s_(p+1) != 2 s_p in the last bit.  K1 renders an element that rounds UP into the next
    binade as 2^(M+1) s_p, the decoder (like the reference's enumerator: one value per code) as 2^M s_(p+1): the same
    grid point, a few fp32 ULP apart (ulp(k - bias) ln 2 relative); everything else round-trips bit for bit.  Found by a random-geometry soak (round 2; git history)."""
    rng = np.random.RandomState(3)
    worst = 0
    for M, mv, sb in ((4, 207.11018, 1), (5, 7.1219254, 1), (6, 12.418949, 0), (4, 305.43573, 1)):
        x = (rng.randn(200000) * mv / 2.5).astype(np.float32)
        if sb == 0:
            x = np.abs(x)
        q = oracle.c_quantize(x, [mv], M, 8, sb)
        rt = oracle.c_decode(oracle.c_encode(x, [mv], M, 8, sb), [mv], M, 8, sb)
        ulp = np.abs(rt.view(np.int32).astype(np.int64) - q.view(np.int32).astype(np.int64))
        assert ulp.max() <= 8, (M, mv, int(ulp.max()))     # < 1e-6 relative
        worst = max(worst, int(ulp.max()))
        # the differing elements sit exactly on binade tops: |q| / s is a power of two there
        diff = ulp > 0
        if diff.any():
            grid = np.unique(np.abs(rt[diff]))
            assert len(grid) <= 8            # a handful of grid points (one per affected binade), many elements
    assert worst >= 1                        # the fixture does exercise the caveat


pytestmark_gpu = pytest.mark.gpu


def dev(a, dtype=np.float32):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("M,sb,mv", [(2, 1, 57344.0), (3, 1, 240.0), (3, 1, 0.7361), (5, 1, 3.0), (3, 0, 2.5),
                                     (1, 0, 0.31), (6, 1, 1.0), (7, 0, 1.0)])
@pytest.mark.parametrize("n", [5, 16, 1000, 4096 + 7, 1 << 20])
def test_hip_codec_per_tensor(M, sb, mv, n):
    import fp8q
    ops = fp8q.ops
    rng = np.random.RandomState(n % 977 + M)
    x = (rng.randn(n) * mv / 2.2).astype(np.float32)
    x[:5] = [0.0, -0.0, mv, -mv * 2, 1e-41]
    xd, mvd = dev(x), dev([mv])
    codes = ops.encode(xd, mvd, M, 8, sb)
    assert codes.dtype == torch.uint8 and codes.shape == xd.shape
    np.testing.assert_array_equal(codes.cpu().numpy(), oracle.c_encode(x, [mv], M, 8, sb))
    y = ops.decode(codes, mvd, M, 8, sb)
    q = ops.quantize(xd, mvd, M, 8, sb)
    assert torch.equal(y.view(torch.int32), q.view(torch.int32))       # decode(encode(x)) == quantize(x)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(64, 3, 7, 7), (1000, 512), (33, 4099), (512, 512, 3, 3)])
def test_hip_codec_per_channel(shape):
    import fp8q
    ops = fp8q.ops
    rng = np.random.RandomState(shape[0])
    mv = (np.abs(rng.randn(shape[0])) + 0.05).astype(np.float32)
    x = (rng.randn(*shape) * (mv.reshape([-1] + [1] * (len(shape) - 1)) / 2)).astype(np.float32)
    xd, mvd = dev(x), dev(mv)
    codes = ops.encode(xd, mvd, 2, 8, 1)
    np.testing.assert_array_equal(codes.cpu().numpy(), oracle.c_encode(x, mv, 2, 8, 1))
    y = ops.decode(codes, mvd, 2, 8, 1)
    assert torch.equal(y.view(torch.int32), ops.quantize(xd, mvd, 2, 8, 1).view(torch.int32))
    np.testing.assert_array_equal(y.cpu().numpy().view(np.int32), oracle.c_decode(codes.cpu().numpy(), mv, 2, 8, 1).view(np.int32))


R18_SHAPES = [(64, 3, 7, 7)] + [(64, 64, 3, 3)] * 2 + [(128, 64, 3, 3), (128, 64, 1, 1), (256, 256, 3, 3), (512, 256, 1, 1),
                                                       (512, 512, 3, 3), (1000, 512)]


@pytest.mark.gpu
@pytest.mark.parametrize("M,sb", [(2, 1), (3, 1), (3, 0), (5, 1)])
def test_hip_multi_tensor_codec_equals_the_single_tensor_entry_points(M, sb):
    """fp8q_multi_minmax_encode_u8 / fp8q_multi_decode_u8 (k_multi_flat<3> / <4>): a model's weight tensors -- odd row
    lengths, rows that straddle 16-byte groups and chunks, ragged tails, > 32 tensors (two launches), a per-tensor range,
    unaligned views (single-tensor fall-back inside the call) -- bit-identical to minmax + encode / decode per tensor"""
    import fp8q
    ops = fp8q.ops
    rng = np.random.RandomState(M * 10 + sb)
    shapes = R18_SHAPES + [(5, 7), (33, 4099), (3, 5, 5, 5), (1, 9)] + [(16, 10)] * 30
    xs = [dev((rng.randn(*sh) * rng.uniform(0.01, 3)).astype(np.float32)) for sh in shapes]
    if sb == 0:
        xs = [x.abs() for x in xs]
    xs[3][5] = 0.0                                                   # an all-zero channel: maxval 0 -> code 0 everywhere
    mvs = [torch.empty(x.shape[0], device="cuda") for x in xs]
    codes = ops.multi_minmax_encode([(x, mv, M, 8, sb) for x, mv in zip(xs, mvs)])
    for x, mv, c in zip(xs, mvs, codes):
        rmv = ops.minmax(x, True, want_maxval=True)[2]
        assert torch.equal(mv.view(torch.int32), rmv.view(torch.int32))
        assert c.dtype == torch.uint8 and torch.equal(c, ops.encode(x, rmv, M, 8, sb)), tuple(x.shape)
    ys = ops.multi_decode([(c, mv, M, 8, sb) for c, mv in zip(codes, mvs)])
    for x, mv, c, y in zip(xs, mvs, codes, ys):
        want = ops.decode(c, mv, M, 8, sb)
        assert torch.equal(torch.isnan(y), torch.isnan(want)) and torch.equal(y[~torch.isnan(y)].view(torch.int32), want[~torch.isnan(want)].view(torch.int32))
    # views into packed buffers at odd offsets (what the bucketed all-gather decodes from): codes 1-byte aligned, values 4-byte
    buf = torch.zeros(3 + 33 * 4099, dtype=torch.uint8, device="cuda")
    cv = buf[3:].view(33, 4099)
    cv.copy_(codes[len(R18_SHAPES) + 1])
    out = torch.zeros(1 + 33 * 4099, device="cuda")[1:].view(33, 4099)
    ops.multi_decode([(cv, mvs[len(R18_SHAPES) + 1], M, 8, sb, out)])
    assert torch.equal(out.view(torch.int32), ys[len(R18_SHAPES) + 1].view(torch.int32))
    # a per-tensor range through the multi-tensor decode
    pt = dev((rng.randn(4, 1000) * 2).astype(np.float32)).abs() if sb == 0 else dev((rng.randn(4, 1000) * 2).astype(np.float32))
    mv1 = torch.tensor([2.5], device="cuda")
    c1 = ops.encode(pt, mv1, M, 8, sb)
    assert torch.equal(ops.multi_decode([(c1, mv1, M, 8, sb)])[0].view(torch.int32), ops.decode(c1, mv1, M, 8, sb).view(torch.int32))


@pytest.mark.gpu
def test_bucketed_weight_exchange_with_codes_on_the_wire_single_process():
    """fp8q.dist.quantize_weights_sharded_bucketed(wire="codes") on the HIP ops (one rank, no process group): the packed
    byte buffer, two launches in, one multi-tensor decode out -- the fp32 wire form's bits"""
    import fp8q
    from fp8q import dist as fd
    rng = np.random.RandomState(5)
    ws = [dev((rng.randn(*sh) * 0.05).astype(np.float32)) for sh in R18_SHAPES + [(1, 7), (13, 3, 3, 3)]]
    a = fd.quantize_weights_sharded_bucketed(ws, 2, 8, 1, wire="codes")
    b = fd.quantize_weights_sharded_bucketed(ws, 2, 8, 1, wire="fp32")
    for (qa, ma), (qb, mb), w in zip(a, b, ws):
        assert qa.shape == w.shape and torch.equal(ma.view(torch.int32), mb.view(torch.int32))
        assert torch.equal(qa.view(torch.int32), qb.view(torch.int32))
    c = fd.quantize_weights_sharded_bucketed(ws, 2, 8, 1, wire="codes", bucket_bytes=1 << 16)     # several buckets
    for (qa, ma), (qc, mc) in zip(a, c):
        assert torch.equal(qa.view(torch.int32), qc.view(torch.int32)) and torch.equal(ma, mc)


@pytest.mark.gpu
def test_hip_codec_short_row_geometries():
    """Per-channel tensors with short rows take the chunked kernel (k_rows_flat encode / decode modes): odd row
    lengths whose rows straddle 16-byte groups and 16 KiB chunks, a ragged tail, tiles of several chunks, every
    format with an exponent bit, unsigned, NaN / inf / -0 inputs -- codes and decoded values equal to the oracle's."""
    import fp8q
    ops = fp8q.ops
    rng = np.random.RandomState(21)
    cases = [(1, 147), (27, 147), (29, 147), (1000, 147), (70001, 147), (333, 255), (7001, 99), (9001, 70), (50021, 41),
             (40000, 5), (77777, 4), (3, 2047), (4099, 27), (20000, 9), (1 << 17, 36)]
    for i, (C, inner) in enumerate(cases):
        M, sb = 1 + i % 6, 1 if i % 4 else 0
        mv = (np.abs(rng.randn(C)) * 2 + 0.05).astype(np.float32)
        x = (rng.randn(C, inner) * (mv[:, None] / 2)).astype(np.float32)
        if sb == 0:
            x = np.abs(x)
        x.reshape(-1)[:4] = [np.nan, np.inf, -0.0, -np.inf]
        xd, mvd = dev(x), dev(mv)
        codes = ops.encode(xd, mvd, M, 8, sb)
        np.testing.assert_array_equal(codes.cpu().numpy(), oracle.c_encode(x, mv, M, 8, sb), err_msg=f"encode {C}x{inner} M={M}")
        y = ops.decode(codes, mvd, M, 8, sb).cpu().numpy()
        ref = oracle.c_decode(codes.cpu().numpy(), mv, M, 8, sb)
        assert np.array_equal(y.view(np.int32), ref.view(np.int32)), f"decode {C}x{inner} M={M}"


@pytest.mark.gpu
def test_hip_codec_nan_and_degenerate():
    import fp8q
    ops = fp8q.ops
    x = np.array([[1.0, np.nan, -2.0, 0.5], [0.0, 0.0, 0.0, 0.0]], np.float32)
    mv = np.array([2.0, 0.0], np.float32)                              # second channel: NaN in K1
    codes = ops.encode(dev(x), dev(mv), 3, 8, 1).cpu().numpy()
    np.testing.assert_array_equal(codes, oracle.c_encode(x, mv, 3, 8, 1))
    assert codes[0, 1] == 0 and (codes[1] == 0).all()                  # no NaN code: documented as 0
    with pytest.raises(Exception):                                     # no exponent bit: refused (EUNSUPPORTED)
        ops.encode(dev(x), dev(mv), 7, 8, 1)
    # a denormal maxval makes every scale underflow to 0 (K1: NaN everywhere): code 0, like the oracle
    mvd = np.array([2.0, 1e-45], np.float32)
    xd = np.array([[1.0, -1.0, 0.25, 3.0], [1.0, -1.0, np.inf, 1e-45]], np.float32)
    cd = ops.encode(dev(xd), dev(mvd), 3, 8, 1).cpu().numpy()
    np.testing.assert_array_equal(cd, oracle.c_encode(xd, mvd, 3, 8, 1))
    assert (cd[1] == 0).all()
    # decoding a degenerate channel follows the reference chain's 2^(k - bias): 0 for maxval 0 (bias = +inf)
    dec = ops.decode(dev(codes, np.uint8), dev(mv), 3, 8, 1).cpu().numpy()
    np.testing.assert_array_equal(dec.view(np.int32), oracle.c_decode(codes, mv, 3, 8, 1).view(np.int32))
    for bad in (np.inf, np.nan):
        mvb = np.array([2.0, bad], np.float32)
        cb = ops.encode(dev(x), dev(mvb), 3, 8, 1)
        got = ops.decode(cb, dev(mvb), 3, 8, 1).cpu().numpy()
        ref = oracle.c_decode(cb.cpu().numpy(), mvb, 3, 8, 1)
        assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(got[~np.isnan(ref)], ref[~np.isnan(ref)])


@pytest.mark.gpu
def test_export_fp8_weights_of_a_model(golden_dir):
    """Model level: after calibration the weights of every FP8 layer can be exported as 1-byte codes + ranges;
    decoding them gives exactly the tensors the layers compute with (4x smaller than the fp32 checkpoint)."""
    import os
    import torch.nn as nn
    from quantization.autoquant_utils import quantize_model
    from quantization.base_quantized_classes import QuantizedModule
    from quantization.base_quantized_model import export_fp8_weights, decode_fp8_weights
    from quantization.hijacker import QuantizationHijacker
    from quantization.quantization_manager import QMethods
    from quantization.range_estimators import RangeEstimators
    g7 = np.load(os.path.join(golden_dir, "g7_tinycnn.npz"))
    net = nn.Sequential(nn.Conv2d(3, 16, 3, padding=1, bias=False), nn.BatchNorm2d(16), nn.ReLU(),
                        nn.Conv2d(16, 24, 3, stride=2, padding=1, bias=True), nn.ReLU6(),
                        nn.Conv2d(24, 24, 3, padding=1, groups=24, bias=False), nn.BatchNorm2d(24), nn.ReLU(),
                        nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(24, 10))
    net.load_state_dict({k[3:]: torch.from_numpy(g7[k]) for k in g7.files if k.startswith("sd_")})
    q = quantize_model(net.eval(), method=QMethods.fp_quantizer.cls,
                       weight_range_method=RangeEstimators.current_minmax.cls,
                       act_range_method=RangeEstimators.allminmax.cls, n_bits=8, per_channel_weights=True,
                       fp8_kwargs=dict(maxval=None, mantissa_bits=3, set_maxval=True)).eval().cuda()
    with torch.no_grad():
        for m in q.modules():
            if isinstance(m, QuantizedModule):
                m.quantized()
        q(torch.from_numpy(g7["calib"]).cuda())
        for m in q.modules():
            if isinstance(m, QuantizedModule):
                m.fix_ranges()
        exported = export_fp8_weights(q)
        layers = {n: m for n, m in q.named_modules() if isinstance(m, QuantizationHijacker)}
        assert set(exported) == set(layers) and len(exported) == 4
        decoded = decode_fp8_weights(exported)
        for name, m in layers.items():
            assert exported[name]["codes"].dtype == torch.uint8 and exported[name]["codes"].shape == m.weight.shape
            assert torch.equal(decoded[name], m.get_params()[0]), name
