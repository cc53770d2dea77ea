"""allow_unsigned without a host round trip: FPQuantizer.set_quant_range decides sign_bits on the device (fp8q_sign_fold_u8
for the reference's `if allow_unsigned and torch.all(x_min >= 0)`, fp8_quantizer.py:216-225), K1 reads the flag
(fp8q_quantize_ds_f32) until the host asks for the attribute."""
import pytest
import torch

pytestmark = pytest.mark.gpu


class _NoSync:
    def __enter__(self):
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")

    def __exit__(self, *exc):
        torch.cuda.set_sync_debug_mode("default")
        return False


def test_sign_fold_follows_the_references_comparison():
    import fp8q
    ops = fp8q.ops
    nan = float("nan")
    cases = [([0.0, 1.0, 2.0], 0), ([-0.0, 3.0], 0), ([-1e-30, 1.0], 1), ([nan, 1.0], 1), ([0.5], 0), ([-0.5], 1), ([], 0),
             ([float("inf")], 0), ([-float("inf"), 1.0], 1)]
    for vals, want in cases:
        xm = torch.tensor(vals, dtype=torch.float32, device="cuda")
        flag = ops.sign_fold(xm)
        ref = 0 if bool(torch.all(xm.cpu() >= 0)) else 1            # the reference's decision
        assert ref == want and int(flag.item()) == want, (vals, int(flag.item()), want)
    # sticky: once unsigned, stays (the reference never sets sign_bits back to 1)
    flag = ops.sign_fold(torch.tensor([-1.0], device="cuda"))
    assert int(flag.item()) == 1
    ops.sign_fold(torch.tensor([0.0, 2.0], device="cuda"), flag)
    assert int(flag.item()) == 0
    ops.sign_fold(torch.tensor([-3.0], device="cuda"), flag)
    assert int(flag.item()) == 0
    # more values than one pass of the workgroup
    xm = torch.rand(5000, device="cuda")
    assert int(ops.sign_fold(xm).item()) == 0
    xm[4321] = -1.0
    assert int(ops.sign_fold(xm).item()) == 1


@pytest.mark.parametrize("shape,per_channel", [((8, 16, 28, 28), False), ((64, 32, 56, 56), False), ((3,), False), ((70001,), False),
                                               ((64, 3, 7, 7), True), ((1280, 320), True), ((5, 70000), True), ((6, 4099), True)])
def test_k1_with_the_sign_in_device_memory(shape, per_channel):
    import fp8q
    ops = fp8q.ops
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(*shape, generator=g) * 2.0).cuda()
    mv = ops.minmax(x, per_channel, want_maxval=True)[2] * 0.8
    for m in (1.0, 3.0, 5.4, 7.0):
        for sb in (1, 0):
            flag = torch.tensor([sb], dtype=torch.uint8, device="cuda")
            want = ops.quantize(x, mv, m, 8, sb)
            got = ops.quantize(x, mv, m, 8, flag)
            assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (shape, m, sb)
    xo = x.reshape(-1)[1:]                   # not 16-byte aligned
    for sb in (1, 0):
        flag = torch.tensor([sb], dtype=torch.uint8, device="cuda")
        assert torch.equal(ops.quantize(xo, mv[:1], 3.0, 8, flag), ops.quantize(xo, mv[:1], 3.0, 8, sb))


def _ref_signs(x_mins):
    """sign_bits after every set_quant_range of the sequence (fp8_quantizer.py:216-225)"""
    out, s = [], 1
    for xm in x_mins:
        if bool(torch.all(xm >= 0)):
            s = 0
        out.append(s)
    return out


@pytest.mark.parametrize("est_name", ["current_minmax", "allminmax", "running_minmax"])
@pytest.mark.parametrize("per_channel", [False, True])
@pytest.mark.parametrize("seq", ["signed", "relu", "signed_then_relu", "relu_then_signed"])
def test_minmax_calibration_with_allow_unsigned_is_sync_free(est_name, per_channel, seq):
    from quantization.quantizers.fp8_quantizer import FPQuantizer
    from quantization.range_estimators import RangeEstimators
    from quantization.quantization_manager import QuantizationManager
    import fp8q
    ops = fp8q.ops
    torch.manual_seed(5)
    shape = (24, 3, 5, 5) if per_channel else (4, 16, 14, 14)
    raw = [torch.randn(shape, device="cuda") * (1 + i) for i in range(3)]
    kinds = {"signed": "sss", "relu": "rrr", "signed_then_relu": "srr", "relu_then_signed": "rss"}[seq]
    batches = [torch.relu(b) if k == "r" else b for b, k in zip(raw, kinds)]
    mgr = QuantizationManager(qmethod=FPQuantizer, init=RangeEstimators[est_name].cls, per_channel=per_channel,
                              qparams=dict(n_bits=8, mantissa_bits=3, set_maxval=True, allow_unsigned=True))
    q = mgr.quantizer
    outs, mins, maxvals = [], [], []
    with _NoSync():
        for b in batches:
            outs.append(mgr(b))
            mins.append(mgr.range_estimator.current_xmin.detach().clone())
            maxvals.append(q.maxval.detach().clone())
    assert q._pending_sign_bits() is not None
    signs = _ref_signs([m.cpu() for m in mins])
    if est_name == "current_minmax":
        assert signs == {"sss": [1, 1, 1], "rrr": [0, 0, 0], "srr": [1, 0, 0], "rss": [0, 0, 0]}[kinds]
    for b, y, mv, sb in zip(batches, outs, maxvals, signs):
        want = ops.quantize(b, mv.reshape(-1), 3.0, 8, sb)
        assert torch.equal(y.view(torch.int32), want.view(torch.int32)), (est_name, per_channel, seq, sb)
    assert q.sign_bits == signs[-1]                  # (this read brings the flag over)
    assert q._pending_sign_bits() is None
    mgr.fix_ranges()
    with _NoSync():
        y = mgr(batches[0])
    assert torch.equal(y, ops.quantize(batches[0], q.maxval.reshape(-1), 3.0, 8, signs[-1]))


def test_an_assignment_of_sign_bits_ends_the_pending_state():
    from quantization.quantizers.fp8_quantizer import FPQuantizer
    q = FPQuantizer(n_bits=8, mantissa_bits=3, set_maxval=True, allow_unsigned=True)
    q.maxval = q.maxval.cuda()
    e0 = q._range_epoch
    q.set_quant_range(torch.tensor(0.0, device="cuda"), torch.tensor(2.0, device="cuda"))
    assert q._pending_sign_bits() is not None and q._range_epoch > e0
    q.sign_bits = 1
    assert q._pending_sign_bits() is None and q.sign_bits == 1
    # python numbers and host tensors decide on the host, as before
    q.set_quant_range(0.0, 2.0)
    assert q._pending_sign_bits() is None and q.sign_bits == 0
    q2 = FPQuantizer(n_bits=8, mantissa_bits=3, set_maxval=True, allow_unsigned=False)
    q2.maxval = q2.maxval.cuda()
    q2.set_quant_range(torch.tensor(0.0, device="cuda"), torch.tensor(2.0, device="cuda"))
    assert q2._pending_sign_bits() is None and q2.sign_bits == 1


def test_model_fix_ranges_brings_all_pending_signs_over_in_one_copy():
    from quantization.model import materialize_sign_bits
    from quantization.quantizers.fp8_quantizer import FPQuantizer
    net = torch.nn.ModuleList([FPQuantizer(n_bits=8, mantissa_bits=3, set_maxval=True, allow_unsigned=True) for _ in range(5)])
    for i, q in enumerate(net):
        q.maxval = q.maxval.cuda()
        if i != 2:
            q.set_quant_range(torch.tensor([0.0 if i % 2 else -1.0], device="cuda"), torch.tensor([2.0], device="cuda"))
    assert materialize_sign_bits(net) == 4
    assert materialize_sign_bits(net) == 0
    with _NoSync():
        assert [q.sign_bits for q in net] == [1, 0, 1, 0, 1]


def test_quantized_model_with_allow_unsigned_calibrates_without_a_round_trip():
    """ResNet-18-style settings with --allow-unsigned: the calibration forward of a small quantized net issues no host
    synchronisation, and ends with the ranges / signs the host-decided flow gives"""
    from quantization.autoquant_utils import quantize_model
    from quantization.base_quantized_classes import QuantizedModule
    from quantization.quantizers.fp8_quantizer import FPQuantizer
    from quantization.quantization_manager import QMethods, QuantizationManager
    from quantization.range_estimators import RangeEstimators
    import fp8q
    torch.manual_seed(3)
    nn = torch.nn
    net = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1, bias=False), nn.BatchNorm2d(8), nn.ReLU(),
                        nn.Conv2d(8, 8, 3, padding=1, bias=True), nn.ReLU6(), nn.Conv2d(8, 4, 1)).eval()
    qnet = quantize_model(net, method=QMethods.fp_quantizer.cls, weight_range_method=RangeEstimators.current_minmax.cls,
                          act_range_method=RangeEstimators.allminmax.cls, n_bits=8, per_channel_weights=True,
                          fp8_kwargs=dict(maxval=None, mantissa_bits=3, set_maxval=True, allow_unsigned=True)).eval().cuda()
    x = torch.randn(4, 3, 12, 12, device="cuda")
    qmods = [m for m in qnet.modules() if isinstance(m, QuantizedModule)]
    mgrs = [m for m in qnet.modules() if isinstance(m, QuantizationManager) and isinstance(m.quantizer, FPQuantizer)]

    def calibrate():
        for m in qmods:
            m.quantized()
            m.estimate_ranges()
        for m in mgrs:
            m.reset_ranges()
            m.quantizer.sign_bits = 1
        return qnet(x)

    with torch.no_grad():
        calibrate()                              # (allocations, code objects)
        for m in qmods:
            m.quantized()
            m.estimate_ranges()
        for m in mgrs:
            m.reset_ranges()
            m.quantizer.sign_bits = 1
        with _NoSync():
            y = qnet(x)
        assert any(m.quantizer._pending_sign_bits() is not None for m in mgrs)
        signs = [m.quantizer.sign_bits for m in mgrs]
        assert 0 in signs and 1 in signs         # post-ReLU activations unsigned, weights signed
        # the same pass with every sign decided on the host (the reference's flow)
        saved = fp8q.ops.sign_fold
        try:
            del fp8q.ops.sign_fold
            y2 = calibrate()
        finally:
            fp8q.ops.sign_fold = saved
    assert [m.quantizer.sign_bits for m in mgrs] == signs
    assert torch.equal(y.view(torch.int32), y2.view(torch.int32))


@pytest.mark.parametrize("per_channel", [False, True])
@pytest.mark.parametrize("search", [True, False])
@pytest.mark.parametrize("seq", ["signed", "relu", "signed_then_relu", "relu_then_signed"])
def test_mse_estimator_with_allow_unsigned_is_sync_free(per_channel, search, seq):
    """FP_MSE_Estimator + allow_unsigned: the reference's `int(torch.any(x < 0))` per batch (range_estimators.py:333) as a device
    flag -- three batches enqueue without a host round trip and leave the tables, the winner, the sign and the quantized
    batches of the flow that decides on the host."""
    from quantization.quantizers.fp8_quantizer import FPQuantizer
    from quantization.range_estimators import FP_MSE_Estimator
    from quantization.quantization_manager import QuantizationManager
    import fp8q
    ops = fp8q.ops
    torch.manual_seed(11)
    shape = (24, 3, 5, 5) if per_channel else (4, 16, 14, 14)
    raw = [torch.randn(shape, device="cuda") * (1 + i) for i in range(3)]
    kinds = {"signed": "sss", "relu": "rrr", "signed_then_relu": "srr", "relu_then_signed": "rss"}[seq]
    batches = [torch.relu(b) if k == "r" else b for b, k in zip(raw, kinds)]
    ops.mse_linspace(torch.ones(1, device="cuda"))       # the once-per-process self-check synchronises

    def make():
        return QuantizationManager(qmethod=FPQuantizer, init=FP_MSE_Estimator, per_channel=per_channel,
                                   qparams=dict(n_bits=8, mantissa_bits=3, set_maxval=True, mse_include_mantissa_bits=search,
                                                allow_unsigned=True))

    mgr = make()
    [mgr(b) for b in batches]                            # (allocations, code objects)
    mgr = make()
    with _NoSync():
        outs = [mgr(b) for b in batches]
    q, est = mgr.quantizer, mgr.range_estimator
    assert q._pending_sign_bits() is not None
    # the host-decided flow (the reference's): the same estimator without the device flag
    saved = ops.sign_fold
    try:
        del ops.sign_fold
        ref = make()
        want = [ref(b) for b in batches]
    finally:
        ops.sign_fold = saved
    rq, rest = ref.quantizer, ref.range_estimator
    assert q.sign_bits == rq.sign_bits == {"sss": 1, "rrr": 0, "srr": 0, "rss": 0}[kinds]
    assert q._pending_sign_bits() is None
    assert torch.equal(est.search_grid, rest.search_grid)
    assert torch.equal(est.mses.view(torch.int32), rest.mses.view(torch.int32)), (per_channel, search, seq)
    assert float(q.mantissa_bits) == float(rq.mantissa_bits)
    assert torch.equal(q.maxval, rq.maxval)
    for i, (a, b) in enumerate(zip(outs, want)):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (per_channel, search, seq, i)
    mgr.fix_ranges()
    with _NoSync():
        y = mgr(batches[0])
    ref.fix_ranges()
    assert torch.equal(y, ref(batches[0]))


def test_k1_with_width_and_sign_in_device_memory():
    import fp8q
    ops = fp8q.ops
    g = torch.Generator().manual_seed(77)
    for shape, pc in (((8, 16, 28, 28), False), ((64, 3, 7, 7), True), ((5, 70000), True), ((70001,), False)):
        x = (torch.randn(*shape, generator=g) * 2.0).cuda()
        mv = ops.minmax(x, pc, want_maxval=True)[2] * 0.8
        for m in (1.0, 2.6, 4.0, 7.0, 8.0):
            for sb in (1, 0):
                flag = torch.tensor([sb], dtype=torch.uint8, device="cuda")
                md = torch.tensor([m], device="cuda")
                want = ops.quantize(x, mv, m, 8, sb)
                got = ops.quantize(x, mv, md, 8, flag)
                assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (shape, m, sb)
