"""The one-call MSE calibration step (fp8q_mse_calibrate_f32 behind FP_MSE_Estimator.calibrate_quantize) against the
protocol calls it replaces (estimator.forward -> set_quant_range -> quantizer.forward: quantization_manager.py:114-122 of the
reference) -- bit for bit -- and the host-side promises around it: no synchronisation in the pass, at most two in fix_ranges().
"""
import os
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _manager(per_channel, search, one_call, mbits=3):
    from quantization.quantizers.fp8_quantizer import FPQuantizer
    from quantization.range_estimators import FP_MSE_Estimator
    from quantization.quantization_manager import QuantizationManager
    mgr = QuantizationManager(qmethod=FPQuantizer, init=FP_MSE_Estimator, per_channel=per_channel,
                              qparams=dict(n_bits=8, mantissa_bits=mbits, set_maxval=True, mse_include_mantissa_bits=search))
    if not one_call:
        mgr.range_estimator.one_call_ok = lambda x: False
    return mgr


def _bits(t):
    return t.detach().contiguous().view(torch.int32)


@pytest.mark.parametrize("shape,per_channel", [((8, 16, 28, 28), False), ((64, 32, 56, 56), False), ((48, 3, 5, 5), True),
                                               ((96, 1, 3, 3), True), ((24, 4100), True), ((7,), False), ((160, 960), True)])
@pytest.mark.parametrize("search", [True, False])
def test_one_call_equals_protocol_calls(shape, per_channel, search):
    torch.manual_seed(sum(shape))
    batches = [torch.randn(shape, device="cuda") * (0.5 + i) for i in range(3)]
    if not per_channel:
        batches[1] = torch.relu(batches[1])
    a, b = _manager(per_channel, search, True), _manager(per_channel, search, False)
    for x in batches:
        ya, yb = a(x), b(x)
        assert a.range_estimator.__dict__.get("_cal") is not None and b.range_estimator.__dict__.get("_cal") is None
        assert torch.equal(_bits(ya), _bits(yb))
        ea, eb = a.range_estimator, b.range_estimator
        assert torch.equal(_bits(ea.search_grid), _bits(eb.search_grid))
        assert torch.equal(_bits(ea.mses), _bits(eb.mses))
        assert torch.equal(_bits(a.quantizer.maxval), _bits(b.quantizer.maxval))
        assert torch.equal(_bits(ea.last_maxval), _bits(eb.last_maxval))
    assert (a.quantizer._pending_mantissa_bits() is not None) == search
    assert float(a.quantizer.mantissa_bits) == float(b.quantizer.mantissa_bits)
    a.fix_ranges()
    b.fix_ranges()
    assert torch.equal(_bits(a(batches[0])), _bits(b(batches[0])))


def test_one_call_survives_a_ragged_batch_and_a_layout_change():
    """the tables depend on the number of channels only: a smaller last batch, or a batch in another memory format (which takes
    the protocol calls), keeps accumulating into the same estimator"""
    torch.manual_seed(5)
    xs = [torch.randn(16, 8, 14, 14, device="cuda"), torch.randn(5, 8, 14, 14, device="cuda"),
          torch.randn(16, 8, 14, 14, device="cuda").to(memory_format=torch.channels_last), torch.randn(16, 8, 14, 14, device="cuda")]
    a, b = _manager(False, True, True), _manager(False, True, False)
    for x in xs:
        ya, yb = a(x), b(x)
        assert torch.equal(_bits(ya), _bits(yb)) and ya.stride() == yb.stride()
        assert torch.equal(_bits(a.range_estimator.mses), _bits(b.range_estimator.mses))
        assert torch.equal(_bits(a.quantizer.maxval), _bits(b.quantizer.maxval))
    # and the other way round: protocol calls first (channels-last), the one-call step adopts their tables
    c = _manager(False, False, True)
    for x in (xs[2], xs[0], xs[3]):
        c(x)
    d = _manager(False, False, False)
    for x in (xs[2], xs[0], xs[3]):
        d(x)
    assert c.range_estimator.__dict__.get("_cal") is not None
    assert torch.equal(_bits(c.range_estimator.mses), _bits(d.range_estimator.mses))
    assert torch.equal(_bits(c.quantizer.maxval), _bits(d.quantizer.maxval))


def test_reset_starts_a_new_search():
    torch.manual_seed(6)
    x1, x2 = torch.randn(4, 8, 9, 9, device="cuda"), torch.randn(4, 8, 9, 9, device="cuda") * 7
    a = _manager(False, True, True)
    a(x1)
    a.reset_ranges()
    a(x2)
    b = _manager(False, True, False)
    b(x2)
    assert torch.equal(_bits(a.range_estimator.mses), _bits(b.range_estimator.mses))
    assert torch.equal(_bits(a.quantizer.maxval), _bits(b.quantizer.maxval))


def test_calibrate_entry_point_validates_before_it_enqueues():
    import ctypes
    import fp8q
    from fp8q._lib import MseState
    L = fp8q.lib()
    x = torch.randn(4, 64, device="cuda")
    cal = fp8q.ops.MseCalibration(4, x.device, [3.0], 8, 1)
    before = cal.mses.clone()
    mb = (ctypes.c_float * 1)(3.0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ws = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    args = lambda n_bits, wsn: (x.data_ptr(), None, 4, 64, ctypes.byref(cal._state), 1, 111, mb, 1, n_bits, 1, None, ws.data_ptr(),      # noqa: E731
                                ws.numel(), ws.data_ptr(), ws.numel(), ws.data_ptr(), wsn, st)
    assert L.fp8q_mse_calibrate_f32(*args(40, ws.numel())) == -1           # FP8Q_EINVAL: format
    assert L.fp8q_mse_calibrate_f32(*args(8, 8)) != 0                      # workspace too small
    torch.cuda.synchronize()
    assert torch.equal(_bits(cal.mses), _bits(before))                     # nothing ran
    nul = MseState()
    assert L.fp8q_mse_calibrate_f32(x.data_ptr(), None, 4, 64, ctypes.byref(nul), 1, 111, mb, 1, 8, 1, None, ws.data_ptr(), ws.numel(),
                                    ws.data_ptr(), ws.numel(), ws.data_ptr(), ws.numel(), st) == -1
    # a pre-stage needs a per-tensor quantizer whose element count is the producer's
    from fp8q._lib import AffinePre
    pre = AffinePre(x.data_ptr(), None, None, 4, 8, 8, 1)
    assert L.fp8q_mse_calibrate_f32(x.data_ptr(), None, 4, 64, ctypes.byref(cal._state), 1, 111, mb, 1, 8, 1, ctypes.byref(pre), ws.data_ptr(),
                                    ws.numel(), ws.data_ptr(), ws.numel(), ws.data_ptr(), ws.numel(), st) == -1


class _CountSyncs:
    """torch's own synchronisations (sync debug mode "warn") + this library's synchronising entry point"""

    def __enter__(self):
        torch.cuda.synchronize()
        self._caught = warnings.catch_warnings(record=True)
        self._log = self._caught.__enter__()
        warnings.simplefilter("always")
        torch.cuda.set_sync_debug_mode("warn")
        return self

    def __exit__(self, *exc):
        torch.cuda.set_sync_debug_mode("default")
        self.torch_syncs = [str(w.message) for w in self._log if "called a synchronizing" in str(w.message)]
        self._caught.__exit__(*exc)
        return False


def test_model_calibration_pass_is_sync_free_and_fix_ranges_syncs_at_most_twice(monkeypatch):
    """BASELINE config 4's procedure on a small net: the MSE calibration forward enqueues only; fix_ranges() -- votes to the
    host, workspace check -- synchronises at most twice."""
    import fp8q
    from quantization.autoquant_utils import quantize_model
    from quantization.base_quantized_model import QuantizedModel
    from quantization.quantization_manager import QMethods
    from quantization.range_estimators import RangeEstimators
    from torch import nn

    class Net(QuantizedModel):
        def __init__(self):
            super().__init__((1, 3, 16, 16))
            seq = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU6(), nn.Conv2d(8, 8, 3, padding=1, groups=8),
                                nn.BatchNorm2d(8), nn.ReLU6(), nn.Conv2d(8, 16, 1), nn.BatchNorm2d(16))
            self.features = quantize_model(seq, method=QMethods.fp_quantizer.cls, n_bits=8, per_channel_weights=True,
                                           weight_range_method=RangeEstimators.MSE.cls, act_range_method=RangeEstimators.MSE.cls,
                                           fp8_kwargs=dict(mantissa_bits=3, set_maxval=True, mse_include_mantissa_bits=True))

        def forward(self, x):
            return self.features(x)

    torch.manual_seed(0)
    net = Net().cuda().eval()
    x = torch.randn(8, 3, 16, 16, device="cuda")
    fp8q.ops.mse_linspace(torch.ones(1, device="cuda"))       # the once-per-process self-check synchronises
    fp8q.ops.check_workspaces()
    fp8q.ops._ws_cache.clear()                                # (workspaces of earlier tests' streams would each get their check)
    with torch.no_grad():
        net.set_quant_state(True, True)
        net.estimate_ranges()
        net(x)                                                # allocations
        with _CountSyncs() as c1:
            net(x)
        assert c1.torch_syncs == []
        checks = []
        real = fp8q.ops.check_workspaces
        monkeypatch.setattr(fp8q.ops, "check_workspaces", lambda *a, **k: (checks.append(1), real(*a, **k))[1])
        with _CountSyncs() as c2:
            net.fix_ranges()
        n_ws = sum(1 for k in fp8q.ops._ws_cache if k[2])      # one synchronising check per live min/max workspace
        assert len(c2.torch_syncs) <= 1, c2.torch_syncs        # the votes: one device-to-host copy
        assert len(checks) == 1 and n_ws <= 1
        with _CountSyncs() as c3:
            net(x)
        assert c3.torch_syncs == []
    widths = [float(m.quantizer.mantissa_bits) for m in net.modules() if hasattr(m, "quantizer") and hasattr(m.quantizer, "maxval")]
    assert widths and all(1.0 <= w <= 6.0 for w in widths)


def test_the_sync_counter_counts():
    t = torch.ones(3, device="cuda")
    with _CountSyncs() as c:
        t.sum().item()
        t.cpu()
    assert len(c.torch_syncs) == 2, c.torch_syncs


def test_votes_live_in_one_arena_and_come_over_in_one_copy():
    import fp8q
    from quantization.model import materialize_mantissa_bits
    mgrs = torch.nn.ModuleList([_manager(False, True, True) for _ in range(6)])
    torch.manual_seed(3)
    for i, m in enumerate(mgrs):
        m(torch.randn(4, 4, 8, 8, device="cuda") * (i + 1))
    where = [fp8q.ops.vote_arena_of(m.quantizer._pending_mantissa_bits()) for m in mgrs]
    assert all(w is not None for w in where) and len({id(w[0]) for w in where}) == 1
    assert len({w[1] for w in where}) == len(mgrs)
    want = [float(m.quantizer._pending_mantissa_bits().cpu()) for m in mgrs]
    with _CountSyncs() as c:
        assert materialize_mantissa_bits(mgrs) == len(mgrs)
    assert len(c.torch_syncs) <= 1
    assert [float(m.quantizer.mantissa_bits) for m in mgrs] == want


@pytest.mark.parametrize("search", [True, False])
@pytest.mark.parametrize("shape,use_bn,use_res,act", [((8, 16, 14, 14), True, False, 2), ((64, 32, 56, 56), True, False, 2),
                                                      ((4, 24, 7, 7), False, True, 0), ((8, 12, 6, 6), True, True, 1),
                                                      ((16, 96, 56, 56), True, False, 2)])
def test_pre_stage_equals_epilogue_then_one_call(shape, use_bn, use_res, act, search):
    """calibrate_quantize(x, pre=(bn_ab, residual, act)) == calibrate_quantize(affine_act(x, ...)): tables, range, width, output --
    first batch (abs-max and grid from the epilogue's own launch) and a second one (plain epilogue pass)"""
    import fp8q
    ops = fp8q.ops
    torch.manual_seed(shape[1])
    C = shape[1]
    ab = None
    if use_bn:
        bn = (torch.randn(C, device="cuda"), 1 / torch.sqrt(torch.rand(C, device="cuda") + 0.5), torch.rand(C, device="cuda") + 0.5,
              torch.randn(C, device="cuda"))
        ab = ops.bn_fold(bn)
    a, b = _manager(False, search, True), _manager(False, search, True)
    for i in range(2):
        x = torch.randn(shape, device="cuda") * (1.0 + i)
        res = torch.randn(shape, device="cuda") if use_res else None
        ya = a.range_estimator.calibrate_quantize(x, pre=(ab, res, act))
        t = ops.affine_act(x, ab, res, act)
        yb = b.range_estimator.calibrate_quantize(t)
        ea, eb = a.range_estimator, b.range_estimator
        assert torch.equal(_bits(ea.search_grid), _bits(eb.search_grid))
        assert torch.equal(_bits(ea.mses), _bits(eb.mses))
        assert torch.equal(_bits(a.quantizer.maxval), _bits(b.quantizer.maxval))
        assert torch.equal(_bits(ya), _bits(yb))
        for name in ("cur_min", "cur_max", "absmax"):
            assert torch.equal(_bits(getattr(ea._cal, name)), _bits(getattr(eb._cal, name))), name
    assert float(a.quantizer.mantissa_bits) == float(b.quantizer.mantissa_bits)


def test_weights_are_calibrated_ahead_on_a_side_stream(monkeypatch):
    """QuantizedModel: the weight quantizers' estimate + quantize of ALL layers is enqueued on a side stream at the start of a
    calibration forward (they do not depend on the data) and every layer waits for its own event -- same ranges, widths and
    outputs, bit for bit, as with each layer calibrating its weight inside its own forward (FP8Q_WEIGHTS_AHEAD=0); nothing of
    it happens once the ranges are fixed."""
    import os
    import fp8q
    from quantization import model as qmodel
    from quantization.autoquant_utils import quantize_model
    from quantization.base_quantized_model import QuantizedModel
    from quantization.quantization_manager import QMethods, QuantizationManager
    from quantization.range_estimators import RangeEstimators
    from torch import nn

    def build(w_est, a_est, search):
        class Net(QuantizedModel):
            def __init__(self):
                super().__init__((1, 3, 16, 16))
                torch.manual_seed(4)
                seq = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1, bias=False), nn.BatchNorm2d(8), nn.ReLU6(),
                                    nn.Conv2d(8, 8, 3, padding=1, groups=8, bias=False), nn.BatchNorm2d(8), nn.ReLU6(),
                                    nn.ConvTranspose2d(8, 6, 2, bias=False), nn.ReLU(), nn.Conv2d(6, 16, 1, bias=False), nn.BatchNorm2d(16))
                self.features = quantize_model(seq, method=QMethods.fp_quantizer.cls, n_bits=8, per_channel_weights=True,
                                               weight_range_method=RangeEstimators[w_est].cls, act_range_method=RangeEstimators[a_est].cls,
                                               fp8_kwargs=dict(mantissa_bits=3, set_maxval=True, mse_include_mantissa_bits=search))

            def forward(self, x):
                return self.features(x)
        return Net().cuda().eval()

    x1, x2 = torch.randn(8, 3, 16, 16, device="cuda"), torch.randn(8, 3, 16, 16, device="cuda") * 2

    def run(net):
        with torch.no_grad():
            net.set_quant_state(True, True)
            net.estimate_ranges()
            ys = [net(x1), net(x2)]
            net.fix_ranges()
            ys.append(net(x1))
        torch.cuda.synchronize()
        state = [(m.quantizer.maxval.clone(), float(m.quantizer.mantissa_bits)) for m in net.modules() if isinstance(m, QuantizationManager)]
        return ys, state

    for w_est, a_est, search in (("MSE", "MSE", True), ("MSE", "MSE", False), ("current_minmax", "allminmax", False)):
        calls = []
        real = qmodel.calibrate_weights_ahead
        monkeypatch.setattr(qmodel, "calibrate_weights_ahead", lambda m: (calls.append(real(m)), calls[-1])[1])
        ya, sa = run(build(w_est, a_est, search))
        assert calls == [4, 4], calls                    # two calibration forwards x 4 weight layers; none with fixed ranges
        monkeypatch.setattr(qmodel, "calibrate_weights_ahead", real)
        os.environ["FP8Q_WEIGHTS_AHEAD"] = "0"
        try:
            yb, sb = run(build(w_est, a_est, search))
        finally:
            os.environ.pop("FP8Q_WEIGHTS_AHEAD")
        for a, b in zip(ya, yb):
            assert torch.equal(_bits(a), _bits(b)), (w_est, search)
        for (mva, ma), (mvb, mb) in zip(sa, sb):
            assert torch.equal(_bits(mva), _bits(mvb)) and ma == mb


def test_weights_ahead_tolerates_a_forward_that_does_not_follow_module_order():
    """six layers registered in one order and executed in the opposite one: the layers whose turn comes before their place in the
    side stream's queue calibrate their own weight (once!), the others pick up what ran ahead -- tables after two batches equal
    those of the plain per-layer flow"""
    import os
    from quantization.autoquant_utils import quantize_model
    from quantization.base_quantized_model import QuantizedModel
    from quantization.quantization_manager import QMethods
    from quantization.range_estimators import RangeEstimators
    from torch import nn

    class Net(QuantizedModel):
        def __init__(self):
            super().__init__((1, 4, 8, 8))
            torch.manual_seed(9)
            self.convs = nn.ModuleList([quantize_model(nn.Conv2d(4, 4, 1, bias=False), method=QMethods.fp_quantizer.cls, n_bits=8,
                                                       per_channel_weights=True, weight_range_method=RangeEstimators.MSE.cls,
                                                       act_range_method=RangeEstimators.MSE.cls,
                                                       fp8_kwargs=dict(mantissa_bits=3, set_maxval=True, mse_include_mantissa_bits=True))
                                        for _ in range(6)])

        def forward(self, x):
            for conv in reversed(self.convs):
                x = conv(x)
            return x

    def run():
        net = Net().cuda().eval()
        with torch.no_grad():
            net.set_quant_state(True, True)
            net.estimate_ranges()
            ys = [net(torch.full((2, 4, 8, 8), 0.5, device="cuda") + i) for i in range(2)]
        torch.cuda.synchronize()
        return ys, [c.weight_quantizer.range_estimator.mses.clone() for c in net.convs]

    ya, ta = run()
    os.environ["FP8Q_WEIGHTS_AHEAD"] = "0"
    try:
        yb, tb = run()
    finally:
        os.environ.pop("FP8Q_WEIGHTS_AHEAD")
    assert all(torch.equal(_bits(a), _bits(b)) for a, b in zip(ya, yb))
    assert all(torch.equal(_bits(a), _bits(b)) for a, b in zip(ta, tb))


def test_weights_ahead_with_nested_quantized_models():
    """a QuantizedModel inside a QuantizedModel: only the outermost forward starts the weight calibrations (a nested start would
    discard and repeat what is already running: tables accumulated twice)"""
    import os
    from quantization import model as qmodel
    from quantization.autoquant_utils import quantize_model
    from quantization.base_quantized_model import QuantizedModel
    from quantization.quantization_manager import QMethods
    from quantization.range_estimators import RangeEstimators
    from torch import nn

    def qconv():
        return quantize_model(nn.Conv2d(4, 4, 1, bias=False), method=QMethods.fp_quantizer.cls, n_bits=8, per_channel_weights=True,
                              weight_range_method=RangeEstimators.MSE.cls, act_range_method=RangeEstimators.MSE.cls,
                              fp8_kwargs=dict(mantissa_bits=3, set_maxval=True, mse_include_mantissa_bits=False))

    class Inner(QuantizedModel):
        def __init__(self):
            super().__init__((1, 4, 8, 8))
            self.c1, self.c2 = qconv(), qconv()

        def forward(self, x):
            return self.c2(self.c1(x))

    class Outer(QuantizedModel):
        def __init__(self):
            super().__init__((1, 4, 8, 8))
            torch.manual_seed(12)
            self.head, self.inner, self.tail = qconv(), Inner(), qconv()

        def forward(self, x):
            return self.tail(self.inner(self.head(x)))

    def run():
        net = Outer().cuda().eval()
        with torch.no_grad():
            net.set_quant_state(True, True)
            net.estimate_ranges()
            y = net(torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(1)).cuda())
        torch.cuda.synchronize()
        assert qmodel._AHEAD_DEPTH[0] == 0
        convs = [net.head, net.inner.c1, net.inner.c2, net.tail]
        return y, [c.weight_quantizer.range_estimator.mses.clone() for c in convs]

    ya, ta = run()
    os.environ["FP8Q_WEIGHTS_AHEAD"] = "0"
    try:
        yb, tb = run()
    finally:
        os.environ.pop("FP8Q_WEIGHTS_AHEAD")
    assert torch.equal(_bits(ya), _bits(yb)) and all(torch.equal(_bits(a), _bits(b)) for a, b in zip(ta, tb))


@pytest.mark.parametrize("w_est,a_est,search", [("MSE", "MSE", True), ("MSE", "MSE", False), ("current_minmax", "allminmax", False),
                                                ("current_minmax", "running_minmax", False)])
def test_calibration_forward_replayed_from_a_hip_graph(w_est, a_est, search):
    """GraphedCalibration: batches 1 and 2 eager, batch 3 captured, batches 4 .. 5 replayed -- the estimators' device state
    (tables, ranges, widths) and every batch's output equal the eager loop's, bit for bit; then fix_ranges() and validation."""
    from quantization.autoquant_utils import quantize_model
    from quantization.base_quantized_model import QuantizedModel, GraphedCalibration
    from quantization.quantization_manager import QMethods, QuantizationManager
    from quantization.range_estimators import RangeEstimators
    from torch import nn

    class Net(QuantizedModel):
        def __init__(self):
            super().__init__((1, 3, 16, 16))
            torch.manual_seed(21)
            seq = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1, bias=False), nn.BatchNorm2d(8), nn.ReLU6(),
                                nn.Conv2d(8, 8, 3, padding=1, groups=8, bias=False), nn.BatchNorm2d(8), nn.ReLU(),
                                nn.Conv2d(8, 12, 1, bias=False), nn.BatchNorm2d(12), nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(12, 5))
            self.features = quantize_model(seq, method=QMethods.fp_quantizer.cls, n_bits=8, per_channel_weights=True,
                                           weight_range_method=RangeEstimators[w_est].cls, act_range_method=RangeEstimators[a_est].cls,
                                           fp8_kwargs=dict(mantissa_bits=3, set_maxval=True, mse_include_mantissa_bits=search))

        def forward(self, x):
            return self.features(x)

    g = torch.Generator().manual_seed(3)
    xs = [torch.randn(8, 3, 16, 16, generator=g).cuda() * (1 + 0.3 * i) for i in range(5)]

    def run(graphed):
        net = Net().cuda().eval()
        with torch.no_grad():
            net.set_quant_state(True, True)
            net.estimate_ranges()
            fwd = GraphedCalibration(net) if graphed else net
            ys = [fwd(x).clone() for x in xs]
            assert (not graphed) or fwd.graph is not None
            net.fix_ranges()
            ys.append(net(xs[0]).clone())
        torch.cuda.synchronize()
        st = [(m.quantizer.maxval.clone(), float(m.quantizer.mantissa_bits)) for m in net.modules() if isinstance(m, QuantizationManager)]
        return ys, st

    ya, sa = run(True)
    yb, sb = run(False)
    for i, (a, b) in enumerate(zip(ya, yb)):
        assert torch.equal(_bits(a), _bits(b)), i
    for (mva, ma), (mvb, mb) in zip(sa, sb):
        assert torch.equal(_bits(mva), _bits(mvb)) and ma == mb


_SEL_SCRIPT = r"""
import sys, numpy as np, torch
sys.path[:0] = [r"%s", r"%s"]
import fp8q
ops = fp8q.ops
torch.manual_seed(11)
out = {}
for name, shape, mb, pre in (("search_pre", (8, 24, 28, 28), [1.0, 2.0, 3.0, 4.0, 5.0, 6.0], True), ("fixed_pre", (8, 24, 28, 28), [3.0], True),
                            ("search_plain", (4, 3, 130, 131), [1.0, 2.0, 3.0, 4.0, 5.0, 6.0], False), ("fixed_big", (64, 32, 112, 112), [3.0], False),
                            ("search_tiny", (2, 5, 9, 7), [1.0, 2.0, 3.0, 4.0, 5.0, 6.0], False)):
    x = torch.randn(*shape, device="cuda") * 2.0
    ab = None
    if pre:
        C = shape[1]
        ab = ops.bn_fold((torch.randn(C, device="cuda") * 0.1, torch.rand(C, device="cuda") + 0.7, torch.rand(C, device="cuda") + 0.5,
                          torch.randn(C, device="cuda") * 0.1))
    cal = ops.MseCalibration(1, x.device, mb, 8, 1)
    for b in range(3):                      # the table accumulates over the batches; the winner is taken after each
        xb = x * (1.0 + 0.25 * b)
        y = cal.step(xb, pre=(ab, None, 2) if pre else None)
        out[f"{name}_y{b}"] = y.cpu().numpy().view(np.uint32)
        out[f"{name}_mv{b}"] = cal.maxval.cpu().numpy().view(np.uint32)
        out[f"{name}_mb{b}"] = cal.mbits.cpu().numpy()
        out[f"{name}_vote{b}"] = cal.vote.cpu().numpy()
        out[f"{name}_xmin{b}"] = cal.xmin.cpu().numpy().view(np.uint32)
np.savez(sys.argv[1], **out)
""" % (ROOT, os.path.join(ROOT, "fp8-quantization_amd"))


def test_selection_in_the_k1_prologue_equals_the_ticket_path(tmp_path):
    """k_quant_rows_sel (the winner selection in the prologue of the K1 launch, the default of fp8q_mse_calibrate_f32 for
    per-tensor quantizers) against FP8Q_SEL_IN_K1=0 (select_one_row behind tickets in the launch that finishes the table):
    quantized batches, clipping value, voted width, its index and xmin after each of three batches -- bit for bit."""
    import subprocess
    import sys
    res = {}
    for mode in ("0", "2"):
        path = os.path.join(str(tmp_path), f"sel{mode}.npz")
        r = subprocess.run([sys.executable, "-c", _SEL_SCRIPT, path], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, FP8Q_SEL_IN_K1=mode))
        assert r.returncode == 0, r.stdout + r.stderr
        res[mode] = np.load(path)
    assert set(res["0"].files) == set(res["2"].files) and len(res["0"].files) == 75
    for k in res["0"].files:
        assert np.array_equal(res["0"][k], res["2"][k]), k
