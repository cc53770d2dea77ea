"""GPU tests of the host-side mirror of the reference API against golden fixtures produced by the
reference's own classes (QuantizationManager, estimators, FP_MSE_Estimator, quantize_model)."""
import os

import numpy as np
import pytest
import torch
from torch import nn

import oracle

from parity import assert_parity, elem_step

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _mgr(init, per_channel=False, **fp8):
    from quantization.quantization_manager import QuantizationManager, QMethods
    from quantization.range_estimators import RangeEstimators
    qp = dict(n_bits=8, mantissa_bits=3, maxval=None, set_maxval=True)
    qp.update(fp8)
    return QuantizationManager(qmethod=QMethods.fp_quantizer.cls, init=RangeEstimators[init].cls,
                               per_channel=per_channel, qparams=qp)


def test_manager_state_machine_vs_reference(golden_dir):
    from quantization.quantization_manager import Qstates
    g6 = np.load(os.path.join(golden_dir, "g6_manager.npz"))
    xs = g6["xs"]
    qm = _mgr("allminmax")
    y0 = qm(dev(xs[0])).cpu().numpy()
    mv0 = qm.quantizer.maxval.cpu().numpy().copy()
    y1 = qm(dev(xs[1])).cpu().numpy()
    np.testing.assert_array_equal(qm.quantizer.maxval.cpu().numpy(), g6["maxval_after2"])
    qm.fix_ranges()
    assert qm.state == Qstates.fix_ranges
    y2 = qm(dev(xs[2])).cpu().numpy()     # larger data, frozen (smaller) range -> clipping
    np.testing.assert_array_equal(qm.quantizer.maxval.cpu().numpy(), g6["maxval_after_fix"])
    for i, (y, mv) in enumerate(((y0, mv0), (y1, g6["maxval_after2"]), (y2, g6["maxval_after_fix"]))):
        assert_parity(y, g6["ys"][i], elem_step(xs[i], mv, 3), what=f"batch {i}")
    # set_maxval=False (CLI default): estimation is a no-op, the format's default range is used
    qm = _mgr("allminmax", mantissa_bits=2, set_maxval=False)
    y = qm(dev(xs[0] * 1e4)).cpu().numpy()
    np.testing.assert_array_equal(qm.quantizer.maxval.cpu().numpy(), g6["nomaxval_maxval"])
    assert_parity(y, g6["nomaxval_y"], elem_step(xs[0] * 1e4, g6["nomaxval_maxval"], 2), what="default maxval")
    assert qm.range_estimator.current_xmax is not None      # the estimate itself is still tracked


def test_estimator_classes_vs_reference(golden_dir):
    from quantization.range_estimators import RangeEstimators
    g3 = np.load(os.path.join(golden_dir, "g3_estimators.npz"))
    for name in ("allminmax", "running_minmax"):
        for pc in (False, True):
            est = RangeEstimators[name].cls(per_channel=pc)
            for b, a in enumerate(g3["acts"]):
                mn, mx = est(dev(a))
                assert mn.dim() == (1 if pc else 0)
                np.testing.assert_array_equal(mn.cpu().numpy().reshape(-1), g3[f"{name}_pc{int(pc)}_min"][b])
                np.testing.assert_array_equal(mx.cpu().numpy().reshape(-1), g3[f"{name}_pc{int(pc)}_max"][b])
            est.reset()
            assert est.current_xmin is None
    est = RangeEstimators.current_minmax.cls(per_channel=True)
    mn, mx = est(dev(g3["w"]))
    np.testing.assert_array_equal(mn.cpu().numpy(), g3["w_cur_pc_min"])
    np.testing.assert_array_equal(est.current_xmax.cpu().numpy(), g3["w_cur_pc_max"])


def test_weight_manager_conv1_config2(golden_dir):
    """BASELINE config 2 through the operator API: per-channel current_minmax + E5M2."""
    g3 = np.load(os.path.join(golden_dir, "g3_estimators.npz"))
    qm = _mgr("current_minmax", per_channel=True, mantissa_bits=2)
    y = qm(dev(g3["w"])).cpu().numpy()
    np.testing.assert_array_equal(qm.quantizer.maxval.cpu().numpy(), g3["w_maxval"])
    np.testing.assert_array_equal(qm.range_estimator.current_xmin.cpu().numpy(), g3["w_cur_pc_min"])
    assert_parity(y, g3["w_q_e5m2"], elem_step(g3["w"], g3["w_maxval"], 2), what="conv1")
    qm.fix_ranges()
    y2 = qm(dev(g3["w"])).cpu().numpy()                     # fixed-range kernel gives the same bits
    assert np.array_equal(y.view(np.int32), y2.view(np.int32))


def test_allow_unsigned_path(golden_dir):
    g3 = np.load(os.path.join(golden_dir, "g3_estimators.npz"))
    qm = _mgr("current_minmax", allow_unsigned=True)
    y = qm(dev(g3["relu_x"])).cpu().numpy()
    assert qm.quantizer.sign_bits == int(g3["relu_sign_bits"]) == 0
    np.testing.assert_array_equal(qm.quantizer.maxval.cpu().numpy(), g3["relu_maxval"])
    assert_parity(y, g3["relu_q"], elem_step(g3["relu_x"], g3["relu_maxval"], 3, 8, 0), what="relu")


@pytest.mark.parametrize("name,pc,incl,M", [("w_pc_fixm", True, False, 3), ("w_pc_srchm", True, True, 3),
                                            ("a_pt_fixm", False, False, 3), ("a_pt_srchm", False, True, 2)])
def test_mse_estimator_vs_reference(golden_dir, name, pc, incl, M):
    from quantization.quantizers.fp8_quantizer import FPQuantizer
    from quantization.range_estimators import RangeEstimators
    g4 = np.load(os.path.join(golden_dir, "g4_mse.npz"))
    q = FPQuantizer(n_bits=8, per_channel=pc, mantissa_bits=M, maxval=None, set_maxval=True,
                    mse_include_mantissa_bits=incl)
    est = RangeEstimators.MSE.cls(per_channel=pc, quantizer=q)
    for b in range(2):
        mn, mx = est(dev(g4[f"{name}_x{b}"]))
        if b == 0:
            np.testing.assert_array_equal(est.search_grid.cpu().numpy(), g4[f"{name}_grid"])
        ref_mses = g4[f"{name}_mses{b}"]
        np.testing.assert_allclose(est.mses.cpu().numpy(), ref_mses, rtol=1e-4)
        assert float(q.mantissa_bits) == float(g4[f"{name}_mbits{b}"])
        got, ref = mx.cpu().numpy().reshape(-1), g4[f"{name}_max{b}"].reshape(-1)
        np.testing.assert_array_equal(mn.cpu().numpy().reshape(-1), -got)
        mb = [1, 2, 3, 4, 5, 6] if incl else [M]
        mi = mb.index(int(float(q.mantissa_bits)))
        grid = g4[f"{name}_grid"]
        for c in range(grid.shape[1]):     # same candidate, or an equally good one (<= 1e-5 rel)
            if got[c] != ref[c]:
                j_got = int(np.argmin(np.abs(grid[:, c] - got[c])))
                j_ref = int(np.argmin(np.abs(grid[:, c] - ref[c])))
                assert abs(ref_mses[mi, j_got, c] - ref_mses[mi, j_ref, c]) <= 1e-5 * ref_mses[mi, j_ref, c]


def test_mse_estimator_unsigned(golden_dir):
    from quantization.quantizers.fp8_quantizer import FPQuantizer
    from quantization.range_estimators import RangeEstimators
    g4 = np.load(os.path.join(golden_dir, "g4_mse.npz"))
    q = FPQuantizer(n_bits=8, per_channel=False, mantissa_bits=3, maxval=None, set_maxval=True,
                    mse_include_mantissa_bits=False, allow_unsigned=True)
    est = RangeEstimators.MSE.cls(per_channel=False, quantizer=q)
    mn, mx = est(dev(g4["relu_x"]))
    assert q.sign_bits == int(g4["relu_sign_bits"])
    np.testing.assert_allclose(est.mses.cpu().numpy(), g4["relu_mses"], rtol=1e-4)
    np.testing.assert_array_equal(mx.cpu().numpy().reshape(-1), g4["relu_max"].reshape(-1))
    np.testing.assert_array_equal(mn.cpu().numpy().reshape(-1), g4["relu_min"].reshape(-1))


def _tiny_cnn(g7):
    net = nn.Sequential(nn.Conv2d(3, 16, 3, padding=1, bias=False), nn.BatchNorm2d(16), nn.ReLU(),
                        nn.Conv2d(16, 24, 3, stride=2, padding=1, bias=True), nn.ReLU6(),
                        nn.Conv2d(24, 24, 3, padding=1, groups=24, bias=False), nn.BatchNorm2d(24), nn.ReLU(),
                        nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(24, 10))
    net.load_state_dict({k[3:]: torch.from_numpy(g7[k]) for k in g7.files if k.startswith("sd_")})
    return net.eval()


@pytest.mark.parametrize("tag,M,act_est", [("e5m2", 2, "allminmax"), ("e4m3", 3, "allminmax"),
                                           ("e4m3_run", 3, "running_minmax")])
def test_quantize_model_tiny_cnn_vs_reference(golden_dir, tag, M, act_est):
    """Wrapper level (BASELINE config 3 procedure): calibrate on one batch, fix ranges, validate.
    The conv/BN themselves run in MIOpen/rocBLAS (fp32): last-ulp differences before a quantizer
    can move an activation by one grid step, hence the logits tolerance."""
    from quantization.autoquant_utils import quantize_model
    from quantization.base_quantized_classes import QuantizedModule
    from quantization.quantization_manager import QuantizationManager, QMethods
    from quantization.range_estimators import RangeEstimators
    g7 = np.load(os.path.join(golden_dir, "g7_tinycnn.npz"))
    qparams = dict(method=QMethods.fp_quantizer.cls, weight_range_method=RangeEstimators.current_minmax.cls,
                   act_range_method=RangeEstimators[act_est].cls, n_bits=8, n_bits_act=8,
                   per_channel_weights=True,
                   fp8_kwargs=dict(maxval=None, mantissa_bits=M, set_maxval=True, learn_maxval=False,
                                   learn_mantissa_bits=False, mse_include_mantissa_bits=False,
                                   allow_unsigned=False))
    q = quantize_model(_tiny_cnn(g7), tie_activation_quantizers=True, **qparams).eval().cuda()

    def each(fn):
        for m in q.modules():
            if isinstance(m, QuantizedModule):
                fn(m)
    calib, val = dev(g7["calib"]), dev(g7["val"])
    with torch.no_grad():
        fp_logits = q(val).cpu().numpy()
        each(lambda m: m.quantized())
        calib_logits = q(calib).cpu().numpy()
        each(lambda m: m.fix_ranges())
        val_logits = q(val).cpu().numpy()
    np.testing.assert_allclose(fp_logits, g7[f"{tag}_fp_logits"], rtol=1e-4, atol=1e-5)
    names = [n for n, m in q.named_modules() if isinstance(m, QuantizationManager)]
    assert names == list(g7[f"{tag}_mgr_names"])
    for n, m in q.named_modules():
        if isinstance(m, QuantizationManager):
            ref = g7[f"{tag}_maxval_{n}"]
            got = m.quantizer.maxval.cpu().numpy()
            if n.endswith("weight_quantizer"):
                np.testing.assert_array_equal(got, ref)            # weights: bit-equal ranges
            else:
                np.testing.assert_allclose(got, ref, rtol=1e-5)    # activations: conv rounding
    for got, ref in ((calib_logits, g7[f"{tag}_calib_logits"]), (val_logits, g7[f"{tag}_val_logits"])):
        assert np.array_equal(got.argmax(1), ref.argmax(1))
        scale = np.abs(ref).max()
        np.testing.assert_allclose(got, ref, rtol=0, atol=0.05 * scale)
        assert np.mean(np.abs(got - ref)) < 0.01 * scale


def test_prequantize_weights_multi_tensor(golden_dir):
    """After fix_ranges all FP8 weights can be quantized in one multi-tensor launch that fills the layers'
    caches: same tensors, bit for bit, as each layer's own quantizer; the forward is unchanged."""
    from quantization.autoquant_utils import quantize_model
    from quantization.base_quantized_classes import QuantizedModule
    from quantization.base_quantized_model import prequantize_weights
    from quantization.hijacker import QuantizationHijacker
    from quantization.quantization_manager import QMethods
    from quantization.range_estimators import RangeEstimators
    g7 = np.load(os.path.join(golden_dir, "g7_tinycnn.npz"))
    qparams = dict(method=QMethods.fp_quantizer.cls, weight_range_method=RangeEstimators.current_minmax.cls,
                   act_range_method=RangeEstimators.allminmax.cls, n_bits=8, per_channel_weights=True,
                   fp8_kwargs=dict(maxval=None, mantissa_bits=2, set_maxval=True))
    q = quantize_model(_tiny_cnn(g7), **qparams).eval().cuda()
    calib, val = dev(g7["calib"]), dev(g7["val"])
    with torch.no_grad():
        for m in q.modules():
            if isinstance(m, QuantizedModule):
                m.quantized()
        q(calib)
        for m in q.modules():
            if isinstance(m, QuantizedModule):
                m.fix_ranges()
        ref = q(val).clone()                  # per-layer launches fill the caches
        per_layer = {n: m._wq_cache.clone() for n, m in q.named_modules() if isinstance(m, QuantizationHijacker)}
        for m in q.modules():
            if isinstance(m, QuantizationHijacker):
                m._wq_key = m._wq_cache = None
        n = prequantize_weights(q)
        assert n == len(per_layer) == 4
        for name, m in q.named_modules():
            if isinstance(m, QuantizationHijacker):
                assert m._wq_key is not None
                assert torch.equal(m._wq_cache, per_layer[name]), name
                assert m.get_params()[0] is m._wq_cache      # the forward uses the prequantized tensor
        assert torch.equal(q(val), ref)


def test_multi_plan_replays_and_follows_in_place_updates():
    """fp8q_multi_plan_*: built once, launch() is bit-identical to fp8q_multi_quantize_f32 / to the oracle; in-place
    changes of the weights or of the ranges are picked up by the next launch; tensors that cannot be batched
    (unaligned storage, >= 64 MiB) get their own launch inside the plan."""
    import fp8q
    ops = fp8q.ops
    rng = np.random.RandomState(5)
    shapes = [(64, 3, 7, 7), (128, 64, 3, 3), (10, 33), (1, 5), (3, 4099)]
    ws = [dev((rng.randn(*sh) * 0.1).astype(np.float32)) for sh in shapes]
    base = torch.zeros(1000 * 7 + 1, device="cuda")
    odd = base[1:].view(1000, 7)                     # 4-byte aligned only: not batchable
    odd.copy_(dev(rng.randn(1000, 7).astype(np.float32)))
    ws.append(odd)
    mvs = [ops.minmax(w, True, want_maxval=True)[2] for w in ws[:-1]] + [torch.tensor([1.25], device="cuda")]
    items = [(w, mv, 2 + (i % 2), 8, 1) for i, (w, mv) in enumerate(zip(ws, mvs))]
    plan = ops.MultiPlan(items)
    # [64,3,7,7] + [128,64,3,3] batched | [10,33]: rows too short for per-row tables next to a chunk, own launch |
    # [1,5] (one range: per tensor) + [3,4099] batched | the unaligned tensor, own launch
    assert plan.launches == 4
    ref = ops.multi_quantize(items)
    outs = plan.launch()
    for (w, mv, M, _, _), a, b in zip(items, outs, ref):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
        want = oracle.c_quantize(w.cpu().numpy(), mv.cpu().numpy(), M, 8, 1)
        assert np.array_equal(a.cpu().numpy().view(np.int32), want.view(np.int32))
    with torch.no_grad():
        ws[1].mul_(0.5)                              # weights updated in place (an optimizer step)
        mvs[1].mul_(0.5)                             # ... and a range tensor
    outs2 = plan.launch()
    assert outs2[1] is outs[1]                       # same buffers
    want = oracle.c_quantize(ws[1].cpu().numpy(), mvs[1].cpu().numpy(), 3, 8, 1)
    assert np.array_equal(outs2[1].cpu().numpy().view(np.int32), want.view(np.int32))
    with pytest.raises(fp8q.Fp8qError):
        ops.MultiPlan([(ws[0].permute(1, 0, 2, 3), mvs[0], 2)])       # non-contiguous: a copy would be quantized


def test_requantize_weights_uses_the_plan(golden_dir):
    """QuantizedModel.requantize_weights(): after in-place weight updates every layer's cached quantized weight is
    refreshed by ONE plan launch and equals the layer's own quantizer."""
    from quantization.autoquant_utils import quantize_model
    from quantization.base_quantized_classes import QuantizedModule
    from quantization.base_quantized_model import prequantize_weights
    from quantization.model import requantize_weights
    from quantization.hijacker import QuantizationHijacker
    from quantization.quantization_manager import QMethods
    from quantization.range_estimators import RangeEstimators
    g7 = np.load(os.path.join(golden_dir, "g7_tinycnn.npz"))
    q = quantize_model(_tiny_cnn(g7), method=QMethods.fp_quantizer.cls,
                       weight_range_method=RangeEstimators.current_minmax.cls,
                       act_range_method=RangeEstimators.allminmax.cls, n_bits=8, per_channel_weights=True,
                       fp8_kwargs=dict(maxval=None, mantissa_bits=2, set_maxval=True)).eval().cuda()
    with torch.no_grad():
        for m in q.modules():
            if isinstance(m, QuantizedModule):
                m.quantized()
        q(dev(g7["calib"]))
        for m in q.modules():
            if isinstance(m, QuantizedModule):
                m.fix_ranges()
        import copy
        from quantization.model import _PLANS
        assert prequantize_weights(q) == 4
        plan = _PLANS[q][0]
        layers = [m for m in q.modules() if isinstance(m, QuantizationHijacker)]
        for m in layers:
            m.weight.mul_(0.9)                       # in place: same storage, new contents
        assert requantize_weights(q) == 4 and _PLANS[q][0] is plan
        for m in layers:
            assert m.get_params()[0] is m._wq_cache
            assert torch.equal(m._wq_cache, m.weight_quantizer(m.weight))
        layers[0].weight.data = layers[0].weight.data.clone()     # storage replaced: the plan is rebuilt
        assert requantize_weights(q) == 4 and _PLANS[q][0] is not plan
        # the plan bakes the format in by value: a new mantissa width (same maxval storage) must not be served from it
        plan = _PLANS[q][0]
        wq = layers[1].weight_quantizer.quantizer
        wq.mantissa_bits = torch.tensor(3.0)
        assert requantize_weights(q) == 4 and _PLANS[q][0] is not plan
        for m in layers:
            assert torch.equal(m._wq_cache, m.weight_quantizer(m.weight))
        want = oracle.c_quantize(layers[1].weight.detach().cpu().numpy(), wq.maxval.cpu().numpy().reshape(-1), 3, 8, 1)
        assert np.array_equal(layers[1]._wq_cache.cpu().numpy().view(np.int32), want.view(np.int32))
        # a layer that left fix_ranges is no longer eligible: the rebuilt plan covers the remaining three
        layers[2].weight_quantizer.estimate_ranges()
        assert requantize_weights(q) == 3
        layers[2].weight_quantizer.fix_ranges()
        assert requantize_weights(q) == 4
        # the plan (a native handle) lives in a side table: models copy and pickle as in the reference's workflows
        q2 = copy.deepcopy(q)
        assert q2 not in _PLANS and q in _PLANS
        import io
        buf = io.BytesIO()
        torch.save(q, buf)
        x = dev(g7["calib"])
        assert torch.equal(q2(x), q(x))
        assert requantize_weights(q2) == 4 and q2 in _PLANS and _PLANS[q2][0] is not _PLANS[q][0]


def test_kernel_sequence_of_the_model_flow(golden_dir, monkeypatch):
    """The manager picks the cheapest kernel sequence (module docstring of quantization/manager.py): calibration =
    ONE fused min/max+quantize launch per weight tensor (weights are Parameters: the fast path must not be lost to
    requires_grad under no_grad) and min/max + quantize epilogues per activation; fix_ranges = one multi-tensor
    launch; validation = no weight launches at all (cache) and one launch per activation quantizer."""
    import collections
    from fp8q import ops
    from quantization.autoquant_utils import quantize_model
    from quantization.base_quantized_classes import QuantizedModule
    from quantization.base_quantized_model import prequantize_weights
    from quantization.quantization_manager import QMethods
    from quantization.range_estimators import RangeEstimators
    cnt = collections.Counter()
    for name in ("minmax", "minmax_quantize", "quantize", "affine_act_quantize", "affine_act_minmax", "multi_quantize"):
        def mk(name, f):
            def w(*a, **k):
                cnt[name + ("_per_channel" if name == "minmax" and a[1] else "")] += 1
                return f(*a, **k)
            return w
        monkeypatch.setattr(ops, name, mk(name, getattr(ops, name)))

    class CountedPlan(ops.MultiPlan):          # fix_ranges(): one prepared multi-tensor launch
        def launch(self):
            cnt["plan_launch"] += 1
            return super().launch()
    monkeypatch.setattr(ops, "MultiPlan", CountedPlan)
    g7 = np.load(os.path.join(golden_dir, "g7_tinycnn.npz"))
    q = quantize_model(_tiny_cnn(g7), method=QMethods.fp_quantizer.cls,
                       weight_range_method=RangeEstimators.current_minmax.cls,
                       act_range_method=RangeEstimators.allminmax.cls, n_bits=8, per_channel_weights=True,
                       fp8_kwargs=dict(maxval=None, mantissa_bits=2, set_maxval=True)).eval().cuda()
    x = dev(g7["calib"])
    with torch.no_grad():
        for m in q.modules():
            if isinstance(m, QuantizedModule):
                m.quantized()
        q(x)
        assert cnt["minmax_quantize"] == 4 and cnt["minmax_per_channel"] == 0, dict(cnt)   # 4 weight tensors, fused
        n_act = cnt["affine_act_quantize"] + cnt["quantize"]
        assert n_act >= 4 and cnt["affine_act_minmax"] + cnt["minmax"] <= n_act
        cnt.clear()
        for m in q.modules():
            if isinstance(m, QuantizedModule):
                m.fix_ranges()
        assert prequantize_weights(q) == 4 and cnt["plan_launch"] == 1
        cnt.clear()
        q(x)
        assert cnt["minmax_quantize"] == cnt["minmax"] == cnt["minmax_per_channel"] == cnt["affine_act_minmax"] == 0
        assert cnt["affine_act_quantize"] + cnt["quantize"] == n_act, dict(cnt)   # activations only: weights are cached


def test_mse_search_sharded_single_process_equals_estimator(golden_dir):
    """fp8q.dist.mse_search_sharded with one rank is FP_MSE_Estimator.forward (HIP ops on both sides)."""
    from fp8q import dist as fd
    from quantization.quantizers.fp8_quantizer import FPQuantizer
    from quantization.range_estimators import RangeEstimators
    g4 = np.load(os.path.join(golden_dir, "g4_mse.npz"))
    x = dev(g4["w_pc_srchm_x0"])
    q = FPQuantizer(n_bits=8, per_channel=True, mantissa_bits=3, set_maxval=True, maxval=None,
                    mse_include_mantissa_bits=True)
    est = RangeEstimators.MSE.cls(per_channel=True, quantizer=q)
    mn, mx = est(x)
    mv, m, state = fd.mse_search_sharded(x, True, [float(v) for v in range(1, 7)], 8, 1, "channel")
    assert torch.equal(mv, mx) and m == float(q.mantissa_bits)
    assert torch.equal(state[1], est.mses)


def test_quantized_forward_in_a_hip_graph(golden_dir):
    """Every entry point only enqueues on the caller's stream (no sync, no allocation of its own), so a
    fixed-range quantized forward can be captured in a HIP graph and replayed; the replay reproduces the
    eager forward bit for bit (launch-bound small batches: 2-3x faster, tools/model_forward_bench.py)."""
    from quantization.autoquant_utils import quantize_model
    from quantization.base_quantized_classes import QuantizedModule
    from quantization.quantization_manager import QMethods
    from quantization.range_estimators import RangeEstimators
    g7 = np.load(os.path.join(golden_dir, "g7_tinycnn.npz"))
    qparams = dict(method=QMethods.fp_quantizer.cls, weight_range_method=RangeEstimators.current_minmax.cls,
                   act_range_method=RangeEstimators.allminmax.cls, n_bits=8, per_channel_weights=True,
                   fp8_kwargs=dict(maxval=None, mantissa_bits=3, set_maxval=True))
    q = quantize_model(_tiny_cnn(g7), **qparams).eval().cuda()
    calib, val = dev(g7["calib"]), dev(g7["val"])
    with torch.no_grad():
        for m in q.modules():
            if isinstance(m, QuantizedModule):
                m.quantized()
        q(calib)
        for m in q.modules():
            if isinstance(m, QuantizedModule):
                m.fix_ranges()
        eager = q(val).clone()
        assert torch.equal(q(val), eager)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            q(val)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = q(val)
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager)
        val.mul_(0.5)                      # new input in the captured buffer -> new result, still equal to eager
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, q(val))


def test_quantized_checkpoint_restores_ranges(golden_dir, tmp_path):
    """a saved quantized model keeps its calibrated FP8 ranges (the reference loses
    them: maxval is not a buffer).  Save after calibration, load into a fresh model, same logits."""
    from quantization.autoquant_utils import quantize_model
    from quantization.base_quantized_model import QuantizedModel, RANGES_KEY
    from quantization.quantization_manager import QMethods, QuantizationManager, Qstates
    from quantization.range_estimators import RangeEstimators
    g7 = np.load(os.path.join(golden_dir, "g7_tinycnn.npz"))
    qparams = dict(method=QMethods.fp_quantizer.cls, weight_range_method=RangeEstimators.current_minmax.cls,
                   act_range_method=RangeEstimators.allminmax.cls, n_bits=8, per_channel_weights=True,
                   fp8_kwargs=dict(maxval=None, mantissa_bits=3, set_maxval=True))

    class Net(QuantizedModel):
        def __init__(self):
            super().__init__(input_size=(1, 3, 16, 16))
            self.body = quantize_model(_tiny_cnn(g7), tie_activation_quantizers=True, **qparams)

        def forward(self, x):
            return self.body(x)

    a = Net().eval().cuda()
    calib, val = dev(g7["calib"]), dev(g7["val"])
    with torch.no_grad():
        a.set_quant_state(True, True)
        a(calib)
        a.fix_ranges()
        ref = a(val)
    sd = a.state_dict_with_ranges()
    assert RANGES_KEY in sd and len(sd[RANGES_KEY]) == 8 and RANGES_KEY not in a.state_dict()
    path = tmp_path / "q.pt"
    torch.save(sd, path)
    b = Net().eval().cuda()
    b.load_state_dict(torch.load(path, weights_only=False))
    assert all(m.state == Qstates.fix_ranges for m in b.modules() if isinstance(m, QuantizationManager))
    with torch.no_grad():
        out = b(val)
    assert torch.equal(out, ref)
    # without the extra entry (a reference-style checkpoint) loading still works, ranges are re-estimated
    c = Net().eval().cuda()
    sd.pop(RANGES_KEY)
    c.load_state_dict(sd)
    assert c.body[0].weight_quantizer.state == Qstates.estimate_ranges


def test_dist_helpers_single_process_on_hip():
    """fp8q.dist with its default backend (the HIP ops), world size 1 (no process group)."""
    import oracle
    from fp8q import dist as fd
    rng = np.random.RandomState(3)
    w = (rng.randn(13, 3, 3, 3) * 0.1).astype(np.float32)
    mn, mx = oracle.c_minmax(w, True)
    mv = oracle.c_absmax(mn, mx)
    ref = oracle.c_quantize(w, mv, 2, 8, 1)
    q, m = fd.quantize_weight_sharded(dev(w), 2, 8, 1)
    assert np.array_equal(q.cpu().numpy().view(np.int32), ref.view(np.int32))
    q2, m2, codes = fd.quantize_weight_sharded_codes(dev(w), 2, 8, 1)
    assert np.array_equal(q2.cpu().numpy().view(np.int32), ref.view(np.int32)) and codes.dtype == torch.uint8
    np.testing.assert_array_equal(m2.cpu().numpy(), mv)
    ws = [w, (rng.randn(8, 13) * 0.3).astype(np.float32), (rng.randn(1, 7)).astype(np.float32),
          (rng.randn(64, 3, 7, 7) * 0.1).astype(np.float32)]
    for bucket_bytes in (None, 1, 4096):                      # one bucket / one per tensor / packed by size
        outs = fd.quantize_weights_sharded_bucketed([dev(t) for t in ws], 2, 8, 1, bucket_bytes=bucket_bytes)
        for t, (qb, mb) in zip(ws, outs):
            tmn, tmx = oracle.c_minmax(t, True)
            tmv = oracle.c_absmax(tmn, tmx)
            np.testing.assert_array_equal(mb.cpu().numpy(), tmv)
            assert np.array_equal(qb.cpu().numpy().view(np.int32), oracle.c_quantize(t, tmv, 2, 8, 1).view(np.int32))
    x = (rng.randn(4, 8, 5, 5)).astype(np.float32)
    y, st = fd.calibrate_quantize_sharded(dev(x), 3, 8, 1)
    pmn, pmx = oracle.c_minmax(x, False)
    assert np.array_equal(y.cpu().numpy().view(np.int32),
                          oracle.c_quantize(x, oracle.c_absmax(pmn, pmx), 3, 8, 1).view(np.int32))


def test_fixed_range_weight_cache_is_invalidated_correctly(golden_dir):
    """Cached quantized weights (fixed ranges) equal the per-forward result and follow weight updates."""
    from quantization.autoquant_utils import QuantLinear
    from quantization.quantization_manager import QMethods
    from quantization.range_estimators import RangeEstimators
    torch.manual_seed(0)
    lin = QuantLinear(32, 16, method=QMethods.fp_quantizer.cls, weight_range_method=RangeEstimators.current_minmax.cls,
                      act_range_method=RangeEstimators.allminmax.cls, per_channel_weights=True,
                      fp8_kwargs=dict(mantissa_bits=3, set_maxval=True, maxval=None)).cuda().eval()
    lin.quantized_weights()
    x = torch.randn(4, 32, device="cuda")
    with torch.no_grad():
        lin(x)                              # estimate state: no caching
        assert getattr(lin, "_wq_key", None) is None
        lin.fix_ranges()
        w1, _ = lin.get_params()
        w2, _ = lin.get_params()
        assert w1 is w2                     # reused
        assert torch.equal(w1, lin.weight_quantizer(lin.weight))
        lin.weight.mul_(0.5)                # in-place update bumps the version -> recomputed
        w3, _ = lin.get_params()
        assert w3 is not w1 and torch.equal(w3, lin.weight_quantizer(lin.weight))
        lin.weight_quantizer.quantizer.maxval = lin.weight_quantizer.quantizer.maxval * 2   # new range tensor
        w4, _ = lin.get_params()
        assert w4 is not w3 and torch.equal(w4, lin.weight_quantizer(lin.weight))


@pytest.mark.parametrize("per_channel", [True, False])
def test_percentile_estimator_close_to_the_references_numpy_percentile(per_channel):
    """CurrentMinMaxEstimator(percentile=p) (range_estimators.py:64-71; unreachable from the reference CLI: hijacker.py:57
    compares a class with an enum member): the reference takes numpy's percentile of a host copy -- a float64 result, which
    would then drag the whole quantizer into float64 --; here torch.quantile on the device, float32.  Pinned to numpy's values
    to float32 accuracy (not to its dtype)."""
    from quantization.range_estimators import CurrentMinMaxEstimator
    rng = np.random.RandomState(5)
    x = rng.randn(48, 300).astype(np.float32)
    for p in (0.1, 1.0, 5.0):
        est = CurrentMinMaxEstimator(percentile=p, per_channel=per_channel)
        lo, hi = est(torch.from_numpy(x).cuda())
        rlo, rhi = np.percentile(x, (p, 100 - p), axis=-1 if per_channel else None)
        np.testing.assert_allclose(lo.cpu().numpy(), rlo, rtol=2e-6, atol=2e-6)      # (float32 interpolation between two order statistics of O(1) values)
        np.testing.assert_allclose(hi.cpu().numpy(), rhi, rtol=2e-6, atol=2e-6)
