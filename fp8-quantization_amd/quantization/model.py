"""Whole-model switches (reference base_quantized_model.py:19-135)."""
import torch
from torch import nn

from .fp8 import QuantizerBase
from .layers import QuantizedModule, _for_managers


RANGES_KEY = "__fp8_quantizer_ranges__"

# model -> (plan, [(layer, weight, quantizer)], [signature]) of prequantize_weights().  A side table, not an attribute:
# the plan wraps a native handle (ctypes), which must neither be pickled / deep-copied with the model
# (copy.deepcopy(model) and torch.save(model) are everyday operations in the reference's workflows,
# autoquant_utils.py:375) nor end up owned by two Python objects.  A copy of a model simply has no plan yet.
import weakref  # noqa: E402

_PLANS = weakref.WeakKeyDictionary()


def _plan_signature(m, q):
    """Everything a prepared plan bakes in BY VALUE for one layer (csrc plan_build: d.f = make_fmt(mbits, n_bits,
    sign_bits)) plus the conditions under which the layer was eligible: if any of it changes, the plan is stale."""
    mgr = m.weight_quantizer
    return (float(q.mantissa_bits), int(q.sign_bits), int(q.n_bits), getattr(q, "_range_epoch", None), mgr.state,
            bool(getattr(m, "_qw", False)))


def materialize_mantissa_bits(model, modules=None):
    """Bring every mantissa width that is still pending on the GPU (the MSE estimators' votes) to the host in ONE
    device-to-host copy; afterwards every quantizer passes its width by value again (the tuned K1 routes).
    `modules`: list(model.modules()) when the caller already has it (one walk of a 500-module model is ~0.4 ms)."""
    pending = [(q, q._pending_mantissa_bits()) for q in (model.modules() if modules is None else modules)
               if hasattr(q, "_pending_mantissa_bits")]
    pending = [(q, t) for q, t in pending if t is not None]
    if not pending:
        return 0
    # votes written by this engine live in a per-device arena (fp8q.ops.vote_slot): ONE device-to-host copy per arena; a width
    # that somebody else put on the device (an assigned CUDA tensor) is gathered the ordinary way
    from fp8q import ops as _ops
    hosts, rest = {}, []
    for q, t in pending:
        where = _ops.vote_arena_of(t)
        if where is None:
            rest.append((q, t))
            continue
        buf, slot = where
        host = hosts.get(id(buf))
        if host is None:
            host = hosts[id(buf)] = (buf, buf.cpu())
        q.__dict__["_mbits_host"] = host[1][slot:slot + 1].clone()       # (not an assignment: the range epoch stays)
    if rest:
        host = torch.cat([t.detach().reshape(1).float() for _, t in rest]).cpu()
        for (q, _), v in zip(rest, host):
            q.__dict__["_mbits_host"] = v.reshape(1).clone()
    return len(pending)


def materialize_sign_bits(model, modules=None):
    """Bring every sign flag that is still pending on the GPU (FPQuantizer.set_quant_range with allow_unsigned) to the host
    in ONE device-to-host copy per device."""
    pending = [(q, q._pending_sign_bits()) for q in (model.modules() if modules is None else modules)
               if hasattr(q, "_pending_sign_bits")]
    pending = [(q, t) for q, t in pending if t is not None]
    by_dev = {}
    for q, t in pending:
        by_dev.setdefault(t.device, []).append((q, t))
    for items in by_dev.values():
        host = torch.cat([t.reshape(1) for _, t in items]).cpu().tolist()
        for (q, _), v in zip(items, host):
            q.__dict__["_sign_host"] = int(v)            # (not an assignment: the range epoch stays)
            if not v:
                q.__dict__["_sign_dev"] = None
    return len(pending)


def quantizer_ranges(model):
    """{manager name: {maxval, mantissa_bits, sign_bits, state}} for every FP8 quantizer of `model`.

    SURVEY.md 8f N4: in the reference `maxval` / `mantissa_bits` are plain attributes, not buffers
    (fp8_quantizer.py:183-184), so a saved "quantized" checkpoint silently loses every calibrated
    range.  This is the missing piece, kept OUT of the regular state-dict keys (which stay
    identical to the reference's) and stored under one extra entry, RANGES_KEY."""
    from .manager import QuantizationManager
    from .fp8 import FPQuantizer
    out, seen = {}, set()
    for name, m in model.named_modules():
        if isinstance(m, QuantizationManager) and isinstance(m.quantizer, FPQuantizer) and id(m) not in seen:
            seen.add(id(m))
            q = m.quantizer
            out[name] = dict(maxval=q.maxval.detach().cpu().clone(),
                             mantissa_bits=q.mantissa_bits.detach().cpu().clone(),
                             sign_bits=int(q.sign_bits), state=m.state.name)
    return out


def load_quantizer_ranges(model, ranges, strict=True):
    from .manager import QuantizationManager, Qstates
    mgrs = {n: m for n, m in model.named_modules() if isinstance(m, QuantizationManager)}
    missing = [n for n in ranges if n not in mgrs]
    if strict and missing:
        raise KeyError(f"quantizer ranges for unknown managers: {missing[:5]}")
    for name, r in ranges.items():
        m = mgrs.get(name)
        if m is None:
            continue
        q = m.quantizer
        dev = q.maxval.device
        q.maxval = r["maxval"].to(dev).clone()
        q.mantissa_bits = r["mantissa_bits"].clone()
        q.sign_bits = int(r["sign_bits"])
        m.state = q.state = Qstates[r["state"]]


def _plan_layers(model, modules=None):
    """[(layer, weight, quantizer, maxval, swap)] of every layer a multi-tensor plan can cover: FP8 weight quantizer with
    fixed ranges on a contiguous CUDA fp32 weight that autograd is not tracking.  swap: a transposed convolution with
    per-channel ranges -- its weight is [in, out, ...] and the quantizer works on the [out, in, ...] copy
    (reference autoquant_utils.py:46-58); the plan quantizes that copy."""
    import os

    from .fp8 import FPQuantizer
    from .layers import QuantizationHijacker, QuantConvTransposeBase
    from .manager import Qstates
    if os.environ.get("FP8Q_CACHE_WEIGHTS", "1") == "0":
        return []
    found = []
    for m in (model.modules() if modules is None else modules):
        if not isinstance(m, QuantizationHijacker) or not getattr(m, "_qw", False):
            continue
        swap = False
        if type(m).quantize_weights is QuantConvTransposeBase.quantize_weights:
            swap = bool(m.per_channel_weights)
        elif type(m).quantize_weights is not QuantizationHijacker.quantize_weights:
            continue   # a layer with its own idea of quantize_weights: leave it to the layer
        mgr = m.weight_quantizer
        q = getattr(mgr, "quantizer", None)
        if not isinstance(q, FPQuantizer) or mgr.state != Qstates.fix_ranges or q.maxval is None:
            continue
        w = m.get_weight_bias()[0]
        if not (isinstance(w, torch.Tensor) and w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()):
            continue
        if w.requires_grad and torch.is_grad_enabled():
            continue
        if swap and w.dim() < 2:
            continue
        mv = q.maxval.detach()
        if not (mv.is_cuda and mv.dtype == torch.float32) or mv.numel() not in (1, w.shape[1] if swap else w.shape[0]):
            continue
        found.append((m, w, q, mv, swap))
    return found


def _store_planned(mods, outs):
    """hand every layer its quantized weight; a swapped layer gets it back in its own [in, out, ...] layout"""
    for (m, w, q, src), y in zip(mods, outs):
        m._wq_cache = y if src is None else y.transpose(1, 0).contiguous()
        m._wq_key = m._weight_cache_key(m.get_weight_bias()[0], q)


def prequantize_weights(model, modules=None):
    """With fixed ranges every layer's weight quantization is independent of the data: instead of one
    small launch per layer in its first forward (the reference: hijacker.py:88-98, every forward), all FP8
    weight tensors go through ONE multi-tensor launch (fp8q_multi_quantize_f32) and fill the layers' caches.
    Bit-identical to the per-layer path; a no-op for anything it does not cover (CPU tensors, INT
    quantizers, layers that override quantize_weights, FP8Q_CACHE_WEIGHTS=0).  Returns the number of layers."""
    found = _plan_layers(model, modules)
    if not found:
        _PLANS.pop(model, None)
        return 0
    # (layer, weight, quantizer, the [out, in, ...] copy of a swapped layer's weight or None)
    mods = [(m, w, q, w.detach().transpose(1, 0).contiguous() if swap else None) for m, w, q, mv, swap in found]
    items = [(w.detach() if src is None else src, mv, float(q.mantissa_bits), int(q.n_bits), int(q.sign_bits))
             for (m, w, q, src), (_, _, _, mv, _) in zip(mods, found)]
    import fp8q
    # a prepared plan (fp8q_multi_plan_*): descriptors validated and packed once here; requantize_weights() replays
    # it with one launch whenever the weights' or the ranges' CONTENTS change (QAT steps, range updates in place)
    plan = fp8q.ops.MultiPlan(items)
    _store_planned(mods, plan.launch())
    _PLANS[model] = (plan, mods, [_plan_signature(m, q) + (w.data_ptr(), tuple(w.shape)) for m, w, q, src in mods])
    return len(mods)


def requantize_weights(model):
    """Refresh every layer's cached quantized weight after the weights (or the range tensors) changed IN PLACE: one
    call into the prepared plan built by prequantize_weights() = one kernel launch for all layers (the reference
    re-quantizes layer by layer in every forward, hijacker.py:88-98).  The plan holds addresses and, BY VALUE, every
    layer's format: it is replayed only if the set of eligible layers, their tensors' storage and shapes, their
    format (mantissa / sign / total bits) and their state are what they were when it was built; anything else --
    a tensor replaced rather than updated, a new mantissa width, a layer that left or entered fix_ranges, a weight that
    became trainable -- rebuilds it (prequantize_weights).  Returns the number of layers refreshed."""
    held = _PLANS.get(model)
    if held is None:
        return prequantize_weights(model)
    plan, mods, sigs = held
    now = _plan_layers(model)
    if len(now) != len(mods):
        return prequantize_weights(model)
    for (m, w, q, src), (x, mv), sig, (m2, w2, q2, mv2, swap2) in zip(mods, plan._keep, sigs, now):
        if (m2 is not m or q2 is not q or w2.data_ptr() != sig[-2] or tuple(w2.shape) != sig[-1] or swap2 != (src is not None)
                or mv2.data_ptr() != mv.data_ptr() or mv2.numel() != mv.numel()):
            return prequantize_weights(model)
        cur = _plan_signature(m, q)
        if cur[:3] != sig[:3] or cur[4:] != sig[4:6]:     # the range epoch (index 3) moves with in-place range updates: fine
            return prequantize_weights(model)
    for m, w, q, src in mods:
        if src is not None:
            src.copy_(m.get_weight_bias()[0].detach().transpose(1, 0))     # the plan reads this copy: refresh it in place
    _store_planned(mods, plan.launch())
    return len(mods)


def export_fp8_weights(model):
    """{layer name: {codes uint8 [like the weight], maxval [1] or [C], mantissa_bits, sign_bits, n_bits}} for every
    layer whose weights go through an FP8 quantizer with fixed ranges: the 1-byte storage form of what the layer
    computes with (the reference only simulates the format; its enumerator
    fp8_quantizer.py:13-41 defines the byte layout).  decode(codes) == the layer's quantized weight (bit for bit for weight-sized
    ranges; within a few ULP on round-ups into the next binade otherwise, see include/fp8q.h)."""
    import torch

    import fp8q
    from .fp8 import FPQuantizer
    from .layers import QuantizationHijacker
    from .manager import Qstates
    out = {}
    for name, m in model.named_modules():
        if not isinstance(m, QuantizationHijacker) or not getattr(m, "_qw", False):
            continue
        if type(m).quantize_weights is not QuantizationHijacker.quantize_weights:
            continue
        mgr = m.weight_quantizer
        q = getattr(mgr, "quantizer", None)
        if not isinstance(q, FPQuantizer) or mgr.state != Qstates.fix_ranges or q.maxval is None:
            continue
        w = m.get_weight_bias()[0].detach()
        if not w.is_cuda:
            continue
        mv = q.maxval.detach().to(device=w.device, dtype=torch.float32).reshape(-1)
        codes = fp8q.ops.encode(w.contiguous(), mv, float(q.mantissa_bits), int(q.n_bits), int(q.sign_bits))
        out[name] = dict(codes=codes.cpu(), maxval=mv.cpu(), mantissa_bits=float(q.mantissa_bits),
                         sign_bits=int(q.sign_bits), n_bits=int(q.n_bits))
    return out


def decode_fp8_weights(exported, device="cuda"):
    """{layer name: fp32 tensor} from export_fp8_weights(): the values the layers compute with."""
    import fp8q
    return {name: fp8q.ops.decode(e["codes"].to(device), e["maxval"].to(device), e["mantissa_bits"], e["n_bits"],
                                  e["sign_bits"]) for name, e in exported.items()}


class GraphedForward:
    """HIP graph of `model(x)` for one input shape.  With FIXED ranges nothing in a quantized forward is decided on
    the host (the engine's entry points only enqueue kernels on the current stream), so the whole forward can be
    captured once and replayed: bit-identical to the eager forward, 2-3x faster where the eager one is
    launch-bound (ResNet-18 batch 1: 1.31 -> 0.56 ms; batch 64 is GPU-bound: no change).
    `gf(x)` copies x into the captured input and replays; the result lives in a buffer that the next call
    overwrites.  Inputs of another shape (a ragged last batch) take the eager path."""

    def __init__(self, model, example):
        import torch
        self.model = model
        self.static_in = example.detach().clone()
        with torch.no_grad():
            side = torch.cuda.Stream(device=example.device)
            side.wait_stream(torch.cuda.current_stream(example.device))
            with torch.cuda.stream(side):
                model(self.static_in)          # warm-up outside the capture: lazy initialisation, weight caches
            torch.cuda.current_stream(example.device).wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.static_out = model(self.static_in)

    def __call__(self, x):
        import torch
        if x.shape != self.static_in.shape or x.dtype != self.static_in.dtype or x.device != self.static_in.device:
            with torch.no_grad():
                return self.model(x)
        self.static_in.copy_(x)
        self.graph.replay()
        return self.static_out


class GraphedCalibration:
    """Calibration forwards (estimate state) replayed from a HIP graph from the THIRD batch of a shape on.  A calibration
    forward of this engine enqueues only: every decision of the estimators (abs-max, search grid, MSE tables, vote, argmin, the
    quantization with the winner) is taken on the device, in state blocks that persist between batches -- so once the first
    batch has created that state (and taken the first-batch launches) the forward is a fixed sequence of launches on fixed
    addresses and can be captured.  Batch 1 and 2 run eagerly (2: steady-state allocations), batch 3 is captured and replayed,
    later batches copy their input into the captured buffer and replay: no Python, no launches from the host (the eager pass is
    ~830 launches + 116 library calls).  Results are those of the eager loop, bit for bit (tests/test_mse_onecall.py).
    Inputs of another shape (a ragged last batch) run eagerly."""

    def __init__(self, model):
        self.model = model
        self.static_in = self.static_out = self.graph = None
        self.seen = 0

    def __call__(self, x):
        if self.graph is not None and x.shape == self.static_in.shape and x.dtype == self.static_in.dtype \
                and x.device == self.static_in.device:
            self.static_in.copy_(x)
            self.graph.replay()
            return self.static_out
        if not x.is_cuda or (self.static_in is not None and x.shape != self.static_in.shape):
            with torch.no_grad():
                return self.model(x)
        self.seen += 1
        if self.seen <= 2:
            self.static_in = x.detach().clone()
            with torch.no_grad():
                return self.model(x)
        self.static_in.copy_(x)
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph):
            self.static_out = self.model(self.static_in)
        self.graph = graph
        graph.replay()                      # (capturing does not run anything: this is batch 3)
        return self.static_out


_SIDE_STREAMS = {}      # device index -> the stream weight calibrations run ahead on
_AHEAD_DEPTH = [0]      # QuantizedModel forwards in progress (nesting depth)


class _WeightsAhead:
    """The weight calibrations of one forward, running ahead of it on a side stream: `pending` = [(layer, weight)] in module
    order, not yet enqueued.  A few are enqueued before the forward starts, one more every time a layer picks its result up
    -- the host alternates between the two streams instead of enqueuing all 53 weight searches (2 ms of host time) before the
    first layer of the forward."""
    __slots__ = ("pending", "side", "main", "dev")
    LOOKAHEAD = 4

    def __init__(self, pending, side, main, dev):
        self.pending, self.side, self.main, self.dev = pending, side, main, dev

    def advance(self, n=1):
        if not self.pending:
            return
        with torch.cuda.stream(self.side):
            while n > 0 and self.pending:
                m, w = self.pending.pop(0)
                n -= 1
                wq = m.quantize_weights(w)
                ev = torch.cuda.Event()
                ev.record(self.side)
                if isinstance(wq, torch.Tensor) and wq.is_cuda:
                    wq.record_stream(self.main)        # allocated on the side stream, consumed on the forward's
                m.__dict__["_wq_ahead"] = (wq, ev, w.data_ptr(), w._version)

    def forget(self, layer):
        """`layer` is about to calibrate its weight itself (its turn came before its place in the queue: a forward that
        does not follow module order): it must not be calibrated a second time"""
        self.pending = [(m, w) for m, w in self.pending if m is not layer]


def calibrate_weights_ahead(model):
    """Weight quantizers in estimate state do not depend on the data: their range estimation + quantization (per layer one
    library call: fp8q_mse_calibrate_f32 / fp8q_minmax_quantize_f32; the reference runs it inside every layer's forward,
    hijacker.py:88-98) runs AHEAD of the forward on a side stream, and each layer waits for its own event.  The small,
    latency-bound launches of the 53 MobileNetV2 weight searches (2 ms of GPU time per batch with a fixed mantissa width,
    4 ms with the search) then overlap the activations' chains instead of standing in line with them: HIP streams, as the
    chip wants them.  Same kernels on the same inputs: bit-identical results; a layer the forward never reaches is never
    calibrated (as in the reference).
    Returns the number of layers queued (0: CPU model, nothing in estimate state, FP8Q_WEIGHTS_AHEAD=0)."""
    import os

    from .layers import QuantizationHijacker
    layers = model.__dict__.get("_hijackers")
    if layers is None:
        layers = model.__dict__["_hijackers"] = [m for m in model.modules() if isinstance(m, QuantizationHijacker)]
    for m in layers:                               # (whatever an earlier forward left behind)
        m.__dict__.pop("_ahead_ctl", None)
        m.__dict__.pop("_wq_ahead", None)
    if os.environ.get("FP8Q_WEIGHTS_AHEAD", "1") == "0":
        return 0
    todo = []
    dev = None
    for m in layers:
        if not m._qw:
            continue
        mgr = m.weight_quantizer
        if not mgr._estimating():
            continue
        w = m.get_weight_bias()[0]
        if not (isinstance(w, torch.Tensor) and w.is_cuda) or (w.requires_grad and torch.is_grad_enabled()):
            continue
        if dev is None:
            dev = w.device
        if w.device == dev:
            todo.append((m, w))
    if not todo:
        return 0
    main = torch.cuda.current_stream(dev)
    side = _SIDE_STREAMS.get(dev.index)
    if side is None:
        side = _SIDE_STREAMS[dev.index] = torch.cuda.Stream(device=dev)
    side.wait_stream(main)                    # whatever wrote the weights on the current stream comes first
    ctl = _WeightsAhead(list(todo), side, main, dev)
    for m, _ in todo:
        m.__dict__["_ahead_ctl"] = ctl
    ctl.advance(_WeightsAhead.LOOKAHEAD)
    return len(todo)


class QuantizedModel(nn.Module):
    def __init__(self, input_size=(1, 3, 224, 224)):
        super().__init__()
        self.input_size = input_size
        self.register_forward_pre_hook(QuantizedModel._weights_ahead_hook)
        self.register_forward_hook(QuantizedModel._weights_ahead_done, always_call=True)

    @staticmethod
    def _weights_ahead_hook(module, args):
        # (ranges fixed -- every validation forward: one attribute test)
        _AHEAD_DEPTH[0] += 1
        if _AHEAD_DEPTH[0] == 1 and not module.__dict__.get("_ranges_fixed", False) and args and isinstance(args[0], torch.Tensor) \
                and args[0].is_cuda and not torch.cuda.is_current_stream_capturing():
            # (only the OUTERMOST QuantizedModel of a forward: a nested one would throw away -- and repeat -- what the outer
            # one has already started for its layers)
            calibrate_weights_ahead(module)

    @staticmethod
    def _weights_ahead_done(module, args, output):
        _AHEAD_DEPTH[0] = max(_AHEAD_DEPTH[0] - 1, 0)

    def state_dict_with_ranges(self, *args, **kwargs):
        """state_dict() plus the calibrated FP8 ranges (see quantizer_ranges)."""
        sd = self.state_dict(*args, **kwargs)
        sd[RANGES_KEY] = quantizer_ranges(self)
        return sd

    def load_state_dict(self, state_dict, strict=True):
        """First restore the _quant_w/_quant_a flags, run one dummy forward so that every None
        buffer (estimator ranges) gets its shape, then load everything (:34-62).  If the dict
        carries RANGES_KEY the FP8 ranges and manager states are restored as well."""
        state_dict = dict(state_dict)
        ranges = state_dict.pop(RANGES_KEY, None)
        res = self._load_reference_state_dict(state_dict, strict)
        if ranges is not None:
            load_quantizer_ranges(self, ranges, strict)
        return res

    def _load_reference_state_dict(self, state_dict, strict=True):
        flags = {k: v for k, v in state_dict.items() if k.endswith("_quant_a") or k.endswith("_quant_w")}
        if not flags:
            raise ValueError("The quantization states of activations or weights should be "
                             "included in the state dict ")
        super().load_state_dict(flags, strict=False)
        device = next(self.parameters()).device
        with torch.no_grad():
            self.forward(torch.rand(*self.input_size, device=device))
        return super().load_state_dict(state_dict, strict)

    def _each(self, method):
        def visit(layer):
            if isinstance(layer, QuantizedModule):
                getattr(layer, method)()
        self.apply(visit)

    def quantized_weights(self):
        self._each("quantized_weights")

    def full_precision_weights(self):
        self._each("full_precision_weights")

    def quantized_acts(self):
        self._each("quantized_acts")

    def full_precision_acts(self):
        self._each("full_precision_acts")

    def quantized(self):
        self._each("quantized")

    def full_precision(self):
        self._each("full_precision")

    def set_quant_state(self, weight_quant, act_quant):
        (self.quantized_acts if act_quant else self.full_precision_acts)()
        (self.quantized_weights if weight_quant else self.full_precision_weights)()

    def grad_scaling(self, grad_scaling=True):
        def visit(m):
            if isinstance(m, QuantizerBase):
                m.grad_scaling = grad_scaling
        self.apply(visit)

    def estimate_ranges(self):
        self.__dict__["_ranges_fixed"] = False
        _for_managers(self, lambda m: m.estimate_ranges(), need_init=False)

    def estimate_ranges_train(self):
        self.__dict__["_ranges_fixed"] = False
        _for_managers(self, lambda m: m.estimate_ranges_train(), need_init=True)

    def learn_ranges(self):
        self.__dict__["_ranges_fixed"] = False
        _for_managers(self, lambda m: m.learn_ranges(), need_init=True)

    def fix_ranges(self):
        self.__dict__["_ranges_fixed"] = True
        for idx, side in _SIDE_STREAMS.items():        # ranges written by calibrate_weights_ahead(): ordered before what follows
            torch.cuda.current_stream(idx).wait_stream(side)
        from .manager import QuantizationManager
        mods = list(self.modules())              # ONE walk for the three steps below (each walk of MobileNetV2: ~0.4 ms)
        for m in mods:                           # (= _for_managers(self, fix_ranges, need_init=True); apply() visits children first,
            if isinstance(m, QuantizationManager) and m.quantizer.is_initialized:      # the managers do not depend on the order)
                m.fix_ranges()
        materialize_mantissa_bits(self, mods)
        materialize_sign_bits(self, mods)
        # end of calibration = the one place where a host sync is free: surface what the enqueue-only min/max
        # launches could not report (a reducer block that timed out -> NaN range; a dirty workspace)
        import fp8q
        fp8q.ops.check_workspaces()
        dev = next((p.device for p in self.parameters() if p.is_cuda), None)
        fp8q.ops.release_workspaces(device=dev)   # the MSE search's scratch (4 B per element of the largest activation): not needed again
        prequantize_weights(self, mods)

    def prequantize_weights(self):
        return prequantize_weights(self)

    def requantize_weights(self):
        return requantize_weights(self)
