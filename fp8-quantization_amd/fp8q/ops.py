"""Torch-facing wrappers of the C ABI: tensors in, tensors out, launched on the current stream.

PyTorch is used for device memory and streams only; all arithmetic is in libfp8q_hip.so.
"""
import torch

from ._lib import Fp8qError, check, lib

FOLD_CURRENT, FOLD_ALL, FOLD_RUNNING = 0, 1, 2

_ws_cache = {}
_ws_retired = []     # outgrown min/max workspaces, not yet inspected by check_workspaces()
_mm_ws_bytes = {}


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t):
    """raw hipStream_t of torch's current stream on t's device (fast path avoids a Stream object)."""
    if _raw_stream is not None:
        return _raw_stream(t.device.index)
    return torch.cuda.current_stream(t.device).cuda_stream


class _on_device:
    """`with _on_device(t):` -- make t's device current for the launch; a no-op (and no torch call
    beyond one integer query) when it already is, which is the single-GPU-per-process case."""
    __slots__ = ("idx", "prev")

    def __init__(self, t):
        self.idx = t.device.index
        self.prev = None

    def __enter__(self):
        cur = torch.cuda.current_device()
        if cur != self.idx:
            self.prev = cur
            torch.cuda.set_device(self.idx)

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)
        return False


def _require(t, name, dtype=torch.float32, like=None):
    """t is a CUDA(HIP) tensor of `dtype` (a dtype or a tuple of them), on `like`'s device when given."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise Fp8qError(f"{name} must be a CUDA(HIP) tensor: the FP8 engine has no CPU path")
    if t.dtype != dtype and not (isinstance(dtype, tuple) and t.dtype in dtype):
        want = " or ".join(str(d).replace("torch.", "") for d in (dtype if isinstance(dtype, tuple) else (dtype,)))
        raise Fp8qError(f"{name} must be {want}, got {t.dtype}")
    if like is not None and t.device != like.device:
        raise Fp8qError(f"{name} is on {t.device}, expected {like.device} (all tensors of a call live on one device)")


def _out(out, like, dtype=None, name="out"):
    """The output tensor of a launch: a fresh one, or the caller's `out=` after checking that the kernel may write
    like.numel() elements of `dtype` through its raw pointer (same device, dtype, element count, contiguous)."""
    dtype = like.dtype if dtype is None else dtype
    if out is None:
        return torch.empty(like.shape, dtype=dtype, device=like.device)
    _require(out, name, dtype, like)
    if out.numel() != like.numel() or not out.is_contiguous():
        raise Fp8qError(f"{name} must be a contiguous tensor of {like.numel()} elements "
                        f"(got {tuple(out.shape)}, contiguous={out.is_contiguous()})")
    return out


def _dense_flat(x):
    """x as a flat [numel] view over its own storage when x is dense but not contiguous (channels-last activations, a
    transposed matrix: every element of the storage span belongs to x exactly once), else None.  Per-tensor K1 and the
    per-tensor min/max are elementwise / order-free, so they can run on the storage as it lies -- no NCHW copy (8 B per
    element) -- and the result keeps x's strides, as the reference's elementwise ATen ops do."""
    if x.is_contiguous() or x.numel() == 0:
        return None
    dims = sorted((st, sz) for st, sz in zip(x.stride(), x.shape) if sz > 1)
    expect = 1
    for st, sz in dims:
        if st != expect:
            return None
        expect *= sz
    return x.as_strided((x.numel(),), (1,), x.storage_offset())


def _rows(x, per_channel):
    """[C, inner] view geometry: channel = dim 0 (reference: x.view(x.shape[0], -1))."""
    if per_channel:
        C = x.shape[0] if x.dim() > 0 else 1
        return C, (x.numel() // C if C else 0)
    return 1, x.numel()


def _workspace(dev, nbytes, zeroed=False, kind=None, stream=None):
    """Per-(device, stream) scratch buffer.  zeroed=True: the min/max entry points' workspace, whose leading ticket
    counters must be zero on first use and are left zero by every call (include/fp8q.h) -- allocated zero-filled and
    never shared with the kernels that scribble over their scratch (MSE partial sums).  kind="select": the winner
    selection's buffer -- allocated zero-filled too (its header holds a ticket that every call leaves zero), but the rest
    of it is ordinary scratch, so it is neither shared with the min/max workspace nor inspected by check_workspaces()."""
    if stream is None:       # (callers that make several requests per launch pass the raw stream they already looked up)
        stream = _raw_stream(dev.index) if _raw_stream is not None else torch.cuda.current_stream(dev).cuda_stream
    key = (dev.index, stream, zeroed, kind)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        if ws is not None and zeroed:
            # a min/max workspace outgrown: its header may hold a time-out count that nobody has looked at yet -- keep it until
            # the next check_workspaces() instead of dropping the report with the buffer.  Never inspected HERE: that would
            # synchronise inside an unrelated enqueue-only op (and raise another call's failure from it).  The list stays
            # short: a workspace only ever grows, and these buffers are tens of kilobytes.
            _ws_retired.append((key, ws))
        alloc = torch.zeros if (zeroed or kind == "select") else torch.empty
        ws = alloc(max(int(nbytes), 1 << 16), dtype=torch.uint8, device=dev)
        _ws_cache[key] = ws
    return ws


def quantize(x, maxval, mbits, n_bits=8, sign_bits=1, out=None):
    """K1: FP8 quantize+dequantize (fp8_quantizer.py:91-133).  maxval: CUDA fp32 tensor [1] or [C]; x float32 or
    float64; mbits: a number, or a 1-element CUDA float32 tensor (read by the kernel: no host round trip)."""
    _require(x, "x", (torch.float32, torch.float64))
    _require(maxval, "maxval", like=x)
    maxval = maxval.contiguous().view(-1)
    n_mv = maxval.numel()
    flat = _dense_flat(x) if (n_mv == 1 and out is None) else None
    if flat is not None:            # per tensor on a dense non-contiguous layout: the storage as it lies, x's strides kept
        res = torch.empty_like(x)                            # (preserve_format: x's strides)
        flat_out = _dense_flat(res) if res.stride() == x.stride() else None
        if flat_out is not None:
            quantize(flat, maxval, mbits, n_bits, sign_bits, out=flat_out)
            return res
    x = x.contiguous()
    C, inner = _rows(x, n_mv != 1)
    if n_mv != 1 and n_mv != C:
        raise Fp8qError(f"maxval has {n_mv} elements, expected 1 or {C}")
    y = _out(out, x)
    if isinstance(mbits, torch.Tensor) and mbits.is_cuda:
        # mantissa width in DEVICE memory (the MSE estimator's vote, not yet brought to the host): fp8q_quantize_dm_f32
        _require(mbits, "mbits", like=x)
        if mbits.numel() != 1 or x.dtype != torch.float32:
            raise Fp8qError("a device-resident mantissa width must be a 1-element float32 tensor, x float32")
        if isinstance(sign_bits, torch.Tensor):     # ... and the sign still a device flag (sign_fold): fp8q_quantize_dms_f32
            _require(sign_bits, "sign_bits", torch.uint8, like=x)
            if sign_bits.numel() != 1:
                raise Fp8qError("a device-resident sign flag must be a 1-element uint8 tensor")
            with _on_device(x):
                rc = lib().fp8q_quantize_dms_f32(x.data_ptr(), y.data_ptr(), C, inner, maxval.data_ptr(), n_mv,
                                                 mbits.data_ptr(), int(n_bits), sign_bits.data_ptr(), _stream(x))
            check(rc, "fp8q_quantize_dms_f32")
            return y
        with _on_device(x):
            rc = lib().fp8q_quantize_dm_f32(x.data_ptr(), y.data_ptr(), C, inner, maxval.data_ptr(), n_mv,
                                            mbits.data_ptr(), int(n_bits), int(sign_bits), _stream(x))
        check(rc, "fp8q_quantize_dm_f32")
        return y
    if isinstance(sign_bits, torch.Tensor):
        # sign_bits still a flag in DEVICE memory (sign_fold: allow_unsigned decided without a host round trip): fp8q_quantize_ds_f32
        _require(sign_bits, "sign_bits", torch.uint8, like=x)
        if sign_bits.numel() != 1 or x.dtype != torch.float32:
            raise Fp8qError("a device-resident sign flag must be a 1-element uint8 tensor, x float32")
        with _on_device(x):
            rc = lib().fp8q_quantize_ds_f32(x.data_ptr(), y.data_ptr(), C, inner, maxval.data_ptr(), n_mv, float(mbits),
                                            int(n_bits), sign_bits.data_ptr(), _stream(x))
        check(rc, "fp8q_quantize_ds_f32")
        return y
    # float64 input (BASELINE config 1): the reference's chain under ATen's type promotion -- bias in float32,
    # everything downstream of x in float64 (fp8q_quantize_f64)
    fn = lib().fp8q_quantize_f64 if x.dtype == torch.float64 else lib().fp8q_quantize_f32
    with _on_device(x):
        rc = fn(x.data_ptr(), y.data_ptr(), C, inner, maxval.data_ptr(), n_mv, float(mbits), int(n_bits),
                int(sign_bits), _stream(x))
    check(rc, "fp8q_quantize_f64" if x.dtype == torch.float64 else "fp8q_quantize_f32")
    return y


def sign_fold(x_min, signed_flag=None):
    """FPQuantizer.set_quant_range's sign decision on the device (fp8_quantizer.py:216-225): the 1-element uint8 flag
    (1 = signed; a fresh one when None) is cleared when every value of x_min is >= 0, and never set again.  Enqueue-only."""
    _require(x_min, "x_min")
    if signed_flag is None:
        signed_flag = torch.ones(1, dtype=torch.uint8, device=x_min.device)
    _require(signed_flag, "signed_flag", torch.uint8, like=x_min)
    xm = x_min.detach().contiguous().view(-1)
    with _on_device(xm):
        rc = lib().fp8q_sign_fold_u8(xm.data_ptr(), xm.numel(), signed_flag.data_ptr(), _stream(xm))
    check(rc, "fp8q_sign_fold_u8")
    return signed_flag


def _pack_descs(items):
    """ctypes descriptor array of (x, maxval, mbits[, n_bits[, sign_bits[, out]]]) items; returns (descs, outs, keep)."""
    from ._lib import TensorDesc
    descs = (TensorDesc * len(items))()
    outs, keep = [], []
    dev0 = None
    for d, it in zip(descs, items):
        x, maxval, mbits = it[0], it[1], it[2]
        n_bits = it[3] if len(it) > 3 else 8
        sign_bits = it[4] if len(it) > 4 else 1
        out = it[5] if len(it) > 5 else None
        _require(x, "x")
        _require(maxval, "maxval", like=x)
        if dev0 is None:
            dev0 = x.device
        elif x.device != dev0:
            raise Fp8qError("multi_quantize: all tensors must be on one device")
        x = x.contiguous()
        maxval = maxval.contiguous().view(-1)
        n_mv = maxval.numel()
        C, inner = _rows(x, n_mv != 1)
        if n_mv != 1 and n_mv != C:
            raise Fp8qError(f"maxval has {n_mv} elements, expected 1 or {C}")
        y = _out(out, x)
        d.x, d.y, d.maxval = x.data_ptr(), y.data_ptr(), maxval.data_ptr()
        d.C, d.inner, d.n_maxval = C, inner, n_mv
        d.mbits, d.n_bits, d.sign_bits = float(mbits), int(n_bits), int(sign_bits)
        outs.append(y)
        keep.append((x, maxval))
    return descs, outs, keep


def multi_quantize(items):
    """Multi-tensor K1: quantize many tensors (all weights of a model) in one launch per 32 tensors.

    items: iterable of (x, maxval, mbits[, n_bits[, sign_bits[, out]]]); every x on the same device.
    Returns the list of outputs (bit-identical to quantize() on each item)."""
    items = [tuple(it) for it in items]
    if not items:
        return []
    descs, outs, keep = _pack_descs(items)
    with _on_device(keep[0][0]):
        rc = lib().fp8q_multi_quantize_f32(descs, len(items), _stream(keep[0][0]))
    check(rc, "fp8q_multi_quantize_f32")
    return outs


def multi_minmax_quantize(items):
    """Multi-tensor K2+K5+K1: per-channel current_minmax ranges + quantize of many tensors in TWO launches
    (fp8q_multi_minmax_quantize_f32).  items: iterable of (x, maxval_out, mbits[, n_bits[, sign_bits[, out]]]);
    maxval_out is a [C] CUDA fp32 tensor that RECEIVES the ranges.  Returns the list of outputs (bit-identical to
    minmax_quantize() on each item)."""
    items = [tuple(it) for it in items]
    if not items:
        return []
    for it in items:
        if it[1].numel() != (it[0].shape[0] if it[0].dim() > 0 else 1) or not it[1].is_contiguous():
            raise Fp8qError("multi_minmax_quantize: maxval_out must be a contiguous [C] tensor")
    descs, outs, keep = _pack_descs(items)
    for (x, mv), it in zip(keep, items):
        if mv.data_ptr() != it[1].data_ptr():
            raise Fp8qError("multi_minmax_quantize: maxval_out must be contiguous (a copy would receive the ranges)")
    import ctypes
    mv_out = (ctypes.c_void_p * len(items))(*[mv.data_ptr() for _, mv in keep])      # where the ranges are WRITTEN
    with _on_device(keep[0][0]):
        rc = lib().fp8q_multi_minmax_quantize_f32(descs, mv_out, len(items), _stream(keep[0][0]))
    check(rc, "fp8q_multi_minmax_quantize_f32")
    return outs


def _pack_codec_descs(items, encode):
    """descriptors of (values-or-codes, maxval, mbits[, n_bits[, sign_bits[, out]]]) items for the multi-tensor codec:
    encode: x float32 -> out uint8; decode: x uint8 -> out float32.  Returns (descs, outs, keep)."""
    from ._lib import TensorDesc
    descs = (TensorDesc * len(items))()
    outs, keep = [], []
    src_dt, dst_dt = (torch.float32, torch.uint8) if encode else (torch.uint8, torch.float32)
    for d, it in zip(descs, items):
        x, maxval, mbits = it[0], it[1], it[2]
        n_bits = it[3] if len(it) > 3 else 8
        sign_bits = it[4] if len(it) > 4 else 1
        out = it[5] if len(it) > 5 else None
        _require(x, "x", src_dt)
        _require(maxval, "maxval", like=x)
        if keep and x.device != keep[0][0].device:
            raise Fp8qError("multi-tensor codec: all tensors must be on one device")
        x = x.contiguous()
        maxval = maxval.contiguous().view(-1)
        n_mv = maxval.numel()
        C, inner = _rows(x, n_mv != 1)
        if n_mv != 1 and n_mv != C:
            raise Fp8qError(f"maxval has {n_mv} elements, expected 1 or {C}")
        y = _out(out, x, dtype=dst_dt)
        d.x, d.y, d.maxval = x.data_ptr(), y.data_ptr(), maxval.data_ptr()
        d.C, d.inner, d.n_maxval = C, inner, n_mv
        d.mbits, d.n_bits, d.sign_bits = float(mbits), int(n_bits), int(sign_bits)
        outs.append(y)
        keep.append((x, maxval))
    return descs, outs, keep


def multi_minmax_encode(items):
    """Per-channel current_minmax ranges + storage codes of many tensors in TWO launches (fp8q_multi_minmax_encode_u8):
    items (x, maxval_out [C] (receives the ranges), mbits[, n_bits[, sign_bits[, out uint8]]]).  Returns the code tensors
    (bit-identical to encode(x, minmax(x, True).maxval) per item)."""
    import ctypes
    items = [tuple(it) for it in items]
    if not items:
        return []
    descs, outs, keep = _pack_codec_descs(items, True)
    for (x, mv), it in zip(keep, items):
        if mv.data_ptr() != it[1].data_ptr() or mv.numel() != (x.shape[0] if x.dim() > 0 else 1):
            raise Fp8qError("multi_minmax_encode: maxval_out must be a contiguous [C] tensor (it receives the ranges)")
    mv_out = (ctypes.c_void_p * len(items))(*[mv.data_ptr() for _, mv in keep])
    with _on_device(keep[0][0]):
        rc = lib().fp8q_multi_minmax_encode_u8(descs, mv_out, len(items), _stream(keep[0][0]))
    check(rc, "fp8q_multi_minmax_encode_u8")
    return outs


def multi_decode(items):
    """Storage codes of many tensors back to float32 in one launch per 32 tensors (fp8q_multi_decode_u8):
    items (codes uint8, maxval, mbits[, n_bits[, sign_bits[, out float32]]])."""
    items = [tuple(it) for it in items]
    if not items:
        return []
    descs, outs, keep = _pack_codec_descs(items, False)
    with _on_device(keep[0][0]):
        rc = lib().fp8q_multi_decode_u8(descs, len(items), _stream(keep[0][0]))
    check(rc, "fp8q_multi_decode_u8")
    return outs


class MultiPlan:
    """Prepared multi-tensor K1 (fp8q_multi_plan_*): the descriptors of `items` are validated and packed once;
    launch() re-quantizes every tensor into its output with one ctypes call and one kernel launch per 32 tensors.
    The plan records addresses: it keeps the tensors alive, their contents may change between launches (in-place
    weight / range updates), their storage and shapes may not.  `outs` are the output tensors, in item order."""

    def __init__(self, items):
        import ctypes
        items = [tuple(it) for it in items]
        if not items:
            raise Fp8qError("MultiPlan needs at least one tensor")
        descs, self.outs, self._keep = _pack_descs(items)
        for (x, mv), it in zip(self._keep, items):
            if x.data_ptr() != it[0].data_ptr() or mv.data_ptr() != it[1].data_ptr():
                raise Fp8qError("MultiPlan needs contiguous tensors (a copy would be quantized instead of the tensor)")
        self._dev = self._keep[0][0]
        self._h = ctypes.c_void_p()
        check(lib().fp8q_multi_plan_create(descs, len(items), ctypes.byref(self._h)), "fp8q_multi_plan_create")
        self._launch = lib().fp8q_multi_plan_launch
        self._idx = self._dev.device.index

    @property
    def launches(self):
        return int(lib().fp8q_multi_plan_launches(self._h))

    def launch(self):
        if torch.cuda.current_device() != self._idx:
            with _on_device(self._dev):
                rc = self._launch(self._h, _stream(self._dev))
        else:
            rc = self._launch(self._h, _raw_stream(self._idx) if _raw_stream is not None else _stream(self._dev))
        if rc:
            check(rc, "fp8q_multi_plan_launch")
        return self.outs

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                lib().fp8q_multi_plan_destroy(h)
            except Exception:
                pass


def new_packed(C, device):
    """[C, 4] fp32 buffer for the packed ranges of batch-sharded calibration (16-byte records)."""
    return torch.empty((C, 4), dtype=torch.float32, device=device)


def _check_packed(packed, C):
    _require(packed, "packed")
    if packed.numel() != 4 * C or not packed.is_contiguous() or packed.data_ptr() % 16:
        raise Fp8qError(f"packed must be a contiguous, 16-byte aligned [{C}, 4] tensor")


def ranges_unpack(packed, cur_min=None, cur_max=None, maxval=None):
    """After all-reduce(MAX) of a packed-ranges buffer (minmax(..., packed=) / affine_act_minmax(..., packed=)):
    cur_min / cur_max / maxval ([C] tensors, written in place; None = allocate).  Returns (cur_min, cur_max, maxval)."""
    _require(packed, "packed")
    n = packed.numel() // 4
    _check_packed(packed, n)
    if cur_min is None or cur_max is None or maxval is None:
        st = torch.empty((3, n), dtype=torch.float32, device=packed.device)
        cur_min = st[0] if cur_min is None else cur_min
        cur_max = st[1] if cur_max is None else cur_max
        maxval = st[2] if maxval is None else maxval
    for t in (cur_min, cur_max, maxval):
        _require(t, "range vector", like=packed)
        if t.numel() != n or not t.is_contiguous():
            raise Fp8qError("range vectors must be contiguous [C] tensors")
    with _on_device(packed):
        rc = lib().fp8q_ranges_unpack_f32(packed.data_ptr(), n, cur_min.data_ptr(), cur_max.data_ptr(),
                                          maxval.data_ptr(), _stream(packed))
    check(rc, "fp8q_ranges_unpack_f32")
    return cur_min, cur_max, maxval


def check_workspaces(clear=True):
    """SYNCHRONISES.  Inspect every min/max workspace this process has used since the last check -- the live ones and
    those a larger request has replaced in the meantime (fp8q_minmax_workspace_check): raises Fp8qError if a reducer
    block timed out (that call's range is NaN) or a workspace is dirty between calls.  Every workspace is inspected
    (and cleared) before the first failure is raised.  Called by QuantizedModel.fix_ranges(), i.e. once after
    calibration."""
    todo = [(k, w) for k, w in list(_ws_cache.items()) if k[2]] + list(_ws_retired)
    del _ws_retired[:]
    failures = []
    for (dev_index, stream, _zeroed, _kind), ws in todo:
        with torch.cuda.device(dev_index):
            rc = lib().fp8q_minmax_workspace_check(ws.data_ptr(), ws.numel(), int(bool(clear)), stream)
        if rc:
            failures.append(rc)
    if failures:
        check(failures[0], f"fp8q_minmax_workspace_check ({len(failures)} of {len(todo)} workspaces failed)")


def release_workspaces(keep_bytes=1 << 24, device=None):
    """Drop the scratch buffers larger than `keep_bytes` (the MSE search's partitioned keys: 4 B per element of the largest
    activation it has seen, ~300 MB for MobileNetV2 at batch 64) -- they are re-allocated on demand.  The zeroed min/max
    workspaces stay (small, and their headers carry state).  device: only that device's buffers (QuantizedModel.fix_ranges()
    passes its own device once calibration is over, so that another model still calibrating on another GPU of the process keeps
    its scratch); None: every device.  Streams are not told apart: a model calibrating on another stream of the SAME device
    re-allocates on its next batch.  Returns the bytes released."""
    idx = None if device is None else torch.device(device).index
    freed = 0
    for key, ws in list(_ws_cache.items()):
        if idx is not None and key[0] != idx:
            continue
        if not key[2] and key[3] in (None, "pre") and ws.numel() > keep_bytes:
            freed += ws.numel()
            del _ws_cache[key]
    return freed


def minmax(x, per_channel, cur_min=None, cur_max=None, mode=FOLD_CURRENT, momentum=0.9,
           want_maxval=False, packed=None):
    """K2/K3(/K5): batch min/max folded into the running estimate (range_estimators.py:61-125).

    cur_min/cur_max: running estimate tensors [C] (updated in place) or None on the first call.
    packed: optional [C, 4] buffer (new_packed) that receives {-min, max, nan flags} of the folded estimate for the
    range all-reduce of batch-sharded calibration (see ranges_unpack).
    Returns (cur_min, cur_max[, maxval]) as [C] tensors.
    """
    _require(x, "x")
    flat = None if per_channel else _dense_flat(x)      # per tensor: any dense layout, no copy
    x = flat if flat is not None else x.contiguous()
    C, inner = _rows(x, per_channel)
    if C == 0 or inner == 0:
        raise Fp8qError("min/max of an empty tensor")
    first = cur_min is None or cur_max is None
    mv = None
    if first:
        # one allocation for the two (three) result vectors: tiny tensors are host-bound, every torch.empty is ~2 us
        stats = torch.empty((3 if want_maxval else 2, C), dtype=torch.float32, device=x.device)
        if want_maxval:
            cur_min, cur_max, mv = stats.unbind(0)
        else:
            cur_min, cur_max = stats.unbind(0)
    else:
        _require(cur_min, "cur_min", like=x)
        _require(cur_max, "cur_max", like=x)
        if cur_min.numel() != C or cur_max.numel() != C or not cur_min.is_contiguous() \
                or not cur_max.is_contiguous():
            raise Fp8qError("running estimate has the wrong shape")
    if want_maxval and mv is None:
        mv = torch.empty(C, dtype=torch.float32, device=x.device)
    L = lib()
    nbytes = _mm_ws_bytes.get((C, inner))
    if nbytes is None:     # a pure function of the shape: one ctypes call per shape, not per launch
        nbytes = _mm_ws_bytes[(C, inner)] = L.fp8q_minmax_workspace_bytes(C, inner)
    ws = _workspace(x.device, nbytes, zeroed=True)
    with _on_device(x):
        if packed is None:
            rc = L.fp8q_minmax_f32(x.data_ptr(), C, inner, cur_min.data_ptr(), cur_max.data_ptr(),
                                   mv.data_ptr() if mv is not None else None, int(mode), float(momentum),
                                   int(first), ws.data_ptr(), ws.numel(), _stream(x))
        else:
            _check_packed(packed, C)
            _require(packed, "packed", like=x)
            rc = L.fp8q_minmax_packed_f32(x.data_ptr(), C, inner, cur_min.data_ptr(), cur_max.data_ptr(),
                                          mv.data_ptr() if mv is not None else None, packed.data_ptr(), int(mode),
                                          float(momentum), int(first), ws.data_ptr(), ws.numel(), _stream(x))
    check(rc, "fp8q_minmax_f32")
    return (cur_min, cur_max, mv) if want_maxval else (cur_min, cur_max)


def fused_max_inner():
    return int(lib().fp8q_fused_max_inner())


def minmax_quantize(x, mbits, n_bits=8, sign_bits=1, out=None):
    """K2+K5+K1 fused per-channel weight quantization (current_minmax, set_maxval=True).

    Returns (y, row_min, row_max, maxval)."""
    _require(x, "x")
    x = x.contiguous()
    C, inner = _rows(x, True)
    y = _out(out, x)
    mn, mx, mv = torch.empty((3, C), dtype=torch.float32, device=x.device).unbind(0)   # one allocation (host-bound sizes)
    with _on_device(x):
        rc = lib().fp8q_minmax_quantize_f32(x.data_ptr(), y.data_ptr(), C, inner, mn.data_ptr(),
                                            mx.data_ptr(), mv.data_ptr(), float(mbits), int(n_bits),
                                            int(sign_bits), _stream(x))
    check(rc, "fp8q_minmax_quantize_f32")
    return y, mn, mx, mv


def copy(x, out=None):
    """float4 copy kernel with K1's launch shape (HBM ceiling yardstick)."""
    _require(x, "x")
    x = x.contiguous()
    y = _out(out, x)
    with _on_device(x):
        rc = lib().fp8q_copy_f32(x.data_ptr(), y.data_ptr(), x.numel(), _stream(x))
    check(rc, "fp8q_copy_f32")
    return y


_mse_ws_bytes = {}


def mse_grid(x, per_channel, grid, mbits_list, n_bits, sign_bits, mses):
    """K4: mses[n_m, n_cand, C] += row-mean((x - q(x; m, grid[i, c]))^2)  (range_estimators.py:337-347).

    grid: CUDA fp32 [n_cand, C]; mbits_list: python floats; mses: CUDA fp32, updated in place.
    A per-tensor call takes any dense layout as it lies (no NCHW copy).  The table then depends on the memory format in its
    last bits (~1e-7 relative): the lane-per-element routes add fp32 squares in storage order (the interval-histogram route's
    integer moments are order-free), so a near-tie between two candidates can fall differently for the same values in NCHW
    and in channels-last -- inside K4's stated contract (include/fp8q.h: 1e-5 per entry), as any other summation order is.
    """
    import ctypes
    _require(x, "x")
    _require(grid, "grid", like=x)
    _require(mses, "mses", like=x)
    flat = None if per_channel else _dense_flat(x)      # a per-tensor mean does not depend on the layout
    x = flat if flat is not None else x.contiguous()
    C, inner = _rows(x, per_channel)
    n_m = len(mbits_list)
    n_cand = grid.shape[0]
    if grid.dim() != 2 or grid.shape[1] != C or not grid.is_contiguous():
        raise Fp8qError(f"grid must be contiguous [n_cand, {C}]")
    if tuple(mses.shape) != (n_m, n_cand, C) or not mses.is_contiguous():
        raise Fp8qError(f"mses must be contiguous [{n_m}, {n_cand}, {C}]")
    L = lib()
    key = (C, inner, n_cand, n_m)
    nbytes = _mse_ws_bytes.get(key)
    if nbytes is None:     # a pure function of the shape: one ctypes call per shape, not per launch
        nbytes = _mse_ws_bytes[key] = L.fp8q_mse_workspace_bytes(C, inner, n_cand, n_m)
    ws = _workspace(x.device, nbytes)
    mb = (ctypes.c_float * n_m)(*[float(v) for v in mbits_list])
    with _on_device(x):
        rc = L.fp8q_mse_grid_f32(x.data_ptr(), C, inner, grid.data_ptr(), n_cand, mb, n_m, int(n_bits),
                                 int(sign_bits), mses.data_ptr(), ws.data_ptr(), ws.numel(), _stream(x))
    check(rc, "fp8q_mse_grid_f32")
    return mses


_linspace_checked = {}     # (steps, fractions) -> True: the device kernel reproduces torch.linspace; False: host fall-back


def _linspace_self_check(steps, lo_frac, hi_frac, device):
    """Once per process and (steps, fractions): the device search-grid formula against torch.linspace itself on a few
    ranges -- the kernel hard-codes ATen's CPU evaluation (the steps / 2 split, one fused multiply-add per element),
    which another ATen build could do differently.  Synchronises (once).  Returns True when the device path may be used;
    on a mismatch the decision is cached and the search grids are made by torch.linspace on the host from then on
    (a host round trip per FIRST calibration batch of a quantizer -- slower, never wrong)."""
    key = (steps, lo_frac, hi_frac)
    ok = _linspace_checked.get(key)
    if ok is not None:
        return ok
    probe = torch.tensor([1.0, 0.7361, 3.3e-5, 123.456, 6.0e4, 0.0131, 2.5], dtype=torch.float32)
    got = torch.empty((steps, probe.numel()), dtype=torch.float32, device=device)
    dev_probe = probe.to(device)
    check(lib().fp8q_mse_linspace_f32(dev_probe.data_ptr(), probe.numel(), steps, lo_frac, hi_frac, got.data_ptr(),
                                      _stream(got)), "fp8q_mse_linspace_f32")
    want = torch.stack([torch.linspace(lo_frac * float(v), hi_frac * float(v), steps) for v in probe.tolist()], 1)
    ok = bool(torch.equal(got.cpu().view(torch.int32), want.view(torch.int32)))
    if not ok:
        import warnings
        warnings.warn("fp8q_mse_linspace_f32 does not reproduce torch.linspace on this PyTorch build: the MSE search "
                      "grids are computed by torch.linspace on the host (one synchronisation per quantizer)")
    _linspace_checked[key] = ok
    return ok


def _linspace_host(mx, steps, lo_frac, hi_frac):
    """the reference's own construction (range_estimators.py:296-305), per channel, on the host"""
    vals = mx.detach().cpu().tolist()
    cols = [torch.linspace(lo_frac * float(v), hi_frac * float(v), steps) for v in vals]
    return torch.stack(cols, 1).contiguous().to(mx.device)


def mse_linspace(mx, steps=111, lo_frac=0.1, hi_frac=1.2):
    """[steps, C] float32 search grid of FP_MSE_Estimator on the device: column c == torch.linspace(lo_frac * mx[c],
    hi_frac * mx[c], steps), bit for bit (range_estimators.py:296-305) -- no host round trip."""
    _require(mx, "mx")
    mx = mx.contiguous().view(-1)
    C = mx.numel()
    if C == 0:
        return torch.empty((steps, 0), dtype=torch.float32, device=mx.device)
    with _on_device(mx):
        if not _linspace_self_check(int(steps), float(lo_frac), float(hi_frac), mx.device):
            return _linspace_host(mx, int(steps), float(lo_frac), float(hi_frac))
        grid = torch.empty((steps, C), dtype=torch.float32, device=mx.device)
        rc = lib().fp8q_mse_linspace_f32(mx.data_ptr(), C, int(steps), float(lo_frac), float(hi_frac), grid.data_ptr(),
                                         _stream(mx))
    check(rc, "fp8q_mse_linspace_f32")
    return grid


def minmax_linspace(x, per_channel, steps=111, lo_frac=0.1, hi_frac=1.2):
    """First calibration batch of FP_MSE_Estimator in ONE launch (fp8q_minmax_linspace_f32): row min / max, max|x| and
    the search grid of that maximum.  Returns (min [C], max [C], absmax [C], grid [steps, C])."""
    _require(x, "x")
    flat = None if per_channel else _dense_flat(x)
    x = flat if flat is not None else x.contiguous()
    C, inner = _rows(x, per_channel)
    if C == 0 or inner == 0:
        raise Fp8qError("min/max of an empty tensor")
    with _on_device(x):
        if not _linspace_self_check(int(steps), float(lo_frac), float(hi_frac), x.device):
            mn, mx, mv = minmax(x, per_channel, want_maxval=True)
            return mn, mx, mv, _linspace_host(mv, int(steps), float(lo_frac), float(hi_frac))
        stats = torch.empty((3 + int(steps), C), dtype=torch.float32, device=x.device)    # one allocation
        L = lib()
        nbytes = _mm_ws_bytes.get((C, inner))
        if nbytes is None:
            nbytes = _mm_ws_bytes[(C, inner)] = L.fp8q_minmax_workspace_bytes(C, inner)
        ws = _workspace(x.device, nbytes, zeroed=True)
        rc = L.fp8q_minmax_linspace_f32(x.data_ptr(), C, inner, stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(),
                                        stats[3].data_ptr(), int(steps), float(lo_frac), float(hi_frac), ws.data_ptr(),
                                        ws.numel(), _stream(x))
    check(rc, "fp8q_minmax_linspace_f32")
    return stats[0], stats[1], stats[2], stats[3:]


def mse_select(mses, grid, mbits_list, sign_bits=1):
    """Winner of the MSE grid search on the device (range_estimators.py:350-369): per channel the mantissa width with the
    smallest minimum, the plurality vote over the channels (torch.mode: smallest value on a tie), per channel the
    winning width's argmin candidate.  Returns (mbits [1] CUDA float32, vote index [1] CUDA int32, maxval [C],
    xmin [C] = -sign_bits * maxval) -- nothing comes back to the host."""
    import ctypes
    _require(mses, "mses")
    _require(grid, "grid", like=mses)
    n_m = len(mbits_list)
    if mses.dim() != 3 or mses.shape[0] != n_m or tuple(mses.shape[1:]) != tuple(grid.shape) or not mses.is_contiguous() \
            or not grid.is_contiguous():
        raise Fp8qError(f"mses must be contiguous [{n_m}, n_cand, C] and grid contiguous [n_cand, C]")
    n_cand, C = grid.shape
    dev = mses.device
    out = torch.empty((2, C), dtype=torch.float32, device=dev)
    mb = vote_slot(dev)                  # the device's vote arena: fix_ranges() brings every pending width over in one copy
    vote = torch.empty(1, dtype=torch.int32, device=dev)
    L = lib()
    ws = _workspace(dev, L.fp8q_mse_select_workspace_bytes(C, n_m), kind="select")     # (zero header: the last-workgroup ticket)
    mbh = (ctypes.c_float * n_m)(*[float(v) for v in mbits_list])
    with _on_device(mses):
        rc = L.fp8q_mse_select_f32(mses.data_ptr(), grid.data_ptr(), C, n_cand, mbh, n_m, int(sign_bits), mb.data_ptr(),
                                   vote.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), ws.data_ptr(), ws.numel(),
                                   _stream(mses))
    check(rc, "fp8q_mse_select_f32")
    return mb, vote, out[0], out[1]


# ---- mantissa-width votes: one arena per device --------------------------------------------------------------------------
# Every MSE estimator's voted width is a device scalar that the host wants exactly once, at fix_ranges().  Scalars that live
# in ONE buffer come over with one device-to-host copy (QuantizedModel.fix_ranges: materialize_mantissa_bits); gathering 116
# separately allocated scalars needed a torch.cat, whose kernel alone took 16 ms to load the first time a process used it.
_VOTE_SLOTS = 4096
_vote_arenas = {}      # device index -> [buffer, next free slot]
_vote_buffers = []     # every arena buffer still referenced by a view (weak): (weakref, base address)


def vote_slot(device):
    """A [1] float32 view into the device's vote arena (a fresh arena when the current one is full)."""
    import weakref
    ent = _vote_arenas.get(device.index)
    if ent is None or ent[1] >= _VOTE_SLOTS:
        buf = torch.zeros(_VOTE_SLOTS, dtype=torch.float32, device=device)
        ent = _vote_arenas[device.index] = [buf, 0]
        _vote_buffers[:] = [(r, a) for r, a in _vote_buffers if r() is not None]
        _vote_buffers.append((weakref.ref(buf), buf.data_ptr()))
    i = ent[1]
    ent[1] = i + 1
    return ent[0][i:i + 1]


def vote_arena_of(t):
    """(arena buffer, slot) if the 1-element CUDA float32 tensor t lies in a vote arena (vote_slot), else None.  By address:
    an arena's range belongs to it for as long as the buffer is alive, whatever chain of views / detach() led to t."""
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.numel() == 1):
        return None
    p = t.data_ptr()
    for r, addr in _vote_buffers:
        if addr <= p < addr + 4 * _VOTE_SLOTS:
            buf = r()
            if buf is not None and buf.device == t.device:
                return buf, (p - addr) // 4
    return None


_cal_ws_bytes = {}


class MseCalibration:
    """The persistent device state of one FP_MSE_Estimator and its quantizer, bound once, stepped with ONE library call per
    batch (fp8q_mse_calibrate_f32): abs-max + search grid (first batch), the MSE table of every (width, candidate), the
    vote / argmin, and the quantization of the batch with the winner -- QuantizationManager.forward in estimate state
    (quantization_manager.py:114-122 around range_estimators.py:318-369).  One allocation holds every result vector:
      block = [cur_min | cur_max | absmax | maxval | xmin] (5 C) | grid (n_cand C) | mses (n_m n_cand C) | vote (1, int32)
    The voted width lives in the device's vote arena (vote_slot)."""

    def __init__(self, C, device, mbits_list, n_bits, sign_bits, n_cand=111, grid=None, mses=None):
        import ctypes
        from ._lib import MseState
        self.C, self.n_cand, self.n_m = int(C), int(n_cand), len(mbits_list)
        self.n_bits, self.sign_bits = int(n_bits), int(sign_bits)
        self.mbits_list = [float(v) for v in mbits_list]
        C, n_m = self.C, self.n_m
        adopt = grid is not None and mses is not None
        n_tab = 0 if adopt else (n_cand * C + n_m * n_cand * C)
        blk = torch.empty(5 * C + n_tab + 1, dtype=torch.float32, device=device)
        self._blk = blk
        if adopt:           # tables made by the generic path (an earlier batch in another layout): keep accumulating into them
            _require(grid, "grid")
            _require(mses, "mses", like=grid)
            if tuple(grid.shape) != (n_cand, C) or tuple(mses.shape) != (n_m, n_cand, C) or not grid.is_contiguous() \
                    or not mses.is_contiguous() or grid.device != blk.device:
                raise Fp8qError("MseCalibration: grid / mses of the wrong shape")
            self.grid, self.mses = grid, mses
        else:
            self.grid = blk[5 * C:5 * C + n_cand * C].view(n_cand, C)
            self.mses = blk[5 * C + n_cand * C:5 * C + n_tab].view(n_m, n_cand, C)
        self.maxval = blk[3 * C:4 * C]
        self.mbits = vote_slot(blk.device)
        self.first = not adopt
        # (the other vectors of the block are views made on demand -- cur_min, cur_max, absmax, xmin, vote: a calibration pass
        # constructs 116 of these objects, every tensor view costs the host a microsecond or two)
        p0 = blk.data_ptr()
        self._state = MseState(p0, p0 + 4 * C, p0 + 8 * C, self.grid.data_ptr(), self.mses.data_ptr(), p0 + 12 * C, p0 + 16 * C,
                               self.mbits.data_ptr(), p0 + 4 * (5 * C + n_tab))
        self._sref = ctypes.byref(self._state)
        self._mb = (ctypes.c_float * n_m)(*self.mbits_list)
        self._fn = lib().fp8q_mse_calibrate_f32
        self._dev = blk.device
        self._idx = blk.device.index
        self._n_tab = n_tab

    cur_min = property(lambda self: self._blk[0:self.C])
    cur_max = property(lambda self: self._blk[self.C:2 * self.C])
    absmax = property(lambda self: self._blk[2 * self.C:3 * self.C])
    xmin = property(lambda self: self._blk[4 * self.C:5 * self.C])
    vote = property(lambda self: self._blk[5 * self.C + self._n_tab:].view(torch.int32))

    def _sizes(self, inner, pre_geo=None):
        import ctypes
        key = (self.C, inner, self.n_cand, self.n_m, pre_geo)
        sz = _cal_ws_bytes.get(key)
        if sz is None:
            a, b = ctypes.c_size_t(), ctypes.c_size_t()
            c = lib().fp8q_mse_calibrate_workspace_bytes(self.C, inner, self.n_cand, self.n_m, ctypes.byref(a), ctypes.byref(b))
            mm = int(a.value)
            if pre_geo is not None:
                mm = max(mm, int(lib().fp8q_affine_act_minmax_workspace_bytes(*pre_geo)))
            sz = _cal_ws_bytes[key] = (mm, int(b.value), int(c))
        return sz

    def step(self, x, quantize=True, pre=None):
        """x: contiguous CUDA float32 with C rows (per tensor: C == 1).  Returns the quantized batch (or None).
        pre = (bn_ab or None, residual or None, act): x is the PRODUCER's output [N, C, ...]; the search and the
        quantization run on act(bn(x) + residual), formed inside the same call (per-tensor quantizers only)."""
        inner = x.numel() // self.C
        dev = self._dev
        st = _raw_stream(self._idx) if _raw_stream is not None else torch.cuda.current_stream(dev).cuda_stream
        pre_ref, keep = None, None
        xin = x
        if pre is not None:
            import ctypes
            from ._lib import AffinePre
            ab, res, act = pre
            N, C, HW = _nchw(x)
            if self.C != 1 or (ab is not None and (ab.numel() != 2 * C or not ab.is_contiguous() or ab.device != x.device)) \
                    or (res is not None and (res.shape != x.shape or not res.is_contiguous() or res.dtype != x.dtype
                                             or res.device != x.device)):
                raise Fp8qError("MseCalibration.step: bad `pre` (per-tensor quantizer, folded [C, 2] BN vector, residual like x)")
            mm, sel, mse = self._sizes(inner, (N, C, HW))
            t = _workspace(dev, 4 * inner, kind="pre", stream=st)   # the tensor the quantizer sees: scratch, consumed by this call
            keep = AffinePre(x.data_ptr(), res.data_ptr() if res is not None else None, ab.data_ptr() if ab is not None else None,
                             N, C, HW, int(act))
            pre_ref = ctypes.byref(keep)
            xin = t
        else:
            mm, sel, mse = self._sizes(inner)
        # (short per-channel rows need no min/max workspace: fp8q_minmax_workspace_bytes says 16 -- none is created then, so a
        # stream that only ever calibrates weights (QuantizedModel: calibrate_weights_ahead) owns no buffer that
        # check_workspaces() would have to synchronise for)
        ws_mm = _workspace(dev, mm, zeroed=True, stream=st) if (mm > 16 or self.C == 1) else None
        ws_sel = _workspace(dev, sel, kind="select", stream=st)
        ws_mse = _workspace(dev, mse, stream=st)
        y = torch.empty_like(x) if quantize else None
        first, self.first = self.first, False
        other = torch.cuda.current_device() != self._idx
        if other:
            prev = torch.cuda.current_device()
            torch.cuda.set_device(self._idx)
        try:
            rc = self._fn(xin.data_ptr(), y.data_ptr() if quantize else None, self.C, inner, self._sref, int(first), self.n_cand,
                          self._mb, self.n_m, self.n_bits, self.sign_bits, pre_ref, ws_mm.data_ptr() if ws_mm is not None else None,
                          ws_mm.numel() if ws_mm is not None else 0, ws_sel.data_ptr(),
                          ws_sel.numel(), ws_mse.data_ptr(), ws_mse.numel(), st)
        finally:
            if other:
                torch.cuda.set_device(prev)
        if rc:
            self.first = first
            check(rc, "fp8q_mse_calibrate_f32")
        return y


def minmax_f64(x, per_channel):
    """Row min / max of a float64 tensor as [C] float64 tensors (fp8q_minmax_f64): what LineSearchEstimator needs of
    its float64 sample (range_estimators.py:205-222: data.min(), data.max()).  NaN anywhere in a row -> NaN."""
    _require(x, "x", torch.float64)
    x = x.contiguous()
    C, inner = _rows(x, per_channel)
    if C == 0 or inner == 0:
        raise Fp8qError("min/max of an empty tensor")
    mn, mx = torch.empty((2, C), dtype=torch.float64, device=x.device).unbind(0)
    L = lib()
    ws = _workspace(x.device, L.fp8q_minmax_f64_workspace_bytes(C, inner))
    with _on_device(x):
        rc = L.fp8q_minmax_f64(x.data_ptr(), C, inner, mn.data_ptr(), mx.data_ptr(), ws.data_ptr(), ws.numel(), _stream(x))
    check(rc, "fp8q_minmax_f64")
    return mn, mx


def mse_grid_f64(x, per_channel, grid, mbits_list, n_bits, sign_bits, out, reduce="sum"):
    """K4 on float64: out[n_m, n_cand, C] (float64, updated in place) += row-sum (reduce="sum": LineSearchEstimator.loss_fx,
    range_estimators.py:161-169) or row-mean (reduce="mean": FP_MSE_Estimator, :337-347) of (x - q(x; m, grid[i, c]))^2.
    grid: CUDA fp32 [n_cand, C] candidate maxvals (what torch.Tensor([x_max]) holds in the reference)."""
    import ctypes
    if reduce not in ("sum", "mean"):
        raise Fp8qError("reduce must be 'sum' or 'mean'")
    _require(x, "x", torch.float64)
    _require(grid, "grid", like=x)
    _require(out, "out", torch.float64, like=x)
    x = x.contiguous()
    C, inner = _rows(x, per_channel)
    n_m = len(mbits_list)
    n_cand = grid.shape[0]
    if grid.dim() != 2 or grid.shape[1] != C or not grid.is_contiguous():
        raise Fp8qError(f"grid must be contiguous [n_cand, {C}]")
    if tuple(out.shape) != (n_m, n_cand, C) or not out.is_contiguous():
        raise Fp8qError(f"out must be contiguous [{n_m}, {n_cand}, {C}]")
    L = lib()
    ws = _workspace(x.device, L.fp8q_mse_f64_workspace_bytes(C, inner, n_cand, n_m))
    mb = (ctypes.c_float * n_m)(*[float(v) for v in mbits_list])
    with _on_device(x):
        rc = L.fp8q_mse_grid_f64(x.data_ptr(), C, inner, grid.data_ptr(), n_cand, mb, n_m, int(n_bits), int(sign_bits),
                                 out.data_ptr(), int(reduce == "sum"), ws.data_ptr(), ws.numel(), _stream(x))
    check(rc, "fp8q_mse_grid_f64")
    return out


def encode(x, maxval, mbits, n_bits=8, sign_bits=1, out=None):
    """N3: uint8 storage codes of quantize(x) ([sign | exponent | fraction], fp8_quantizer.py:13-41)."""
    _require(x, "x")
    _require(maxval, "maxval", like=x)
    x = x.contiguous()
    maxval = maxval.contiguous().view(-1)
    n_mv = maxval.numel()
    C, inner = _rows(x, n_mv != 1)
    if n_mv != 1 and n_mv != C:
        raise Fp8qError(f"maxval has {n_mv} elements, expected 1 or {C}")
    codes = _out(out, x, torch.uint8)
    with _on_device(x):
        rc = lib().fp8q_encode_u8(x.data_ptr(), codes.data_ptr(), C, inner, maxval.data_ptr(), n_mv, float(mbits),
                                  int(n_bits), int(sign_bits), _stream(x))
    check(rc, "fp8q_encode_u8")
    return codes


def decode(codes, maxval, mbits, n_bits=8, sign_bits=1, out=None):
    """N3: fp32 values of uint8 storage codes; decode(encode(x)) == quantize(x) bit for bit when the channel's
    scale table is exactly geometric in fp32 (all weight-sized ranges), else within a few ULP (<= 5e-6 relative) on elements that round up
    into the next binade (see include/fp8q.h)."""
    if not isinstance(codes, torch.Tensor) or not codes.is_cuda or codes.dtype != torch.uint8:
        raise Fp8qError("codes must be a CUDA(HIP) uint8 tensor")
    _require(maxval, "maxval", like=codes)
    codes = codes.contiguous()
    maxval = maxval.contiguous().view(-1)
    n_mv = maxval.numel()
    C, inner = _rows(codes, n_mv != 1)
    if n_mv != 1 and n_mv != C:
        raise Fp8qError(f"maxval has {n_mv} elements, expected 1 or {C}")
    y = _out(out, codes, torch.float32)
    with _on_device(codes):
        rc = lib().fp8q_decode_u8(codes.data_ptr(), y.data_ptr(), C, inner, maxval.data_ptr(), n_mv, float(mbits),
                                  int(n_bits), int(sign_bits), _stream(codes))
    check(rc, "fp8q_decode_u8")
    return y


def _nchw(x):
    """[N, C, *spatial] -> (N, C, HW)."""
    if x.dim() < 2:
        raise Fp8qError("fused epilogue expects [N, C, ...] tensors")
    N, C = x.shape[0], x.shape[1]
    return N, C, (x.numel() // (N * C) if N * C else 0)


def _bn_ptrs(bn, C, dev):
    if bn is None:
        return (None, None, None, None), ()
    keep = []
    for t in bn:
        _require(t, "bn parameter")
        t = t.contiguous()
        if t.numel() != C or t.device != dev:
            raise Fp8qError("batch-norm vectors must be [C] tensors on x's device")
        keep.append(t)
    return tuple(t.data_ptr() for t in keep), tuple(keep)


def affine_act_supported(x):
    """fp8q_affine_act_* needs C*HW % 4 == 0 (16-byte groups never straddle images)."""
    N, C, HW = _nchw(x)
    return N > 0 and (C * HW) % 4 == 0 and x.data_ptr() % 16 == 0


def bn_fold(bn):
    """[C, 2] fp32 {alpha, beta'} = {invstd * gamma, fma(-mean, alpha, beta)} of an eval-mode batch norm
    (fp8q_bn_fold_f32): the per-channel constants of the fused epilogue, folded once per parameter set."""
    mean = bn[0]
    _require(mean, "bn parameter")
    C = mean.numel()
    ptrs, keep = _bn_ptrs(bn, C, mean.device)
    ab = torch.empty((C, 2), dtype=torch.float32, device=mean.device)
    with _on_device(mean):
        rc = lib().fp8q_bn_fold_f32(ptrs[0], ptrs[1], ptrs[2], ptrs[3], C, ab.data_ptr(), _stream(mean))
    check(rc, "fp8q_bn_fold_f32")
    return ab


PREP_FLOATS = 264      # FP8Q_PREP_BYTES / 4


def quantizer_prepare(maxval, mbits, n_bits=8, sign_bits=1):
    """[264] fp32 block of a FIXED-range per-tensor quantizer: its channel constants and {s, 1/s} table, as the fused
    epilogue rebuilds them from maxval on every launch (fp8q_quantizer_prepare_f32).  Pass as affine_act_quantize(prep=)."""
    _require(maxval, "maxval")
    if maxval.numel() != 1:
        raise Fp8qError("quantizer_prepare: per-tensor quantizers only (one maxval)")
    prep = torch.empty(PREP_FLOATS, dtype=torch.float32, device=maxval.device)
    with _on_device(maxval):
        rc = lib().fp8q_quantizer_prepare_f32(maxval.contiguous().data_ptr(), float(mbits), int(n_bits), int(sign_bits),
                                              prep.data_ptr(), _stream(maxval))
    check(rc, "fp8q_quantizer_prepare_f32")
    return prep


def affine_act_quantize(x, maxval, mbits, n_bits=8, sign_bits=1, bn=None, residual=None, act=0, out=None, bn_ab=None,
                        prep=None):
    """N2: quantize(act(bn(x) + residual)) in one pass.  bn = (mean, invstd, gamma, beta), each [C] -- or bn_ab = the
    folded [C, 2] vector of bn_fold(bn) (same result, fewer loads); act: 0 none, 1 ReLU, 2 ReLU6; per-tensor maxval [1];
    prep = quantizer_prepare(maxval, mbits, n_bits, sign_bits) of the SAME quantizer (fixed ranges: the table is not rebuilt)."""
    _require(x, "x")
    _require(maxval, "maxval", like=x)
    x = x.contiguous()
    N, C, HW = _nchw(x)
    if residual is not None:
        _require(residual, "residual", like=x)
        residual = residual.contiguous()
        if residual.shape != x.shape:
            raise Fp8qError("residual must have x's shape")
    if maxval.numel() != 1:
        raise Fp8qError("the fused epilogue quantizes per tensor: maxval must have one element")
    y = _out(out, x)
    if prep is not None:
        _require(prep, "prep", like=x)
        if prep.numel() != PREP_FLOATS or not prep.is_contiguous() or prep.data_ptr() % 16:
            raise Fp8qError("prep must be the [264] fp32 block of fp8q.ops.quantizer_prepare")
    if bn_ab is not None or (prep is not None and bn is None):
        if bn_ab is not None:
            _require(bn_ab, "bn_ab", like=x)
            if bn_ab.numel() != 2 * C or not bn_ab.is_contiguous():
                raise Fp8qError("bn_ab must be a contiguous [C, 2] tensor (fp8q.ops.bn_fold)")
        with _on_device(x):
            rc = lib().fp8q_affine_act_quantize_ab_f32(
                x.data_ptr(), residual.data_ptr() if residual is not None else None, y.data_ptr(), N, C, HW,
                bn_ab.data_ptr() if bn_ab is not None else None, int(act), maxval.contiguous().data_ptr(),
                prep.data_ptr() if prep is not None else None, float(mbits), int(n_bits), int(sign_bits), _stream(x))
        check(rc, "fp8q_affine_act_quantize_ab_f32")
        return y
    ptrs, keep = _bn_ptrs(bn, C, x.device)
    with _on_device(x):
        rc = lib().fp8q_affine_act_quantize_f32(
            x.data_ptr(), residual.data_ptr() if residual is not None else None, y.data_ptr(), N, C, HW,
            ptrs[0], ptrs[1], ptrs[2], ptrs[3], int(act), maxval.contiguous().data_ptr(), float(mbits),
            int(n_bits), int(sign_bits), _stream(x))
    check(rc, "fp8q_affine_act_quantize_f32")
    return y


def affine_act(x, bn_ab=None, residual=None, act=0, out=None):
    """act(bn(x) + residual) without a quantizer (fp8q_affine_act_f32): what an MSE estimator behind a BN + activation searches
    on.  bn_ab: the folded [C, 2] vector of bn_fold(), or None."""
    _require(x, "x")
    x = x.contiguous()
    N, C, HW = _nchw(x)
    if residual is not None:
        _require(residual, "residual", like=x)
        residual = residual.contiguous()
        if residual.shape != x.shape:
            raise Fp8qError("residual must have x's shape")
    if bn_ab is not None:
        _require(bn_ab, "bn_ab", like=x)
        if bn_ab.numel() != 2 * C or not bn_ab.is_contiguous():
            raise Fp8qError("bn_ab must be a contiguous [C, 2] tensor (fp8q.ops.bn_fold)")
    y = _out(out, x)
    with _on_device(x):
        rc = lib().fp8q_affine_act_f32(x.data_ptr(), residual.data_ptr() if residual is not None else None, y.data_ptr(), N, C, HW,
                                       bn_ab.data_ptr() if bn_ab is not None else None, int(act), _stream(x))
    check(rc, "fp8q_affine_act_f32")
    return y


def affine_act_minmax(x, cur_min=None, cur_max=None, mode=FOLD_CURRENT, momentum=0.9, bn=None, residual=None,
                      act=0, packed=None):
    """N2: per-tensor min/max of act(bn(x) + residual), folded into the running estimate.
    packed: optional [1, 4] buffer for the range all-reduce (see minmax).
    Returns (cur_min, cur_max, maxval) as [1] tensors."""
    _require(x, "x")
    x = x.contiguous()
    N, C, HW = _nchw(x)
    if residual is not None:
        _require(residual, "residual", like=x)
        residual = residual.contiguous()
        if residual.shape != x.shape:
            raise Fp8qError("residual must have x's shape")
    ptrs, keep = _bn_ptrs(bn, C, x.device)
    first = cur_min is None or cur_max is None
    if first:
        cur_min = torch.empty(1, dtype=torch.float32, device=x.device)
        cur_max = torch.empty(1, dtype=torch.float32, device=x.device)
    else:
        for t in (cur_min, cur_max):
            _require(t, "running estimate", like=x)
            if t.numel() != 1 or not t.is_contiguous():
                raise Fp8qError("running estimate has the wrong shape")
    mv = torch.empty(1, dtype=torch.float32, device=x.device)
    L = lib()
    ws = _workspace(x.device, L.fp8q_affine_act_minmax_workspace_bytes(N, C, HW), zeroed=True)
    with _on_device(x):
        if packed is None:
            rc = L.fp8q_affine_act_minmax_f32(
                x.data_ptr(), residual.data_ptr() if residual is not None else None, N, C, HW, ptrs[0], ptrs[1],
                ptrs[2], ptrs[3], int(act), cur_min.data_ptr(), cur_max.data_ptr(), mv.data_ptr(), int(mode),
                float(momentum), int(first), ws.data_ptr(), ws.numel(), _stream(x))
        else:
            _check_packed(packed, 1)
            _require(packed, "packed", like=x)
            rc = L.fp8q_affine_act_minmax_packed_f32(
                x.data_ptr(), residual.data_ptr() if residual is not None else None, N, C, HW, ptrs[0], ptrs[1],
                ptrs[2], ptrs[3], int(act), cur_min.data_ptr(), cur_max.data_ptr(), mv.data_ptr(), packed.data_ptr(),
                int(mode), float(momentum), int(first), ws.data_ptr(), ws.numel(), _stream(x))
    check(rc, "fp8q_affine_act_minmax_f32")
    return cur_min, cur_max, mv
