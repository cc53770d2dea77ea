"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on the GPU box; "gloo" in the CPU tests).

The path has exactly two exchange steps; everything else is embarrassingly parallel:

  * activation calibration is batch-sharded: every rank folds its own batches into a running
    min/max on the device, and the ranks exchange 2 floats per quantizer with ONE fused
    all-reduce(MAX) over [-min, max] for all quantizers of a model (min/max are associative and
    commutative, so the result equals the single-process estimate over the union of the batches);
  * weight quantization is sharded per output channel (channels are independent); the
    quantized shards and their per-channel ranges are re-assembled with one all-gather each.
    On an 8-GPU xGMI mesh every GPU pair has its own link, so the all-gather is a direct
    exchange of C/8-channel shards (ResNet-18: 5.8 MB per peer) rather than a ring.

`ops` is the local compute backend (default: the HIP engine fp8q.ops).  The CPU tests inject an
oracle-backed object with the same three functions so that the sharding / collective logic runs
under gloo without a GPU; the product default never touches the oracle.
"""
import os

import torch
import torch.distributed as dist


def _multi(group=None):
    """True when the collective code path has to run: more than one rank -- or FP8Q_DIST_FORCE=1, which runs it on a
    single rank too (hardware smoke test of the RCCL call sequence on a one-GPU box: every collective of this module
    executes on the "nccl" backend, with itself as the only peer)."""
    if not dist.is_initialized():
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("FP8Q_DIST_FORCE") == "1"


def _default_ops():
    from . import ops
    return ops


def _mark(timing):
    """Optional phase timing for bench.py: append a recorded event of the current stream to timing["events"]
    (collectives issued with async_op=False make the current stream wait for them, so the events bracket them)."""
    if timing is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        timing.setdefault("events", []).append(ev)


def _local_minmax_quantize(ops, shard, mbits, n_bits, sign_bits, out=None):
    """current_minmax range + quantize of this rank's channels: the fused launch when the rows fit it
    (fp8q_fused_max_inner), else min/max then quantize (e.g. Linear(25088, 4096): rows of 25088 elements).
    `out`: where the quantized shard should land (a view of a send buffer); a backend may ignore it (the caller
    checks and copies)."""
    inner = shard.numel() // max(shard.shape[0], 1)
    limit = getattr(ops, "fused_max_inner", None)
    if limit is None or inner <= limit():
        q, _, _, mv = ops.minmax_quantize(shard, mbits, n_bits, sign_bits, out=out)
        return q, mv
    mv = ops.minmax(shard, True, want_maxval=True)[2]
    return ops.quantize(shard, mv, mbits, n_bits, sign_bits, out=out), mv


def codes_wire_ok(mbits, n_bits, sign_bits):
    """True when the format has a 1-byte storage code (fp8q_encode_u8: n_bits <= 8 and at least one exponent bit).  K1 and the
    reference's clamp (fp8_quantizer.py:105-106: M = clamp(round(mbits), 1, n_bits - sign_bits)) also accept E = 0 -- a plain
    fixed-point grid --, which the codec refuses (FP8Q_EUNSUPPORTED): such formats travel as fp32 values."""
    if n_bits > 8:
        return False
    import numpy as np
    hi = n_bits - sign_bits
    m = int(min(max(float(np.round(np.float32(mbits))), 1.0), float(hi)))       # round half to even, as torch.round
    return hi - m >= 1


def channel_partition(n_channels, world_size):
    """[lo, hi) of every rank; the first (C mod W) ranks get one extra channel."""
    base, extra = divmod(n_channels, world_size)
    bounds, lo = [], 0
    for r in range(world_size):
        hi = lo + base + (1 if r < extra else 0)
        bounds.append((lo, hi))
        lo = hi
    return bounds


def allreduce_ranges(mins, maxs, group=None):
    """In-place global min of `mins` and max of `maxs` with a single all-reduce(MAX).

    Tensor-op version for ranges that already sit in separate tensors (sync_activation_ranges: once per model).  The
    per-batch exchange of calibration does not come through here: there the min/max kernel itself writes the packed
    operand and one kernel unpacks it (ops.minmax(packed=) -> all_reduce -> ops.ranges_unpack)."""
    if not _multi(group):
        return mins, maxs
    n = mins.numel()
    buf = torch.cat([-mins.reshape(-1), maxs.reshape(-1)])
    # NaN must win on every rank (torch.min/max semantics): MAX over an all-NaN-or-not flag
    nan = torch.isnan(buf).to(buf.dtype)
    buf = torch.nan_to_num(buf, nan=float("-inf"))
    packed = torch.cat([buf, nan])
    dist.all_reduce(packed, op=dist.ReduceOp.MAX, group=group)
    buf, nan = packed[: 2 * n], packed[2 * n:]
    buf = torch.where(nan > 0, torch.full_like(buf, float("nan")), buf)
    mins.copy_((-buf[:n]).reshape(mins.shape))
    maxs.copy_(buf[n:].reshape(maxs.shape))
    return mins, maxs


def sync_activation_ranges(model, group=None):
    """After a batch-sharded calibration pass: make every min/max estimator of `model` (and the
    quantizer range derived from it) identical on all ranks.  One collective for the whole model.

    Only estimators whose fold is order-independent are synchronised (current / all min-max);
    an EMA (running_minmax) depends on batch order and stays per-rank ("replicas only")."""
    from quantization.manager import QuantizationManager
    from quantization.estimators import AllMinMaxEstimator, CurrentMinMaxEstimator
    mgrs = [m for m in model.modules() if isinstance(m, QuantizationManager)
            and isinstance(m.range_estimator, (AllMinMaxEstimator, CurrentMinMaxEstimator))
            and m.range_estimator.current_xmin is not None and not m.per_channel]
    seen, uniq = set(), []
    for m in mgrs:                    # tied quantizers appear more than once
        if id(m) not in seen:
            seen.add(id(m))
            uniq.append(m)
    if not uniq:
        return 0
    mins = torch.stack([m.range_estimator.current_xmin.reshape(()) for m in uniq])
    maxs = torch.stack([m.range_estimator.current_xmax.reshape(()) for m in uniq])
    allreduce_ranges(mins, maxs, group)
    for i, m in enumerate(uniq):
        est = m.range_estimator
        est.current_xmin, est.current_xmax = mins[i].clone(), maxs[i].clone()
        m.set_quant_range(est.current_xmin, est.current_xmax)
    return len(uniq)


def enable_distributed_calibration(model, group=None, enable=True):
    """Batch-sharded (data-parallel) calibration that is EXACTLY the single-process calibration on the
    concatenated batch: every per-tensor min/max estimator of `model` all-reduces its estimate right after each
    update, i.e. before the batch is quantized with it (the reference's order, quantization_manager.py:119-122),
    so downstream layers see the same activations as in one process.

    Cost: one 16-byte all-reduce per activation quantizer and batch (ResNet-18: 30).  The MSE estimator exchanges
    its grid maximum once and its partial MSE table (<= 2.7 KB) per batch; the combined table equals the
    single-process one up to the fp32 rounding of the per-rank means.  Weights need nothing: every rank holds all
    channels.  Use sync_activation_ranges() instead for a single collective at the end, when it is acceptable that
    the activations seen DURING calibration differ between ranks.  Returns the number of estimators switched."""
    from quantization.manager import QuantizationManager
    from quantization.estimators import (AllMinMaxEstimator, CurrentMinMaxEstimator, RunningMinMaxEstimator,
                                         FP_MSE_Estimator)
    n = 0
    for m in model.modules():
        if isinstance(m, QuantizationManager) and not m.per_channel and isinstance(
                m.range_estimator, (AllMinMaxEstimator, CurrentMinMaxEstimator, RunningMinMaxEstimator,
                                    FP_MSE_Estimator)):
            m.range_estimator.dist_group = (group if group is not None else True) if enable else None
            n += 1
    return n


def calibrate_quantize_sharded(x_local, mbits, n_bits=8, sign_bits=1, state=None, group=None, ops=None, out=None,
                               timing=None):
    """BASELINE config 5: this rank's slab of a batch-sharded activation tensor.

    local allminmax fold -> all-reduce of the 2-float running range -> quantize the slab with the
    GLOBAL range (the reference updates the range from the current batch before quantizing it,
    quantization_manager.py:119-122).  Returns (y_local, state); state = (min, max) tensors [1]."""
    ops = ops or _default_ops()
    cur_min, cur_max = state if state is not None else (None, None)
    multi = _multi(group)
    packed = ops.new_packed(1, x_local.device) if multi else None
    _mark(timing)
    # the kernel leaves the folded range, K5's maxval and (multi-rank) the packed all-reduce operand
    if multi:
        cur_min, cur_max, maxval = ops.minmax(x_local, False, cur_min, cur_max, mode=1, want_maxval=True, packed=packed)
    else:
        cur_min, cur_max, maxval = ops.minmax(x_local, False, cur_min, cur_max, mode=1, want_maxval=True)
    _mark(timing)
    if multi:
        dist.all_reduce(packed, op=dist.ReduceOp.MAX, group=group)
        ops.ranges_unpack(packed, cur_min, cur_max, maxval)         # + fp8_quantizer.py:236 on the global range
    _mark(timing)
    y = ops.quantize(x_local, maxval, mbits, n_bits, sign_bits, out=out)
    _mark(timing)
    return y, (cur_min, cur_max)


def quantize_weight_sharded_codes(w, mbits, n_bits=8, sign_bits=1, group=None, ops=None, timing=None, maxval=None):
    """Channel-sharded weight quantization that ships 1-byte storage codes instead of fp32 values
    (SURVEY.md 8f N3): rank r finds the ranges of its channels and encodes them (fp8q_encode_u8),
    the ranks all-gather codes (1 B/element: 4x less xGMI traffic than fp32) and per-channel
    ranges, and every rank decodes the full tensor locally (fp8q_decode_u8).  The result is
    bit-identical to quantize_weight_sharded / to the single-process quantizer.
    maxval [C]: FIXED ranges that every rank already holds (after fix_ranges): nothing is estimated and only the codes
    travel.  Returns (w_q [C, ...], maxval [C], codes uint8 [C, ...])."""
    ops = ops or _default_ops()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    C = w.shape[0]
    inner = w.numel() // max(C, 1)
    lo, hi = channel_partition(C, world)[rank]
    shard = w[lo:hi].contiguous()
    fixed = maxval is not None
    _mark(timing)
    if hi > lo:
        mv_shard = maxval[lo:hi].contiguous() if fixed else ops.minmax(shard, True, want_maxval=True)[2]
        c_shard = ops.encode(shard, mv_shard, mbits, n_bits, sign_bits)
    else:
        mv_shard = w.new_empty(0)
        c_shard = torch.empty(0, dtype=torch.uint8, device=w.device)
    _mark(timing)
    if fixed and _multi(group) and C % world == 0:
        codes = torch.empty(w.shape, dtype=torch.uint8, device=w.device)
        dist.all_gather_into_tensor(codes.view(-1), c_shard.reshape(-1), group=group)       # the ranges are everywhere already
    elif fixed and world > 1:
        per = -(-C // world)
        send_c = torch.zeros(per * inner, dtype=torch.uint8, device=w.device)
        send_c[: (hi - lo) * inner] = c_shard.reshape(-1)
        recv_c = torch.empty(world * per * inner, dtype=torch.uint8, device=w.device)
        dist.all_gather_into_tensor(recv_c, send_c, group=group)
        codes = torch.cat([recv_c[r * per * inner: r * per * inner + (b - a) * inner]
                           for r, (a, b) in enumerate(channel_partition(C, world))]).view(w.shape)
    elif fixed:
        codes = c_shard.view(w.shape)
    elif _multi(group) and C % world == 0:
        # even split: gather straight into the final tensors (no padding, no re-assembly copies)
        codes = torch.empty(w.shape, dtype=torch.uint8, device=w.device)
        maxval = w.new_empty(C)
        dist.all_gather_into_tensor(codes.view(-1), c_shard.reshape(-1), group=group)
        dist.all_gather_into_tensor(maxval, mv_shard, group=group)
    elif world > 1:
        per = -(-C // world)
        send_c = torch.zeros(per * inner, dtype=torch.uint8, device=w.device)
        send_c[: (hi - lo) * inner] = c_shard.reshape(-1)
        send_m = w.new_zeros(per)
        send_m[: hi - lo] = mv_shard
        recv_c = torch.empty(world * per * inner, dtype=torch.uint8, device=w.device)
        recv_m = w.new_empty(world * per)
        dist.all_gather_into_tensor(recv_c, send_c, group=group)
        dist.all_gather_into_tensor(recv_m, send_m, group=group)
        parts_c, parts_m = [], []
        for r, (a, b) in enumerate(channel_partition(C, world)):
            parts_c.append(recv_c[r * per * inner: r * per * inner + (b - a) * inner])
            parts_m.append(recv_m[r * per: r * per + (b - a)])
        codes, maxval = torch.cat(parts_c).view(w.shape), torch.cat(parts_m)
    else:
        codes, maxval = c_shard.view(w.shape), mv_shard
    _mark(timing)
    y = ops.decode(codes, maxval, mbits, n_bits, sign_bits)
    _mark(timing)
    return y, maxval, codes


def quantize_weight_sharded(w, mbits, n_bits=8, sign_bits=1, maxval=None, group=None, ops=None,
                            gather=True, timing=None):
    """Per-output-channel weight quantization sharded over the ranks of `group`.

    Every rank holds the full fp32 weight `w` ([C, ...]); rank r quantizes channels
    channel_partition(C, W)[r] (current_minmax range unless `maxval` [C] is given) and the shards
    are re-assembled with one all-gather (+ one for the ranges).  Returns (w_q [C, ...], maxval [C]);
    with gather=False only this rank's shard is returned (no collective)."""
    ops = ops or _default_ops()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    C = w.shape[0]
    inner = w.numel() // max(C, 1)
    lo, hi = channel_partition(C, world)[rank]
    shard = w[lo:hi].contiguous()
    _mark(timing)
    if hi > lo:
        if maxval is None:
            q_shard, mv_shard = _local_minmax_quantize(ops, shard, mbits, n_bits, sign_bits)
        else:
            mv_shard = maxval[lo:hi].contiguous()
            q_shard = ops.quantize(shard, mv_shard, mbits, n_bits, sign_bits)
    else:
        q_shard = shard
        mv_shard = w.new_empty(0)
    _mark(timing)
    if not gather or not _multi(group):
        _mark(timing)
        return q_shard, mv_shard
    if C % world == 0:
        # even split: gather straight into the final tensors (no padding, no re-assembly copies)
        out, mv_out = torch.empty(w.shape, dtype=w.dtype, device=w.device), w.new_empty(C)   # contiguous for any layout of w
        dist.all_gather_into_tensor(out.view(-1), q_shard.reshape(-1), group=group)
        dist.all_gather_into_tensor(mv_out, mv_shard, group=group)
        _mark(timing)
        return out, mv_out
    # uneven split: equal-size exchange, every shard padded to ceil(C / W) channels
    per = -(-C // world)
    send = w.new_zeros(per * (inner + 1))
    send[: (hi - lo) * inner] = q_shard.reshape(-1)
    send[per * inner: per * inner + (hi - lo)] = mv_shard
    recv = w.new_empty(world * per * (inner + 1))
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(world, per * (inner + 1))
    parts, mvs = [], []
    for r, (a, b) in enumerate(channel_partition(C, world)):
        parts.append(recv[r, : (b - a) * inner])
        mvs.append(recv[r, per * inner: per * inner + (b - a)])
    out = torch.cat(parts).view_as(w), torch.cat(mvs)
    _mark(timing)
    return out


def quantize_weights_sharded_bucketed(weights, mbits, n_bits=8, sign_bits=1, group=None, ops=None, bucket_bytes=None,
                                      wire=None):
    """All weight tensors of a model, channel-sharded, with ONE collective per bucket: every rank quantizes its
    channels of every tensor (current_minmax ranges) straight into a packed send buffer, the ranks exchange that
    buffer with a single all-gather (one large transfer instead of 21 small ones), and every rank unpacks the full
    quantized tensors and ranges.

    wire="codes" (the default with more than one rank): the send buffer holds 1-byte STORAGE CODES and
    the fp32 ranges -- ResNet-18: 11.7 MB in total, 1.46 MB per rank, a quarter of the fp32 form -- written by one
    multi-tensor range launch + one multi-tensor encode launch per bucket (fp8q_multi_minmax_encode_u8) and decoded,
    straight out of the receive buffer into the result tensors, by one multi-tensor decode launch per 32 (tensor, rank)
    parts (fp8q_multi_decode_u8).  wire="fp32": the quantized fp32 values travel (46.7 MB / 5.8 MB per rank; the form of
    round 4, and what a single rank uses: there is nothing to ship).  Both give bit-identical tensors.

    bucket_bytes=None puts everything into one bucket.  With a limit (per-rank send bytes, e.g. 8 << 20) the tensors
    are packed into consecutive buckets and every bucket's all-gather is launched asynchronously (on RCCL's own
    stream) while the next bucket is being quantized on the compute stream; all handles are waited for before
    unpacking.  Result per tensor identical to quantize_weight_sharded.  Returns [(w_q, maxval), ...]."""
    ops = ops or _default_ops()
    if wire is None:
        wire = "codes" if (_multi(group) and hasattr(ops, "encode") and codes_wire_ok(mbits, n_bits, sign_bits)) else "fp32"
    if wire not in ("codes", "fp32"):
        raise ValueError(f"wire must be 'codes' or 'fp32', got {wire!r}")
    if wire == "codes":
        return _bucketed_codes(weights, mbits, n_bits, sign_bits, group, ops, bucket_bytes)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if not weights:
        return []
    dev0, dt = weights[0].device, weights[0].dtype
    esz = weights[0].element_size()
    # geometry per tensor: (bucket, C, inner, per, offset of values, offset of ranges) -- offsets within the bucket
    geo, totals = [], [0]
    for w in weights:
        C = w.shape[0]
        inner = w.numel() // max(C, 1)
        per = -(-C // world)
        need = -(-(per * (inner + 1)) // 4) * 4        # regions start on 16-byte boundaries: the aligned kernels apply
        if bucket_bytes is not None and totals[-1] > 0 and (totals[-1] + need) * esz > bucket_bytes:
            totals.append(0)
        geo.append((len(totals) - 1, C, inner, per, totals[-1], totals[-1] + per * inner))
        totals[-1] += need
    sends = [torch.zeros(t, dtype=dt, device=dev0) for t in totals]
    recvs, handles = [None] * len(totals), []

    def exchange(bi):
        if _multi(group):
            recvs[bi] = torch.empty(world * totals[bi], dtype=dt, device=dev0)
            h = dist.all_gather_into_tensor(recvs[bi], sends[bi], group=group, async_op=len(totals) > 1)
            if h is not None:
                handles.append(h)
            recvs[bi] = recvs[bi].view(world, totals[bi])
        else:
            recvs[bi] = sends[bi].view(1, totals[bi])

    multi = getattr(ops, "multi_minmax_quantize", None)   # ranges + quantize of a whole bucket in two launches
    pending = []
    for i, (w, (bi, C, inner, per, off_v, off_m)) in enumerate(zip(weights, geo)):
        lo, hi = channel_partition(C, world)[rank]
        if hi > lo:
            shard = w[lo:hi].contiguous()
            dst = sends[bi][off_v: off_v + (hi - lo) * inner].view(shard.shape)    # quantize straight into the send buffer
            if multi is not None:
                pending.append((shard, sends[bi][off_m: off_m + (hi - lo)], mbits, n_bits, sign_bits, dst))
            else:
                q, mv = _local_minmax_quantize(ops, shard, mbits, n_bits, sign_bits, out=dst)
                if q.data_ptr() != dst.data_ptr():
                    dst.copy_(q)
                sends[bi][off_m: off_m + (hi - lo)] = mv
        if i + 1 == len(weights) or geo[i + 1][0] != bi:   # bucket complete: ship it, go on quantizing the next
            if pending:
                multi(pending)      # every shard of the bucket: one range launch + one quantize launch, ranges land in place
                pending = []
            exchange(bi)
    for h in handles:
        h.wait()
    out = []
    for w, (bi, C, inner, per, off_v, off_m) in zip(weights, geo):
        if world == 1 and not _multi(group):      # nothing to re-assemble: views of the packed buffer
            out.append((recvs[bi][0, off_v: off_v + C * inner].view_as(w), recvs[bi][0, off_m: off_m + C]))
            continue
        parts, mvs = [], []
        for r, (a_, b_) in enumerate(channel_partition(C, world)):
            parts.append(recvs[bi][r, off_v: off_v + (b_ - a_) * inner])
            mvs.append(recvs[bi][r, off_m: off_m + (b_ - a_)])
        out.append((torch.cat(parts).view_as(w), torch.cat(mvs)))
    return out


def _bucketed_codes(weights, mbits, n_bits, sign_bits, group, ops, bucket_bytes):
    """quantize_weights_sharded_bucketed with 1-byte codes on the wire.  Bucket layout (uint8), per tensor:
    [codes of `per` channels, padded to 16 B | `per` fp32 ranges, padded to 16 B], per = ceil(C / world)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if not weights:
        return []
    dev0 = weights[0].device
    pad16 = lambda v: -(-v // 16) * 16
    geo, totals = [], [0]
    for w in weights:
        C = w.shape[0]
        inner = w.numel() // max(C, 1)
        per = -(-C // world)
        need = pad16(per * inner) + pad16(per * 4)
        if bucket_bytes is not None and totals[-1] > 0 and totals[-1] + need > bucket_bytes:
            totals.append(0)
        geo.append((len(totals) - 1, C, inner, per, totals[-1], totals[-1] + pad16(per * inner)))
        totals[-1] += need
    sends = [torch.zeros(t, dtype=torch.uint8, device=dev0) for t in totals]
    recvs, handles = [None] * len(totals), []

    def exchange(bi):
        if _multi(group):
            recvs[bi] = torch.empty(world * totals[bi], dtype=torch.uint8, device=dev0)
            h = dist.all_gather_into_tensor(recvs[bi], sends[bi], group=group, async_op=len(totals) > 1)
            if h is not None:
                handles.append(h)
            recvs[bi] = recvs[bi].view(world, totals[bi])
        else:
            recvs[bi] = sends[bi].view(1, totals[bi])

    multi = getattr(ops, "multi_minmax_encode", None)     # ranges + codes of a whole bucket in two launches
    pending = []
    for i, (w, (bi, C, inner, per, off_c, off_m)) in enumerate(zip(weights, geo)):
        lo, hi = channel_partition(C, world)[rank]
        if hi > lo:
            shard = w[lo:hi].contiguous()
            dst_c = sends[bi][off_c: off_c + (hi - lo) * inner].view((hi - lo,) + tuple(shard.shape[1:]))
            dst_m = sends[bi][off_m: off_m + (hi - lo) * 4].view(torch.float32)
            if multi is not None:
                pending.append((shard, dst_m, mbits, n_bits, sign_bits, dst_c))
            else:
                mv = ops.minmax(shard, True, want_maxval=True)[2]
                dst_m.copy_(mv)
                dst_c.copy_(ops.encode(shard, mv, mbits, n_bits, sign_bits).view_as(dst_c))
        if i + 1 == len(weights) or geo[i + 1][0] != bi:   # bucket complete: ship it, go on with the next
            if pending:
                multi(pending)
                pending = []
            exchange(bi)
    for h in handles:
        h.wait()
    # decode every (tensor, rank) part from where it was received into its place in the result: no re-assembly copies
    outs, items = [], []
    for w, (bi, C, inner, per, off_c, off_m) in zip(weights, geo):
        y = torch.empty(w.shape, dtype=torch.float32, device=dev0)
        mvs = []
        for r, (a_, b_) in enumerate(channel_partition(C, world)):
            if b_ == a_:
                continue
            codes = recvs[bi][r, off_c: off_c + (b_ - a_) * inner].view((b_ - a_,) + tuple(w.shape[1:]))
            mv = recvs[bi][r, off_m: off_m + (b_ - a_) * 4].view(torch.float32)
            items.append((codes, mv, mbits, n_bits, sign_bits, y[a_:b_]))
            mvs.append(mv)
        outs.append((y, torch.cat(mvs) if len(mvs) != 1 else mvs[0].clone()))
    mdec = getattr(ops, "multi_decode", None)
    if mdec is not None:
        mdec(items)
    else:
        for codes, mv, mb, nb, sb, dst in items:
            dst.copy_(ops.decode(codes, mv, mb, nb, sb).view_as(dst))
    return outs


N_MSE_GRID = 111   # candidates of FP_MSE_Estimator (range_estimators.py:305)


def mse_search_sharded(x_local, per_channel, mbit_list, n_bits=8, sign_bits=1, shard="batch", state=None,
                       group=None, ops=None):
    """FP_MSE_Estimator.forward (range_estimators.py:318-369) over a sharded tensor.

    shard="batch"    activations, per tensor: every rank holds some images.  Two exchange steps: all-reduce(MAX)
                     of the first batch's abs-max (it defines the 111-point search grid) and all-reduce(SUM) of
                     the element-count-weighted partial MSEs [|m|, 111, 1] (<= 2.7 KB, combined in float64).
    shard="channel"  weights, per channel: x_local = this rank's channels (channel_partition).  Channels are
                     independent; the only exchange is the all-gather of the per-channel best mantissa width
                     for the plurality vote (:350-354), C ints.
    state            (grid, mses) from the previous call on this estimator (MSEs accumulate over batches)
    Returns (maxval [C_local], best_mbits, state)."""
    ops = ops or _default_ops()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    n_m = len(mbit_list)
    empty = shard == "channel" and per_channel and x_local.shape[0] == 0   # more ranks than channels: this rank has none
    if empty:
        # no local work, but the vote exchange below is a collective: take part with zero votes
        grid = x_local.new_zeros(N_MSE_GRID, 0)
        mses = x_local.new_zeros(n_m, N_MSE_GRID, 0)
    elif state is None:
        mx = ops.minmax(x_local, per_channel, want_maxval=True)[2]
        if shard == "batch" and _multi(group):
            dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
        grid = ops.mse_linspace(mx, N_MSE_GRID)                # [111, C] on the device, == torch.linspace per channel
        mses = torch.zeros(n_m, N_MSE_GRID, grid.shape[1], device=x_local.device)
    else:
        grid, mses = state
    inc = torch.zeros_like(mses)
    if not empty:
        ops.mse_grid(x_local, per_channel, grid, list(mbit_list), n_bits, sign_bits, inc)
    if shard == "batch" and _multi(group):
        n_local = float(x_local.numel() // grid.shape[1])
        packed = torch.cat([inc.double().reshape(-1) * n_local, torch.tensor([n_local], dtype=torch.float64,
                                                                               device=inc.device)])
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
        inc = (packed[:-1] / packed[-1]).to(mses.dtype).view_as(mses)
    mses += inc
    best_m_per_ch = mses.min(1)[0].argmin(0)                                           # [C_local]
    votes = best_m_per_ch
    if shard == "channel" and _multi(group):
        sizes = [torch.zeros(1, dtype=torch.int64, device=votes.device) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([votes.numel()], dtype=torch.int64, device=votes.device), group=group)
        per = int(max(int(s.item()) for s in sizes))
        send = torch.full((max(per, 1),), -1, dtype=torch.int64, device=votes.device)
        send[: votes.numel()] = votes
        recv = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(recv, send, group=group)
        votes = torch.cat([r[: int(s.item())] for r, s in zip(recv, sizes)])
    best_idx = int(torch.mode(votes).values.item())
    arg = mses[best_idx].argmin(0)
    maxval = grid.gather(0, arg.unsqueeze(0)).squeeze(0)
    return maxval, float(mbit_list[best_idx]), (grid, mses)
