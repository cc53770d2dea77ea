"""World-size-2 tests of the sharding / collective logic on CPU (gloo).  The local compute is an
oracle-backed stand-in for fp8q.ops (same call signatures): tests may use the oracle, the product
default (HIP) is exercised by the -m gpu tests."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle


class OracleOps:
    """fp8q.ops look-alike on CPU tensors, for the gloo tests only."""

    @staticmethod
    def minmax(x, per_channel, cur_min=None, cur_max=None, mode=0, momentum=0.9, want_maxval=False, packed=None):
        import oracle_ops
        return oracle_ops.minmax(x, per_channel, cur_min, cur_max, mode, momentum, want_maxval, packed)

    @staticmethod
    def new_packed(C, device):
        import oracle_ops
        return oracle_ops.new_packed(C, device)

    @staticmethod
    def ranges_unpack(packed, cur_min=None, cur_max=None, maxval=None):
        import oracle_ops
        return oracle_ops.ranges_unpack(packed, cur_min, cur_max, maxval)

    @staticmethod
    def quantize(x, maxval, mbits, n_bits=8, sign_bits=1, out=None):
        return torch.from_numpy(oracle.c_quantize(x.numpy(), maxval.numpy(), mbits, n_bits, sign_bits))

    @staticmethod
    def minmax_quantize(x, mbits, n_bits=8, sign_bits=1, out=None):
        mn, mx = oracle.c_minmax(x.numpy(), True)
        mv = oracle.c_absmax(mn, mx)
        y = oracle.c_quantize(x.numpy(), mv, mbits, n_bits, sign_bits)
        return torch.from_numpy(y), torch.from_numpy(mn), torch.from_numpy(mx), torch.from_numpy(mv)


    @staticmethod
    def mse_grid(x, per_channel, grid, mbits_list, n_bits, sign_bits, mses):
        mses += torch.from_numpy(oracle.c_mse_grid(x.numpy(), per_channel, grid.numpy(), mbits_list, n_bits, sign_bits))
        return mses

    @staticmethod
    def mse_linspace(mx, steps=111, lo_frac=0.1, hi_frac=1.2):
        import oracle_ops
        return oracle_ops.mse_linspace(mx, steps, lo_frac, hi_frac)

    @staticmethod
    def mse_select(mses, grid, mbits_list, sign_bits=1):
        import oracle_ops
        return oracle_ops.mse_select(mses, grid, mbits_list, sign_bits)

    @staticmethod
    def encode(x, maxval, mbits, n_bits=8, sign_bits=1, out=None):
        return torch.from_numpy(oracle.c_encode(x.numpy(), maxval.numpy(), mbits, n_bits, sign_bits))

    @staticmethod
    def decode(codes, maxval, mbits, n_bits=8, sign_bits=1, out=None):
        return torch.from_numpy(oracle.c_decode(codes.numpy(), maxval.numpy(), mbits, n_bits, sign_bits))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def run(fn, world=2):
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _weights():
    rng = np.random.RandomState(0)
    w = (rng.randn(13, 3, 3, 3) * 0.1).astype(np.float32)   # 13 channels: uneven split 7 + 6
    w[4] = 0                                                # NaN channel quirk survives the gather
    return w


def _w_job(rank, world):
    from fp8q import dist as fd
    w = torch.from_numpy(_weights())
    q, mv = fd.quantize_weight_sharded(w, 2, 8, 1, ops=OracleOps)
    return q.numpy(), mv.numpy()


def test_weight_channel_shard_allgather_bit_equal():
    from fp8q.dist import channel_partition
    assert channel_partition(13, 2) == [(0, 7), (7, 13)]
    assert channel_partition(1000, 8)[0] == (0, 125) and channel_partition(3, 8)[5] == (3, 3)
    w = _weights()
    mn, mx = oracle.c_minmax(w, True)
    mv = oracle.c_absmax(mn, mx)
    ref = oracle.c_quantize(w, mv, 2, 8, 1)
    for q, m in run(_w_job):
        assert np.array_equal(np.isnan(q), np.isnan(ref))
        assert np.array_equal(np.nan_to_num(q).view(np.int32), np.nan_to_num(ref).view(np.int32))
        np.testing.assert_array_equal(m, mv)


def _w_even_job(rank, world):
    """12 channels over 2 ranks: the even split gathers straight into the final tensors (no padding); plus rows longer
    than the fused kernel's limit (min/max then quantize instead of the fused launch)."""
    from fp8q import dist as fd
    w = torch.from_numpy(_weights()[:12])

    class ShortFused(OracleOps):
        calls = []

        @staticmethod
        def fused_max_inner():
            return 8           # 27-element rows do not fit: the guard must take the two-launch path

        @staticmethod
        def minmax_quantize(*a, **k):
            raise AssertionError("fused launch used for rows longer than fused_max_inner()")

    q, mv = fd.quantize_weight_sharded(w, 2, 8, 1, ops=OracleOps)
    q2, mv2 = fd.quantize_weight_sharded(w, 2, 8, 1, ops=ShortFused)
    qc, mvc, codes = fd.quantize_weight_sharded_codes(w[[0, 1, 2, 3, 5, 6, 7, 8]].contiguous(), 3, 8, 1, ops=OracleOps)
    return q.numpy(), mv.numpy(), q2.numpy(), mv2.numpy(), qc.numpy(), mvc.numpy(), codes.numpy()


def test_weight_even_split_and_long_rows():
    w = _weights()[:12]
    mn, mx = oracle.c_minmax(w, True)
    mv = oracle.c_absmax(mn, mx)
    ref = oracle.c_quantize(w, mv, 2, 8, 1)
    w8 = w[[0, 1, 2, 3, 5, 6, 7, 8]]
    mn8, mx8 = oracle.c_minmax(w8, True)
    mv8 = oracle.c_absmax(mn8, mx8)
    for q, m, q2, m2, qc, mc, codes in run(_w_even_job):
        for got, gm in ((q, m), (q2, m2)):
            assert np.array_equal(np.isnan(got), np.isnan(ref))
            assert np.array_equal(np.nan_to_num(got).view(np.int32), np.nan_to_num(ref).view(np.int32))
            np.testing.assert_array_equal(gm, mv)
        assert np.array_equal(qc.view(np.int32), oracle.c_quantize(w8, mv8, 3, 8, 1).view(np.int32))
        np.testing.assert_array_equal(mc, mv8)
        np.testing.assert_array_equal(codes, oracle.c_encode(w8, mv8, 3, 8, 1))


def _w_codes_job(rank, world):
    from fp8q import dist as fd
    w = torch.from_numpy(_weights()[[0, 1, 2, 3, 5, 6, 7, 8, 9, 10, 11]])   # 11 channels, no zero channel
    q, mv, codes = fd.quantize_weight_sharded_codes(w, 3, 8, 1, ops=OracleOps)
    return q.numpy(), mv.numpy(), codes.numpy()


def test_weight_shard_allgather_of_storage_codes():
    """1-byte codes on the wire, decode after the gather: same bits as the single-process quantizer."""
    w = _weights()[[0, 1, 2, 3, 5, 6, 7, 8, 9, 10, 11]]
    mn, mx = oracle.c_minmax(w, True)
    mv = oracle.c_absmax(mn, mx)
    ref = oracle.c_quantize(w, mv, 3, 8, 1)
    for q, m, codes in run(_w_codes_job):
        assert codes.dtype == np.uint8 and codes.shape == w.shape
        assert np.array_equal(q.view(np.int32), ref.view(np.int32))
        np.testing.assert_array_equal(m, mv)
        np.testing.assert_array_equal(codes, oracle.c_encode(w, mv, 3, 8, 1))


def _batches():
    rng = np.random.RandomState(1)
    return [(rng.randn(4, 8, 5, 5) * (1 + i)).astype(np.float32) for i in range(4)]


def _c5_job(rank, world):
    from fp8q import dist as fd
    state, outs = None, []
    for b in _batches():
        x = torch.from_numpy(b[rank * 2:(rank + 1) * 2].copy())      # batch-sharded: 2 of 4 images
        y, state = fd.calibrate_quantize_sharded(x, 3, 8, 1, state=state, ops=OracleOps)
        outs.append(y.numpy())
    return outs, state[0].numpy(), state[1].numpy()


def test_batch_sharded_calibration_equals_single_process():
    """allminmax is associative/commutative: the sharded run equals the 1-process run bit for bit."""
    res = run(_c5_job)
    cur = None
    for i, b in enumerate(_batches()):
        mn, mx = oracle.c_minmax(b, False)
        cur = (mn, mx) if cur is None else oracle.c_fold(cur[0], cur[1], mn, mx, 1)
        mv = oracle.c_absmax(*cur)
        ref = oracle.c_quantize(b, mv, 3, 8, 1)
        got = np.concatenate([res[0][0][i], res[1][0][i]])
        assert np.array_equal(got.view(np.int32), ref.view(np.int32)), f"batch {i}"
    for r in range(2):
        np.testing.assert_array_equal(res[r][1], cur[0])
        np.testing.assert_array_equal(res[r][2], cur[1])


def _nan_job(rank, world):
    from fp8q import dist as fd
    mins = torch.tensor([-1.0 - rank, float("nan") if rank == 1 else -3.0, 0.5])
    maxs = torch.tensor([2.0 + rank, 4.0, float("nan") if rank == 0 else 1.0])
    fd.allreduce_ranges(mins, maxs)
    return mins.numpy(), maxs.numpy()


def test_allreduce_ranges_nan_propagates():
    for mins, maxs in run(_nan_job):
        assert mins[0] == -2.0 and maxs[0] == 3.0
        assert np.isnan(mins[1]) and maxs[1] == 4.0
        assert mins[2] == 0.5 and np.isnan(maxs[2])


def _mse_acts():
    rng = np.random.RandomState(3)
    return [(rng.randn(6, 4, 5, 5) * s).astype(np.float32) for s in (1.0, 1.7)]   # two calibration batches


def _mse_batch_job(rank, world):
    from fp8q import dist as fd
    state, out = None, None
    for b in _mse_acts():
        shard = torch.from_numpy(b[rank::world].copy())          # images of this rank (uneven: 3 + 3 of 6)
        out = fd.mse_search_sharded(shard, False, [1.0, 2.0, 3.0, 4.0], 8, 1, "batch", state, ops=OracleOps)
        state = out[2]
    return out[0].numpy(), out[1], state[1].numpy()


def test_mse_search_batch_sharded_matches_single_process():
    """Activation MSE calibration, batch-sharded: grid from the all-reduced max, MSEs from the all-reduced
    weighted partial sums -> same grid, same MSE table (fp32 rounding of the per-rank means), same winner."""
    from fp8q import dist as fd
    state = None
    for b in _mse_acts():
        ref = fd.mse_search_sharded(torch.from_numpy(b), False, [1.0, 2.0, 3.0, 4.0], 8, 1, "batch", state, ops=OracleOps)
        state = ref[2]
    for mv, m, mses in run(_mse_batch_job):
        np.testing.assert_allclose(mses, state[1].numpy(), rtol=2e-6)
        assert m == ref[1]
        np.testing.assert_array_equal(mv, ref[0].numpy())


def _mse_weights():
    rng = np.random.RandomState(4)
    return (rng.randn(11, 3, 3, 3) * np.exp(rng.uniform(-2, 0, (11, 1, 1, 1)))).astype(np.float32)


def _mse_channel_job(rank, world):
    from fp8q import dist as fd
    w = _mse_weights()
    lo, hi = fd.channel_partition(w.shape[0], world)[rank]
    mv, m, st = fd.mse_search_sharded(torch.from_numpy(w[lo:hi].copy()), True, [1.0, 2.0, 3.0, 4.0, 5.0, 6.0], 8, 1,
                                      "channel", None, ops=OracleOps)
    return lo, hi, mv.numpy(), m


def test_mse_search_channel_sharded_matches_single_process():
    """Weight MSE search, channel-sharded (11 channels: 6 + 5): per-channel results are local, the mantissa
    width is the plurality vote over ALL channels (one all-gather of C ints) -> identical to one process."""
    from fp8q import dist as fd
    w = _mse_weights()
    ref_mv, ref_m, _ = fd.mse_search_sharded(torch.from_numpy(w), True, [1.0, 2.0, 3.0, 4.0, 5.0, 6.0], 8, 1,
                                             "channel", None, ops=OracleOps)
    for lo, hi, mv, m in run(_mse_channel_job):
        assert m == ref_m
        np.testing.assert_array_equal(mv, ref_mv.numpy()[lo:hi])


def _mse_one_channel_job(rank, world):
    from fp8q import dist as fd
    w = _mse_weights()[:1]                      # one channel, two ranks: rank 1 owns nothing
    lo, hi = fd.channel_partition(1, world)[rank]
    mv, m, st = fd.mse_search_sharded(torch.from_numpy(w[lo:hi].copy()), True, [2.0, 3.0], 8, 1, "channel", None,
                                      ops=OracleOps)
    return lo, hi, mv.numpy(), m


def test_mse_search_channel_sharded_rank_without_channels():
    """More ranks than channels: the rank that owns none still takes part in the vote exchange (no hang, no error) and
    every rank reports the single-process mantissa width."""
    from fp8q import dist as fd
    w = _mse_weights()[:1]
    ref_mv, ref_m, _ = fd.mse_search_sharded(torch.from_numpy(w), True, [2.0, 3.0], 8, 1, "channel", None, ops=OracleOps)
    res = run(_mse_one_channel_job)
    assert [r[:2] for r in res] == [(0, 1), (1, 1)]
    assert all(r[3] == ref_m for r in res)
    np.testing.assert_array_equal(res[0][2], ref_mv.numpy())
    assert res[1][2].size == 0


def _codes_fixed_job(rank, world):
    """the bench's N > 1 headline flow: fixed ranges held by every rank, the shards travel as 1-byte codes, nothing else"""
    from fp8q import dist as fd
    out = []
    for C in (16, 13, 3):            # even split, uneven split, more ranks than some ranks' channels
        rng = np.random.RandomState(C)
        w = torch.from_numpy((rng.randn(C, 3, 5) * 0.2).astype(np.float32))
        mn, mx = oracle.c_minmax(w.numpy(), True)
        mv = torch.from_numpy(oracle.c_absmax(mn, mx))
        q, mv_out, codes = fd.quantize_weight_sharded_codes(w, 2, 8, 1, ops=OracleOps, maxval=mv)
        assert mv_out is mv and codes.dtype == torch.uint8 and codes.shape == w.shape
        out.append((w.numpy(), mv.numpy(), q.numpy()))
    return out


def test_codes_wire_with_fixed_ranges_equals_the_single_process_quantizer():
    for world in (2, 4):
        for res in run(_codes_fixed_job, world=world):
            for w, mv, q in res:
                ref = oracle.c_quantize(w, mv, 2, 8, 1)
                assert np.array_equal(q.view(np.int32), ref.view(np.int32))


def _bucket_weights():
    rng = np.random.RandomState(7)
    return [(rng.randn(*shp) * 0.1).astype(np.float32) for shp in ((13, 3, 3, 3), (8, 13, 1, 1), (5, 8), (1, 7), (16, 4, 3, 3))]


def _bucket_job(rank, world, bucket_bytes=None):
    from fp8q import dist as fd
    out = fd.quantize_weights_sharded_bucketed([torch.from_numpy(w) for w in _bucket_weights()], 2, 8, 1, ops=OracleOps,
                                               bucket_bytes=bucket_bytes, wire="fp32")
    return [(q.numpy(), mv.numpy()) for q, mv in out]


def _bucket_job_codes(rank, world, bucket_bytes=None, ops=None):
    """the default wire form at world > 1: 1-byte storage codes + fp32 ranges"""
    from fp8q import dist as fd
    out = fd.quantize_weights_sharded_bucketed([torch.from_numpy(w) for w in _bucket_weights()], 2, 8, 1, ops=ops or OracleOps,
                                               bucket_bytes=bucket_bytes)
    return [(q.numpy(), mv.numpy()) for q, mv in out]


def _bucket_job_codes_small(rank, world):
    return _bucket_job_codes(rank, world, bucket_bytes=200)       # several buckets, async all-gathers


class MultiCodecOracleOps(OracleOps):
    """+ the multi-tensor range+encode / decode entry points (what fp8q.ops has): two launches per bucket, one decode"""
    calls = [0, 0]

    @classmethod
    def multi_minmax_encode(cls, items):
        cls.calls[0] += 1
        for x, mv_out, mbits, n_bits, sign_bits, out in items:
            mv = OracleOps.minmax(x, True, want_maxval=True)[2]
            mv_out.copy_(mv)
            out.copy_(OracleOps.encode(x, mv, mbits, n_bits, sign_bits).view_as(out))

    @classmethod
    def multi_decode(cls, items):
        cls.calls[1] += 1
        for codes, mv, mbits, n_bits, sign_bits, out in items:
            out.copy_(OracleOps.decode(codes.contiguous(), mv.contiguous(), mbits, n_bits, sign_bits).view_as(out))


def _bucket_job_codes_multi(rank, world):
    res = _bucket_job_codes(rank, world, ops=MultiCodecOracleOps)
    assert MultiCodecOracleOps.calls == [1, 1], MultiCodecOracleOps.calls       # one bucket: one encode call, one decode call
    return res


class MultiOracleOps(OracleOps):
    """+ the multi-tensor range+quantize entry point (what fp8q.ops has): the bucketed path takes its two-launch branch"""
    calls = [0]

    @classmethod
    def multi_minmax_quantize(cls, items):
        cls.calls[0] += 1
        outs = []
        for x, mv_out, mbits, n_bits, sign_bits, out in items:
            q, _, _, mv = OracleOps.minmax_quantize(x, mbits, n_bits, sign_bits)
            mv_out.copy_(mv)
            out.copy_(q)
            outs.append(out)
        return outs


def _bucket_job_multi(rank, world):
    from fp8q import dist as fd
    out = fd.quantize_weights_sharded_bucketed([torch.from_numpy(w) for w in _bucket_weights()], 2, 8, 1, ops=MultiOracleOps,
                                               bucket_bytes=600, wire="fp32")
    assert MultiOracleOps.calls[0] == 3, MultiOracleOps.calls      # one call per bucket, not one per tensor
    return [(q.numpy(), mv.numpy()) for q, mv in out]


def _bucket_job_small(rank, world):
    return _bucket_job(rank, world, bucket_bytes=600)      # 5 tensors -> 3 buckets, async all-gathers


def _bucket_job_tiny(rank, world):
    return _bucket_job(rank, world, bucket_bytes=1)        # every tensor its own bucket


def test_bucketed_weight_quantization_one_all_gather():
    """All layers' channel shards in one packed buffer and ONE all-gather (uneven splits, a 1-channel tensor that
    only rank 0 owns), or packed into several buckets whose all-gathers are launched asynchronously while the next
    bucket is quantized: every rank ends up with exactly the single-process quantized tensors and ranges."""
    ws = _bucket_weights()
    for job in (_bucket_job, _bucket_job_small, _bucket_job_tiny, _bucket_job_multi, _bucket_job_codes, _bucket_job_codes_small,
                _bucket_job_codes_multi):
        for res in run(job):
            for w, (q, mv) in zip(ws, res):
                mn, mx = oracle.c_minmax(w, True)
                rmv = oracle.c_absmax(mn, mx)
                np.testing.assert_array_equal(mv, rmv)
                ref = oracle.c_quantize(w, rmv, 2, 8, 1)
                assert np.array_equal(np.isnan(q), np.isnan(ref)) and np.array_equal(q[~np.isnan(ref)], ref[~np.isnan(ref)])


class RefusingCodecOps(OracleOps):
    """a codec that would refuse the format (what fp8q_encode_u8 answers for a format without an exponent bit)"""

    @staticmethod
    def encode(*a, **k):
        raise AssertionError("the codes wire was chosen for a format without an exponent bit")


def _bucket_job_no_exponent_bit(rank, world):
    """mantissa_bits = n_bits - sign_bits: E = 0, which K1 and the reference's clamp accept (fp8_quantizer.py:105-106) and
    the storage codec refuses -- the default wire form must fall back to fp32 values instead of raising"""
    from fp8q import dist as fd
    assert not fd.codes_wire_ok(7, 8, 1) and not fd.codes_wire_ok(9.0, 8, 1) and not fd.codes_wire_ok(3, 12, 1)
    assert fd.codes_wire_ok(6.4, 8, 1) and fd.codes_wire_ok(2, 8, 1) and fd.codes_wire_ok(7, 8, 0) and not fd.codes_wire_ok(6.5, 7, 1)
    out = fd.quantize_weights_sharded_bucketed([torch.from_numpy(w) for w in _bucket_weights()], 7, 8, 1, ops=RefusingCodecOps,
                                               bucket_bytes=600)
    return [(q.numpy(), mv.numpy()) for q, mv in out]


def test_bucketed_default_wire_falls_back_to_fp32_without_an_exponent_bit():
    ws = _bucket_weights()
    for res in run(_bucket_job_no_exponent_bit):
        for w, (q, mv) in zip(ws, res):
            mn, mx = oracle.c_minmax(w, True)
            rmv = oracle.c_absmax(mn, mx)
            np.testing.assert_array_equal(mv, rmv)
            ref = oracle.c_quantize(w, rmv, 7, 8, 1)
            assert np.array_equal(np.isnan(q), np.isnan(ref)) and np.array_equal(q[~np.isnan(ref)], ref[~np.isnan(ref)])


def _dp_model():
    import torch.nn as nn
    from quantization.autoquant_utils import quantize_model
    from quantization.quantization_manager import QMethods
    from quantization.range_estimators import RangeEstimators
    torch.manual_seed(5)
    net = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1, bias=False), nn.BatchNorm2d(8), nn.ReLU(),
                        nn.Conv2d(8, 12, 3, stride=2, padding=1, bias=True), nn.ReLU6(),
                        nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(12, 5)).eval()
    with torch.no_grad():
        net[1].running_mean.uniform_(-0.2, 0.2)
        net[1].running_var.uniform_(0.5, 1.5)
    return net, dict(method=QMethods.fp_quantizer.cls, weight_range_method=RangeEstimators.current_minmax.cls,
                     n_bits=8, per_channel_weights=True, fp8_kwargs=dict(maxval=None, mantissa_bits=3, set_maxval=True))


def _dp_batches():
    rng = np.random.RandomState(11)
    return [(rng.randn(6, 3, 10, 10) * s).astype(np.float32) for s in (1.0, 1.6)]      # two calibration steps


def _dp_calibrate(act_est, shard=None, world=1):
    """Calibrate on two steps; with `shard` every rank sees only its images of each step."""
    import oracle_ops
    from fp8q import dist as fd
    from quantization.autoquant_utils import quantize_model
    from quantization.base_quantized_classes import QuantizedModule
    from quantization.quantization_manager import QuantizationManager
    from quantization.range_estimators import RangeEstimators
    net, qp = _dp_model()
    q = quantize_model(net, act_range_method=RangeEstimators[act_est].cls, **qp).eval()
    if shard is not None:
        assert fd.enable_distributed_calibration(q) > 0
    outs = []
    with oracle_ops.patched(), torch.no_grad():
        for m in q.modules():
            if isinstance(m, QuantizedModule):
                m.quantized()
        for b in _dp_batches():
            xb = torch.from_numpy(b if shard is None else b[shard::world].copy())
            outs.append(q(xb).numpy())
    ranges = {n: (m.range_estimator.current_xmin.numpy().copy(), m.range_estimator.current_xmax.numpy().copy(),
                  m.quantizer.maxval.numpy().copy())
              for n, m in q.named_modules() if isinstance(m, QuantizationManager) and m.range_estimator is not None
              and m.range_estimator.current_xmin is not None}
    return outs, ranges


def _dp_job_all(rank, world):
    return _dp_calibrate("allminmax", rank, world)


def _dp_job_run(rank, world):
    return _dp_calibrate("running_minmax", rank, world)


@pytest.mark.parametrize("est,job", [("allminmax", _dp_job_all), ("running_minmax", _dp_job_run)])
def test_data_parallel_calibration_equals_single_process(est, job):
    """enable_distributed_calibration: two ranks, each with half of the images of every calibration step, end up with
    exactly the ranges -- and produce exactly the activations / logits for their images -- that one process computes
    on the whole batches (per-layer all-reduce before the batch is quantized; exact for min/max AND for the EMA)."""
    ref_outs, ref_ranges = _dp_calibrate(est)
    for rank, (outs, ranges) in enumerate(run(job)):
        assert ranges.keys() == ref_ranges.keys() and len(ranges) >= 6
        for name in ranges:
            for a, b in zip(ranges[name], ref_ranges[name]):
                np.testing.assert_array_equal(a, b, err_msg=name)
        for step, (o, r) in enumerate(zip(outs, ref_outs)):
            np.testing.assert_array_equal(o, r[rank::2], err_msg=f"step {step}")


def _dp_job_mse(rank, world):
    return _dp_calibrate("MSE", rank, world)


def test_data_parallel_mse_calibration():
    """The MSE estimator under enable_distributed_calibration: grid from the all-reduced maximum, MSE table from the
    all-reduced weighted partial sums -> the same ranges as one process on the whole batches (the fixture has no
    near-ties), hence the same logits."""
    ref_outs, ref_ranges = _dp_calibrate("MSE")
    for rank, (outs, ranges) in enumerate(run(_dp_job_mse)):
        for name in ref_ranges:
            np.testing.assert_allclose(ranges[name][2], ref_ranges[name][2], rtol=1e-6, err_msg=name)
        for step, (o, r) in enumerate(zip(outs, ref_outs)):
            np.testing.assert_allclose(o, r[rank::2], rtol=1e-5, atol=1e-6, err_msg=f"step {step}")


# ---- world size 4: uneven channel partitions, a rank without channels, 4-way batch shards -------------------------

def _w4_job(rank, world):
    from fp8q import dist as fd
    w = torch.from_numpy(_weights())                     # 13 channels over 4 ranks: 4 + 3 + 3 + 3
    q, mv = fd.quantize_weight_sharded(w, 2, 8, 1, ops=OracleOps)
    w3 = torch.from_numpy(_weights()[:3].copy())         # 3 channels over 4 ranks: rank 3 owns nothing
    q3, mv3 = fd.quantize_weight_sharded(w3, 3, 8, 1, ops=OracleOps)
    qc, mvc, codes = fd.quantize_weight_sharded_codes(w3, 3, 8, 1, ops=OracleOps)
    buck = fd.quantize_weights_sharded_bucketed([torch.from_numpy(t) for t in _bucket_weights()], 2, 8, 1, ops=OracleOps,
                                                bucket_bytes=600)
    # the default above ships 1-byte codes (4 ranks: uneven partitions, ranks without channels); the fp32 wire form must
    # give the same bits
    buck32 = fd.quantize_weights_sharded_bucketed([torch.from_numpy(t) for t in _bucket_weights()], 2, 8, 1, ops=OracleOps,
                                                  wire="fp32")
    for (a, am), (b, bm) in zip(buck, buck32):
        assert torch.equal(am, bm) and torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(a[~torch.isnan(a)], b[~torch.isnan(b)])
    return (q.numpy(), mv.numpy(), q3.numpy(), mv3.numpy(), qc.numpy(), mvc.numpy(), codes.numpy(),
            [(a.numpy(), b.numpy()) for a, b in buck])


def _assert_same_quant(got, ref):
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    assert np.array_equal(np.nan_to_num(got).view(np.int32), np.nan_to_num(ref).view(np.int32))


def test_four_ranks_uneven_partitions_and_empty_rank():
    from fp8q.dist import channel_partition
    assert channel_partition(13, 4) == [(0, 4), (4, 7), (7, 10), (10, 13)]
    assert channel_partition(3, 4) == [(0, 1), (1, 2), (2, 3), (3, 3)]
    w = _weights()
    mn, mx = oracle.c_minmax(w, True)
    mv = oracle.c_absmax(mn, mx)
    ref = oracle.c_quantize(w, mv, 2, 8, 1)
    w3 = w[:3]
    mv3 = oracle.c_absmax(*oracle.c_minmax(w3, True))
    ref3 = oracle.c_quantize(w3, mv3, 3, 8, 1)
    res = run(_w4_job, world=4)
    assert len(res) == 4
    for q, m, q3, m3, qc, mc, codes, buck in res:
        _assert_same_quant(q, ref)
        np.testing.assert_array_equal(m, mv)
        _assert_same_quant(q3, ref3)
        np.testing.assert_array_equal(m3, mv3)
        _assert_same_quant(qc, ref3)
        np.testing.assert_array_equal(mc, mv3)
        np.testing.assert_array_equal(codes, oracle.c_encode(w3, mv3, 3, 8, 1))
        for wb, (qb, mvb) in zip(_bucket_weights(), buck):
            rmv = oracle.c_absmax(*oracle.c_minmax(wb, True))
            np.testing.assert_array_equal(mvb, rmv)
            _assert_same_quant(qb, oracle.c_quantize(wb, rmv, 2, 8, 1))


def _c5_job4(rank, world):
    from fp8q import dist as fd
    state, outs = None, []
    for i, b in enumerate(_batches()):
        x = torch.from_numpy(b[rank:rank + 1].copy())               # 4 images, one per rank
        if i == 2 and rank == 3:
            x = x.clone()
            x[0, 0, 0, 0] = float("nan")                            # a NaN on ONE rank must reach every rank
        y, state = fd.calibrate_quantize_sharded(x, 3, 8, 1, state=state, ops=OracleOps)
        outs.append(y.numpy())
    return outs, state[0].numpy(), state[1].numpy()


def test_four_rank_batch_sharded_calibration_packed_exchange():
    """Config 5's flow on 4 ranks through the packed exchange (kernel-written {-min, max, nan flags} -> ONE
    all-reduce(MAX) -> unpack): ranges and quantized shards equal the single-process run; a NaN seen by one rank
    makes the running range NaN everywhere from that batch on (torch.min / torch.max semantics)."""
    res = run(_c5_job4, world=4)
    cur = None
    for i, b in enumerate(_batches()):
        b = b.copy()
        if i == 2:
            b[3, 0, 0, 0] = np.nan
        mn, mx = oracle.c_minmax(b, False)
        cur = (mn, mx) if cur is None else oracle.c_fold(cur[0], cur[1], mn, mx, 1)
        mv = oracle.c_absmax(*cur)
        ref = oracle.c_quantize(b, mv, 3, 8, 1)
        got = np.concatenate([res[r][0][i] for r in range(4)])
        _assert_same_quant(got, ref)
    assert np.isnan(cur[0]).all() and np.isnan(cur[1]).all()
    for r in range(4):
        assert np.isnan(res[r][1]).all() and np.isnan(res[r][2]).all()


def test_pack_unpack_ranges_roundtrip():
    """the stand-in's packed format (what fold_store writes on the device; the -m gpu tests compare the two)"""
    import oracle_ops
    mn = np.array([-1.5, np.nan, 0.0, -0.0, 3.0], np.float32)
    mx = np.array([2.0, 4.0, np.nan, 0.0, 7.0], np.float32)
    p = oracle_ops.new_packed(5, "cpu")
    oracle_ops.pack_ranges(mn, mx, p)
    assert np.isfinite(p.numpy()[:, :2]).sum() == 8 and not np.isnan(p.numpy()).any()
    a, b, mv = oracle_ops.ranges_unpack(p)
    np.testing.assert_array_equal(a.numpy(), mn)
    np.testing.assert_array_equal(b.numpy(), mx)
    np.testing.assert_array_equal(mv.numpy(), oracle.c_absmax(mn, mx))
