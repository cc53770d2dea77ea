"""Sync-free MSE calibration: device search grid, device winner selection, K1 with a device-resident mantissa width --
each against the torch ops the reference uses, and FP_MSE_Estimator.forward issuing no host synchronisation."""
import numpy as np
import pytest
import torch

from oracle import torch_eager as te

pytestmark = pytest.mark.gpu


def test_linspace_kernel_equals_torch_linspace():
    import fp8q
    rng = np.random.RandomState(0)
    mx = np.concatenate([np.abs(rng.standard_normal(3000)) * 10 ** rng.uniform(-6, 5, 3000), [0.0, 1.0, 2.0e38, 1e-38, 6e4]])
    mx = mx.astype(np.float32)
    got = fp8q.ops.mse_linspace(torch.from_numpy(mx).cuda(), 111).cpu()
    with np.errstate(all="ignore"):
        want = torch.stack([torch.linspace(0.1 * float(v), 1.2 * float(v), 111) for v in mx.tolist()], 1)
    assert got.shape == (111, mx.size)
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    got = fp8q.ops.mse_linspace(torch.from_numpy(mx[:50]).cuda(), 12, 0.25, 1.0).cpu()
    want = torch.stack([torch.linspace(0.25 * float(v), 1.0 * float(v), 12) for v in mx[:50].tolist()], 1)
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))


@pytest.mark.parametrize("shape,per_channel", [((64, 3, 7, 7), True), ((1280, 320), True), ((32, 1, 3, 3), True), ((512, 4608), True),
                                               ((5, 70000), True), ((8, 16, 28, 28), False), ((64, 32, 56, 56), False), ((3,), False)])
def test_minmax_linspace_in_one_launch(shape, per_channel):
    """the first calibration batch's abs-max launch also writes the search grid (fp8q_minmax_linspace_f32): every min/max
    route (rows in registers, staged, direct, split rows with the reducer block) against minmax + torch.linspace"""
    import fp8q
    g = torch.Generator().manual_seed(len(shape) + shape[0])
    x = torch.randn(*shape, generator=g) * 3.0
    mn, mx, mv, grid = fp8q.ops.minmax_linspace(x.cuda(), per_channel, 111)
    rmn, rmx, rmv = fp8q.ops.minmax(x.cuda(), per_channel, want_maxval=True)
    for a, b in ((mn, rmn), (mx, rmx), (mv, rmv)):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    want = torch.stack([torch.linspace(0.1 * float(v), 1.2 * float(v), 111) for v in rmv.cpu().tolist()], 1)
    assert grid.shape == want.shape and torch.equal(grid.cpu().view(torch.int32), want.view(torch.int32))


def test_linspace_falls_back_to_the_host_when_the_kernel_formula_does_not_match(monkeypatch):
    """another ATen build could evaluate torch.linspace differently: then the grids come from torch.linspace on the host
    (a cached decision, a warning) instead of an exception that would stop every MSE calibration"""
    import fp8q
    ops = fp8q.ops
    monkeypatch.setattr(ops, "_linspace_checked", {(111, 0.1, 1.2): False})
    mx = torch.tensor([0.5, 2.0, 3.3e-5], device="cuda")
    got = ops.mse_linspace(mx, 111)
    want = torch.stack([torch.linspace(0.1 * float(v), 1.2 * float(v), 111) for v in mx.cpu().tolist()], 1)
    assert got.is_cuda and torch.equal(got.cpu().view(torch.int32), want.view(torch.int32))
    x = torch.randn(6, 40, device="cuda")
    mn, mxx, mv, grid = ops.minmax_linspace(x, True, 111)
    want = torch.stack([torch.linspace(0.1 * float(v), 1.2 * float(v), 111) for v in x.abs().amax(1).cpu().tolist()], 1)
    assert torch.equal(grid.cpu().view(torch.int32), want.view(torch.int32))


@pytest.mark.parametrize("C,n_m,n_cand", [(1, 6, 111), (1, 1, 111), (32, 6, 111), (1280, 6, 111), (5, 3, 7), (2049, 2, 130), (16, 2, 9),
                                           (17, 2, 4), (64, 1, 111), (65, 6, 5)])
def test_select_kernel_equals_torch_ops(C, n_m, n_cand):
    import fp8q
    rng = np.random.RandomState(C + n_m)
    mbits = [float(m) for m in range(1, n_m + 1)]
    for trial in range(4):
        mses = rng.uniform(0.1, 1.0, (n_m, n_cand, C)).astype(np.float32)
        if trial == 1:      # ties everywhere: coarse values (first index must win; mode tie -> smallest)
            mses = np.round(mses * 4) / 4
        if trial == 2 and C > 1:      # NaNs: a NaN counts as the minimum, the first one wins
            mses[rng.randint(n_m), rng.randint(n_cand), rng.randint(C)] = np.nan
            mses[0, 3, 0] = np.nan
        if trial == 3:      # exactly tied vote between two widths
            mses[:] = 1.0
            for c in range(C):
                mses[(c % 2) * (n_m - 1), c % n_cand, c] = 0.5
        grid = rng.uniform(0.1, 5.0, (n_cand, C)).astype(np.float32)
        mb, vote, maxval, xmin = fp8q.ops.mse_select(torch.from_numpy(mses).cuda(), torch.from_numpy(grid).cuda(), mbits, 1)
        r_mb, r_maxval, r_idx = te.mse_select(torch.from_numpy(mses), torch.from_numpy(grid), mbits)
        assert float(mb) == r_mb and mbits[int(vote)] == r_mb, (C, n_m, trial)
        assert torch.equal(maxval.cpu(), r_maxval) and torch.equal(xmin.cpu(), -r_maxval)
    # unsigned: xmin = 0 * -1.0 * maxval = -0.0
    _, _, maxval, xmin = fp8q.ops.mse_select(torch.from_numpy(mses).cuda(), torch.from_numpy(grid).cuda(), mbits, 0)
    assert torch.equal(xmin.cpu().view(torch.int32), (0 * -1.0 * maxval.cpu()).view(torch.int32))


@pytest.mark.parametrize("shape,per_channel", [((64, 32, 14, 14), False), ((1000, 147), True), ((37, 1, 3, 3), True),
                                               ((5, 4099), True), ((3, 1001), False), ((70000, 9), True)])
def test_quantize_with_device_mantissa_bits(shape, per_channel):
    import fp8q
    torch.manual_seed(1)
    x = torch.randn(shape, device="cuda")
    mv = (torch.rand(shape[0], device="cuda") * 3 + 0.1) if per_channel else torch.tensor([2.3], device="cuda")
    for sb in (1, 0):
        for m in (1.0, 2.0, 3.0, 4.4, 5.5, 7.0, 9.0):
            want = fp8q.ops.quantize(x, mv, m, 8, sb)
            got = fp8q.ops.quantize(x, mv, torch.tensor([m], device="cuda"), 8, sb)
            assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (shape, m, sb)
    xo = x.reshape(-1)[1:]                   # not 16-byte aligned
    want = fp8q.ops.quantize(xo, mv[:1], 3.0)
    assert torch.equal(fp8q.ops.quantize(xo, mv[:1], torch.tensor([3.0], device="cuda")), want)


class _NoSync:
    """torch.cuda.set_sync_debug_mode("error"): any host synchronisation inside the block raises."""

    def __enter__(self):
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")

    def __exit__(self, *exc):
        torch.cuda.set_sync_debug_mode("default")
        return False


@pytest.mark.parametrize("per_channel", [False, True])
@pytest.mark.parametrize("search", [True, False])
def test_mse_estimator_forward_is_sync_free(per_channel, search):
    from quantization.quantizers.fp8_quantizer import FPQuantizer
    from quantization.range_estimators import FP_MSE_Estimator
    from quantization.quantization_manager import QuantizationManager
    from quantization.manager import Qstates
    import fp8q
    torch.manual_seed(2)
    shape = (48, 3, 5, 5) if per_channel else (8, 16, 28, 28)
    batches = [torch.randn(shape, device="cuda") * (1 + i) for i in range(3)]
    fp8q.ops.mse_linspace(torch.ones(1, device="cuda"))       # the once-per-process self-check synchronises
    mgr = QuantizationManager(qmethod=FPQuantizer, init=FP_MSE_Estimator, per_channel=per_channel,
                              qparams=dict(n_bits=8, mantissa_bits=3, set_maxval=True, mse_include_mantissa_bits=search))
    with _NoSync():                          # even the FIRST batch: grid, search, vote, quantize -- no host round trip
        outs = [mgr(b) for b in batches]
    # the same three batches through the reference's torch op sequence for the selection (with host syncs)
    est = mgr.range_estimator
    mbits_list = [float(m) for m in range(1, 7)] if search else [3.0]
    r_mb, r_maxval, _ = te.mse_select(est.mses.cpu(), est.search_grid.cpu(), mbits_list)
    q = mgr.quantizer
    assert (q._pending_mantissa_bits() is not None) == search
    assert float(q.mantissa_bits) == r_mb            # (this read brings the pending value over)
    assert q._pending_mantissa_bits() is None
    assert torch.equal(q.maxval.cpu(), r_maxval)
    want = fp8q.ops.quantize(batches[-1], q.maxval, r_mb, 8, 1)
    assert torch.equal(outs[-1].view(torch.int32), want.view(torch.int32))
    mgr.fix_ranges()
    with _NoSync():
        y = mgr(batches[0])
    assert torch.equal(y, fp8q.ops.quantize(batches[0], q.maxval, r_mb, 8, 1))


def test_sync_debug_mode_catches_a_host_round_trip():
    """the guard used above is live on this build: an .item() inside it raises"""
    t = torch.ones(1, device="cuda")
    with pytest.raises(RuntimeError):
        with _NoSync():
            t.item()


def test_model_fix_ranges_materializes_all_votes_once():
    from quantization.model import materialize_mantissa_bits
    from quantization.quantizers.fp8_quantizer import FPQuantizer
    net = torch.nn.ModuleList([FPQuantizer(n_bits=8, mantissa_bits=3) for _ in range(5)])
    for i, q in enumerate(net):
        if i != 2:
            q.mantissa_bits = torch.tensor([float(i + 1)], device="cuda")
    assert materialize_mantissa_bits(net) == 4
    assert materialize_mantissa_bits(net) == 0
    with _NoSync():
        assert [float(q.mantissa_bits) for q in net] == [1.0, 2.0, 3.0, 4.0, 5.0]
