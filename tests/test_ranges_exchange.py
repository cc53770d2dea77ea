"""Round-3 additions to the min/max family on the GPU: the packed-ranges operand of the calibration all-reduce (written
by the min/max kernels themselves, every kernel path), its unpack kernel, and the workspace status / debug checks."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

import oracle
import oracle_ops

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _expect_packed(mn, mx):
    p = oracle_ops.new_packed(mn.size, "cpu")
    oracle_ops.pack_ranges(mn, mx, p)
    return p.numpy()


# (C, inner, per_channel): one shape per kernel that ends in fold_store -- k_minmax_partial with and without a
# reducer block, k_rows_reg K2 mode, k_rows_staged_mm, k_rows_direct<2>
SHAPES = [(1, 3 * 1000 * 1000 + 5, False), (1, 777, False), (6, 1 << 20, True), (300, 1024, True), (4096, 147, True),
          (513, 1000, True), (70000, 3, True)]


@pytest.mark.parametrize("C,inner,pc", SHAPES)
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_minmax_packed_matches_the_folded_estimate(C, inner, pc, mode):
    from fp8q import ops
    rng = np.random.RandomState(C + inner + mode)
    n_rows = C if pc else 1
    a = (rng.randn(C, inner) * 0.7).astype(np.float32)
    b = (rng.randn(C, inner) * 1.3 + 0.2).astype(np.float32)
    if n_rows > 2:
        b[1, inner // 2] = np.nan                       # a NaN row: flag set, value -inf
    xa, xb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    mn, mx = ops.minmax(xa, pc)
    packed = ops.new_packed(n_rows, xa.device)
    mn, mx, mv = ops.minmax(xb, pc, mn, mx, mode=mode, momentum=0.9, want_maxval=True, packed=packed)
    rmn, rmx = oracle.c_minmax(a, pc)
    bmn, bmx = oracle.c_minmax(b, pc)
    if mode:
        rmn, rmx = oracle.c_fold(rmn, rmx, bmn, bmx, mode, 0.9)
    else:
        rmn, rmx = bmn, bmx
    got_mn, got_mx = mn.cpu().numpy(), mx.cpu().numpy()
    for got_v, ref_v in ((got_mn, rmn), (got_mx, rmx)):
        assert np.array_equal(np.isnan(got_v), np.isnan(ref_v))
        assert np.array_equal(got_v[~np.isnan(ref_v)].view(np.int32), ref_v[~np.isnan(ref_v)].view(np.int32))
    want = _expect_packed(got_mn, got_mx)
    got = packed.cpu().numpy()
    assert not np.isnan(got).any()                      # nothing the collective could trip over
    np.testing.assert_array_equal(got, want)
    # unpack: back to exactly the folded estimate (+ K5)
    umn, umx, umv = ops.ranges_unpack(packed)
    for u, ref in ((umn, got_mn), (umx, got_mx), (umv, mv.cpu().numpy())):
        u = u.cpu().numpy()
        assert np.array_equal(np.isnan(u), np.isnan(ref)) and np.array_equal(u[~np.isnan(ref)], ref[~np.isnan(ref)])


def test_packed_exchange_equals_tensor_op_exchange():
    """MAX over two 'ranks' of the packed records, unpacked, == min / max over the ranks with NaN winning (what
    fp8q.dist.allreduce_ranges computes with ~8 tensor ops)."""
    from fp8q import ops
    mins = [np.array([-1.0, np.nan, 0.5, -0.0], np.float32), np.array([-2.0, -3.0, 0.25, 0.0], np.float32)]
    maxs = [np.array([2.0, 4.0, np.nan, 0.0], np.float32), np.array([3.0, 1.0, 1.0, -0.0], np.float32)]
    packs = []
    for mn, mx in zip(mins, maxs):
        p = oracle_ops.new_packed(4, "cpu")
        oracle_ops.pack_ranges(mn, mx, p)
        packs.append(p)
    red = torch.maximum(packs[0], packs[1]).cuda()
    mn, mx, mv = (t.cpu().numpy() for t in ops.ranges_unpack(red))
    assert mn[0] == -2.0 and mx[0] == 3.0 and mv[0] == 3.0
    assert np.isnan(mn[1]) and mx[1] == 4.0 and np.isnan(mv[1])
    assert mn[2] == 0.25 and np.isnan(mx[2]) and np.isnan(mv[2])
    assert mn[3] == 0.0 and mx[3] == 0.0 and mv[3] == 0.0


def test_affine_act_minmax_packed():
    from fp8q import ops
    rng = np.random.RandomState(5)
    x = torch.from_numpy(rng.randn(8, 16, 28, 28).astype(np.float32)).cuda()
    bn = tuple(torch.from_numpy(v.astype(np.float32)).cuda() for v in
               (rng.randn(16) * 0.1, 1 / np.sqrt(rng.rand(16) + 0.5), rng.rand(16) + 0.5, rng.randn(16) * 0.1))
    packed = ops.new_packed(1, x.device)
    mn, mx, mv = ops.affine_act_minmax(x, bn=bn, act=1, packed=packed)
    mn2, mx2, mv2 = ops.affine_act_minmax(x, bn=bn, act=1)
    assert torch.equal(mn, mn2) and torch.equal(mx, mx2) and torch.equal(mv, mv2)
    np.testing.assert_array_equal(packed.cpu().numpy(), _expect_packed(mn.cpu().numpy(), mx.cpu().numpy()))


def test_workspace_check_reports_timeouts_and_dirty_granules():
    import fp8q
    from fp8q import ops
    L = fp8q.lib()
    x = torch.randn(1 << 22, device="cuda")
    ops.minmax(x, False)
    ops.check_workspaces()                                         # clean after a normal call
    ws = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert L.fp8q_minmax_workspace_check(ws.data_ptr(), ws.numel(), 0, st) == 0
    ws.view(torch.int32)[0] = 3                                    # what a reducer leaves behind when it gives up
    rc = L.fp8q_minmax_workspace_check(ws.data_ptr(), ws.numel(), 0, st)
    assert rc == -6 and b"timed out" in L.fp8q_strerror(rc)
    assert L.fp8q_minmax_workspace_check(ws.data_ptr(), ws.numel(), 1, st) == -6      # reported once more, then cleared
    assert L.fp8q_minmax_workspace_check(ws.data_ptr(), ws.numel(), 0, st) == 0
    ws.view(torch.int64)[5] = 12345                                # a granule someone scribbled over
    assert L.fp8q_minmax_workspace_check(ws.data_ptr(), ws.numel(), 1, st) == -3
    assert L.fp8q_minmax_workspace_check(ws.data_ptr(), ws.numel(), 0, st) == 0
    assert int(ws.view(torch.int64).abs().sum()) == 0


def _sub(code, **env):
    e = dict(os.environ, **env)
    e["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "fp8-quantization_amd"), e.get("PYTHONPATH", "")])
    return subprocess.run([sys.executable, "-c", textwrap.dedent(code)], capture_output=True, text=True, timeout=600, env=e)


def test_reducer_timeout_is_visible():
    """A streaming block that never publishes (fault injection) makes the reducer give up: that call's range is NaN --
    and the workspace remembers it, so the next synchronising check (QuantizedModel.fix_ranges) raises."""
    r = _sub("""
        import torch, fp8q
        from fp8q import ops
        x = torch.randn(1 << 24, device="cuda")
        mn, mx = ops.minmax(x, False)
        torch.cuda.synchronize()
        assert torch.isnan(mn).all() and torch.isnan(mx).all(), (mn, mx)
        try:
            ops.check_workspaces()
        except fp8q.Fp8qError as e:
            assert "timed out" in str(e), e
            print("TIMEOUT_REPORTED")
        ops.check_workspaces()          # cleared by the failing check: usable again
        print("CLEAN_AGAIN")
    """, FP8Q_TEST_FAULT="drop_publish", FP8Q_K3_SPIN_LIMIT="20000")
    assert r.returncode == 0 and "TIMEOUT_REPORTED" in r.stdout and "CLEAN_AGAIN" in r.stdout, r.stdout + r.stderr


def test_debug_ws_refuses_a_dirty_workspace():
    r = _sub("""
        import torch, fp8q
        from fp8q import ops
        x = torch.randn(1 << 22, device="cuda")
        a = ops.minmax(x, False)
        ws = [w for (d, s, z, _k), w in ops._ws_cache.items() if z][0]
        ws.view(torch.int64)[7] = 99            # violate the contract
        try:
            ops.minmax(x, False)
        except fp8q.Fp8qError as e:
            assert "workspace" in str(e), e
            print("REFUSED")
        ws.zero_()
        b = ops.minmax(x, False)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        print("OK_AFTER_CLEAR")
    """, FP8Q_DEBUG_WS="1")
    assert r.returncode == 0 and "REFUSED" in r.stdout and "OK_AFTER_CLEAR" in r.stdout, r.stdout + r.stderr


def test_rccl_runs_every_collective_of_the_path_single_rank():
    """The one-GPU box cannot host two RCCL ranks, but it can run every collective this repo issues on the "nccl"
    backend with one rank (FP8Q_DIST_FORCE=1 takes the multi-rank code path): the packed range all-reduce of config 5
    and of the estimators, the fp32 / code all-gathers of the channel-sharded weights, the bucketed exchange with
    asynchronous handles, both MSE exchanges.  Results must equal the plain single-process ones bit for bit."""
    r = _sub("""
        import os, torch, torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300))
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        assert dist.get_backend() == "nccl"
        import fp8q
        from fp8q import dist as fd, ops
        g = torch.Generator(device=dev).manual_seed(3)
        x1 = torch.randn(32, 512, 512, device=dev, generator=g)
        x2 = torch.randn(32, 512, 512, device=dev, generator=g) * 1.5
        os.environ["FP8Q_DIST_FORCE"] = "0"
        ref, st = [], None
        for x in (x1, x2):
            y, st = fd.calibrate_quantize_sharded(x, 3, 8, 1, state=st)
            ref.append(y.clone())
        ref_state = (st[0].clone(), st[1].clone())
        w = torch.randn(130, 3, 7, 7, device=dev, generator=g) * 0.1
        rq, rmv = fd.quantize_weight_sharded(w, 2, 8, 1)
        ws = [torch.randn(*s, device=dev, generator=g) * 0.05 for s in ((64, 3, 7, 7), (128, 64, 3, 3), (10, 128))]
        rb = fd.quantize_weights_sharded_bucketed(ws, 2, 8, 1)
        rm = fd.mse_search_sharded(x1[:2], False, [2.0, 3.0], 8, 1, "batch", None)
        rmc = fd.mse_search_sharded(w, True, [2.0, 3.0], 8, 1, "channel", None)
        os.environ["FP8Q_DIST_FORCE"] = "1"            # same calls, every collective now runs on RCCL
        st = None
        for x, want in zip((x1, x2), ref):
            y, st = fd.calibrate_quantize_sharded(x, 3, 8, 1, state=st)
            assert torch.equal(y.view(torch.int32), want.view(torch.int32))
        assert torch.equal(st[0], ref_state[0]) and torch.equal(st[1], ref_state[1])
        q, mv = fd.quantize_weight_sharded(w, 2, 8, 1)
        assert torch.equal(q.view(torch.int32), rq.view(torch.int32)) and torch.equal(mv, rmv)
        qc, mvc, codes = fd.quantize_weight_sharded_codes(w, 2, 8, 1)
        assert torch.equal(qc.view(torch.int32), rq.view(torch.int32)) and torch.equal(mvc, rmv) and codes.dtype == torch.uint8
        # the bench's N > 1 headline: FIXED ranges, only the 1-byte codes travel (one uint8 all-gather on RCCL)
        qf, mvf, cf = fd.quantize_weight_sharded_codes(w, 2, 8, 1, maxval=rmv)
        assert torch.equal(qf.view(torch.int32), rq.view(torch.int32)) and mvf is rmv and torch.equal(cf, codes)
        for bb in (None, 1 << 16):
            for (a, b), (c, d) in zip(fd.quantize_weights_sharded_bucketed(ws, 2, 8, 1, bucket_bytes=bb), rb):
                assert torch.equal(a.view(torch.int32), c.view(torch.int32)) and torch.equal(b, d)
        m = fd.mse_search_sharded(x1[:2], False, [2.0, 3.0], 8, 1, "batch", None)
        assert m[1] == rm[1] and torch.allclose(m[0], rm[0], rtol=1e-6)
        mc = fd.mse_search_sharded(w, True, [2.0, 3.0], 8, 1, "channel", None)
        assert mc[1] == rmc[1] and torch.equal(mc[0], rmc[0])
        # the estimators' per-batch exchange (enable_distributed_calibration) on RCCL
        from quantization.range_estimators import RangeEstimators
        e1 = RangeEstimators.allminmax.cls(per_channel=False)
        e2 = RangeEstimators.allminmax.cls(per_channel=False)
        e2.dist_group = True
        for x in (x1, x2):
            a = e1(x); b = e2(x)
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(e1.last_maxval, e2.last_maxval)
        torch.cuda.synchronize()
        dist.destroy_process_group()
        print("RCCL_PATH_OK")
    """, HSA_ENABLE_IPC_MODE_LEGACY="0")
    assert r.returncode == 0 and "RCCL_PATH_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]
