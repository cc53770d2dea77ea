"""GPU parity: HIP kernels (through the C ABI) vs the CPU oracle and the golden fixtures.

Bar: K1 and the min/max estimators are BIT-EXACT against oracle/fp8q_oracle.c (same arithmetic
contract); against the reference's own output (golden fixtures) they meet the north_star
tolerance (<= 1 step of the emulated FP8 grid; measured: <= 2 fp32 ULP, no tie flips).
"""
import os

import numpy as np
import pytest
import torch

import oracle
from parity import assert_parity, compare, elem_step

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import fp8q
    fp8q.lib()  # raises if libfp8q_hip.so is missing: no fallback
    return fp8q.ops


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.int32)


def assert_bit_exact(y, y_ref, what=""):
    y, y_ref = np.asarray(y, np.float32), np.asarray(y_ref, np.float32)
    nan_a, nan_b = np.isnan(y), np.isnan(y_ref)
    assert np.array_equal(nan_a, nan_b), f"{what}: NaN pattern"
    bad = (bits(y) != bits(y_ref)) & ~nan_a
    assert not bad.any(), f"{what}: {bad.sum()} / {y.size} elements differ, first at {np.argwhere(bad)[:5].tolist()}"


def test_library_loaded(ops):
    import fp8q
    assert fp8q.lib().fp8q_version() == 601
    assert os.path.exists(fp8q.so_path())


def test_bulk_flip_rate_vs_reference(ops, golden_dir):
    """HIP vs the REFERENCE on bulk data (g1b: 4 x 4 M seeded normals, {E5M2, E4M3} x {per-tensor, per-channel}):
    <= 1 grid step everywhere, no more than 1e-5 of the elements a grid step off, the rest within 2 fp32 ULP -- and
    bit-exact against the C oracle.  The reference's output is rebuilt from the fixture (hash-checked)."""
    from test_oracle_golden import bulk_cases, bulk_input, reference_from_oracle
    g = np.load(os.path.join(golden_dir, "g1b_bulk.npz"))
    for name, seed, M, shape, mv in bulk_cases():
        x = bulk_input(seed, shape)
        y_orc = oracle.c_quantize(x, mv, M, 8, 1)
        y = ops.quantize(dev(x), dev(mv), M, 8, 1).cpu().numpy()
        assert_bit_exact(y, y_orc, f"{name} vs oracle")
        y_ref, _, _ = reference_from_oracle(g, name, y_orc)
        r = assert_parity(y, y_ref, elem_step(x, mv, M, 8, 1), max_flip_frac=1e-5, max_ulp=2, what=f"{name} vs reference")
        if name.endswith("channel"):       # the fused estimate-state launch on the same rows, ranges from the data
            yf, mn, mx, mvf = ops.minmax_quantize(dev(x), M, 8, 1)
            rmn, rmx = oracle.c_minmax(x, True)
            rmv = oracle.c_absmax(rmn, rmx)
            assert_bit_exact(mvf.cpu().numpy(), rmv, f"{name} fused maxval")
            assert_bit_exact(yf.cpu().numpy(), oracle.c_quantize(x, rmv, M, 8, 1), f"{name} fused vs oracle")
        print(f"\n{name}: HIP vs reference bit-exact {r['exact_frac']:.4%}, flips {r['n_flips']}, max ULP {r['max_ulp_nonflip']}")


def test_quantize_golden_and_oracle(ops, golden_dir):
    g1 = np.load(os.path.join(golden_dir, "g1_quantize.npz"))
    tot = exact = 0
    for cid, mbits, mv, sb, nmv in g1["cases"]:
        cid, sb = int(cid), int(sb)
        x, y_ref = g1[f"c{cid}_x"], g1[f"c{cid}_y"]
        maxval = g1[f"c{cid}_maxval"] if mv < 0 else np.array([mv], np.float32)
        y = ops.quantize(dev(x), dev(maxval), float(mbits), 8, sb).cpu().numpy()
        assert_bit_exact(y, oracle.c_quantize(x, maxval, float(mbits), 8, sb), f"case {cid} vs oracle")
        r = assert_parity(y, y_ref, elem_step(x, maxval, float(mbits), 8, sb), max_flip_frac=0.0,
                          max_ulp=2, what=f"case {cid} vs reference")
        tot += r["n"]
        exact += r["exact_frac"] * r["n"]
    print(f"\nHIP vs reference golden: {tot} elems, bit-exact {exact / tot:.4%}")
    assert exact / tot > 0.90


@pytest.mark.parametrize("M,sb,mv", [(2, 1, 57344.0), (3, 1, 240.0), (3, 1, 0.7361), (5, 1, 3.0),
                                     (1, 1, 1.0), (6, 1, 0.0123), (3, 0, 2.5), (1, 0, 0.31), (7, 1, 1.0),
                                     (8, 0, 1.0)])
@pytest.mark.parametrize("n", [1, 3, 4, 1023, 4096 + 5, 1 << 20])
def test_quantize_per_tensor_bit_exact(ops, M, sb, mv, n):
    rng = np.random.RandomState(n % 1000 + M)
    x = (rng.randn(n) * mv / 2.5).astype(np.float32)
    if n > 16:
        x[:8] = [0.0, -0.0, np.inf, -np.inf, np.nan, mv, -mv, 1e-41]
        # exact powers of two and ties sit on the p / rounding boundaries
        x[8:16] = mv * 2.0 ** -np.arange(8)
    y = ops.quantize(dev(x), dev([mv]), M, 8, sb).cpu().numpy()
    assert_bit_exact(y, oracle.c_quantize(x, [mv], M, 8, sb), f"M={M} sb={sb} mv={mv} n={n}")


@pytest.mark.parametrize("shape", [(64, 3, 7, 7), (32, 1, 3, 3), (7, 5), (1000, 512), (16, 2048),
                                   (3, 4099), (512, 512, 3, 3), (5, 1), (300, 66), (4, 70000)])
@pytest.mark.parametrize("M,sb", [(2, 1), (3, 1), (5, 1), (1, 0)])
def test_quantize_per_channel_bit_exact(ops, shape, M, sb):
    rng = np.random.RandomState(sum(shape) + M)
    C = shape[0]
    mv = (np.abs(rng.randn(C)) + 0.05).astype(np.float32)
    x = (rng.randn(*shape) * (mv.reshape([-1] + [1] * (len(shape) - 1)) / 2)).astype(np.float32)
    y = ops.quantize(dev(x), dev(mv), M, 8, sb).cpu().numpy()
    assert_bit_exact(y, oracle.c_quantize(x, mv, M, 8, sb), f"shape={shape} M={M}")


def test_quantize_misaligned_views(ops):
    rng = np.random.RandomState(5)
    base = dev(rng.randn(10007))
    for off in (1, 2, 3):
        x = base[off:off + 9001]
        y = ops.quantize(x, dev([1.3]), 3, 8, 1).cpu().numpy()
        assert_bit_exact(y, oracle.c_quantize(x.cpu().numpy(), [1.3], 3, 8, 1), f"offset {off}")
    # per-channel rows whose starts are not 16-byte aligned
    xw = base[1:1 + 4 * 2049].view(4, 2049)
    mv = np.array([0.5, 1.0, 2.0, 3.0], np.float32)
    y = ops.quantize(xw, dev(mv), 2, 8, 1).cpu().numpy()
    assert_bit_exact(y, oracle.c_quantize(xw.cpu().numpy(), mv, 2, 8, 1), "rows offset")


def test_quantize_degenerate_maxval(ops):
    x = np.random.RandomState(0).randn(3, 64).astype(np.float32)
    x[1] = 0
    mv = np.array([1.0, 0.0, np.inf], np.float32)
    y = ops.quantize(dev(x), dev(mv), 3, 8, 1).cpu().numpy()
    ref = oracle.c_quantize(x, mv, 3, 8, 1)
    assert_bit_exact(y, ref, "degenerate")
    assert np.isnan(y[1]).all() and np.isnan(y[2]).all()  # reference quirk: all-zero channel -> NaN


def test_quantize_empty_and_errors(ops):
    import fp8q
    y = ops.quantize(torch.empty(0, device="cuda"), dev([1.0]), 3)
    assert y.numel() == 0
    with pytest.raises(fp8q.Fp8qError):
        ops.quantize(torch.zeros(4), dev([1.0]), 3)          # CPU tensor: no fallback
    with pytest.raises(fp8q.Fp8qError):
        ops.quantize(dev(np.zeros((4, 4))), dev([1.0, 2.0]), 3)  # maxval neither 1 nor C
    with pytest.raises(fp8q.Fp8qError):
        ops.quantize(dev(np.zeros(4)), dev([1.0]), 3, n_bits=16, sign_bits=1)  # E > 7


def test_out_and_device_arguments_are_validated(ops):
    """A raw pointer goes to the kernel: an `out=` of the wrong dtype / size / layout / device, or a range vector on another
    device, must be refused before the launch (Fp8qError), never written through."""
    import fp8q
    x = torch.randn(8, 3, 7, 7, device="cuda")
    mv1, mvc = dev([1.5]), dev(np.full(8, 1.5))
    bad_outs = [torch.empty(8, 3, 7, 7, dtype=torch.float64, device="cuda"),      # dtype
                torch.empty(8, 3, 7, 6, device="cuda"),                             # too small
                torch.empty(8, 3, 7, 14, device="cuda")[..., ::2],                  # not contiguous
                torch.empty(8, 3, 7, 7)]                                            # CPU
    calls = [lambda o: ops.quantize(x, mv1, 3, out=o), lambda o: ops.quantize(x, mvc, 3, out=o),
             lambda o: ops.minmax_quantize(x, 2, out=o), lambda o: ops.copy(x, out=o),
             lambda o: ops.affine_act_quantize(x, mv1, 3, act=1, out=o),
             lambda o: ops.multi_quantize([(x, mvc, 2, 8, 1, o)]),
             lambda o: ops.decode(ops.encode(x, mvc, 3), mvc, 3, out=o)]
    for call in calls:
        for o in bad_outs:
            with pytest.raises(fp8q.Fp8qError):
                call(o)
    with pytest.raises(fp8q.Fp8qError):
        ops.encode(x, mvc, 3, out=torch.empty(8, 3, 7, 7, device="cuda"))           # codes must be uint8
    with pytest.raises(fp8q.Fp8qError):
        ops.encode(x, mvc, 3, out=torch.empty(8 * 147 - 1, dtype=torch.uint8, device="cuda"))
    with pytest.raises(fp8q.Fp8qError):
        ops.quantize(x, torch.tensor([1.5]), 3)                                     # maxval on the CPU
    with pytest.raises(fp8q.Fp8qError):
        ops.affine_act_quantize(x, mvc, 3)                                          # per-channel maxval into the per-tensor epilogue
    with pytest.raises(fp8q.Fp8qError):
        ops.affine_act_quantize(x, mv1, 3, residual=torch.zeros(8, 3, 7, 6, device="cuda"))
    with pytest.raises(fp8q.Fp8qError):
        ops.minmax(x, True, torch.zeros(8), torch.zeros(8))                         # running estimate on the CPU
    with pytest.raises(fp8q.Fp8qError):
        ops.mse_grid(x, True, torch.ones(4, 8, device="cuda"), [3.0], 8, 1, torch.zeros(1, 4, 8))   # table on the CPU
    if torch.cuda.device_count() > 1:
        with pytest.raises(fp8q.Fp8qError):
            ops.quantize(x, mv1.to("cuda:1"), 3)
    # a good `out=` of another SHAPE with the same element count is fine (the kernel sees [C, inner] either way)
    o = torch.empty(8 * 147, device="cuda")
    assert torch.equal(ops.quantize(x, mv1, 3, out=o).view_as(x), ops.quantize(x, mv1, 3))


def test_per_tensor_ops_run_on_dense_layouts_without_a_copy(ops, monkeypatch):
    """channels-last activations / transposed matrices: per-tensor K1, min/max and the MSE table are layout-agnostic, so
    they run on the storage as it lies (no .contiguous() copy: 8 B per element) and K1's result keeps the input's strides,
    as the reference's elementwise ATen ops do"""
    torch.manual_seed(3)
    x = torch.randn(6, 16, 9, 11, device="cuda").contiguous(memory_format=torch.channels_last)
    mv = torch.tensor([1.7], device="cuda")
    calls = []
    orig = torch.Tensor.contiguous
    monkeypatch.setattr(torch.Tensor, "contiguous", lambda self, *a, **k: (calls.append(tuple(self.shape)), orig(self, *a, **k))[1])
    y = ops.quantize(x, mv, 3, 8, 1)
    mn, mx = ops.minmax(x, False)
    grid = torch.linspace(0.2, 3.0, 111, device="cuda").reshape(111, 1)
    mses = ops.mse_grid(x, False, grid, [3.0], 8, 1, torch.zeros(1, 111, 1, device="cuda"))
    monkeypatch.undo()
    assert (6, 16, 9, 11) not in calls, calls                              # the 4-D tensor was never copied
    assert y.stride() == x.stride() and y.is_contiguous(memory_format=torch.channels_last)
    xc = x.contiguous()
    assert torch.equal(y, ops.quantize(xc, mv, 3, 8, 1))                   # element for element the same values
    rmn, rmx = ops.minmax(xc, False)
    assert torch.equal(mn, rmn) and torch.equal(mx, rmx)
    want = ops.mse_grid(xc, False, grid, [3.0], 8, 1, torch.zeros(1, 111, 1, device="cuda"))
    torch.testing.assert_close(mses, want, rtol=1e-6, atol=0)              # (another summation order)
    t = torch.randn(300, 70, device="cuda").t()                            # a transposed matrix
    yt = ops.quantize(t, mv, 2, 8, 1)
    assert yt.stride() == t.stride() and torch.equal(yt, ops.quantize(t.contiguous(), mv, 2, 8, 1))
    s = torch.randn(8, 64, device="cuda")[:, ::2]                          # NOT dense: the copy path, contiguous result
    assert torch.equal(ops.quantize(s, mv, 3, 8, 1), ops.quantize(s.contiguous(), mv, 3, 8, 1))
    pc = ops.quantize(x, torch.rand(6, device="cuda") + 0.5, 3, 8, 1)      # per channel: rows are dim 0 -> the NCHW view
    assert pc.is_contiguous()


def test_signed_zero_rows_follow_the_pinned_contract(ops, golden_dir):
    """g3b (rows mixing -0.0 and +0.0): the kernels' min / max equal the oracle's bit for bit on every route (per channel,
    per tensor, running fold, fused min/max + quantize), maxval and the quantized rows equal the reference's"""
    g = np.load(os.path.join(golden_dir, "g3b_signed_zero.npz"))
    x = g["x"]
    xd = torch.from_numpy(x).cuda()
    for pc in (True, False):
        mn, mx, mv = ops.minmax(xd, pc, want_maxval=True)
        rmn, rmx = oracle.c_minmax(x, pc)
        assert np.array_equal(mn.cpu().numpy().view(np.int32), rmn.view(np.int32))
        assert np.array_equal(mx.cpu().numpy().view(np.int32), rmx.view(np.int32))
        assert np.array_equal(mn.cpu().numpy(), g[f"pc{int(pc)}_min"]) and np.array_equal(mx.cpu().numpy(), g[f"pc{int(pc)}_max"])
    mn, mx = ops.minmax(xd, True)
    mn, mx = ops.minmax(torch.from_numpy(x[:, ::-1].copy()).cuda(), True, mn, mx, mode=ops.FOLD_ALL)
    assert np.array_equal(mn.cpu().numpy(), g["all_pc1_min"]) and np.array_equal(mx.cpu().numpy(), g["all_pc1_max"])
    y, rmn, rmx, mv = ops.minmax_quantize(xd, 3, 8, 1)
    assert np.array_equal(mv.cpu().numpy().view(np.int32), g["maxval"].view(np.int32))
    q = y.cpu().numpy()
    assert np.array_equal(np.isnan(q), np.isnan(g["q"]))
    ok = ~np.isnan(q)
    assert np.array_equal(np.signbit(q[ok]), np.signbit(g["q"][ok]))                     # signed zeros kept
    assert np.abs(q[ok].view(np.int32).astype(np.int64) - g["q"][ok].view(np.int32)).max() <= 2   # (log2 / 2^x: <= 2 ulp)


def test_retired_workspace_is_still_checked(ops):
    """check_workspaces() also inspects min/max workspaces that a larger request has replaced since the last check."""
    import fp8q
    ops.check_workspaces()
    small = torch.randn(1 << 16, device="cuda")
    ops.minmax(small, False)
    key = [k for k in ops._ws_cache if k[2]][0]
    old = ops._ws_cache[key]
    old.view(torch.int32)[0] = 3                    # pretend three reducers timed out on it
    ops._ws_cache[key] = torch.zeros(16, dtype=torch.uint8, device="cuda")     # force the next call to outgrow the buffer
    ops._ws_retired.append((key, old))
    ops._ws_cache[key] = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    with pytest.raises(fp8q.Fp8qError, match="timed out"):
        ops.check_workspaces()
    ops.check_workspaces()                          # reported once, cleared, forgotten
    assert not ops._ws_retired


def test_spin_waiting_workgroups_make_progress_next_to_a_saturating_stream(ops):
    """The single-launch min/max (its reducer workgroup polls the streaming workgroups' granules, fp8q_common.h) and the fused
    interval scan + evaluation of K4 (evaluating workgroups poll the scan workgroups' counter, k_scan_eval) while a SECOND
    stream keeps every CU busy with back-to-back 1 GiB copies: same results as on an idle GPU, no reducer time-out
    (check_workspaces), no NaN table entry (the bounded spins give up with NaN rather than hang)."""
    ops.check_workspaces()
    x = torch.randn(64, 64, 112, 112, device="cuda")
    big = torch.randn(1 << 28, device="cuda")
    bout = torch.empty_like(big)
    a = torch.clamp(torch.randn(64, 24, 56, 56, device="cuda") * 2.5, 0, 6)
    s_busy, s_work = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s_work):
        mn0, mx0 = ops.minmax(x, False)
        cal0 = ops.MseCalibration(1, a.device, [1.0, 2.0, 3.0, 4.0, 5.0, 6.0], 8, 1)
        y0 = cal0.step(a)
    torch.cuda.synchronize()
    results = []
    for it in range(12):
        with torch.cuda.stream(s_busy):
            for _ in range(3):
                ops.copy(big, out=bout)            # 65536-workgroup launches: the CUs stay full
        with torch.cuda.stream(s_work):
            mn, mx = ops.minmax(x, False)
            cal = ops.MseCalibration(1, a.device, [1.0, 2.0, 3.0, 4.0, 5.0, 6.0], 8, 1)
            y = cal.step(a)
            results.append((mn, mx, cal, y))
    torch.cuda.synchronize()
    for mn, mx, cal, y in results:
        assert torch.equal(mn, mn0) and torch.equal(mx, mx0)
        assert torch.isfinite(cal.mses).all()
        assert torch.equal(cal.mses.view(torch.int32), cal0.mses.view(torch.int32))
        assert torch.equal(cal.maxval, cal0.maxval) and torch.equal(cal.mbits, cal0.mbits) and torch.equal(y, y0)
    ops.check_workspaces()                         # raises if any reducer timed out


@pytest.mark.parametrize("shape,pc", [((64, 3, 7, 7), True), ((64, 3, 7, 7), False), ((4, 8, 6, 6), False),
                                      ((1000, 512), True), ((8, 300000), True), ((3, 5), True),
                                      ((64, 64, 56, 56), False), ((2, 2049), True)])
def test_minmax_bit_exact(ops, shape, pc):
    rng = np.random.RandomState(len(shape) + shape[0])
    x = rng.randn(*shape).astype(np.float32)
    mn, mx, mv = ops.minmax(dev(x), pc, want_maxval=True)
    rmn, rmx = oracle.c_minmax(x, pc)
    np.testing.assert_array_equal(mn.cpu().numpy(), rmn)
    np.testing.assert_array_equal(mx.cpu().numpy(), rmx)
    np.testing.assert_array_equal(mv.cpu().numpy(), oracle.c_absmax(rmn, rmx))


def test_minmax_fold_modes_and_nan(ops, golden_dir):
    g3 = np.load(os.path.join(golden_dir, "g3_estimators.npz"))
    for mode, name in ((1, "allminmax"), (2, "running_minmax")):
        for pc in (False, True):
            cur = (None, None)
            for b, a in enumerate(g3["acts"]):
                cur = ops.minmax(dev(a), pc, cur[0], cur[1], mode=mode, momentum=0.9)
                np.testing.assert_array_equal(cur[0].cpu().numpy(), g3[f"{name}_pc{int(pc)}_min"][b])
                np.testing.assert_array_equal(cur[1].cpu().numpy(), g3[f"{name}_pc{int(pc)}_max"][b])
    a = g3["acts"][0].copy()
    a[1, 2, 3, 4] = np.nan
    mn, mx = ops.minmax(dev(a), False)
    assert np.isnan(mn.item()) and np.isnan(mx.item())
    big = np.random.RandomState(1).randn(3, 50000).astype(np.float32)
    big[1, 40000] = np.nan
    mn, mx = ops.minmax(dev(big), True)
    rmn, rmx = oracle.c_minmax(big, True)
    np.testing.assert_array_equal(np.isnan(mn.cpu().numpy()), np.isnan(rmn))
    np.testing.assert_array_equal(mn.cpu().numpy()[[0, 2]], rmn[[0, 2]])


@pytest.mark.parametrize("shape", [(64, 3, 7, 7), (32, 1, 3, 3), (1000, 512), (512, 512, 3, 3), (7, 5),
                                   (3, 4099), (5, 16384), (130, 66), (2048, 3, 7, 7)])
@pytest.mark.parametrize("M", [2, 3])
def test_fused_minmax_quantize_bit_exact(ops, shape, M):
    rng = np.random.RandomState(shape[0] + M)
    x = (rng.randn(*shape) * 0.1).astype(np.float32)
    if shape[0] > 4:
        x[2] = 0  # all-zero channel -> NaN channel
    y, mn, mx, mv = ops.minmax_quantize(dev(x), M, 8, 1)
    rmn, rmx = oracle.c_minmax(x, True)
    rmv = oracle.c_absmax(rmn, rmx)
    np.testing.assert_array_equal(mn.cpu().numpy(), rmn)
    np.testing.assert_array_equal(mx.cpu().numpy(), rmx)
    np.testing.assert_array_equal(mv.cpu().numpy(), rmv)
    assert_bit_exact(y.cpu().numpy(), oracle.c_quantize(x, rmv, M, 8, 1), f"fused {shape}")


@pytest.mark.parametrize("shape", [(64, 147), (16, 1), (33, 64), (50, 65), (100, 129), (64, 256), (20, 300), (32, 512), (1, 512),
                                   (257, 63), (5, 3, 1, 1)])
@pytest.mark.parametrize("M,sign", [(2, 1), (3, 1), (5, 0), (7, 1)])
def test_small_tensor_fused_route_bit_exact(ops, shape, M, sign):
    """tensors of <= 64 KB with rows of <= 512 elements: a wave per row, the row in registers (k_small_rows_fused, BASELINE
    config 2's literal size) -- every row length class (1 .. 8 elements per lane), special values, degenerate rows"""
    rng = np.random.RandomState(shape[0] * 31 + shape[1] + M)
    x = (rng.randn(*shape) * rng.choice([1e-3, 0.1, 3.0, 200.0])).astype(np.float32)
    flat = x.reshape(shape[0], -1)
    if shape[0] > 8:
        flat[1] = 0                        # all-zero row -> NaN row
        flat[2, 0] = np.nan                # a NaN: the row's range is NaN
        flat[3, -1] = np.inf
        flat[4, 0] = -0.0
        flat[5] = np.abs(flat[5])          # one-sided row
        flat[6] = 1e-41                    # denormals only
    y, mn, mx, mv = ops.minmax_quantize(dev(x), M, 8, sign)
    rmn, rmx = oracle.c_minmax(x, True)
    rmv = oracle.c_absmax(rmn, rmx)
    for got, ref, what in ((mn, rmn, "min"), (mx, rmx, "max"), (mv, rmv, "maxval")):
        got = got.cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(got[~np.isnan(ref)].view(np.int32),
                                                                                ref[~np.isnan(ref)].view(np.int32)), what
    assert_bit_exact(y.cpu().numpy(), oracle.c_quantize(x, rmv, M, 8, sign), f"small fused {shape} M={M} sign={sign}")


def test_config2_conv1_vs_reference(ops, golden_dir):
    """BASELINE config 2: conv1 [64,3,7,7] per-channel E5M2, current_minmax, against the reference."""
    g3 = np.load(os.path.join(golden_dir, "g3_estimators.npz"))
    w = g3["w"]
    y, mn, mx, mv = ops.minmax_quantize(dev(w), 2, 8, 1)
    np.testing.assert_array_equal(mn.cpu().numpy(), g3["w_cur_pc_min"])
    np.testing.assert_array_equal(mx.cpu().numpy(), g3["w_cur_pc_max"])
    np.testing.assert_array_equal(mv.cpu().numpy(), g3["w_maxval"])
    r = assert_parity(y.cpu().numpy(), g3["w_q_e5m2"], elem_step(w, g3["w_maxval"], 2), what="conv1")
    print("\nconv1 E5M2 vs reference:", r)


@pytest.mark.parametrize("name,pc,incl,M", [("w_pc_fixm", True, False, 3), ("w_pc_srchm", True, True, 3),
                                            ("a_pt_fixm", False, False, 3), ("a_pt_srchm", False, True, 2)])
def test_mse_grid_vs_oracle_and_golden(ops, golden_dir, name, pc, incl, M):
    g4 = np.load(os.path.join(golden_dir, "g4_mse.npz"))
    mb = [1.0, 2.0, 3.0, 4.0, 5.0, 6.0] if incl else [float(M)]
    grid = g4[f"{name}_grid"]
    C = grid.shape[1]
    mses = torch.zeros(len(mb), grid.shape[0], C, device="cuda")
    ref_o = None
    for b in range(2):
        x = g4[f"{name}_x{b}"]
        ops.mse_grid(dev(x), pc, dev(grid), mb, 8, 1, mses)
        ref_o = oracle.c_mse_grid(x, pc, grid, mb, 8, 1, ref_o)
        got = mses.cpu().numpy()
        np.testing.assert_allclose(got, ref_o, rtol=1e-5, atol=0)   # vs CPU oracle (near-tie flips move a 27-element MSE by ~2e-6)
        np.testing.assert_allclose(got, g4[f"{name}_mses{b}"], rtol=1e-4, atol=0)  # vs reference


def test_mse_grid_large_tensor(ops):
    rng = np.random.RandomState(3)
    x = (rng.randn(1, 300001) * 0.7).astype(np.float32)
    grid = np.linspace(0.1 * 3, 1.2 * 3, 111, dtype=np.float32).reshape(111, 1)
    mses = torch.zeros(2, 111, 1, device="cuda")
    ops.mse_grid(dev(x), False, dev(grid), [2.0, 3.0], 8, 1, mses)
    ref = oracle.c_mse_grid(x, False, grid, [2.0, 3.0], 8, 1)
    np.testing.assert_allclose(mses.cpu().numpy(), ref, rtol=2e-6)
    assert mses.cpu().numpy().argmin(1).tolist() == ref.argmin(1).tolist()


def test_mse_row_kernel_all_paths(ops):
    """k_mse_row (lane = element; rows >= 2048 elements) over ranges that take each of its per-candidate paths: one scale
    mantissa (the usual case), two (ranges whose low binades round k - bias differently), the exact path (tiny / huge /
    degenerate ranges, many exponent bits), unsigned formats, several rows with their own grids, a ragged last tile."""
    rng = np.random.RandomState(11)
    for C, inner, sign in ((1, 70001, 1), (3, 5000, 1), (4, 300, 1), (2, 4099, 0)):   # (4, 300): the lane-per-candidate kernel
        x = (rng.randn(C, inner) * rng.uniform(0.3, 3.0, (C, 1))).astype(np.float32)
        if sign == 0:
            x = np.abs(x)
        base = np.exp(np.linspace(np.log(2e-3), np.log(3e3), 61)).astype(np.float32)
        grid = np.concatenate([base, np.float32([1e-30, 1e-12, 3e37, 0.0, 1.0, 2.0, 4.0])])[:, None] * np.ones((1, C), np.float32)
        grid = (grid * rng.uniform(0.9, 1.1, grid.shape)).astype(np.float32)
        mb = [1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0]
        mses = torch.zeros(len(mb), grid.shape[0], C, device="cuda")
        ops.mse_grid(dev(x), C > 1, dev(grid), mb, 8, sign, mses)
        ref = oracle.c_mse_grid(x if C > 1 else x.reshape(-1), C > 1, grid, mb, 8, sign)
        got = mses.cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(ref)), (C, inner)
        ok = np.isfinite(ref)
        np.testing.assert_allclose(got[ok], ref[ok], rtol=1e-5, atol=1e-37)
        assert np.array_equal(np.isinf(got), np.isinf(ref))
    # a second call accumulates
    mses2 = mses.clone()
    ops.mse_grid(dev(x), True, dev(grid), mb, 8, 0, mses2)
    ok = np.isfinite(ref)
    np.testing.assert_allclose(mses2.cpu().numpy()[ok], 2 * ref[ok], rtol=1e-5, atol=1e-37)


def test_mse_row_kernel_tile_dependent_paths(ops):
    """Round 3: k_mse_row picks, per (tile, candidate), between integer rounding on the bits of t (no nonzero element below
    the candidate's first binade), the float magic-number rounding (fixed-step subnormal range present) and their
    clamp-free twins (candidate range covers the tile).  Tensors built so that neighbouring 2048-element tiles take
    different paths: sparse tiles (exact zeros, -0), tiles with a few tiny / denormal nonzeros, tiles beyond every
    candidate, exact rounding ties, one-sided data on an unsigned format, +-inf (clamps) and NaN (NaN table)."""
    rng = np.random.RandomState(23)
    n = 2048 * 24
    for sign, special in ((1, False), (0, False), (1, True)):
        x = rng.randn(n).astype(np.float32)
        t = x.reshape(24, 2048)
        t[1] *= (rng.rand(2048) < 0.5)                       # ReLU-like: half exact zeros
        t[2, ::7] = -0.0
        t[3, :5] = np.float32([1e-6, -3e-7, 1e-20, 1e-38, 1e-45])      # tiny and denormal nonzeros: float path for every candidate
        t[4] *= 8.0                                          # far beyond most candidates: everything clamps
        t[5] *= 0.01                                         # every candidate covers the tile: clamp-free
        t[6] = np.round(t[6] * 8) / 8 + 1 / 16               # exact ties of coarse grids
        t[7] = 1.25                                          # constant tile
        t[8, 100] = 1e-3
        t[8, 101:] = 0.0
        if sign == 0:
            x = np.abs(x)
        if special:
            x[9 * 2048 + 3] = np.inf
            x[10 * 2048 + 5] = -np.inf
        grid = np.linspace(0.1 * 5.5, 1.2 * 5.5, 111).astype(np.float32)[:, None]
        mb = [1.0, 2.0, 3.0, 4.0, 5.0, 6.0]
        mses = torch.zeros(len(mb), 111, 1, device="cuda")
        ops.mse_grid(dev(x), False, dev(grid), mb, 8, sign, mses)
        ref = oracle.c_mse_grid(x, False, grid, mb, 8, sign)
        np.testing.assert_allclose(mses.cpu().numpy(), ref, rtol=1e-5, atol=1e-37, err_msg=f"sign {sign} special {special}")
        # the decision the estimator takes from the table: equal, or within 1e-6 of the oracle's minimum
        got = mses.cpu().numpy()[:, :, 0]
        gi, ri = np.unravel_index(got.argmin(), got.shape), np.unravel_index(ref[:, :, 0].argmin(), got.shape)
        assert gi == ri or ref[:, :, 0][gi] <= ref[:, :, 0][ri] * (1 + 1e-6), (gi, ri)
    xn = rng.randn(n).astype(np.float32)
    xn[12345] = np.nan
    mses = torch.zeros(2, 111, 1, device="cuda")
    ops.mse_grid(dev(xn), False, dev(grid), [2.0, 3.0], 8, 1, mses)
    assert torch.isnan(mses).all()


def test_full_size_properties(ops):
    """Size-independent properties at a BASELINE-scale tensor ([2^20,3,7,7], 154 M elements)."""
    n_ch = 1 << 20
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(n_ch, 147, device="cuda", generator=g) * 0.1
    y, mn, mx, mv = ops.minmax_quantize(x, 2, 8, 1)
    # idempotence: quantizing a quantized tensor with the same ranges changes nothing
    y2 = ops.quantize(y, mv, 2, 8, 1)
    assert torch.equal(y, y2)
    # fused == two-pass
    mn2, mx2, mv2 = ops.minmax(x, True, want_maxval=True)
    assert torch.equal(mn, mn2) and torch.equal(mx, mx2) and torch.equal(mv, mv2)
    assert torch.equal(ops.quantize(x, mv, 2, 8, 1), y)
    # range: |y| <= maxval, error bounded by half a step of the top binade (2^-M * maxval / (2-2^-M) / 2 ...)
    # (the top grid point is rint(maxval / s) * s and s carries the fp32 rounding of `bias`: as in the
    # reference it can exceed maxval by ~1e-6 relative)
    assert bool((y.abs() <= mv[:, None] * (1 + 4e-6)).all())
    err = (y - x).abs().amax(1)
    assert bool((err <= mv * 2.0 ** -2 / 1.75 * 0.5 * 1.0001).all())
    # at most 2^8 distinct values per channel (spot check) and symmetry q(-x) = -q(x)
    assert all(torch.unique(y[i]).numel() <= 256 for i in range(0, n_ch, n_ch // 8))
    assert torch.equal(ops.quantize(-x, mv, 2, 8, 1), -y)
    # spot-check 4096 random channels against the CPU oracle, bit for bit (also exercises the
    # table-driven per-channel log2 / exp2 of the short-row kernels on many distinct maxvals)
    idx = torch.randint(0, n_ch, (4096,), generator=torch.Generator().manual_seed(1))
    xs = x[idx.cuda()].cpu().numpy()
    assert_bit_exact(y[idx.cuda()].cpu().numpy(),
                     oracle.c_quantize(xs, mv[idx.cuda()].cpu().numpy(), 2, 8, 1), "spot check")


def test_per_channel_constants_fast_vs_libm(ops):
    """Short-row kernels derive bias / scale constants with table-driven double log2 / exp2; long-row
    kernels use device libm.  Both must give the oracle's bits over a wide range of maxval."""
    rng = np.random.RandomState(11)
    C = 50000
    mv = np.exp(rng.uniform(-40, 40, C)).astype(np.float32)
    mv[:64] = np.float32(2.0) ** np.arange(-32, 32)           # exact powers of two
    mv[64:72] = [0.0, np.inf, 1e-45, 1.17e-38, 3.4e38, 240.0, 57344.0, 1.0]
    for M, inner in ((2, 72), (3, 40), (5, 8)):               # LUT and direct variants of k_rows_direct
        x = (rng.randn(C, inner).astype(np.float32) * (np.where(np.isfinite(mv), mv, 1.0)[:, None] / 2))
        y = ops.quantize(dev(x), dev(mv), M, 8, 1).cpu().numpy()
        assert_bit_exact(y, oracle.c_quantize(x, mv, M, 8, 1), f"M={M} inner={inner}")


@pytest.mark.parametrize("C,inner", [(1, 4), (3, 5), (1, 4097), (2, 2049), (29, 147), (57, 147), (4096, 4), (700, 21),
                                     (333, 36), (1000, 147), (9, 1023), (17, 1024), (5, 1025), (7, 2047),
                                     (4, 1536), (100, 576), (31, 333)])
@pytest.mark.parametrize("M", [2, 3, 4])
def test_flat_short_row_kernel_geometries(ops, C, inner, M):
    """k_rows_flat / k_rows_staged cut the tensor into aligned 4096-element chunks regardless of the rows: chunk borders
    inside rows, rows longer than a chunk, a partial last chunk, <= 3 tail scalars (C*inner % 4 != 0),
    16-byte groups that straddle two rows (inner % 4 != 0).  K1 (also in place) and the fused
    min/max + quantize, bit-exact against the oracle."""
    rng = np.random.RandomState(C * 7 + inner + M)
    x = (rng.randn(C, inner) * np.exp(rng.uniform(-3, 3, (C, 1)))).astype(np.float32)
    x.reshape(-1)[:: 97] = 0.0
    mn, mx = oracle.c_minmax(x, True)
    mv = oracle.c_absmax(mn, mx)
    ref = oracle.c_quantize(x, mv, M, 8, 1)
    xd = dev(x)
    assert_bit_exact(ops.quantize(xd, dev(mv), M, 8, 1).cpu().numpy(), ref, f"K1 {C}x{inner} M={M}")
    y, gmn, gmx, gmv = ops.minmax_quantize(xd, M, 8, 1)
    np.testing.assert_array_equal(gmn.cpu().numpy(), mn)
    np.testing.assert_array_equal(gmx.cpu().numpy(), mx)
    np.testing.assert_array_equal(gmv.cpu().numpy(), mv)
    assert_bit_exact(y.cpu().numpy(), ref, f"fused {C}x{inner} M={M}")
    xi = xd.clone()
    ops.quantize(xi, dev(mv), M, 8, 1, out=xi)                 # in place
    assert_bit_exact(xi.cpu().numpy(), ref, f"K1 in place {C}x{inner} M={M}")
    xi = xd.clone()
    ops.minmax_quantize(xi, M, 8, 1, out=xi)                   # in place: row-tiled kernel (rows owned whole)
    assert_bit_exact(xi.cpu().numpy(), ref, f"fused in place {C}x{inner} M={M}")


@pytest.mark.parametrize("C,inner,M", [(200003, 147, 2), (150001, 99, 3), (70001, 255, 2), (300007, 39, 3), (90001, 201, 1)])
def test_staged_fused_short_rows_many_chunks(ops, C, inner, M):
    """k_rows_staged (fused min/max + quantize of rows <= 256 elements, the chunk parked in LDS and fetched once): tensors
    of more chunks than the persistent grid has blocks, so every block runs the software-pipelined loop several times
    (next chunk's loads in flight during the range / table / patch phase); NaN rows, all-zero rows, rows cut by chunk
    borders, tail scalars.  Bit-exact against the oracle, reported ranges included."""
    rng = np.random.RandomState(C % 1000 + inner + M)
    x = (rng.randn(C, inner) * np.exp(rng.uniform(-3, 3, (C, 1)))).astype(np.float32)
    x[1] = 0.0
    x[C // 2, inner // 2] = np.nan
    x[C - 1, inner - 1] = np.nan
    x[27:31, 0] = 1e30                                  # rows around the first chunk border
    mn, mx = oracle.c_minmax(x, True)
    mv = oracle.c_absmax(mn, mx)
    ref = oracle.c_quantize(x, mv, M, 8, 1)
    y, gmn, gmx, gmv = ops.minmax_quantize(dev(x), M, 8, 1)
    np.testing.assert_array_equal(gmn.cpu().numpy(), mn)
    np.testing.assert_array_equal(gmx.cpu().numpy(), mx)
    np.testing.assert_array_equal(gmv.cpu().numpy(), mv)
    assert_bit_exact(y.cpu().numpy(), ref, f"staged fused {C}x{inner} M={M}")


@pytest.mark.parametrize("C,inner", [(2, 147), (29, 147), (70001, 147), (333, 255), (7001, 99), (50021, 41), (100003, 9),
                                     (40000, 5), (77777, 4), (5, 250), (300007, 27), (28, 256)])
def test_staged_minmax_short_rows(ops, C, inner):
    """k_rows_staged_mm (per-channel min/max of rows <= 256 elements through aligned chunks parked in LDS): rows cut by
    chunk borders, several rows-per-pass geometries (1..8 lanes per row), tail scalars, NaN rows; overwrite, all-min/max
    and EMA folds into a running estimate.  Equal to the oracle bit for bit."""
    rng = np.random.RandomState(C % 977 + inner)
    x = rng.randn(C, inner).astype(np.float32)
    x[C // 2, inner // 2] = np.nan
    x[0] = 0.0
    rmn, rmx = oracle.c_minmax(x, True)
    mn, mx, mv = ops.minmax(dev(x), True, want_maxval=True)
    np.testing.assert_array_equal(mn.cpu().numpy(), rmn)
    np.testing.assert_array_equal(mx.cpu().numpy(), rmx)
    np.testing.assert_array_equal(mv.cpu().numpy(), oracle.c_absmax(rmn, rmx))
    x2 = (rng.randn(C, inner) * 1.5).astype(np.float32)
    r2 = oracle.c_minmax(x2, True)
    for mode in (1, 2):
        cur = ops.minmax(dev(x), True, mode=mode, momentum=0.9)
        cur = ops.minmax(dev(x2), True, cur[0], cur[1], mode=mode, momentum=0.9)
        emn, emx = oracle.c_fold(rmn, rmx, r2[0], r2[1], mode, 0.9)
        np.testing.assert_array_equal(cur[0].cpu().numpy(), emn)
        np.testing.assert_array_equal(cur[1].cpu().numpy(), emx)


@pytest.mark.parametrize("C,inner", [(9, 512), (33, 1024), (7, 1152), (5, 2048), (6, 2052), (3, 4608), (5, 8192),
                                     (2, 8196), (13, 576), (4, 260), (1, 1024), (257, 1280), (50, 128), (35, 192),
                                     (19, 288), (70, 384), (3, 7168), (17, 124)])
@pytest.mark.parametrize("M", [2, 4])
def test_fused_rows_in_registers(ops, C, inner, M):
    """k_rows_reg (rows of 257..8192 elements held in registers between the min/max and the quantize pass): one
    wave per row and one block per row, full and partly filled lanes, rows that fall back to the row-tiled
    kernel, NaN / all-zero rows, in place; bit-exact against the oracle."""
    rng = np.random.RandomState(C + inner + M)
    x = (rng.randn(C, inner) * np.exp(rng.uniform(-3, 3, (C, 1)))).astype(np.float32)
    if C > 2:
        x[1] = 0.0
        x[2, inner // 2] = np.nan
    mn, mx = oracle.c_minmax(x, True)
    mv = oracle.c_absmax(mn, mx)
    ref = oracle.c_quantize(x, mv, M, 8, 1)
    xd = dev(x)
    y, gmn, gmx, gmv = ops.minmax_quantize(xd, M, 8, 1)
    np.testing.assert_array_equal(gmn.cpu().numpy(), mn)
    np.testing.assert_array_equal(gmx.cpu().numpy(), mx)
    np.testing.assert_array_equal(gmv.cpu().numpy(), mv)
    assert_bit_exact(y.cpu().numpy(), ref, f"fused {C}x{inner} M={M}")
    xi = xd.clone()
    ops.minmax_quantize(xi, M, 8, 1, out=xi)
    assert_bit_exact(xi.cpu().numpy(), ref, f"fused in place {C}x{inner} M={M}")


@pytest.mark.parametrize("C,inner", [(35, 8193), (2, 8195), (3, 16385), (1, 8193), (1, 2048 * 4 * 7 + 1), (5, 24577)])
def test_minmax_with_an_empty_split(ops, C, inner):
    """Rows a few elements longer than a whole number of 8192-element steps leave the last split of the
    two-stage min/max without any aligned group: its partial is {+inf, -inf} and must not leak into the result
    (found by a round-2 random-geometry soak: all-positive rows came back with max = +inf)."""
    rng = np.random.RandomState(C + inner)
    for x in (np.abs(rng.randn(C, inner)).astype(np.float32) + 0.5, -np.abs(rng.randn(C, inner)).astype(np.float32) - 0.5,
              rng.randn(C, inner).astype(np.float32)):
        for pc in (True, False):
            mn, mx = ops.minmax(dev(x), pc)
            rmn, rmx = oracle.c_minmax(x, pc)
            np.testing.assert_array_equal(mn.cpu().numpy(), rmn)
            np.testing.assert_array_equal(mx.cpu().numpy(), rmx)


def test_fuzz_geometries_against_oracle(ops):
    """Seeded fuzz over the routing space of the per-channel kernels (k_rows_flat / k_rows_reg / k_rows_direct /
    k_quant_rows / k_quant_scalar): random channel counts, row lengths, formats, sign bits, pointer phases (views
    that start 0..3 elements into an allocation), in place or not; K1, fused min/max+quantize and the folding
    min/max, all bit-exact against the oracle."""
    rng = np.random.RandomState(2024)
    lengths = [1, 2, 3, 4, 5, 7, 9, 16, 20, 21, 35, 36, 37, 64, 100, 124, 128, 132, 147, 192, 255, 256, 257, 288, 300,
               384, 388, 448, 511, 512, 513, 576, 640, 896,
               1000, 1023, 1024, 1028, 1152, 2044, 2047, 2048, 2049, 2052, 3000, 4096, 4100, 4608, 8192, 8193, 8195, 8196, 9000,
               16385]
    for case in range(200):
        inner = int(lengths[rng.randint(len(lengths))])
        C = int(rng.choice([1, 2, 3, 5, 17, 64, 130, 301])) if inner > 600 else int(rng.randint(1, 700))
        M = int(rng.randint(1, 7))
        sb = int(rng.rand() < 0.85)
        off = int(rng.choice([0, 0, 0, 1, 2, 3]))
        x = (rng.randn(C, inner) * np.exp(rng.uniform(-4, 4, (C, 1)))).astype(np.float32)
        if sb == 0:
            x = np.abs(x)
        if rng.rand() < 0.3 and x.size > 10:
            x.reshape(-1)[rng.randint(x.size, size=3)] = [0.0, -0.0, np.float32(1e-40)]
        base = torch.empty(x.size + 4, device="cuda")
        xd = base[off: off + x.size].view(C, inner)
        xd.copy_(torch.from_numpy(x))
        mn, mx = oracle.c_minmax(x, True)
        mv = oracle.c_absmax(mn, mx)
        what = f"case {case}: C={C} inner={inner} M={M} sb={sb} off={off}"
        ref = oracle.c_quantize(x, mv, M, 8, sb)
        y = ops.quantize(xd, dev(mv), M, 8, sb)
        assert_bit_exact(y.cpu().numpy(), ref, "K1 " + what)
        fusable = inner <= ops.fused_max_inner()   # longer rows: the manager calls min/max + quantize
        if fusable:
            yf, gmn, gmx, gmv = ops.minmax_quantize(xd, M, 8, sb)
            np.testing.assert_array_equal(gmn.cpu().numpy(), mn, err_msg=what)
            np.testing.assert_array_equal(gmx.cpu().numpy(), mx, err_msg=what)
            assert_bit_exact(yf.cpu().numpy(), ref, "fused " + what)
        kmn, kmx = ops.minmax(xd, True)
        np.testing.assert_array_equal(kmn.cpu().numpy(), mn, err_msg=what)
        np.testing.assert_array_equal(kmx.cpu().numpy(), mx, err_msg=what)
        if case % 3 == 0 and fusable:   # in place, through a view with the same phase
            xi = base.clone()[off: off + x.size].view(C, inner)
            ops.minmax_quantize(xi, M, 8, sb, out=xi)
            assert_bit_exact(xi.cpu().numpy(), ref, "fused in place " + what)
            pmn, pmx = ops.minmax(xd * 0.5, True)
            fmn, fmx = ops.minmax(xd, True, pmn, pmx, mode=1)
            rmn, rmx = oracle.c_fold(*oracle.c_minmax(x * np.float32(0.5), True), mn, mx, 1, 0.9)
            np.testing.assert_array_equal(fmn.cpu().numpy(), rmn, err_msg=what)
            np.testing.assert_array_equal(fmx.cpu().numpy(), rmx, err_msg=what)
        one = np.array([np.abs(x).max() + np.float32(0.01)], np.float32)
        assert_bit_exact(ops.quantize(xd, dev(one), M, 8, sb).cpu().numpy(), oracle.c_quantize(x, one, M, 8, sb),
                         "per-tensor " + what)


@pytest.mark.parametrize("C,inner,M", [(1 << 17, 147, 2), (1 << 18, 147, 3), (30000, 1152, 2), (100000, 576, 2), (70001, 99, 3)])
def test_flat_kernel_resident_grid_mid_size(ops, C, inner, M):
    """Cache-sized per-channel tensors (1024 < tiles, <= 16384 chunks) run k_rows_flat on a RESIDENT grid of 1024 blocks
    that stride over the chunks (several tiles per block): every row band of the result against the oracle."""
    g = torch.Generator(device="cuda").manual_seed(C + inner)
    x = torch.randn(C, inner, device="cuda", generator=g) * (torch.rand(C, 1, device="cuda", generator=g) * 2 + 0.05)
    mv = ops.minmax(x, True, want_maxval=True)[2]
    y = ops.quantize(x, mv, M, 8, 1)
    mvh = mv.cpu().numpy()
    for lo in (0, C // 3, C // 2 + 17, C - 2048):
        sl = slice(lo, lo + 2048)
        ref = oracle.c_quantize(x[sl].cpu().numpy(), mvh[sl], M, 8, 1)
        assert_bit_exact(y[sl].cpu().numpy(), ref, f"rows {lo}..{lo + 2048} of [{C},{inner}]")
    # and the whole tensor against the fused kernel's output (another kernel, same arithmetic)
    if inner <= ops.fused_max_inner():
        y2 = ops.minmax_quantize(x, M, 8, 1)[0]
        assert torch.equal(y.view(torch.int32), y2.view(torch.int32))


def test_multi_tensor_quantize(ops):
    """fp8q_multi_quantize_f32: every weight tensor of a model in one launch, bit-identical to one
    fp8q_quantize_f32 per tensor (and to the oracle); mixed formats, per-tensor entries, tensors that fall back
    to their own launch (depthwise 3x3 rows, unaligned views), empty tensors, > 32 tensors."""
    rng = np.random.RandomState(5)
    shapes = [(64, 3, 7, 7)] + [(64, 64, 3, 3)] * 4 + [(128, 64, 3, 3), (128, 128, 3, 3), (128, 64, 1, 1)] + \
             [(128, 128, 3, 3)] * 2 + [(256, 128, 3, 3), (256, 256, 3, 3), (256, 128, 1, 1)] + [(256, 256, 3, 3)] * 2 + \
             [(512, 256, 3, 3), (512, 512, 3, 3), (512, 256, 1, 1)] + [(512, 512, 3, 3)] * 2 + [(1000, 512)]
    shapes += [(32, 1, 3, 3), (7, 5), (3, 4099), (1, 4), (0, 9), (96, 1, 3, 3), (24, 144, 1, 1), (1280, 320, 1, 1)]
    shapes += [(16, 10 + i) for i in range(12)]
    items, refs = [], []
    for i, shp in enumerate(shapes):
        x = (rng.randn(*shp) * 0.1).astype(np.float32)
        M = (2, 3, 4)[i % 3]
        per_tensor = i % 7 == 3
        if per_tensor:
            mv = np.array([np.abs(x).max() if x.size else 1.0], np.float32)
        else:
            mv = (np.abs(x.reshape(shp[0], -1)).max(1) if x.size else np.zeros(0)).astype(np.float32)
        items.append((dev(x) if x.size else torch.empty(shp, device="cuda"), dev(mv) if mv.size else torch.empty(0, device="cuda"), M, 8, 1))
        refs.append(oracle.c_quantize(x, mv, M, 8, 1) if x.size else x)
    # one unaligned view (own launch)
    base = dev((rng.randn(64 * 147 + 1) * 0.1).astype(np.float32))
    xv = base[1:].view(64, 147)
    mvv = xv.abs().amax(1)
    items.append((xv, mvv, 2, 8, 1))
    refs.append(oracle.c_quantize(xv.cpu().numpy(), mvv.cpu().numpy(), 2, 8, 1))
    outs = ops.multi_quantize(items)
    assert len(outs) == len(items)
    for i, (it, y, ref) in enumerate(zip(items, outs, refs)):
        if ref.size == 0:
            assert y.numel() == 0
            continue
        assert_bit_exact(y.cpu().numpy(), ref, f"multi item {i} shape {tuple(it[0].shape)}")
        assert_bit_exact(y.cpu().numpy(), ops.quantize(it[0], it[1], it[2], 8, 1).cpu().numpy(), f"multi vs single {i}")
    with pytest.raises(Exception):
        ops.multi_quantize([(items[0][0], items[0][1][:3], 2, 8, 1)])   # wrong maxval length


def test_multi_tensor_minmax_quantize(ops):
    """fp8q_multi_minmax_quantize_f32: per-channel current_minmax ranges + quantize of a whole model's weight tensors in
    two launches -- ranges and values bit-identical to the oracle and to one fused fp8q_minmax_quantize_f32 per tensor;
    short / odd / long rows, a zero channel (NaN quirk), a NaN element, an unaligned view, an empty tensor, > 32 tensors."""
    rng = np.random.RandomState(6)
    shapes = [(64, 3, 7, 7), (64, 64, 3, 3), (128, 64, 1, 1), (512, 512, 3, 3), (1000, 512), (32, 1, 3, 3), (7, 5), (3, 4099),
              (1, 4), (0, 9), (5, 20000), (24, 144, 1, 1)] + [(16, 10 + i) for i in range(24)]
    items, xs = [], []
    for i, shp in enumerate(shapes):
        x = (rng.randn(*shp) * 0.1).astype(np.float32)
        if i == 1:
            x[3] = 0.0                      # all-zero channel: maxval 0 -> NaN channel, as in the reference
        if i == 3:
            x[7, 5, 1, 1] = np.nan          # NaN anywhere in a row: the row's range is NaN
        xs.append(x)
        items.append((dev(x) if x.size else torch.empty(shp, device="cuda"),
                      torch.full((shp[0],), -7.0, device="cuda"), (2, 3, 4)[i % 3], 8, 1))
    base = dev((rng.randn(64 * 147 + 1) * 0.1).astype(np.float32))
    items.append((base[1:].view(64, 147), torch.empty(64, device="cuda"), 2, 8, 1))
    xs.append(base[1:].view(64, 147).cpu().numpy())
    outs = ops.multi_minmax_quantize(items)
    assert len(outs) == len(items)
    for i, (it, y, x) in enumerate(zip(items, outs, xs)):
        if x.size == 0:
            assert y.numel() == 0
            continue
        mn, mx = oracle.c_minmax(x, True)
        mv = oracle.c_absmax(mn, mx)
        got_mv = it[1].cpu().numpy()
        assert np.array_equal(np.isnan(got_mv), np.isnan(mv)) and np.array_equal(got_mv[~np.isnan(mv)], mv[~np.isnan(mv)]), i
        assert_bit_exact(y.cpu().numpy(), oracle.c_quantize(x, mv, it[2], 8, 1), f"multi minmax item {i} {x.shape}")
        if x.shape[1:] and int(np.prod(x.shape[1:])) <= ops.fused_max_inner():
            y1, _, _, mv1 = ops.minmax_quantize(it[0], it[2], 8, 1)
            assert_bit_exact(y.cpu().numpy(), y1.cpu().numpy(), f"multi vs fused {i}")
    with pytest.raises(Exception):
        ops.multi_minmax_quantize([(items[0][0], torch.empty(3, device="cuda"), 2, 8, 1)])   # wrong range length


def test_more_than_2_31_elements(ops):
    """Maximum sizes: a per-tensor tensor with > 2^31 elements (8.6 GB in, 8.6 GB out) exercises the
    64-bit indexing; chunks quantized separately must give the same bits, min/max must see the planted
    extremes at both ends, and the tail is checked against the oracle."""
    n = (1 << 31) + 12345
    free, _ = torch.cuda.mem_get_info()
    if free < 3 * n * 4:
        pytest.skip("not enough free HBM for the 2^31-element case")
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.empty(n, device="cuda")
    step = 1 << 28
    for i in range(0, n, step):
        x[i:i + step].normal_(generator=g)
    x[7] = -123.0
    x[n - 3] = 77.0
    mv = torch.tensor([4.0], device="cuda")
    y = ops.quantize(x, mv, 3, 8, 1)
    for lo in (0, (1 << 31) - 4096, n - 65536):
        hi = min(lo + 65536, n)
        assert torch.equal(y[lo:hi], ops.quantize(x[lo:hi].clone(), mv, 3, 8, 1)), lo
    tail = x[n - 4096:].cpu().numpy()
    assert_bit_exact(y[n - 4096:].cpu().numpy(), oracle.c_quantize(tail, [4.0], 3, 8, 1), "tail")
    mn, mx = ops.minmax(x, False)
    assert mn.item() == -123.0 and mx.item() == 77.0
    del y
    # per-channel rows kernel across the 2^31 boundary: [2, n/2]
    half = n // 2
    xr = x[: 2 * half].view(2, half)
    mvr = torch.tensor([3.0, 5.0], device="cuda")
    yr = ops.quantize(xr, mvr, 2, 8, 1)
    assert torch.equal(yr[1, half - 8192:], ops.quantize(xr[1, half - 8192:].clone(), mvr[1:], 2, 8, 1))
