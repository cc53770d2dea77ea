"""Pin the CPU oracle against golden vectors produced by the reference itself (CPU-only test)."""
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import torch_eager as te
from parity import assert_parity, compare, elem_step


@pytest.fixture(scope="module")
def g1(golden_dir):
    return np.load(os.path.join(golden_dir, "g1_quantize.npz"))


def _cases(g1):
    for cid, mbits, mv, sb, nmv in g1["cases"]:
        cid = int(cid)
        x, y = g1[f"c{cid}_x"], g1[f"c{cid}_y"]
        maxval = g1[f"c{cid}_maxval"] if mv < 0 else np.array([mv], np.float32)
        yield cid, float(mbits), maxval, int(sb), x, y


def test_c_oracle_quantize_vs_reference(g1):
    """C restatement (correctly rounded log2/pow) vs the reference's torch CPU output."""
    tot, exact, flips = 0, 0.0, 0
    for cid, mbits, maxval, sb, x, y_ref in _cases(g1):
        y = oracle.c_quantize(x, maxval, mbits, 8, sb)
        step = elem_step(x, maxval, mbits, 8, sb)
        # the hand-picked inputs sit exactly on ties / binade edges; measured: not one of them flips
        r = assert_parity(y, y_ref, step, max_flip_frac=0.0, max_ulp=2, what=f"case {cid}")
        tot += r["n"]
        exact += r["exact_frac"] * r["n"]
        flips += r["n_flips"]
    print(f"\nC oracle vs reference: {tot} elems, bit-exact {exact / tot:.4%}, tie flips {flips}")
    assert exact / tot > 0.90


def bulk_cases():
    """mirror of tests/golden/make_golden.py:bulk_cases (the generator imports the reference and cannot travel)"""
    cases = []
    for M, tag in ((2, "e5m2"), (3, "e4m3")):
        cases.append((f"{tag}_tensor", 4100 + M, M, (1 << 22,), np.array([2.7361], np.float32)))
        rng = np.random.RandomState(4200 + M)
        cases.append((f"{tag}_channel", 4300 + M, M, (4096, 1024),
                      (np.abs(rng.standard_normal(4096)) * 2 + 0.05).astype(np.float32)))
    return cases


def bulk_input(seed, shape):
    return np.random.RandomState(seed).standard_normal(int(np.prod(shape))).astype(np.float32).reshape(shape)


def reference_from_oracle(g, name, y_orc):
    """The reference's output rebuilt from the oracle's + the stored sparse difference; its SHA-256 must be the stored
    one -- i.e. the fixture pins all 4 M reference values bit for bit.  Returns (y_ref, ulp deltas, n flips)."""
    import hashlib
    y_ref = np.ascontiguousarray(y_orc, np.float32).reshape(-1).copy()
    idx = np.cumsum(g[f"{name}_idx_delta"].astype(np.int64))
    dulp = g[f"{name}_ulp_delta"].astype(np.int64)
    key = y_ref[idx].view(np.int32).astype(np.int64)          # only the differing elements are touched (keeps -0)
    key = np.where(key < 0, np.int64(-2147483648) - key, key) + dulp
    y_ref[idx] = np.where(key < 0, np.int64(-2147483648) - key, key).astype(np.int32).view(np.float32)
    y_ref[g[f"{name}_big_idx"]] = g[f"{name}_big_val"]
    assert hashlib.sha256(y_ref.tobytes()).digest() == g[f"{name}_sha256"].tobytes(), \
        f"{name}: oracle output + stored difference is not the reference's output"
    return y_ref.reshape(y_orc.shape), dulp, int(g[f"{name}_big_idx"].size)


def test_c_oracle_bulk_flip_rate_vs_reference(golden_dir):
    """SURVEY 8(c) metric on bulk data: 4 x 4 M seeded normals ({E5M2, E4M3} x {per-tensor arbitrary maxval,
    per-channel}).  C oracle vs the reference: <= 1 grid step everywhere, grid-step flips <= 1e-5 of the elements
    (measured: none), everything else within 2 fp32 ULP (measured: per-tensor bit-identical, per-channel 1.7-1.9 %
    of the elements off by <= 2 ULP -- the last-bit differences of torch's vectorised log2 / pow)."""
    g = np.load(os.path.join(golden_dir, "g1b_bulk.npz"))
    for name, seed, M, shape, mv in bulk_cases():
        x = bulk_input(seed, shape)
        y = oracle.c_quantize(x, mv, M, 8, 1)
        y_ref, dulp, n_big = reference_from_oracle(g, name, y)
        r = assert_parity(y, y_ref, elem_step(x, mv, M, 8, 1), max_flip_frac=1e-5, max_ulp=2, what=name)
        assert n_big == 0 and (dulp.size == 0 or np.abs(dulp).max() <= 2), (name, n_big, np.abs(dulp).max())
        print(f"\n{name}: {r['n']} elements, bit-exact {r['exact_frac']:.4%}, flips {r['n_flips']}, "
              f"max ULP {r['max_ulp_nonflip']}")


def test_torch_eager_oracle_quantize_vs_reference(g1):
    """Same ATen op chain -> bit-identical on the torch build that made the fixtures."""
    all_exact = True
    for cid, mbits, maxval, sb, x, y_ref in _cases(g1):
        y = te.fake_quant(torch.from_numpy(x), 8, torch.from_numpy(maxval),
                          torch.tensor([mbits]), sb).numpy()
        r = compare(y, y_ref, elem_step(x, maxval, mbits, 8, sb))
        assert r["nan_equal"] and r["max_steps"] <= 1.0 + 1e-6 and r["max_ulp_nonflip"] <= 2, r
        all_exact &= r["exact_frac"] == 1.0
    if torch.__version__.startswith("2.10"):
        assert all_exact, "torch-eager restatement should be bit-identical on torch 2.10 CPU"


def test_fp_grids(golden_dir):
    g2 = np.load(os.path.join(golden_dir, "g2_grids.npz"))
    for key in g2.files:
        if key.startswith("scaled_"):
            e = int(key.rsplit("e", 1)[1])
            grid = oracle.c_fp_grid(8, e, 2 ** (e - 1))
            np.testing.assert_array_equal(grid / (np.abs(grid).max() / 3.0), g2[key])
        else:
            e, b = key[1:].split("_b")
            np.testing.assert_array_equal(oracle.c_fp_grid(8, int(e), int(b)), g2[key])


def test_quantizer_output_lies_on_enumerated_grid():
    """Independent format definition (fp8_quantizer.py:13-50): outputs are grid points."""
    rng = np.random.RandomState(0)
    for M, mv in ((2, 57344.0), (3, 240.0), (3, 1.7), (4, 0.31)):
        E = 7 - M
        grid = oracle.c_fp_grid(8, E, 2 ** (E - 1))
        grid = grid / (np.abs(grid).max() / mv)
        x = (rng.randn(20000) * mv / 3).astype(np.float32)
        y = oracle.c_quantize(x, mv, M).astype(np.float64)
        d = np.abs(y[:, None] - grid[None, :]).min(1)
        assert np.all(d <= 4e-7 * np.maximum(np.abs(y), mv * 2.0 ** -20)), (M, mv, d.max())


def test_estimators_minmax(golden_dir):
    g3 = np.load(os.path.join(golden_dir, "g3_estimators.npz"))
    w = g3["w"]
    mn, mx = oracle.c_minmax(w, True)
    np.testing.assert_array_equal(mn, g3["w_cur_pc_min"])
    np.testing.assert_array_equal(mx, g3["w_cur_pc_max"])
    mn, mx = oracle.c_minmax(w, False)
    np.testing.assert_array_equal(mn[0], g3["w_cur_pt_min"])
    np.testing.assert_array_equal(mx[0], g3["w_cur_pt_max"])
    mv = oracle.c_absmax(g3["w_cur_pc_min"], g3["w_cur_pc_max"])
    np.testing.assert_array_equal(mv, g3["w_maxval"])
    # conv1 [64,3,7,7] E5M2 per-channel = BASELINE config 2
    y = oracle.c_quantize(w, mv, 2)
    assert_parity(y, g3["w_q_e5m2"], elem_step(w, mv, 2), what="conv1 e5m2")
    for mode, name in ((1, "allminmax"), (2, "running_minmax")):
        for pc in (False, True):
            cur = None
            for b, a in enumerate(g3["acts"]):
                mn, mx = oracle.c_minmax(a, pc)
                cur = (mn, mx) if cur is None else oracle.c_fold(cur[0], cur[1], mn, mx, mode)
                ref_mn, ref_mx = g3[f"{name}_pc{int(pc)}_min"][b], g3[f"{name}_pc{int(pc)}_max"][b]
                if mode == 1:
                    np.testing.assert_array_equal(cur[0], ref_mn)
                    np.testing.assert_array_equal(cur[1], ref_mx)
                else:  # EMA: fp32 a*b+c*d, no FMA contraction in either -> exact too
                    np.testing.assert_array_equal(cur[0], ref_mn)
                    np.testing.assert_array_equal(cur[1], ref_mx)
    a = g3["acts"][0].copy()
    a[1, 2, 3, 4] = np.nan
    mn, mx = oracle.c_minmax(a, False)
    assert np.isnan(mn[0]) and np.isnan(mx[0]) and np.isnan(g3["nan_min"]) and np.isnan(g3["nan_max"])
    # unsigned (ReLU) range -> sign_bits 0
    assert int(g3["relu_sign_bits"]) == 0
    y = oracle.c_quantize(g3["relu_x"], g3["relu_maxval"], 3, 8, 0)
    assert_parity(y, g3["relu_q"], elem_step(g3["relu_x"], g3["relu_maxval"], 3, 8, 0), what="relu")


def ieee_min_max_rows(x):
    """IEEE 754-2019 minimum / maximum per row (-0 < +0), written out independently of the oracle"""
    mn, mx = x.min(1), x.max(1)            # numpy: correct values, unspecified sign of zero
    neg0 = (np.signbit(x) & (x == 0)).any(1)
    pos0 = (~np.signbit(x) & (x == 0)).any(1)
    mn = np.where(mn == 0, np.where(neg0, np.float32(-0.0), np.float32(0.0)), mn).astype(np.float32)
    mx = np.where(mx == 0, np.where(pos0, np.float32(0.0), np.float32(-0.0)), mx).astype(np.float32)
    return mn, mx


def test_signed_zero_contract_of_minmax_is_pinned_by_a_fixture(golden_dir):
    """g3b: rows mixing -0.0 and +0.0.  ATen's min / max of such a row takes the sign of whichever zero comes first (rows
    0 / 1 and 6 / 7 of the fixture hold the same values in opposite orders and get opposite signs), so there is nothing
    order-independent to reproduce bit for bit; the contract of oracle and kernels is IEEE 754-2019 minimum / maximum.
    Pinned here: the reference's VALUES (== treats the zeros alike), K5's maxval bit for bit (it takes abs), the quantized
    rows bit for bit (NaN rows for maxval 0, signed zeros kept), the running fold, and the contract itself."""
    g = np.load(os.path.join(golden_dir, "g3b_signed_zero.npz"))
    x = g["x"]
    assert not np.array_equal(np.signbit(g["pc1_max"][[0, 6]]), np.signbit(g["pc1_max"][[1, 7]]))   # the reference: order-dependent
    mn, mx = oracle.c_minmax(x, True)
    assert np.array_equal(mn, g["pc1_min"]) and np.array_equal(mx, g["pc1_max"])                    # values
    emn, emx = ieee_min_max_rows(x)
    assert np.array_equal(mn.view(np.int32), emn.view(np.int32)) and np.array_equal(mx.view(np.int32), emx.view(np.int32))
    tmn, tmx = oracle.c_minmax(x, False)
    assert np.array_equal(tmn, g["pc0_min"]) and np.array_equal(tmx, g["pc0_max"])
    # allminmax over the batch and its mirror image: same values, the contract's signs
    fmn, fmx = oracle.c_fold(mn, mx, *oracle.c_minmax(x[:, ::-1].copy(), True), 1)
    assert np.array_equal(fmn, g["all_pc1_min"]) and np.array_equal(fmx, g["all_pc1_max"])
    assert np.array_equal(fmn.view(np.int32), emn.view(np.int32)) and np.array_equal(fmx.view(np.int32), emx.view(np.int32))
    mv = oracle.c_absmax(mn, mx)
    assert np.array_equal(mv.view(np.int32), g["maxval"].view(np.int32))
    with np.errstate(all="ignore"):
        q = oracle.c_quantize(x, mv, 3, 8, 1)
    assert np.array_equal(np.isnan(q), np.isnan(g["q"]))
    ok = ~np.isnan(q)
    assert np.array_equal(np.signbit(q[ok]), np.signbit(g["q"][ok]))                     # signed zeros kept
    assert np.abs(q[ok].view(np.int32).astype(np.int64) - g["q"][ok].view(np.int32)).max() <= 2   # (log2 / 2^x: <= 2 ulp)


@pytest.mark.parametrize("name,pc,incl,M", [("w_pc_fixm", True, False, 3), ("w_pc_srchm", True, True, 3),
                                            ("a_pt_fixm", False, False, 3), ("a_pt_srchm", False, True, 2)])
def test_mse_estimator(golden_dir, name, pc, incl, M):
    g4 = np.load(os.path.join(golden_dir, "g4_mse.npz"))
    mb = [1, 2, 3, 4, 5, 6] if incl else [M]
    grid_ref = g4[f"{name}_grid"]
    x0 = g4[f"{name}_x0"]
    grid = te.mse_search_grid(torch.from_numpy(x0), pc).numpy()
    np.testing.assert_array_equal(grid, grid_ref)
    mses = None
    for b in range(2):
        x = g4[f"{name}_x{b}"]
        mses = oracle.c_mse_grid(x, pc, grid, mb, 8, 1, mses)
        ref = g4[f"{name}_mses{b}"]
        np.testing.assert_allclose(mses, ref, rtol=1e-4, atol=0)  # 27-elem channels: a 1-ulp scale shift moves a tiny MSE by ~1e-5
        mbits, maxval, idx = te.mse_select(torch.from_numpy(mses), torch.from_numpy(grid), mb)
        assert mbits == float(g4[f"{name}_mbits{b}"])
        ref_max = g4[f"{name}_max{b}"].reshape(-1)
        # chosen candidate equal, or its MSE within 1e-5 relative of the reference's minimum
        mi = mb.index(int(mbits))
        for c in range(grid.shape[1]):
            if maxval[c].item() != ref_max[c]:
                j_ref = int(np.argmin(np.abs(grid[:, c] - ref_max[c])))
                assert abs(ref[mi, idx[c], c] - ref[mi, j_ref, c]) <= 1e-5 * ref[mi, j_ref, c]


def test_torch_eager_mse_bit_exact(golden_dir):
    g4 = np.load(os.path.join(golden_dir, "g4_mse.npz"))
    x = torch.from_numpy(g4["w_pc_fixm_x0"])
    grid = te.mse_search_grid(x, True)
    mses = te.mse_grid(x, True, grid, [3.0], 8, 1, torch.zeros(1, 111, 32))
    if torch.__version__.startswith("2.10"):
        np.testing.assert_array_equal(mses.numpy(), g4["w_pc_fixm_mses0"])
    else:
        np.testing.assert_allclose(mses.numpy(), g4["w_pc_fixm_mses0"], rtol=1e-5)
