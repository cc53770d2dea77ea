#!/usr/bin/env python
"""(test infrastructure: lives under tests/ because it drives the oracle; not collected by pytest)
Time-bounded random parity soak of every C-ABI entry point against the CPU oracle (GPU box only).

    python tests/soak.py [--seconds 300] [--seed 0]  > gpurun_out/r04_soak.txt

Every case draws a shape, a format, a range and special values at random, runs the HIP path and the oracle on the same
inputs and demands BIT equality (fp32 / fp64 values, codes, ranges).  The search kernel's sort-once route is compared
with the lane-per-element kernel and the oracle at K4's stated tolerance.  Prints one line per family with the number of
cases and elements, and exits non-zero on the first mismatch (with the case that produced it)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fp8-quantization_amd"))
sys.path.insert(0, ROOT)

import fp8q  # noqa: E402
import oracle  # noqa: E402

ops = fp8q.ops
INNERS = [1, 2, 3, 4, 5, 7, 8, 9, 16, 27, 49, 64, 99, 147, 201, 256, 257, 576, 1152, 2047, 2048, 2049, 4096, 4608, 10007]


def bits(a):
    a = np.ascontiguousarray(a)
    v = a.view(np.int64 if a.dtype == np.float64 else np.int32)
    return np.where(np.isnan(a), -1, v)


def fmt(rng):
    n_bits = int(rng.choice([8, 8, 8, 8, 6, 5, 4]))
    sb = int(rng.choice([1, 1, 0]))
    M = float(rng.choice([1, 2, 3, 4, 5, 6, 7, 2.5, 0.2, 9.0]))
    if n_bits - sb - min(max(round(M), 1), n_bits - sb) < 0:
        M = 1.0
    return M, n_bits, sb


LAST_KIND = [None]


def data(rng, C, inner, dtype=np.float32):
    kind = rng.randint(6)
    LAST_KIND[0] = int(kind)
    scale = 10.0 ** rng.uniform(-4, 3)
    if kind == 0:      # exact powers of two and ties
        x = np.ldexp(rng.choice([1.0, 1.5, 1.25, 1.75, 1.125]), rng.randint(-20, 8, size=(C, inner))) * rng.choice([-1.0, 1.0], size=(C, inner))
    elif kind == 1:    # heavy tails
        x = rng.standard_cauchy((C, inner)) * scale
    elif kind == 2:    # post-ReLU: half zeros
        x = np.maximum(rng.standard_normal((C, inner)), 0.0) * scale
    else:
        x = rng.standard_normal((C, inner)) * scale
    x = x.astype(dtype)
    if rng.randint(5) == 0 and x.size:
        k = min(x.size, 4)
        x.reshape(-1)[rng.randint(x.size, size=k)] = np.array([np.nan, np.inf, -np.inf, -0.0], dtype)[:k]
    if rng.randint(7) == 0 and x.size:
        x.reshape(-1)[rng.randint(x.size, size=min(x.size, 3))] = dtype(1e-42 if dtype == np.float32 else 1e-310)
    return x


def maxvals(rng, x, n):
    fin = np.abs(x[np.isfinite(x)])
    top = float(fin.max()) if fin.size and fin.max() > 0 else 1.0
    mv = (np.abs(rng.standard_normal(n)) * top + top * 1e-3).astype(np.float32)
    r = rng.randint(12)
    if r == 0:
        mv[rng.randint(n)] = 0.0
    elif r == 1:
        mv[rng.randint(n)] = np.float32(1e-39)
    elif r == 2:
        mv[rng.randint(n)] = np.float32(3e38)
    elif r == 3:
        mv[rng.randint(n)] = np.nan
    return mv


def case_k1(rng):
    pc = bool(rng.randint(2))
    C = int(rng.choice([1, 2, 3, 17, 64, 129, 1000])) if pc else int(rng.choice([1, 4]))
    inner = int(rng.choice(INNERS)) * (1 if pc else int(rng.choice([1, 13, 500])))
    x = data(rng, C, inner)
    M, nb, sb = fmt(rng)
    mv = maxvals(rng, x, C if pc else 1)
    xd, mvd = torch.from_numpy(x).cuda(), torch.from_numpy(mv).cuda()
    ref = oracle.c_quantize(x, mv, M, nb, sb)
    y = ops.quantize(xd, mvd, M, nb, sb).cpu().numpy()
    assert np.array_equal(bits(y), bits(ref)), ("quantize", C, inner, M, nb, sb, pc)
    md = torch.tensor([M], dtype=torch.float32, device="cuda")
    y2 = ops.quantize(xd, mvd, md, nb, sb).cpu().numpy()
    assert np.array_equal(bits(y2), bits(ref)), ("quantize, device mbits", C, inner, M, nb, sb, pc)
    # sign_bits as the device flag of allow_unsigned (fp8q_sign_fold_u8 / fp8q_quantize_ds_f32 / _dms_f32)
    flag = torch.tensor([sb], dtype=torch.uint8, device="cuda")
    y3 = ops.quantize(xd, mvd, M, nb, flag).cpu().numpy()
    assert np.array_equal(bits(y3), bits(ref)), ("quantize, device sign", C, inner, M, nb, sb, pc)
    y4 = ops.quantize(xd, mvd, md, nb, flag).cpu().numpy()
    assert np.array_equal(bits(y4), bits(ref)), ("quantize, device mbits + sign", C, inner, M, nb, sb, pc)
    xm = x.reshape(C, -1).min(1) if x.size else np.zeros(0, np.float32)
    with np.errstate(invalid="ignore"):
        want = 0 if bool(np.all(xm >= 0)) else 1                       # fp8_quantizer.py:216-225 (NaN: not >= 0)
    got = int(ops.sign_fold(torch.from_numpy(np.ascontiguousarray(xm, np.float32)).cuda()).item())
    assert got == want, ("sign_fold", C, inner, got, want)
    return x.size


def case_fused(rng):
    C = int(rng.choice([1, 2, 3, 17, 64, 129, 1000, 4099]))
    inner = int(rng.choice(INNERS))
    x = data(rng, C, inner)
    M, nb, sb = fmt(rng)
    mn, mx = oracle.c_minmax(x, True)
    mv = oracle.c_absmax(mn, mx)
    ref = oracle.c_quantize(x, mv, M, nb, sb)
    y, gmn, gmx, gmv = ops.minmax_quantize(torch.from_numpy(x).cuda(), M, nb, sb)
    gmn, gmx = gmn.cpu().numpy(), gmx.cpu().numpy()
    bad = np.flatnonzero((bits(gmn) != bits(mn)) | (bits(gmx) != bits(mx)))
    assert bad.size == 0, ("fused ranges", C, inner, "row", int(bad[0]), "hip", float(gmn[bad[0]]), float(gmx[bad[0]]), "oracle",
                           float(mn[bad[0]]), float(mx[bad[0]]), "row data", x[bad[0]].tolist()[:32])
    assert np.array_equal(bits(gmv.cpu().numpy()), bits(mv)), ("fused maxval", C, inner)
    assert np.array_equal(bits(y.cpu().numpy()), bits(ref)), ("fused quantize", C, inner, M, nb, sb)
    return x.size


def case_codes(rng):
    pc = bool(rng.randint(2))
    C = int(rng.choice([1, 3, 64, 1000])) if pc else 1
    inner = int(rng.choice(INNERS)) * (1 if pc else 97)
    x = data(rng, C, inner)
    M, nb, sb = fmt(rng)
    nb = 8
    if 8 - sb - min(max(round(M), 1), 8 - sb) < 1:     # the codec refuses formats without an exponent bit
        M = 3.0
    mv = maxvals(rng, x, C if pc else 1)
    mv = np.where(np.isfinite(mv) & (mv > 1e-30) & (mv < 1e30), mv, np.float32(1.0)).astype(np.float32)
    xd, mvd = torch.from_numpy(x).cuda(), torch.from_numpy(mv).cuda()
    ref = oracle.c_encode(x, mv, M, nb, sb)
    got = ops.encode(xd, mvd, M, nb, sb)
    assert np.array_equal(got.cpu().numpy().reshape(-1), ref.reshape(-1)), ("encode", C, inner, M, sb, pc)
    dref = oracle.c_decode(ref, mv, M, nb, sb)
    dec = ops.decode(got, mvd, M, nb, sb).cpu().numpy()
    assert np.array_equal(bits(dec.reshape(-1)), bits(dref.reshape(-1))), ("decode", C, inner, M, sb, pc)
    return x.size


def case_epilogue(rng):
    N = int(rng.choice([1, 2, 8, 64]))
    C = int(rng.choice([1, 3, 16, 96, 320]))
    HW = int(rng.choice([1, 4, 49, 196, 784, 3136]))
    if (C * HW) % 4:
        HW *= 4
    while N * C * HW > 8e6 and N > 1:
        N //= 2
    x = data(rng, N * C, HW).reshape(N, C, HW)
    res = data(rng, N * C, HW).reshape(N, C, HW) if rng.randint(2) else None
    act = int(rng.randint(3))
    bn = None
    if rng.randint(4):
        bn = tuple(t.astype(np.float32) for t in (rng.standard_normal(C), np.abs(rng.standard_normal(C)) + 0.1,
                                                   rng.standard_normal(C), rng.standard_normal(C)))
    M, nb, sb = fmt(rng)
    z = oracle.c_affine_act(x, bn, res, act)
    mv = maxvals(rng, z, 1)
    ref = oracle.c_quantize(z.reshape(1, -1), mv, M, nb, sb).reshape(z.shape)
    xd, mvd = torch.from_numpy(x).cuda(), torch.from_numpy(mv).cuda()
    rd = torch.from_numpy(res).cuda() if res is not None else None
    bnd = tuple(torch.from_numpy(t).cuda() for t in bn) if bn is not None else None
    y = ops.affine_act_quantize(xd, mvd, M, nb, sb, bn=bnd, residual=rd, act=act).cpu().numpy()
    assert np.array_equal(bits(y), bits(ref)), ("epilogue", N, C, HW, act, bn is not None, res is not None, M, nb, sb)
    ab = ops.bn_fold(bnd) if bnd is not None else None
    prep = ops.quantizer_prepare(mvd, M, nb, sb)
    y2 = ops.affine_act_quantize(xd, mvd, M, nb, sb, residual=rd, act=act, bn_ab=ab, prep=prep).cpu().numpy()
    assert np.array_equal(bits(y2), bits(ref)), ("epilogue, folded + prepared", N, C, HW, act, bn is not None, res is not None, M, nb, sb)
    mn, mx, gmv = ops.affine_act_minmax(xd, bn=bnd, residual=rd, act=act)
    rmn, rmx = oracle.c_minmax(z.reshape(1, -1), False)
    assert np.array_equal(bits(mn.cpu().numpy()), bits(rmn)) and np.array_equal(bits(mx.cpu().numpy()), bits(rmx)), ("epilogue min/max", N, C, HW)
    return x.size


def case_multi(rng):
    n = int(rng.randint(1, 40))
    items, refs = [], []
    tot = 0
    for _ in range(n):
        pc = bool(rng.randint(4))
        C = int(rng.choice([1, 8, 64, 512]))
        inner = int(rng.choice([4, 9, 27, 147, 576, 4608, 1000]))
        x = data(rng, C, inner)
        M, nb, sb = fmt(rng)
        mv = maxvals(rng, x, C if pc else 1)
        items.append((torch.from_numpy(x).cuda(), torch.from_numpy(mv).cuda(), M, nb, sb))
        refs.append(oracle.c_quantize(x if pc else x.reshape(1, -1), mv, M, nb, sb).reshape(x.shape))
        tot += x.size
    outs = ops.multi_quantize(items)
    for i, (o, r) in enumerate(zip(outs, refs)):
        assert np.array_equal(bits(o.cpu().numpy()), bits(r)), ("multi_quantize", i, tuple(items[i][0].shape), items[i][2:])
    return tot


def case_f64(rng):
    pc = bool(rng.randint(2))
    C = int(rng.choice([1, 3, 64, 300]))
    inner = int(rng.choice(INNERS))
    x = data(rng, C, inner, np.float64)
    M, nb, sb = fmt(rng)
    mv = maxvals(rng, x, C if pc else 1)
    xd = torch.from_numpy(x).cuda()
    y = ops.quantize(xd, torch.from_numpy(mv).cuda(), M, nb, sb).cpu().numpy()
    ref = oracle.c_quantize_f64(x, mv, M, nb, sb)
    assert np.array_equal(bits(y), bits(ref)), ("quantize f64", C, inner, M, nb, sb, pc)
    return x.size


def case_sorted(rng):
    n = int(rng.choice([1 << 20, (1 << 20) + 1, (1 << 20) + 255, 1500007, (1 << 21) + 77]))
    x = data(rng, 1, n)
    if not np.isfinite(x).all():
        x = np.nan_to_num(x, nan=0.0, posinf=1.0, neginf=-1.0)
    xd = torch.from_numpy(x).cuda()
    top = float(np.abs(x).max()) or 1.0
    grid = torch.linspace(0.1 * top, 1.2 * top, 111, device="cuda").reshape(111, 1).contiguous()
    widths = [1.0, 2.0, 3.0, 4.0, 5.0, 6.0]
    sb = int(rng.choice([1, 1, 0]))                                # unsigned formats: negative elements clip to 0 (error x^2)
    srt = torch.zeros(6, 111, 1, device="cuda")
    ops.mse_grid(xd, False, grid, widths, 8, sb, srt)              # 666 pairs on >= 2^20 elements: the interval-histogram route
    row = torch.zeros(6, 111, 1, device="cuda")
    for i in range(6):                                             # 111 pairs per call on < 4 M elements: the lane-per-element kernel
        ops.mse_grid(xd, False, grid, widths[i:i + 1], 8, sb, row[i:i + 1])
    s, r = srt.cpu().numpy().astype(np.float64), row.cpu().numpy().astype(np.float64)
    floor = 1e-24 * max(r.max(), 1e-300)
    rel = np.abs(s - r) / (np.abs(r) + floor)
    w = int(np.unravel_index(np.argmax(rel), rel.shape)[0]) if rel.max() > 1e-5 else int(rng.randint(6))   # the worst width, if any
    o = oracle.c_mse_grid(x, False, grid.cpu().numpy(), [widths[w]], 8, sb).astype(np.float64)[0]
    es, er = np.abs(s[w] - o) / (np.abs(o) + floor), np.abs(r[w] - o) / (np.abs(o) + floor)
    # include/fp8q.h: both routes within 1e-5 relative of the oracle on EVERY entry (measured ~1e-7): since round 5 the
    # lane-per-element kernels re-evaluate elements near a rounding tie with the reference's division
    if es.max() > 1e-5 or er.max() > 1e-5:
        i = int(np.argmax(np.maximum(es, er)))
        np.savez(os.path.join(ROOT, "gpurun_out", "soak_fail_k4.npz"), x=x, grid=grid.cpu().numpy(), sorted=s, row=r, oracle_w=o, w=w)
        raise AssertionError(("K4 vs oracle", n, "width", widths[w], "candidate", i, "sorted", float(s[w, i, 0]), "row", float(r[w, i, 0]),
                              "sign_bits", sb, "oracle", float(o[i, 0]), "rel sorted", float(es.max()), "rel row", float(er.max()), "data kind", LAST_KIND[0]))
    return n * 666


def case_ranges(rng):
    """K2/K3/K5 with the three folds (+ the packed record of the range all-reduce)"""
    pc = bool(rng.randint(2))
    C = int(rng.choice([1, 2, 3, 17, 64, 129, 1000, 4099])) if pc else 1
    inner = int(rng.choice(INNERS)) * (1 if pc else int(rng.choice([1, 97, 3001])))
    x = data(rng, C, inner)
    xd = torch.from_numpy(x).cuda()
    rows = C if pc else 1
    mode = int(rng.randint(3))
    mom = float(rng.choice([0.9, 0.5, 0.99]))
    mn, mx = oracle.c_minmax(x, pc)
    if rng.randint(2):      # a previous estimate to fold into
        pmn = (-np.abs(rng.standard_normal(rows))).astype(np.float32)
        pmx = np.abs(rng.standard_normal(rows)).astype(np.float32)
        if rng.randint(6) == 0:
            pmn[rng.randint(rows)] = np.nan
        rmn, rmx = oracle.c_fold(pmn, pmx, mn, mx, mode, mom)
        cmn, cmx = torch.from_numpy(pmn.copy()).cuda(), torch.from_numpy(pmx.copy()).cuda()
        packed = ops.new_packed(rows, "cuda") if rng.randint(2) else None
        gmn, gmx, gmv = ops.minmax(xd, pc, cmn, cmx, mode=mode, momentum=mom, want_maxval=True, packed=packed)
    else:
        rmn, rmx = mn, mx
        packed = ops.new_packed(rows, "cuda") if rng.randint(2) else None
        gmn, gmx, gmv = ops.minmax(xd, pc, mode=mode, momentum=mom, want_maxval=True, packed=packed)
    rmv = oracle.c_absmax(rmn, rmx)
    got = [t.cpu().numpy().reshape(-1) for t in (gmn, gmx, gmv)]
    for name, g, r in zip(("min", "max", "maxval"), got, (rmn, rmx, rmv)):
        bad = np.flatnonzero(bits(g) != bits(r))
        assert bad.size == 0, ("ranges " + name, C, inner, pc, mode, "row", int(bad[0]), float(g[bad[0]]), float(r[bad[0]]))
    if packed is not None:
        a, b, c = ops.ranges_unpack(packed)
        for name, g, r in zip(("min", "max", "maxval"), (a, b, c), (rmn, rmx, rmv)):
            g = g.cpu().numpy().reshape(-1)
            # the record carries -min: a zero minimum comes back as the negated zero of what was stored
            assert np.array_equal(np.where(np.isnan(g), np.float32(7), g), np.where(np.isnan(r), np.float32(7), r)), ("packed " + name, C, inner, pc, mode)
    return x.size


def case_mse_small(rng):
    """K4 on per-channel weights and short / mid rows (k_mse_grid, k_mse_row), against the oracle at the stated tolerance"""
    pc = bool(rng.randint(3))
    C = int(rng.choice([1, 3, 32, 96, 320])) if pc else 1
    inner = int(rng.choice([1, 9, 27, 147, 576, 2047, 2048, 4608, 50000]))
    x = data(rng, C, inner)
    x = np.nan_to_num(x, nan=0.25, posinf=3.0, neginf=-3.0)
    M, nb, sb = fmt(rng)
    n_cand = int(rng.choice([1, 7, 111]))
    widths = [M] if rng.randint(2) else [1.0, 3.0, 4.0]
    rows = C if pc else 1
    top = np.abs(x.reshape(rows, -1)).max(1) + 1e-6
    grid = (np.linspace(0.1, 1.2, n_cand)[:, None] * top[None, :]).astype(np.float32)
    out = torch.zeros(len(widths), n_cand, rows, device="cuda")
    ops.mse_grid(torch.from_numpy(x).cuda(), pc, torch.from_numpy(grid).cuda(), widths, nb, sb, out)
    ref = oracle.c_mse_grid(x, pc, grid, widths, nb, sb).astype(np.float64)
    got = out.cpu().numpy().astype(np.float64)
    badn = np.argwhere(np.isnan(got) != np.isnan(ref))
    if badn.size:
        wi, ci, ri = (int(v) for v in badn[0])
        np.savez(os.path.join(ROOT, "gpurun_out", "soak_fail_k4_small.npz"), x=x, grid=grid, got=got, ref=ref)
        raise AssertionError(("K4 NaN pattern", C, inner, pc, widths, nb, sb, "at (width, cand, row)", (wi, ci, ri), "hip", float(got[wi, ci, ri]),
                              "oracle", float(ref[wi, ci, ri]), "candidate", float(grid[ci, ri]), "row data", x.reshape(rows, -1)[ri].tolist()[:16],
                              "mismatches", int(badn.shape[0])))
    ok = ~np.isnan(ref)
    rel = np.abs(got[ok] - ref[ok]) / (np.abs(ref[ok]) + 1e-30 * (np.abs(ref[ok]).max() if ok.any() else 1.0) + 1e-300)
    assert not rel.size or rel.max() <= 1e-5, ("K4 small rows vs oracle", C, inner, pc, widths, nb, sb, n_cand, float(rel.max()))
    return x.size * n_cand * len(widths)


def case_select(rng):
    """device-side search grid (bit-equal to torch.linspace) and selection (torch.min / argmin / mode semantics)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ops
    C = int(rng.choice([1, 2, 7, 64, 1000]))
    n_m = int(rng.choice([1, 2, 6]))
    n_cand = int(rng.choice([2, 111]))
    mx = (np.abs(rng.standard_normal(C)) * 10.0 ** rng.uniform(-6, 6) + 1e-30).astype(np.float32)
    if rng.randint(5) == 0:
        mx[rng.randint(C)] = 0.0
    g = ops.mse_linspace(torch.from_numpy(mx).cuda(), n_cand)
    gr = oracle_ops.mse_linspace(torch.from_numpy(mx), n_cand)
    assert np.array_equal(bits(g.cpu().numpy()), bits(gr.numpy())), ("linspace", C, n_cand)
    mses = np.abs(rng.standard_normal((n_m, n_cand, C))).astype(np.float32)
    r = rng.randint(4)
    if r == 0:      # ties everywhere: a few distinct values
        mses = np.round(mses * 2) / 2
    elif r == 1:
        mses.reshape(-1)[rng.randint(mses.size, size=3)] = np.nan
    elif r == 2:
        mses.reshape(-1)[rng.randint(mses.size, size=3)] = np.inf
    widths = [float(i + 1) for i in range(n_m)]
    got = ops.mse_select(torch.from_numpy(mses).cuda(), g, widths, 1)
    ref = oracle_ops.mse_select(torch.from_numpy(mses), gr, widths, 1)
    for name, a, b in zip(("mbits", "vote", "maxval", "minval"), got, ref):
        assert np.array_equal(bits(a.cpu().numpy().astype(np.float32)), bits(b.numpy().astype(np.float32))), ("select " + name, C, n_m, n_cand, r)
    return mses.size


def case_multi_ranges(rng):
    """per-channel ranges + quantize of many tensors in two launches"""
    n = int(rng.randint(1, 30))
    items, refs = [], []
    tot = 0
    for _ in range(n):
        C = int(rng.choice([1, 8, 64, 512]))
        inner = int(rng.choice([4, 9, 27, 147, 576, 4608, 1000]))
        x = data(rng, C, inner)
        M, nb, sb = fmt(rng)
        mn, mx = oracle.c_minmax(x, True)
        mv = oracle.c_absmax(mn, mx)
        refs.append((oracle.c_quantize(x, mv, M, nb, sb), mv))
        items.append((torch.from_numpy(x).cuda(), torch.empty(C, device="cuda"), M, nb, sb))
        tot += x.size
    outs = ops.multi_minmax_quantize(items)
    for i, (o, (r, mv)) in enumerate(zip(outs, refs)):
        assert np.array_equal(bits(items[i][1].cpu().numpy()), bits(mv)), ("multi ranges", i, tuple(items[i][0].shape))
        assert np.array_equal(bits(o.cpu().numpy()), bits(r)), ("multi_minmax_quantize", i, tuple(items[i][0].shape), items[i][2:])
    return tot


def case_f64_search(rng):
    """float64 lane: row min/max and the candidate search (sum and mean)"""
    pc = bool(rng.randint(2))
    C = int(rng.choice([1, 3, 64]))
    inner = int(rng.choice([1, 5, 147, 2049, 10007, 200000]))
    x = data(rng, C, inner, np.float64)
    x = np.nan_to_num(x, nan=0.25, posinf=3.0, neginf=-3.0)
    xd = torch.from_numpy(x).cuda()
    mn, mx = ops.minmax_f64(xd, pc)
    rmn, rmx = oracle.c_minmax_f64(x, pc)
    assert np.array_equal(bits(mn.cpu().numpy()), bits(rmn)) and np.array_equal(bits(mx.cpu().numpy()), bits(rmx)), ("f64 min/max", C, inner, pc)
    rows = C if pc else 1
    n_cand = int(rng.choice([3, 100]))
    top = np.abs(x.reshape(rows, -1)).max(1) + 1e-9
    grid = (np.linspace(0.05, 1.1, n_cand)[:, None] * top[None, :]).astype(np.float32)
    widths = [float(rng.choice([1, 2, 3, 4, 5]))]
    red = str(rng.choice(["sum", "mean"]))
    out = torch.zeros(1, n_cand, rows, dtype=torch.float64, device="cuda")
    ops.mse_grid_f64(xd, pc, torch.from_numpy(grid).cuda(), widths, 8, 1, out, reduce=red)
    ref = oracle.c_sse_grid_f64(x, pc, grid, widths, 8, 1, reduce=red)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-12, atol=1e-300, err_msg=str(("f64 search", C, inner, pc, widths, red)))
    return x.size * n_cand


def case_one_call(rng):
    """the one-call MSE calibration step (fp8q_mse_calibrate_f32, with and without the BN + activation pre-stage) over two batches
    against the entry points it replaces called one after the other: tables, grid, range, width and output bit for bit"""
    pc = bool(rng.randint(2))
    pre = (not pc) and bool(rng.randint(2))
    if pre:
        N, C, HW = int(rng.choice([1, 3, 16])), int(rng.choice([4, 12, 96])), int(rng.choice([1, 9, 49, 196, 3136]))
        shape, rows = (N, C, HW), 1
    else:
        rows = int(rng.choice([1, 3, 32, 96, 320])) if pc else 1
        inner = int(rng.choice([1, 9, 147, 576, 2047, 2048, 4608, 70000, 300000, 1200000])) if rows <= 32 else int(rng.choice([1, 9, 147, 576, 2047]))
        shape = (rows, inner)
    n_bits = int(rng.choice([8, 8, 6]))
    sb = int(rng.choice([1, 1, 0]))
    hi = n_bits - sb
    widths = [float(m) for m in range(1, hi)] if rng.randint(2) else [float(rng.randint(1, hi))]
    ab = res = None
    act = 0
    if pre:
        C = shape[1]
        ab = torch.from_numpy(np.stack([rng.uniform(0.5, 1.5, C), rng.standard_normal(C) * 0.2], 1).astype(np.float32)).cuda()
        act = int(rng.randint(3))
        if rng.randint(3) == 0:
            res = True
        if rng.randint(4) == 0:
            ab = None
            if res is None and act == 0:
                act = 1
    a = ops.MseCalibration(rows, torch.device("cuda", 0), widths, n_bits, sb)
    g = m = None
    n_el = 0
    for batch in range(2):
        x = data(rng, int(np.prod(shape[:-1])), shape[-1]).reshape(shape)
        x = np.nan_to_num(x, nan=0.25, posinf=3.0, neginf=-3.0)
        xd = torch.from_numpy(x).cuda()
        rd = torch.from_numpy(data(rng, int(np.prod(shape[:-1])), shape[-1]).reshape(shape)).cuda().nan_to_num(0.5, 1.0, -1.0) if res else None
        ya = a.step(xd, pre=(ab, rd, act) if pre else None)
        t = ops.affine_act(xd, ab, rd, act) if pre else xd
        if batch == 0:
            _, _, mv0, g = ops.minmax_linspace(t, pc, 111)
            g = g.contiguous()
            m = torch.zeros(len(widths), 111, rows, device="cuda")
            assert np.array_equal(bits(a.absmax.cpu().numpy()), bits(mv0.cpu().numpy())), ("one-call absmax", shape, pc, pre)
        ops.mse_grid(t, pc, g, widths, n_bits, sb, m)
        mb, vote, maxval, xmin = ops.mse_select(m, g, widths, sb)
        yb = ops.quantize(t, maxval, mb if len(widths) > 1 else widths[0], n_bits, sb)
        what = ("one-call step", shape, pc, pre, widths, n_bits, sb, act, batch)
        assert np.array_equal(bits(a.grid.cpu().numpy()), bits(g.cpu().numpy())), what + ("grid",)
        assert np.array_equal(bits(a.mses.cpu().numpy()), bits(m.cpu().numpy())), what + ("table",)
        assert np.array_equal(bits(a.maxval.cpu().numpy()), bits(maxval.cpu().numpy())) and float(a.mbits.cpu()) == float(mb.cpu()), what + ("choice",)
        assert np.array_equal(bits(ya.cpu().numpy()), bits(yb.cpu().numpy())), what + ("output",)
        n_el += x.size
    return n_el * 111 * len(widths)


FAMILIES = [("one-call MSE calibration step", case_one_call, 3), ("K1 fp32 (+ device mantissa width / sign)", case_k1, 6), ("fused min/max + quantize", case_fused, 4),
            ("storage codes", case_codes, 3), ("epilogue (bn / folded / prepared / min-max)", case_epilogue, 4),
            ("multi-tensor K1", case_multi, 1), ("K1 fp64", case_f64, 3), ("K4 interval histogram vs row kernel vs oracle", case_sorted, 1), ("ranges: min/max, folds, packed record", case_ranges, 4),
            ("K4 per-channel / short rows", case_mse_small, 3), ("search grid + selection", case_select, 2),
            ("multi-tensor ranges + K1", case_multi_ranges, 1), ("fp64 min/max + search", case_f64_search, 2)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    fp8q.lib()
    rng = np.random.RandomState(args.seed)
    stats = {name: [0, 0] for name, _, _ in FAMILIES}
    t0 = time.time()
    it = 0
    while time.time() - t0 < args.seconds:
        for name, fn, weight in FAMILIES:
            for _ in range(weight):
                try:
                    stats[name][1] += fn(rng)
                except Exception as e:     # a mismatch (AssertionError) or an error code the oracle did not raise
                    print(f"FAILED after {it} rounds in '{name}': {type(e).__name__}: {e.args[0] if e.args else e}", flush=True)
                    sys.exit(1)
                stats[name][0] += 1
        ops.check_workspaces()
        it += 1
    print(f"# tests/soak.py --seconds {args.seconds:g} --seed {args.seed}: {it} rounds in {time.time() - t0:.0f} s on "
          f"{torch.cuda.get_device_name(0)}; every comparison bit-exact against oracle/ (K4: within its stated tolerance)")
    for name, (cases, elems) in stats.items():
        print(f"{name:48s} cases {cases:6d}   elements {elems:.3e}")
    print("OK")


if __name__ == "__main__":
    main()
