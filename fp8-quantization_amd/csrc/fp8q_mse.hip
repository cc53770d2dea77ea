// fp8q_mse.hip -- K4: FP-MSE grid search (FP_MSE_Estimator / LineSearchEstimator candidates in one pass over x).
#include "fp8q_common.h"
#include "fp8q_select.h"

namespace {

// ---------------------------------------------------------------------------------------------
// K4: FP-MSE grid search (range_estimators.py:337-347), ALU-bound.
// One lane = one candidate maxval; every lane walks the SAME x tile, broadcast out of LDS, so a
// candidate's squared error accumulates in one register and no cross-lane reduction exists.
// Block = 128 lanes (candidates i0..i0+127 of one mantissa width m, one row c, one split of the
// row: MseArgs::tile elements per trip, mse_tile()).  Per-candidate scale table s_p in LDS, the
// same numbers K1 uses (scale_exact() / its ldexp shortcut); the quotient is formed from the bits
// of xc * 2^bf and every element within 5 ulps of a rounding tie is redone with the reference's
// own binade decision and IEEE division (see the loop).
// Dynamic LDS: float xs[kMseTile] | float lut[128 * stride], stride = (pmax + 1) | 1
// ---------------------------------------------------------------------------------------------
constexpr int kMseBlock = 128;
constexpr int kMseTile = 2048;
constexpr int kMseMaxM = 8;

struct MseArgs {
    QFmt fmt[kMseMaxM];
    int n_m;
    int n_cand;
    int cgroups;   // ceil(n_cand / 128)
    int nsplit;
    int tile;      // k_mse_grid: elements of a row per workgroup and trip (<= kMseTile, a multiple of 32): mse_tile()
    int64_t inner;
    int64_t C;
    int overwrite;   // the table entries are written, not added to (first batch of fp8q_mse_calibrate_f32: no cleared table needed;
                     // same bits as adding to +0: an entry is never -0)
    // k_mse_grid on the FIRST calibration batch of a per-channel quantizer (fp8q_mse_calibrate_f32): the search grid does not
    // exist yet -- every workgroup finds its row's min / max itself (rows < 2048 elements: a few loads per thread from L2),
    // takes its candidate from linspace_at(max|row|, 0.1, 1.2, n_cand, cand), and the row's first workgroup writes the range
    // and the grid column for the batches to come.  Saves the separate abs-max + grid launch (12 us x 53 weight tensors).
    float *first_min, *first_max, *first_absmax, *first_grid;     // all NULL: the grid is given
};

// (an agent-scope store: the last workgroup of the launch may read the entry for the winner selection, fp8q_select.h)
__device__ __forceinline__ void table_add(float *p, float v, int overwrite) { agent_store(p, overwrite ? v : *p + v); }

__global__ void __launch_bounds__(kMseBlock)
k_mse_grid(const float *__restrict__ x, const float *__restrict__ grid, double *__restrict__ ws,
           MseArgs a, float *__restrict__ mses, double inv_inner)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *xs = reinterpret_cast<float *>(smem);
    const int tid = threadIdx.x;
    const int split = blockIdx.x;
    const int m = blockIdx.y / a.cgroups;
    const int cand = (blockIdx.y - m * a.cgroups) * kMseBlock + tid;
    const int64_t c = blockIdx.z;
    const QFmt f = a.fmt[m];
    const int stride = (f.pmax + 1) | 1;       // odd: lanes at the same p hit different banks
    float *lut = xs + kMseTile + tid * stride;   // s_p of this candidate, exact (scale_exact)
    const bool active = cand < a.n_cand;

    float gv;
    if (a.first_grid) {
        // the row's min / max as fp8q_minmax_f32 forms them (mm_acc: NaN flag, IEEE min / max), |max(|min|, max)|, the grid point
        __shared__ float s_mn[kMseBlock / 64], s_mx[kMseBlock / 64];
        __shared__ int s_nan[kMseBlock / 64];
        const float *xrow = x + c * a.inner;
        MinMax mm;
        mm_init(mm);
        for (int64_t i = tid; i < a.inner; i += kMseBlock) mm_acc(mm, xrow[i]);
        mm_wave_reduce(mm);
        if ((tid & 63) == 0) {
            s_mn[tid >> 6] = mm.mn;
            s_mx[tid >> 6] = mm.mx;
            s_nan[tid >> 6] = mm.nan;
        }
        __syncthreads();
        float mn = fminf(s_mn[0], s_mn[1]), mx = fmaxf(s_mx[0], s_mx[1]);
        if (s_nan[0] | s_nan[1]) mn = mx = __builtin_nanf("");
        const float absmax = fabsf(tmax(fabsf(mn), mx));               // fp8_quantizer.py:236
        gv = active ? linspace_at(absmax, 0.1, 1.2, a.n_cand, cand) : 1.0f;
        if (split == 0 && m == 0) {
            if (active) a.first_grid[(int64_t)cand * a.C + c] = gv;
            if (blockIdx.y == 0 && tid == 0) {
                a.first_min[c] = mn;
                a.first_max[c] = mx;
                a.first_absmax[c] = absmax;
            }
        }
    } else {
        // set_quant_range(-g, g): maxval = |max(|-g|, g)|  (fp8_quantizer.py:236)
        gv = active ? grid[(int64_t)cand * a.C + c] : 1.0f;
    }
    const Chan ch = make_chan(fabsf(fmaxf(fabsf(-gv), gv)), f);
    lut[0] = __builtin_nanf("");
    {
        // lut_row()'s shortcut: when fl32(k - bias) is exact at both ends of the table it is exact in between and every
        // entry is ldexp(fl32(g), k - bi) -- one instruction instead of scale_exact()'s ~30 double-precision operations
        // (for short rows -- MobileNetV2's [C, 1, 3, 3] and [C_out, C_in] weights -- building 64-entry tables per
        // candidate was most of the kernel: 70 us per call with the mantissa search)
        const float k1 = 1.0f - f.M, kp = (float)f.pmax - f.M;
        const bool lin = ch.pthr >= 0.0f && (k1 - (k1 - ch.bias)) == ch.bias && (kp - (kp - ch.bias)) == ch.bias;
        if (lin) {
            const int j0 = (int)k1 - ch.bi - 1;
            for (int p = 1; p <= f.pmax; ++p) lut[p] = ldexpf(ch.m0, j0 + p);
        } else {
            for (int p = 1; p <= f.pmax; ++p) lut[p] = scale_exact(ch, (float)p, f.M);
        }
    }
    // p = floor(log2|xc| + bias) without a logarithm: with bias = bi + bf, log2|xc| + bias = log2(|xc| 2^bf) + bi,
    // so p is the exponent field of fl32(|xc| * 2^bf) plus a constant.  An element within a few ulps of a binade
    // border can land on either side; both sides give the same grid point there (2^(M+1) steps of s_p = 2^M
    // steps of s_(p+1)), so the squared error is unaffected beyond fp32 rounding.  A non-finite bias makes c1
    // NaN -> exponent 255 -> p = pmax, whose entry is NaN, like the reference.
    // 1/s_p is not tabulated (the table is what limits occupancy): 1/s_p = 2^bf * 2^(M + bi - p) up to the
    // fp32 rounding of the table entry (<= 3e-6 relative), which can only move r = rint(xc / s_p) at an exact
    // tie, where both neighbours are equally far from x.
    const float c1 = (float)(1.0 / ch.g);      // 2^bf in [1, 2)
    // in terms of the raw exponent field e8 of t = xc * c1:  p = clamp(e8 + koff, 1, pmax), koff = bi - 127
    const int koff = ch.bi - 127;
    const int e_lo = 1 - koff, e_hi = f.pmax - koff;          // clamp bounds for e8
    const float *lutk = lut + koff;                             // lutk[e8] == lut[p]
    const int jk = (int)f.M + ch.bi - koff;                     // ldexp exponent M + bi - p == jk - e8
    const float *xr = x + c * a.inner;
    // E = 0 formats: see the loop.  Ranges outside the fast path's preconditions (Chan::pthr < 0: |bias| >= 100, i.e. a
    // scale or its reciprocal may be denormal, zero or infinite -- E = 7 with maxval < ~2^(M-22): s_1 underflows to 0 and
    // the reference's 0 / 0 makes every zero element NaN) divide as the reference does, like k_mse_row's exact path.
    const bool exact_div = f.pmax == 1 || ch.pthr < 0.0f;
    const float tie_thr = 0.5f - 5.0f * __builtin_ldexpf(1.0f, (int)f.M - 23);   // (kTieW of the row kernel; f.M <= 16)
    double acc = 0.0;

    const int tile = a.tile;
    for (int64_t t0 = (int64_t)split * tile; t0 < a.inner; t0 += (int64_t)a.nsplit * tile) {
        const int n = (int)((a.inner - t0) < tile ? (a.inner - t0) : tile);
        __syncthreads();
        for (int i = tid; i < tile; i += kMseBlock) xs[i] = i < n ? xr[t0 + i] : 0.0f;
        __syncthreads();
        // zero padding: q(0) = 0 exactly, contributes nothing (the exact-division path masks it: there q(0) can be NaN)
        const int n32 = (n + 31) & ~31;
        for (int j = 0; j < n32; j += 32) {
            // Branch-free pass over the group: the fast quotient for every element + how close any of them came to a tie.
            // (One compare-and-branch per element -- v_cmp -> s_and_saveexec -> s_cbranch, a VALU -> SALU round trip the
            // next element's work cannot overlap at the 2 waves per SIMD the per-lane tables allow -- made the loop ~3x
            // slower than its 13 VALU operations.)  A lane whose sub-group held a near-tie, or whose candidate needs the exact
            // division, repeats it element by element below with the reference's own division.
            // Four sub-groups of 8 elements, each with its own partial sum and tie distance: a redo costs 8 elements, not 32
            // (the search grid's last candidate puts the row's largest element on a tie for M = 1, 3, 5 -- every row pays
            // one redo --, and at M = 6 / 7 a tenth of all 32-element groups would hold a near-tie in some lane).
            float ps[4], nd[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float pa = 0.0f, dmax = exact_div ? __builtin_inff() : 0.0f;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float4 v = *reinterpret_cast<const float4 *>(xs + j + g * 8 + u * 4);
                    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float xv = e[q];
                        const float xc = __builtin_amdgcn_fmed3f(xv, ch.minv, ch.maxv);
                        const float tt = xc * c1;
                        int e8 = (int)__builtin_amdgcn_ubfe(__float_as_uint(tt), 23u, 8u);
                        e8 = max(min(e8, e_hi), e_lo);
                        const float sc = lutk[e8];
                        const float qf = ldexpf(tt, jk - e8);
                        const float r = rintf(qf);
                        dmax = fmaxf(dmax, fabsf(qf - r));      // (a NaN quotient is not "near": it takes the fast formula, as before)
                        const float d = xv - r * sc;
                        pa = fmaf(d, d, pa);
                    }
                }
                ps[g] = pa;
                nd[g] = dmax;
            }
            if (__builtin_expect(fmaxf(fmaxf(nd[0], nd[1]), fmaxf(nd[2], nd[3])) >= tie_thr, 0)) {
#pragma unroll 1
                for (int g = 0; g < 4; ++g) {
                    const float dmax = g == 0 ? nd[0] : g == 1 ? nd[1] : g == 2 ? nd[2] : nd[3];
                    if (!(dmax >= tie_thr)) continue;
                    float pa = 0.0f;
#pragma unroll 1
                    for (int u = 0; u < 2; ++u) {
                        const float4 v = *reinterpret_cast<const float4 *>(xs + j + g * 8 + u * 4);
                        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float xv = e[q];
                            const float xc = __builtin_amdgcn_fmed3f(xv, ch.minv, ch.maxv);
                            const float tt = xc * c1;
                            int e8 = (int)__builtin_amdgcn_ubfe(__float_as_uint(tt), 23u, 8u);
                            e8 = max(min(e8, e_hi), e_lo);
                            float sc = lutk[e8];
                            // E = 0 (mantissa bits = n_bits - sign_bits): maxval / s_1 = 2^M - 0.5 is an exact TIE, so every
                            // clipped element sits on one and the fp32 rounding of the quotient decides all of them at
                            // once: there the IEEE division of the reference is reproduced
                            float r;
                            if (exact_div) {
                                // (bias > 128: binade 1 lies below the exponent field's range, a zero or tiny t would read
                                // e8 = 0 as binade bi - 127 -- K1's exact decision instead)
                                const float ls = __builtin_amdgcn_fmed3f(floorf(log2_tab(fabsf(xc), kFastTab) + ch.bias), 1.0f, (float)f.pmax);
                                sc = lut[(int)ls];
                                r = rintf(xc / sc);
                                if (j + g * 8 + u * 4 + q >= n) r = sc = 0.0f;   // zero padding: here q(0) may be NaN (s_1 = 0), a real zero's is
                            } else {
                                const float qf = ldexpf(tt, jk - e8);
                                r = rintf(qf);
                                // within kTieW ulps of a rounding tie (t carries up to 4 ulps of error against the reference's
                                // quotient): the reference's own binade decision and division for this element (rare:
                                // ~1e-5 2^M of the (element, candidate) pairs)
                                if (fabsf(qf - r) >= tie_thr) {
                                    const float ls = __builtin_amdgcn_fmed3f(floorf(log2_tab(fabsf(xc), kFastTab) + ch.bias), 1.0f, (float)f.pmax);
                                    sc = lut[(int)ls];
                                    r = rintf(xc / sc);
                                }
                            }
                            const float d = xv - r * sc;
                            pa = fmaf(d, d, pa);
                        }
                    }
                    if (g == 0) ps[0] = pa; else if (g == 1) ps[1] = pa; else if (g == 2) ps[2] = pa; else ps[3] = pa;
                }
            }
            const float pa = (ps[0] + ps[1]) + (ps[2] + ps[3]);
            acc += (double)pa;
        }
    }
    if (!active) return;
    // a row that one workgroup covers (nsplit == 1: depthwise and narrow pointwise weights, half of MobileNetV2's tensors)
    // needs no second launch: the table entry is this lane's alone -- the same (float)(sum * inv_inner) k_mse_final_tile forms
    if (a.nsplit == 1)
        table_add(mses + ((int64_t)m * a.n_cand + cand) * a.C + c, (float)((0.0 + acc) * inv_inner), a.overwrite);
    else
        ws[((c * a.n_m + m) * a.n_cand + cand) * a.nsplit + split] = acc;
}

// ---------------------------------------------------------------------------------------------
// K4 for LONG rows (per-tensor activations: 99 % of config 4's grid-search work): lane = ELEMENT.
// One wave = one block owns tiles of 4096 consecutive elements of one row (64 per lane, in registers: with the 4-7-slot
// loops the per-candidate reduction and constant hand-over were a fifth of the work at 32 per lane) and walks the
// candidates of its group one after the other; a candidate's constants are wave-uniform (broadcast LDS reads), its
// squared error is summed over the lane's 64 elements in two fp32 accumulators, reduced across the wave with DPP adds
// and added to the candidate's double accumulator in LDS.  No per-candidate table and no logarithm: with
// bias = bi + bf, t = xc * 2^bf has the element's binade p - bi in its exponent field, so rounding xc to the format's
// grid is rounding t to M fraction bits -- the float magic-number trick (t + C) - C with C = 1.5 * 2^(e' + 23 - M),
// e' = max(exponent(t), exponent of binade 1) (subnormal range: fixed step) -- and y = round(t) * fl32(2^-bf) is the
// very product r * s_p K1 forms whenever the channel's scales are ldexp(fl32(2^-bf), .) ("lin", lut_row(): every
// range with |k - bias| inside bias's binade).  Per candidate-element: 1 v_med3 + 3 integer ops + 6 fp32 ops that run
// two-wide (v_pk_mul / v_pk_add / v_pk_fma) = 7 issue slots instead of 12 + a ds_read; candidates that are not "lin"
// (or out of the fast range) take an exact per-element path -- same results as the table kernel either way.
// Rounding differs from IEEE x / s_p only where the quotient is within ~2.4e-7 relative of a tie (see k_mse_grid).
// ---------------------------------------------------------------------------------------------
constexpr int kMseRowTile = 4096;     // elements per wave and tile: 64 per lane
constexpr int kMseRowEpl = 64;
constexpr int kMseRowGroup = 128;     // candidates per block (LDS: 48 B each; two double accumulators per lane)
constexpr int kMseRowMinInner = 2048; // shorter rows: k_mse_grid (lane = candidate)
constexpr bool kIntRound = true;      // bit-level round-half-UP-on-the-magnitude fast loops (mse_cand_int: differs from half-to-even only on exact ties, same squared error); false: the float magic-number loops only

typedef float vf2 __attribute__((ext_vector_type(2)));

struct __attribute__((aligned(16))) CandK {
    float maxv, minv, c1, m0;     // clamp bounds, 2^bf, scale mantissa: s_p = m0 * 2^(p - M - bi)
    uint32_t lo, kadd;            // exponent field of binade 1 in t's domain (<< 23); ((23 - M) << 23) | 0x400000
    int fast;                     // 1: one scale mantissa; 2: two (m0b below exponent field `thr`); 0: exact path
    int m;                        // index into MseArgs::fmt
    float m0b;
    uint32_t thr;
    int pad[2];
};

// The scales of a channel are s_p = 2^fl32((p - M) - bias).  fl32() is exact while |(p - M) - bias| stays inside bias's
// binade; beyond it the rounding error is the same for every p whose difference has the same sign and binade (p - M is
// an integer: it does not touch the fraction bits that get rounded).  So s_p / 2^(p - M - bi) takes one value per such
// run of p -- one for 90 % of all ranges, two for 8 % (checked over 60 000 random ranges x formats on the CPU) -- and
// scale_exact() is needed once per run, not once per p.  Returns the number of distinct consecutive values (1, 2, or 3 =
// "more": exact path), the first two of them and the first p of the second run.
__device__ __forceinline__ int scale_mantissa_runs(const Chan &ch, const QFmt &f, float &m_first, float &m_second, int &pswitch)
{
    const uint32_t expb = (__float_as_uint(ch.bias) >> 23) & 0xffu;
    auto run_id = [&](int p) -> uint32_t {
        const float e = ((float)p - f.M) - ch.bias;
        const uint32_t ie = (__float_as_uint(e) >> 23) & 0xffu;
        return ie > expb ? (ie | ((__float_as_uint(e) >> 31) << 8)) : 0u;
    };
    auto mant = [&](int p) -> float { return ldexpf(scale_exact(ch, (float)p, f.M), -((p - (int)f.M) - ch.bi)); };
    m_first = m_second = mant(1);
    pswitch = 0;
    int nruns = 1;
    uint32_t prev = run_id(1);
    for (int p = 2; p <= f.pmax; ++p) {
        const uint32_t cur = run_id(p);
        if (cur != prev) {
            prev = cur;
            const float mm = mant(p);
            if (mm != m_second) {
                if (++nruns > 2) return 3;
                m_second = mm;
                pswitch = p;
            }
        }
    }
    return nruns;
}

__device__ __forceinline__ float wave_sum(float v)
{
    // DPP adds inside each row of 16 lanes, then the four row sums through readlane (wave-uniform result)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    const int b = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
}

// wave-uniform min / max of a per-lane value (once per tile: the shuffle's LDS-crossbar latency does not matter here)
__device__ __forceinline__ float wave_min_f(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

__device__ __forceinline__ float wave_max_f(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// Squared error of one candidate over the lane's 64 elements (kMseRowEpl) when NO nonzero element of the tile lies below the
// candidate's first binade (the caller checks it with the tile's smallest nonzero magnitude): rounding t = xc * 2^bf to
// M fraction bits is then rounding on the BITS of t -- r = (bits + half) & ~mask (v_add_u32, v_and_b32) -- with no
// exponent extraction, no clamp of it to binade 1 and no float add / sub pair: 5 issue slots per element instead of 7, 4
// when the candidate's range covers the whole tile (CLAMP = false: the v_med3 goes too).  Zero stays zero; a carry out of the fraction moves the value into the next binade, as rounding
// up must.  TWO = the candidate has two scale mantissas (m0b below |t| = thr).
//
// Near ties (round 5).  t carries up to 4 ulps of error against the reference's fl32(xc / s_p) (2^bf and 1 / fl32(2^-bf) differ
// by a rounding each, the product and the reference's quotient are rounded once more), so an element whose quotient lies
// within 4 ulps of r + 1/2 may take the other neighbour -- its squared error moves by up to 4 * 2^-21 * 2^(M+1) of itself,
// which shows in a table entry that a few elements carry (the search grid's last candidate 1.2 max|x| puts the largest element
// on such a tie for M = 1, 3, 5).  Both loops therefore track how close the lane's elements come to a tie -- `near`: the
// smallest distance in ulps of t, biased by kTieW, as an unsigned number shifted to the top of the word (one v_lshl_add_u32 +
// one v_min_u32 per element) / `nearf`: the largest |t - round(t)| - (half step - kTieW ulps) -- and the caller exchanges
// the squared error of the elements that came within kTieW ulps for the reference's (mse_tie_patch).  Exact ties round half up here
// and half to even there: same distance, same squared error to the last bits of r * m0 -- they are re-evaluated too.
constexpr uint32_t kTieW = 5u;    // ulps of t around a tie that count as "near" (the bound above is 4)

// (a << k) + c in one instruction (written as C the compiler re-associates it into an add and a shift; one SGPR operand at most)
__device__ __forceinline__ uint32_t lshl_add(uint32_t a, uint32_t k, uint32_t c)
{
    uint32_t d;
    asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(k), "v"(c));
    return d;
}

template <bool CLAMP, bool TWO>
__device__ __forceinline__ vf2 mse_cand_int(const vf2 (&xv)[kMseRowEpl / 2], float minv, float maxv, float c1s, float m0s, float m0b, float thr,
                                            uint32_t half, uint32_t msk, uint32_t ksh, uint32_t nadd, uint32_t &near)
{
    const vf2 c1 = {c1s, c1s}, m0 = {m0s, m0s};
    vf2 pa = {0.0f, 0.0f};
    uint32_t mn = 0xffffffffu;
#pragma unroll
    for (int u = 0; u < kMseRowEpl / 2; ++u) {
        const vf2 xx = xv[u];
        vf2 xc = xx;
        if (CLAMP) xc = vf2{__builtin_amdgcn_fmed3f(xx.x, minv, maxv), __builtin_amdgcn_fmed3f(xx.y, minv, maxv)};
        const vf2 tt = xc * c1;
        const uint32_t b0 = __float_as_uint(tt.x), b1 = __float_as_uint(tt.y);
        // round half UP on the magnitude bits (2 ops) instead of half to even (3): the two differ only on an exact tie,
        // where both neighbours are equally far from x -- the squared error is the same (to the last bits of r * m0)
        const uint32_t r0 = (b0 + half) & msk;
        const uint32_t r1 = (b1 + half) & msk;
        // (the dropped bits + kTieW) mod 2^sh, moved to the top of the word: <= 2 kTieW << ksh within kTieW ulps of a tie
        mn = min(mn, min(lshl_add(b0, ksh, nadd), lshl_add(b1, ksh, nadd)));
        const vf2 rr = {__uint_as_float(r0), __uint_as_float(r1)};
        vf2 ms = m0;
        if (TWO) ms = vf2{fabsf(tt.x) < thr ? m0b : m0s, fabsf(tt.y) < thr ? m0b : m0s};
        const vf2 d = xx - rr * ms;
        pa = __builtin_elementwise_fma(d, d, pa);
    }
    near = mn;
    return pa;
}

// The general fast loop: the float magic-number rounding (t + C) - C, C = 1.5 * 2^(max(exponent(t), exponent of binade 1)
// + 23 - M) -- the max is the subnormal range's fixed step, which the integer loop above cannot do.  7 issue slots per
// element, 6 without the clamp (+ ~3 for the near-tie watch: half step = 2^(exponent(C) - 24) from C's bits, kf = 1 - kTieW 2^(M-22)).
template <bool CLAMP, bool TWO>
__device__ __forceinline__ vf2 mse_cand_magic(const vf2 (&xv)[kMseRowEpl / 2], float minv, float maxv, float c1s, float m0s, float m0b, float thr,
                                              uint32_t lo, uint32_t kadd, float kf, float &nearf)
{
    const vf2 c1 = {c1s, c1s}, m0 = {m0s, m0s};
    vf2 pa = {0.0f, 0.0f};
    float mx = -__builtin_inff();
#pragma unroll
    for (int u = 0; u < kMseRowEpl / 2; ++u) {
        const vf2 xx = xv[u];
        vf2 xc = xx;
        if (CLAMP) xc = vf2{__builtin_amdgcn_fmed3f(xx.x, minv, maxv), __builtin_amdgcn_fmed3f(xx.y, minv, maxv)};
        const vf2 tt = xc * c1;
        const uint32_t b0 = max(__float_as_uint(tt.x) & 0x7f800000u, lo) + kadd;
        const uint32_t b1 = max(__float_as_uint(tt.y) & 0x7f800000u, lo) + kadd;
        const vf2 cc = {__uint_as_float(b0), __uint_as_float(b1)};
        const vf2 rr = (tt + cc) - cc;          // t rounded to M fraction bits, half to even
        const vf2 dt = tt - rr;
        const vf2 hs = {__uint_as_float(b0 - 0x0C400000u), __uint_as_float(b1 - 0x0C400000u)};   // half a rounding step
        const vf2 hk = hs * vf2{kf, kf};
        mx = fmaxf(mx, fmaxf(fabsf(dt.x) - hk.x, fabsf(dt.y) - hk.y));
        vf2 ms = m0;
        if (TWO) ms = vf2{fabsf(tt.x) < thr ? m0b : m0s, fabsf(tt.y) < thr ? m0b : m0s};
        const vf2 d = xx - rr * ms;
        pa = __builtin_elementwise_fma(d, d, pa);
    }
    nearf = mx;
    return pa;
}

// A candidate that came near a tie on this lane: walk the lane's 64 elements once more -- read again from memory, so that this
// rare path neither keeps the caller's register tile alive nor indexes it dynamically (k_mse_row is compiled for 3 waves per
// SIMD: variants that held the tile here cost the fast loops a third of their occupancy) -- with the fast loop's own test, and exchange the squared error of every element within kTieW ulps of a tie,
// as the fast loop formed it, for the reference's: K1's exact binade decision, the exact scale, IEEE division, round half
// even.  Returns the correction of the lane's sum.  A few 1e-4 of the (lane, candidate) pairs on continuous data.
__device__ __forceinline__ float mse_tie_patch(const float *__restrict__ xt /* the tile */, int lane, int64_t left /* elements from the tile's start */,
                                            QFmt f, float maxv, float c1s, float m0s, float m0b, float thr, uint32_t lo, uint32_t kadd,
                                            int use_int, int two)
{
    const Chan ch = make_chan(maxv, f);
    const uint32_t sh = kadd >> 23;
    const uint32_t half = 1u << (sh - 1u), msk = ~((1u << sh) - 1u), ksh = 32u - sh, nadd = (half + kTieW) << ksh;
    const uint32_t nthr = (2u * kTieW) << ksh;
    const float kf = 1.0f - (float)kTieW * __builtin_ldexpf(1.0f, (int)f.M - 22);
    float delta = 0.0f;
    // 16 elements per trip, their four 16-byte loads in flight together (one element per trip made a small tensor wait 64
    // memory round trips -- ~100 us -- for the structural tie of its last candidate)
#pragma unroll 1
    for (int g = 0; g < kMseRowEpl / 16; ++g) {
        float xe[16];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int off = (g * 4 + v) * 256 + lane * 4;
            if (off + 3 < left) {
                const vf4 w = ld16u<false>(xt + off);
                xe[4 * v] = w.x, xe[4 * v + 1] = w.y, xe[4 * v + 2] = w.z, xe[4 * v + 3] = w.w;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) xe[4 * v + q] = off + q < left ? xt[off + q] : 0.0f;   // zero padding: q(0) = 0 either way
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float x = xe[e];
            const float xc = __builtin_amdgcn_fmed3f(x, ch.minv, ch.maxv);   // (a no-op where the fast loop skipped the clamp)
            const float tt = xc * c1s;
            float rr;
            bool near;
            if (use_int) {
                rr = __uint_as_float((__float_as_uint(tt) + half) & msk);
                near = (__float_as_uint(tt) << ksh) + nadd <= nthr;
            } else {
                const uint32_t b = max(__float_as_uint(tt) & 0x7f800000u, lo) + kadd;
                const float cc = __uint_as_float(b);
                rr = (tt + cc) - cc;
                near = fabsf(tt - rr) - __uint_as_float(b - 0x0C400000u) * kf >= 0.0f;
            }
            if (__builtin_expect(near, 0)) {
                const float ms = (two && fabsf(tt) < thr) ? m0b : m0s;
                const float df = x - rr * ms;
                const float ls = __builtin_amdgcn_fmed3f(floorf(log2_tab(fabsf(xc), kFastTab) + ch.bias), 1.0f, (float)f.pmax);
                const float sc = scale_exact(ch, ls, f.M);
                const float de = x - rintf(xc / sc) * sc;
                delta += de * de - df * df;
            }
        }
    }
    return delta;
}

__global__ void __launch_bounds__(64, 3)   // 3 waves per SIMD: the rare near-tie path may spill, the fast loops must not lose occupancy to it
k_mse_row(const float *__restrict__ x, const float *__restrict__ grid, double *__restrict__ ws, MseArgs a,
          int64_t ntiles, int tpb, int ngroup, int gsize)
{
    __shared__ CandK cst[kMseRowGroup];
    const int lane = threadIdx.x;
    const int64_t c = blockIdx.z;
    const int total = a.n_m * a.n_cand;
    const int j0 = blockIdx.y * gsize;
    const int ng = min(gsize, total - j0);
    // ---- candidate constants of this group (lane <-> candidate) ----
    for (int j = lane; j < ng; j += 64) {
        const int jj = j0 + j, m = jj / a.n_cand, cand = jj - m * a.n_cand;
        const QFmt f = a.fmt[m];
        const float gv = grid[(int64_t)cand * a.C + c];
        const Chan ch = make_chan(fabsf(fmaxf(fabsf(-gv), gv)), f);   // set_quant_range(-g, g): fp8_quantizer.py:236
        CandK k;
        k.maxv = ch.maxv;
        k.minv = ch.minv;
        k.c1 = (float)(1.0 / ch.g);
        k.m0 = ch.m0;
        const int e_lo = 128 - ch.bi;                 // exponent field of t for binade p = 1
        k.lo = (uint32_t)e_lo << 23;
        k.kadd = ((uint32_t)(23 - (int)f.M) << 23) | 0x00400000u;
        k.m = m;
        k.fast = 0;
        k.m0b = ch.m0;
        k.thr = 0u;
        k.pad[0] = k.pad[1] = 0;
        // magic constant stays a normal float: exponent field of t (<= that of binade pmax) + 23 - M <= 254.
        // E = 0 formats (pmax == 1) always take the exact path: maxval / s_1 = 2^M - 0.5 is an exact tie, every clipped
        // element sits on it, and only the reference's own IEEE division decides them the reference's way.
        if (ch.pthr >= 0.0f && f.pmax > 1 && e_lo >= 24 && f.M <= 22.0f && (e_lo - 1 + f.pmax) + 23 - (int)f.M <= 254) {
            float m1, m2;
            int psw;
            const int runs = scale_mantissa_runs(ch, f, m1, m2, psw);
            if (runs <= 2) {
                k.fast = runs;
                k.m0 = m2;                      // p >= pswitch (all p when there is one run)
                k.m0b = m1;                     // p <  pswitch
                k.thr = runs == 2 ? (uint32_t)(psw + 127 - ch.bi) << 23 : 0u;   // exponent field of t at p == pswitch
            }
        }
        cst[j] = k;
    }
    __syncthreads();   // one wave: this is only the LDS ordering point
    // candidate (64 q + L) of the group accumulates, in double, in register dacc[q] of lane L: no LDS read-modify-write
    // (and no wait for one) per candidate
    double dacc[kMseRowGroup / 64];
#pragma unroll
    for (int q = 0; q < kMseRowGroup / 64; ++q) dacc[q] = 0.0;
    const float *xr = x + c * a.inner;
    const int64_t t_begin = (int64_t)blockIdx.x * tpb;
    const int64_t t_end = min(t_begin + tpb, ntiles);
    for (int64_t t = t_begin; t < t_end; ++t) {
        const int64_t e0 = t * kMseRowTile;
        vf2 xv[kMseRowEpl / 2];
        if (e0 + kMseRowTile <= a.inner) {
#pragma unroll
            for (int u = 0; u < kMseRowEpl / 4; ++u) {
                const vf4 v = ld16u<false>(xr + e0 + u * 256 + lane * 4);
                xv[2 * u] = vf2{v.x, v.y};
                xv[2 * u + 1] = vf2{v.z, v.w};
            }
        } else {   // last tile of the row: zero padding (q(0) = 0: contributes nothing)
#pragma unroll
            for (int u = 0; u < kMseRowEpl / 4; ++u) {
                const int64_t i = e0 + u * 256 + lane * 4;
                float e[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) e[q] = i + q < a.inner ? xr[i + q] : 0.0f;
                xv[2 * u] = vf2{e[0], e[1]};
                xv[2 * u + 1] = vf2{e[2], e[3]};
            }
        }
        // tile statistics (wave-uniform), amortised over the >= 111 candidates: does a candidate's range cover the tile
        // (no clamp needed), and does any nonzero element fall below a candidate's first binade (integer rounding needs
        // none to)?  NaN elements are ignored here: their squared error is NaN on every path.
        float tmn = __builtin_inff(), tmx = -__builtin_inff(), anz = __builtin_inff();
#pragma unroll
        for (int u = 0; u < kMseRowEpl / 2; ++u) {
            tmn = fminf(tmn, fminf(xv[u].x, xv[u].y));
            tmx = fmaxf(tmx, fmaxf(xv[u].x, xv[u].y));
            const float a0 = fabsf(xv[u].x), a1 = fabsf(xv[u].y);
            anz = fminf(anz, fminf(a0 > 0.0f ? a0 : __builtin_inff(), a1 > 0.0f ? a1 : __builtin_inff()));
        }
        tmn = wave_min_f(tmn);
        tmx = wave_max_f(tmx);
        anz = wave_min_f(anz);
#pragma unroll
        for (int jq = 0; jq < kMseRowGroup / 64; ++jq) {
            const int jb = jq * 64;
            if (jb >= ng) break;   // wave-uniform
            const int jn = min(64, ng - jb);
            double acc = dacc[jq];
            CandK k = cst[jb];
            for (int jl = 0; jl < jn; ++jl) {
                const CandK kn = cst[jb + min(jl + 1, jn - 1)];   // next candidate's constants: in flight during this one
                vf2 pa = {0.0f, 0.0f};
                // integer rounding: no nonzero |t| below binade 1 (|xc| >= min(anz, maxv), and maxv sits in the top binade)
                const bool int_ok = kIntRound && k.fast != 0 && anz * k.c1 >= __uint_as_float(k.lo) && k.maxv >= anz;
                if (int_ok) {
                    const uint32_t sh = k.kadd >> 23;                       // 23 - M: position of the last kept fraction bit
                    const uint32_t half = 1u << (sh - 1u), msk = ~((1u << sh) - 1u);
                    const uint32_t ksh = 32u - sh, nadd = (half + kTieW) << ksh;
                    const bool cover = tmn >= k.minv && tmx <= k.maxv;   // the candidate's range covers the tile: nothing to clamp
                    const float thr = __uint_as_float(k.thr);
                    uint32_t near;
                    if (k.fast == 1) {
                        pa = cover ? mse_cand_int<false, false>(xv, k.minv, k.maxv, k.c1, k.m0, k.m0b, thr, half, msk, ksh, nadd, near)
                                   : mse_cand_int<true, false>(xv, k.minv, k.maxv, k.c1, k.m0, k.m0b, thr, half, msk, ksh, nadd, near);
                    } else {
                        pa = cover ? mse_cand_int<false, true>(xv, k.minv, k.maxv, k.c1, k.m0, k.m0b, thr, half, msk, ksh, nadd, near)
                                   : mse_cand_int<true, true>(xv, k.minv, k.maxv, k.c1, k.m0, k.m0b, thr, half, msk, ksh, nadd, near);
                    }
                    if (__builtin_expect(near <= (((2u * kTieW) << ksh) | ((1u << ksh) - 1u)), 0)) {   // some lanes, rarely
                        const QFmt f = a.fmt[k.m];
                        pa.x += mse_tie_patch(xr + e0, lane, a.inner - e0, f, k.maxv, k.c1, k.m0, k.m0b, thr, k.lo, k.kadd, 1, k.fast == 2);
                    }
                } else if (k.fast != 0) {
                    const bool cover = tmn >= k.minv && tmx <= k.maxv;   // nothing to clamp for this candidate on this tile
                    const float thr = __uint_as_float(k.thr);
                    const float kf = 1.0f - (float)kTieW * __builtin_ldexpf(1.0f, 1 - (int)(k.kadd >> 23));   // 1 - kTieW 2^(M-22)
                    float nearf;
                    if (k.fast == 1) {
                        pa = cover ? mse_cand_magic<false, false>(xv, k.minv, k.maxv, k.c1, k.m0, k.m0b, thr, k.lo, k.kadd, kf, nearf)
                                   : mse_cand_magic<true, false>(xv, k.minv, k.maxv, k.c1, k.m0, k.m0b, thr, k.lo, k.kadd, kf, nearf);
                    } else {   // two scale mantissas: the low binades (|t| below thr) use m0b
                        pa = cover ? mse_cand_magic<false, true>(xv, k.minv, k.maxv, k.c1, k.m0, k.m0b, thr, k.lo, k.kadd, kf, nearf)
                                   : mse_cand_magic<true, true>(xv, k.minv, k.maxv, k.c1, k.m0, k.m0b, thr, k.lo, k.kadd, kf, nearf);
                    }
                    if (__builtin_expect(nearf >= 0.0f, 0)) {
                        const QFmt f = a.fmt[k.m];
                        pa.x += mse_tie_patch(xr + e0, lane, a.inner - e0, f, k.maxv, k.c1, k.m0, k.m0b, thr, k.lo, k.kadd, 0, k.fast == 2);
                    }
                } else {
                    // exact per-element path (scales not exactly geometric in fp32, tiny / huge / degenerate ranges)
                    const QFmt f = a.fmt[k.m];
                    const Chan ch = make_chan(k.maxv, f);
#pragma unroll 1
                    for (int u = 0; u < kMseRowEpl / 2; ++u) {
                        const float e[2] = {xv[u].x, xv[u].y};
                        float dd[2];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const float xc = __builtin_amdgcn_fmed3f(e[q], ch.minv, ch.maxv);
                            // K1's exact binade decision (the exponent field of xc * 2^bf cannot express binade 1 when
                            // bias > 128, and non-geometric scales give no "either side of a border" equivalence)
                            const float ls = __builtin_amdgcn_fmed3f(floorf(log2_tab(fabsf(xc), kFastTab) + ch.bias), 1.0f, (float)f.pmax);
                            const float sc = scale_exact(ch, ls, f.M);
                            dd[q] = e[q] - rintf(xc / sc) * sc;   // IEEE division, as the reference: exact at ties too
                            // zero padding of the row's last tile: q(0) may be NaN here (s_1 = 0), and only a real zero's counts
                            if (e0 + (u >> 1) * 256 + lane * 4 + (u & 1) * 2 + q >= a.inner) dd[q] = 0.0f;
                        }
                        pa = __builtin_elementwise_fma(vf2{dd[0], dd[1]}, vf2{dd[0], dd[1]}, pa);
                    }
                }
                const float s = wave_sum(pa.x + pa.y);       // wave-uniform
                acc += lane == jl ? (double)s : 0.0;
                k = kn;
            }
            dacc[jq] = acc;
        }
    }
    const int64_t nblk = gridDim.x;
#pragma unroll
    for (int jq = 0; jq < kMseRowGroup / 64; ++jq) {
        const int j = jq * 64 + lane;
        if (j < ng) ws[((c * total) + (j0 + j)) * nblk + blockIdx.x] = dacc[jq];   // j0 + j == m * n_cand + cand
    }
}

// mses[m, i, c] += (sum over the splits of row (c, m, i)) / inner: one wave per row of partial sums, in double
__global__ void __launch_bounds__(kBlock)
k_mse_final(const double *__restrict__ ws, float *mses, int64_t C, int n_m, int n_cand,
            int64_t nsplit, double inv_inner, int overwrite, const float *__restrict__ grid, SelOne so)
{
    const int64_t total = C * n_m * n_cand;
    const int lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // ws row: ((c * n_m + m) * n_cand + i)
    if (j < total) {
        double sum = 0.0;
        for (int64_t s2 = lane; s2 < nsplit; s2 += 64) sum += ws[j * nsplit + s2];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
        if (lane == 0) {
            const int64_t c = j / ((int64_t)n_m * n_cand);
            const int64_t mi = j - c * n_m * n_cand;   // m * n_cand + i
            table_add(mses + mi * C + c, (float)(sum * inv_inner), overwrite);
        }
    }
    // per-tensor quantizers: the last workgroup also selects the winner (fp8q_select.h)
    if (so.enabled && last_workgroup(so.ticket, gridDim.x, blockIdx.x)) select_one_row(mses, grid, n_m, n_cand, so);
}

// The same for FEW splits per row (per-channel weights: 1-32 partial sums, but C x n_m x n_cand rows -- 852 K for a
// [1280, 320] layer with the mantissa search): a wave per row left 63 lanes idle and wrote one strided float per wave
// (33-41 us per call, 2.2 ms per MobileNetV2 search batch).  Here a workgroup takes a tile of 16 channels x 64 (width,
// candidate) pairs: a thread per row sums its splits reading ws in its own order (contiguous), the tile turns through LDS,
// and mses (channel fastest) is updated in 64-byte runs.
constexpr int kFinTC = 16, kFinTM = 64;

__global__ void __launch_bounds__(kBlock)
k_mse_final_tile(const double *__restrict__ ws, float *mses, int64_t C, int64_t NM /* n_m * n_cand */, int nsplit,
                 double inv_inner, int overwrite, const float *__restrict__ grid, int n_m, int n_cand, SelOne so)
{
    __shared__ float tile[kFinTC][kFinTM + 1];
    const int64_t c0 = (int64_t)blockIdx.y * kFinTC, m0 = (int64_t)blockIdx.x * kFinTM;
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < kFinTC * kFinTM / kBlock; ++k) {
        const int lc = (tid >> 6) + 4 * k, lm = tid & 63;
        const int64_t c = c0 + lc, mi = m0 + lm;
        double sum = 0.0;
        if (c < C && mi < NM) {
            const double *row = ws + (c * NM + mi) * nsplit;
            for (int s2 = 0; s2 < nsplit; ++s2) sum += row[s2];
        }
        tile[lc][lm] = (float)(sum * inv_inner);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kFinTC * kFinTM / kBlock; ++k) {
        const int lm = (tid >> 4) + 16 * k, lc = tid & 15;
        const int64_t c = c0 + lc, mi = m0 + lm;
        if (c < C && mi < NM) table_add(mses + mi * C + c, tile[lc][lm], overwrite);
    }
    if (so.enabled && last_workgroup(so.ticket, gridDim.x * gridDim.y, blockIdx.y * gridDim.x + blockIdx.x)) select_one_row(mses, grid, n_m, n_cand, so);
}

// ---------------------------------------------------------------------------------------------
// Device-side search grid and winner selection of FP_MSE_Estimator: with these two, a calibration batch needs no host
// round trip (the reference synchronises at range_estimators.py:305, :353 and :360).
// ---------------------------------------------------------------------------------------------
// grid[i, c] = torch.linspace(0.1 * mx_c, 1.2 * mx_c, steps)[i] bit for bit: linspace_at() (fp8q_common.h; fp8q.ops checks it
// against torch.linspace itself once per process).  The first calibration batch gets its grid from the abs-max pass itself
// (fp8q_minmax_linspace_f32); this kernel serves the batch-sharded case, where the maximum is all-reduced first.
__global__ void __launch_bounds__(kBlock)
k_mse_linspace(const float *__restrict__ mx, int64_t C, int steps, double lo_frac, double hi_frac, float *__restrict__ grid)
{
    const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= C * steps) return;
    const int i = (int)(idx / C);
    const int64_t c = idx - (int64_t)i * C;
    grid[idx] = linspace_at(mx[c], lo_frac, hi_frac, steps, i);
}

// Winner selection in ONE launch (round 4: two dependent ones, 930 launches of 3-34 us per MobileNetV2 calibration batch).
// Phase 1, one wave per channel: for every mantissa width m the minimum over the candidates and its first index, then the
// channel's best width (range_estimators.py:350-351: mses.min(1)[0].argmin(0)) ->
//     sel: int32 [C, 1 + n_m] = {best width index, argmin_i for width 0, ..., argmin_i for width n_m - 1}
// Phase 2, the LAST workgroup to finish phase 1 (a ticket in the workspace header: zero between calls -- atomicInc wraps it
// back; per-tensor quantizers, C <= 4, are a single workgroup and need none): plurality vote over the channels' best widths
// (torch.mode: the most frequent value, the smallest one on a tie -- :352-354), then per channel the winning width's argmin
// candidate and its maxval (:356-362).
__global__ void __launch_bounds__(kBlock)
k_mse_select(const float *__restrict__ mses, const float *__restrict__ grid, int64_t C, int n_m, int n_cand, int *sel,
             unsigned *__restrict__ ticket, MseArgs a /* fmt[m].M = width m */, float *__restrict__ mbits_out, int *__restrict__ vote_out,
             float *__restrict__ maxval_out, float *__restrict__ xmin_out, float sign)
{
    __shared__ int hist[kMseMaxM];
    __shared__ int s_vote;
    const int lane = threadIdx.x & 63;
    const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c < C) {
        ArgMin best_m = {__builtin_inff(), 0x7fffffff};
        for (int m = 0; m < n_m; ++m) {
            ArgMin am = {__builtin_inff(), 0x7fffffff};
            for (int i = lane; i < n_cand; i += 64) {
                const ArgMin o = {mses[((int64_t)m * n_cand + i) * C + c], i};
                if (argmin_less(o, am)) am = o;
            }
            am = wave_argmin(am);
            if (lane == 0) agent_store(&sel[c * (1 + n_m) + 1 + m], am.idx);
            const ArgMin o = {am.v, m};
            if (argmin_less(o, best_m)) best_m = o;
        }
        if (lane == 0) agent_store(&sel[c * (1 + n_m)], best_m.idx);
    }
    // (Reading the table in 64-byte runs does not pay.  Round 5: lane = channel with the waves splitting the candidates: ~28
    // dependent trips per width, 36 us per call instead of 12.7.  Round 6: 16 channels x 16 candidate slots per workgroup, all
    // n_m * 7 loads of a thread independent, the slots combined through LDS: 12.4 us instead of 6.7 with one width, 20.5
    // instead of 14.3 with six (profiles/r06_select_fence_ab.txt).  Measured, dropped: a wave per channel it stays.)
    // (no fences: sel travels through agent-scope stores / loads, fp8q_select.h)
    if (!last_workgroup(ticket, gridDim.x, blockIdx.x)) return;
    if (threadIdx.x < kMseMaxM) hist[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t cc = threadIdx.x; cc < C; cc += kBlock) atomicAdd(&hist[agent_load(&sel[cc * (1 + n_m)])], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        int v = 0;
        for (int m = 1; m < n_m; ++m)
            if (hist[m] > hist[v]) v = m;
        s_vote = v;
        mbits_out[0] = a.fmt[v].M;
        if (vote_out) vote_out[0] = v;
    }
    __syncthreads();
    const int v = s_vote;
    for (int64_t cc = threadIdx.x; cc < C; cc += kBlock) {
        const float mv = grid[(int64_t)agent_load(&sel[cc * (1 + n_m) + 1 + v]) * C + cc];
        maxval_out[cc] = mv;
        if (xmin_out) xmin_out[cc] = sign * mv;       // sign_bits * -1.0 * maxval (:369)
    }
}

}  // namespace

// fp8q_mse_hist.hip: partition-once / interval-histogram evaluation of all candidates on one long row
size_t fp8q_mse_hist_workspace_bytes(int64_t n, int64_t n_pairs);
bool fp8q_mse_hist_supported(const QFmt *fmts, int n_m, int n_bits);
int fp8q_mse_hist_launch(const float *x, int64_t n, const float *grid, int64_t n_cand, const QFmt *fmts, int n_m, float *mses,
                         void *ws, size_t ws_bytes, hipStream_t st, int brute, int overwrite, const SelOne *sel);
// (fp8q_quant.hip) K1 of a per-tensor quantizer with the winner selection in its prologue
int fp8q_quantize_select_f32(const float *x, float *y, int64_t n, const float *mses, const float *grid, int n_m, int n_cand,
                             const SelOne *so, int n_bits, int sign_bits, hipStream_t st);

// FP8Q_MSE_HIST: 1 (default) = long per-tensor rows of a signed format of <= 8 bits go through the interval-histogram
// evaluation; 0 = never (the lane-per-element kernel everywhere); 2 = same routing with every candidate evaluated element by
// element (the self-check of the cell logic the tests use); 3 = the route for every per-tensor row of >= 2^12 elements (tuning)
static int mse_hist_mode()
{
    static const int v = [] {
        const char *e = getenv("FP8Q_MSE_HIST");
        return e ? atoi(e) : 1;
    }();
    return v;
}

// The route's cost does not depend on the data: ~60 us of small launches + ~0.1 us per (width, candidate) pair for the borders
// + ~4.3 ps per element (partition at the copy rate + the moments).  k_mse_row costs ~0.2 ps per (element, pair) on top of a
// floor that round 6's per-shape timelines put at ~75 us on activation-like data (ReLU6 outputs, whose many elements at the
// clipping value share every near-tie and are all re-evaluated, are its worst case and the histogram's best: 79.6 us at 0.5 M
// elements, 83.6 at 0.8 M -- profiles/r06_calib_timeline_*.txt), so with the moments kernel no longer latency-bound on small
// tensors (hist_slice_min) every per-tensor row from 256 K elements takes the histogram: MobileNetV2's 14 activations of
// 0.5-1.2 M elements, 100-104 us per calibration step through k_mse_row, take ~80.
// Which route a row takes depends on its shape only, so a tensor is evaluated the same way on every call.
static bool mse_use_hist_shape(int64_t C, int64_t inner, int64_t n_cand, int n_m)
{
    if (mse_hist_mode() == 0 || C != 1 || inner >= (1ll << 31)) return false;
    if (mse_hist_mode() == 3) return inner >= (1 << 12);
    // Round 6, after the chain lost two launches and its sort / selection / moments their serial parts: ~48 us + 0.05 us per pair
    // + 4.3 ps per element, below k_mse_row's own floor (~62 us) on every row it is built for -- measured down to
    // [64, 1280] = 82 K elements (MobileNetV2's pooled features: the calibration step 78.4 -> 66.3 us with 111 pairs,
    // 115.9 -> 93.3 with 666) and on to 16 K elements (66.8 -> 65.0 / 103.6 -> 91.0; at 4 K elements the six-width search is
    // faster element by element: 82.3 against 90.2).  From 16 K elements every per-tensor row takes the histogram.
    return inner >= (1 << 14) && (int64_t)n_m * n_cand <= 4096;            // (8 KB of border-count table per pair)
}

extern "C" {

int fp8q_mse_linspace_f32(const float *mx, int64_t C, int n_cand, double lo_frac, double hi_frac, float *grid,
                          fp8q_stream_t stream)
{
    if (!mx || !grid || C <= 0 || n_cand < 2 || n_cand > (1 << 20)) return FP8Q_EINVAL;
    hipLaunchKernelGGL(k_mse_linspace, dim3((unsigned)cdiv(C * n_cand, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, mx, C,
                       n_cand, lo_frac, hi_frac, grid);
    return launch_rc();
}

size_t fp8q_mse_select_workspace_bytes(int64_t C, int n_m) { return kTicketBytes + (C > 0 && n_m > 0 ? (size_t)C * (1 + n_m) * sizeof(int) : 0); }

int fp8q_mse_select_f32(const float *mses, const float *grid, int64_t C, int64_t n_cand, const float *mbits_host, int n_m,
                        int sign_bits, float *mbits_out, int *vote_out, float *maxval_out, float *xmin_out, void *ws,
                        size_t ws_bytes, fp8q_stream_t stream)
{
    if (!mses || !grid || !mbits_host || !mbits_out || !maxval_out || C <= 0 || n_cand <= 0 || n_cand > (1 << 20) ||
        n_m <= 0 || n_m > kMseMaxM || (sign_bits != 0 && sign_bits != 1))
        return FP8Q_EINVAL;
    if (!ws || ws_bytes < fp8q_mse_select_workspace_bytes(C, n_m) || ((uintptr_t)ws & 3)) return FP8Q_EWORKSPACE;
    MseArgs a;
    memset(&a, 0, sizeof(a));
    for (int m = 0; m < n_m; ++m) a.fmt[m].M = mbits_host[m];   // the candidate widths as given (the vote returns one of them)
    // ws: ticket block (kTicketBytes: zero between calls) | sel
    hipLaunchKernelGGL(k_mse_select, dim3((unsigned)cdiv(C, 4)), dim3(kBlock), 0, (hipStream_t)stream, mses, grid, C, n_m, (int)n_cand,
                       (int *)((char *)ws + kTicketBytes), (unsigned *)ws, a, mbits_out, vote_out, maxval_out, xmin_out, -(float)sign_bits);
    return launch_rc();
}

// k_mse_grid's tile: a lane walks its candidate over a whole tile, one element after the other (~0.085 us per element when
// a SIMD holds a single wave: dependent LDS lookups), so a per-channel weight with few channels and rows of a few hundred
// elements -- MobileNetV2's [160, 960] pointwise convolutions: 160 workgroups, 82 us -- left most of the chip idle.  Rows
// are cut finer (down to 64 elements) until the launch has ~4096 workgroups; every cut repeats the candidate set-up.
static int mse_tile(int64_t C, int64_t inner, int64_t n_cand, int n_m)
{
    static const int env = [] {   // FP8Q_MSE_GRID_TILE=2048: whole rows per workgroup, as before round 5 (A/B)
        const char *e = getenv("FP8Q_MSE_GRID_TILE");
        const int v = e ? atoi(e) : 0;
        return v >= 32 && v <= kMseTile ? (v & ~31) : 0;
    }();
    if (env) return env;
    const int64_t base = C * n_m * cdiv(n_cand, kMseBlock);
    int tl = kMseTile;
    while (tl > 64 && base * cdiv(inner, tl) < 4096) tl >>= 1;
    return tl;
}

static int mse_nsplit(int64_t C, int64_t inner, int64_t n_cand, int n_m)
{
    const int64_t cg = cdiv(n_cand, kMseBlock);
    int64_t ns = cdiv(inner, mse_tile(C, inner, n_cand, n_m));
    int64_t cap = (4 * kTargetBlocks) / (C * n_m * cg > 0 ? C * n_m * cg : 1);
    if (cap < 1) cap = 1;
    if (ns > cap) ns = cap;
    if (ns < 1) ns = 1;
    return (int)ns;
}

// k_mse_row geometry: tiles per block (so that a row is cut into <= 16384 blocks) and candidate groups
struct RowGeo {
    int64_t ntiles, nblk;
    int tpb, ngroup, gsize;
};

static bool mse_use_row(int64_t C, int64_t inner)
{
    static const int env = [] {   // FP8Q_MSE_ROW=0: the lane-per-candidate kernel everywhere (A/B)
        const char *e = getenv("FP8Q_MSE_ROW");
        return e ? atoi(e) : 1;
    }();
    return env && inner >= kMseRowMinInner && C <= 65535;
}

static RowGeo mse_row_geo(int64_t C, int64_t inner, int64_t n_cand, int n_m)
{
    RowGeo g;
    g.ntiles = cdiv(inner, kMseRowTile);
    int64_t cap = 16384 / (C > 0 ? C : 1);
    if (cap < 64) cap = 64;
    g.tpb = (int)cdiv(g.ntiles, cap);
    g.nblk = cdiv(g.ntiles, g.tpb);
    const int64_t total = n_cand * n_m;
    // Every block does the same amount of work, so the launch runs in ceil(blocks / resident) rounds of equal length
    // and a nearly empty last round is pure loss (111 candidates on [64,32,112,112]: 12 544 blocks on 4 096 resident
    // waves = 3.06 -> 4 rounds, 77 %).  Cutting the candidates into more, smaller groups gives finer rounds at the
    // price of one more constant-setup pass per block: pick the split with the best product of the two.
    const int64_t resident = 256 * 12;   // CUs x waves (144 VGPRs: 3 per SIMD)
    const int64_t g0 = cdiv(total, kMseRowGroup);
    double best = -1.0;
    for (int64_t ng = g0; ng <= g0 * 4 && ng <= total; ++ng) {
        const int64_t gs = cdiv(total, ng), waves = g.nblk * ng * (C > 0 ? C : 1);
        const double rounds = (double)cdiv(waves, resident);
        const double fill = (double)waves / (rounds * (double)resident);
        const double work = (double)gs * (6.0 * kMseRowEpl) * g.tpb, setup = 250.0 * (double)cdiv(gs, 64);
        const double score = fill * work / (work + setup);
        if (score > best * 1.005) {   // prefer fewer groups on ties
            best = score;
            g.ngroup = (int)ng;
            g.gsize = (int)gs;
        }
    }
    return g;
}

size_t fp8q_mse_workspace_bytes(int64_t C, int64_t inner, int64_t n_cand, int n_m)
{
    if (C <= 0 || inner <= 0 || n_cand <= 0 || n_m <= 0) return 16;
    // (the format is not known here: the histogram route's size is returned whenever the shape could take it)
    if (mse_use_hist_shape(C, inner, n_cand, n_m)) {
        const size_t a = fp8q_mse_hist_workspace_bytes(inner, n_cand * n_m);
        const int64_t ns0 = mse_use_row(C, inner) ? mse_row_geo(C, inner, n_cand, n_m).nblk : mse_nsplit(C, inner, n_cand, n_m);
        const size_t b = (size_t)C * n_m * n_cand * ns0 * sizeof(double) + 16;
        return a > b ? a : b;
    }
    const int64_t ns = mse_use_row(C, inner) ? mse_row_geo(C, inner, n_cand, n_m).nblk : mse_nsplit(C, inner, n_cand, n_m);
    return (size_t)C * n_m * n_cand * ns * sizeof(double) + 16;
}

// true when mse_grid_impl will run k_mse_grid for this shape (per-channel rows below 2048 elements: the weights) -- the kernel
// that can also make the first batch's ranges and grid itself (MseArgs::first_grid)
static bool mse_first_batch_in_grid_kernel(const float *x, int64_t C, int64_t inner, int64_t n_cand, int n_m)
{
    static const bool on = [] {   // FP8Q_MSE_FIRST_IN_GRID=0: the separate abs-max + grid launch (A/B)
        const char *e = getenv("FP8Q_MSE_FIRST_IN_GRID");
        return !e || atoi(e) != 0;
    }();
    if (!on || C <= 1 || mse_use_hist_shape(C, inner, n_cand, n_m)) return false;
    return !(mse_use_row(C, inner) && ((uintptr_t)x & 3) == 0);
}

static int mse_grid_impl(const float *x, int64_t C, int64_t inner, const float *grid, int64_t n_cand,
                         const float *mbits_host, int n_m, int n_bits, int sign_bits, float *mses,
                         void *ws, size_t ws_bytes, fp8q_stream_t stream, int overwrite, const SelOne *sel = nullptr,
                         int *sel_done = nullptr, const fp8q_mse_state *first_state = nullptr)
{
    SelOne so;
    memset(&so, 0, sizeof(so));
    if (sel && C == 1) so = *sel;      // the winner selection rides on the launch that finishes the table (one row only)
    if (sel_done) *sel_done = 0;
    if (!x || !grid || !mbits_host || !mses || C <= 0 || inner <= 0 || n_cand <= 0 || n_m <= 0 ||
        n_m > kMseMaxM || n_cand > (1 << 20))
        return FP8Q_EINVAL;
    if (C > 65535) return FP8Q_ETOOMANY;   // gridDim.z; no model of this path has that many channels
    if (!ws || ws_bytes < fp8q_mse_workspace_bytes(C, inner, n_cand, n_m) || ((uintptr_t)ws & 7))
        return FP8Q_EWORKSPACE;
    MseArgs a;
    int pmax_all = 0;
    for (int m = 0; m < n_m; ++m) {
        if (int rc = make_fmt(mbits_host[m], n_bits, sign_bits, &a.fmt[m])) return rc;
        if (a.fmt[m].pmax > pmax_all) pmax_all = a.fmt[m].pmax;
    }
    a.n_m = n_m;
    a.n_cand = (int)n_cand;
    a.cgroups = (int)cdiv(n_cand, kMseBlock);
    a.nsplit = mse_nsplit(C, inner, n_cand, n_m);
    a.tile = mse_tile(C, inner, n_cand, n_m);
    a.inner = inner;
    a.C = C;
    a.overwrite = overwrite;
    a.first_min = a.first_max = a.first_absmax = a.first_grid = nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (mse_use_hist_shape(C, inner, n_cand, n_m) && fp8q_mse_hist_supported(a.fmt, n_m, n_bits))
    {
        if (sel_done) *sel_done = so.enabled;
        return fp8q_mse_hist_launch(x, inner, grid, n_cand, a.fmt, n_m, mses, ws, ws_bytes, st, mse_hist_mode() == 2, overwrite, &so);
    }
    int64_t nsplit = a.nsplit;
    if (mse_use_row(C, inner) && ((uintptr_t)x & 3) == 0) {
        const RowGeo g = mse_row_geo(C, inner, n_cand, n_m);
        nsplit = g.nblk;
        hipLaunchKernelGGL(k_mse_row, dim3((unsigned)g.nblk, (unsigned)g.ngroup, (unsigned)C), dim3(64), 0, st, x, grid,
                           (double *)ws, a, g.ntiles, g.tpb, g.ngroup, g.gsize);
    } else {
        if (first_state) {     // (the caller asked mse_first_batch_in_grid_kernel() before: this is the route it promised)
            a.first_min = first_state->cur_min;
            a.first_max = first_state->cur_max;
            a.first_absmax = first_state->absmax;
            a.first_grid = first_state->grid;
        }
        const size_t shmem = (size_t)kMseTile * 4 + (size_t)kMseBlock * ((pmax_all + 1) | 1) * sizeof(float);
        if (shmem > 64 * 1024) {
            // per device, cheap: a process may drive several GPUs (fp8q.ops._on_device)
            // (the kernel also has a few words of static LDS -- the first-batch row range --: 160 KiB is the CU's total)
            hipError_t e = hipFuncSetAttribute((const void *)k_mse_grid, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               159 * 1024);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(k_mse_grid, dim3((unsigned)a.nsplit, (unsigned)(n_m * a.cgroups), (unsigned)C),
                           dim3(kMseBlock), shmem, st, x, grid, (double *)ws, a, mses, 1.0 / (double)inner);
        if (a.nsplit == 1) return launch_rc();
    }
    if (int rc = launch_rc()) return rc;
    const int64_t rows = C * n_m * n_cand;
    if (nsplit <= 32 && cdiv(C, kFinTC) <= 65535)
        hipLaunchKernelGGL(k_mse_final_tile, dim3((unsigned)cdiv(n_m * n_cand, kFinTM), (unsigned)cdiv(C, kFinTC)), dim3(kBlock), 0, st,
                           (const double *)ws, mses, C, (int64_t)n_m * n_cand, (int)nsplit, 1.0 / (double)inner, overwrite, grid, n_m,
                           (int)n_cand, so);
    else
        hipLaunchKernelGGL(k_mse_final, dim3((unsigned)cdiv(rows, 4)), dim3(kBlock), 0, st, (const double *)ws, mses, C,
                           n_m, (int)n_cand, nsplit, 1.0 / (double)inner, overwrite, grid, so);
    if (sel_done) *sel_done = so.enabled;
    return launch_rc();
}

int fp8q_mse_grid_f32(const float *x, int64_t C, int64_t inner, const float *grid, int64_t n_cand,
                      const float *mbits_host, int n_m, int n_bits, int sign_bits, float *mses,
                      void *ws, size_t ws_bytes, fp8q_stream_t stream)
{
    return mse_grid_impl(x, C, inner, grid, n_cand, mbits_host, n_m, n_bits, sign_bits, mses, ws, ws_bytes, stream, 0);
}

size_t fp8q_mse_calibrate_workspace_bytes(int64_t C, int64_t inner, int64_t n_cand, int n_m, size_t *minmax_bytes, size_t *select_bytes)
{
    if (minmax_bytes) *minmax_bytes = fp8q_minmax_workspace_bytes(C, inner);
    if (select_bytes) *select_bytes = fp8q_mse_select_workspace_bytes(C, n_m);
    return fp8q_mse_workspace_bytes(C, inner, n_cand, n_m);
}

// QuantizationManager.forward in estimate_ranges state with FP_MSE_Estimator (quantization_manager.py:114-122 around
// range_estimators.py:318-369) as ONE call into the library: what used to be four entry points called from Python
// (fp8q_minmax_linspace_f32, fp8q_mse_grid_f32, fp8q_mse_select_f32, fp8q_quantize[_dm]_f32: ~45 us of host time each
// through ctypes + torch allocations) is enqueued from here; the host side of a calibration batch was the critical path
// of BASELINE config 4 (profiles/r06_host_profile.txt).
int fp8q_mse_calibrate_f32(float *x, float *y, int64_t C, int64_t inner, const fp8q_mse_state *s, int first, int n_cand,
                           const float *mbits_host, int n_m, int n_bits, int sign_bits, const fp8q_affine_pre *pre, void *ws_minmax,
                           size_t ws_minmax_bytes, void *ws_select, size_t ws_select_bytes, void *ws_mse, size_t ws_mse_bytes,
                           fp8q_stream_t stream)
{
    if (!s || !x || !s->grid || !s->mses || !s->maxval || !s->mbits || !mbits_host || C <= 0 || inner <= 0 || n_cand < 2 ||
        n_m <= 0 || n_m > kMseMaxM)
        return FP8Q_EINVAL;
    for (int m = 0; m < n_m; ++m) {   // every format is checked before the first launch
        QFmt f;
        if (int rc = make_fmt(mbits_host[m], n_bits, sign_bits, &f)) return rc;
    }
    if (C > 65535) return FP8Q_ETOOMANY;
    bool first_in_grid = false;
    if (pre && (C != 1 || !pre->x || pre->N <= 0 || pre->N * pre->C * pre->HW != inner)) return FP8Q_EINVAL;
    if (!ws_mse || ws_mse_bytes < fp8q_mse_workspace_bytes(C, inner, n_cand, n_m) || ((uintptr_t)ws_mse & 7)) return FP8Q_EWORKSPACE;
    if (!ws_select || ws_select_bytes < fp8q_mse_select_workspace_bytes(C, n_m) || ((uintptr_t)ws_select & 3)) return FP8Q_EWORKSPACE;
    if (first) {
        // max|x| per row and the search grid of that maximum in one launch (:295-316); the table needs no clearing: the
        // first batch's entries are written, not added.  Behind a BN + activation the same launch also writes t.
        if (!s->cur_min || !s->cur_max || !s->absmax) return FP8Q_EINVAL;
        first_in_grid = !pre && mse_first_batch_in_grid_kernel(x, C, inner, n_cand, n_m);
        const int rc = first_in_grid ? FP8Q_OK : pre ? fp8q_affine_act_minmax_linspace_f32(pre->x, pre->residual, x, pre->N, pre->C, pre->HW, pre->alpha_beta, pre->act,
                                                                 s->cur_min, s->cur_max, s->absmax, s->grid, n_cand, 0.1, 1.2, ws_minmax,
                                                                 ws_minmax_bytes, stream)
                           : fp8q_minmax_linspace_f32(x, C, inner, s->cur_min, s->cur_max, s->absmax, s->grid, n_cand, 0.1, 1.2, ws_minmax,
                                                      ws_minmax_bytes, stream);
        if (rc) return rc;
    } else if (pre) {
        if (int rc = fp8q_affine_act_f32(pre->x, pre->residual, x, pre->N, pre->C, pre->HW, pre->alpha_beta, pre->act, stream)) return rc;
    }
    SelOne so;
    memset(&so, 0, sizeof(so));
    so.mbits_out = s->mbits;
    so.vote_out = s->vote;
    so.maxval_out = s->maxval;
    so.xmin_out = s->xmin;
    so.ticket = (unsigned *)ws_select;          // the selection workspace's ticket block: zero between calls
    so.sign = -(float)sign_bits;
    for (int m = 0; m < n_m; ++m) so.M[m] = mbits_host[m];
    so.enabled = 1;
    // per-tensor quantizer, the batch quantized right away: the selection rides in the prologue of that K1 launch
    // (k_quant_rows_sel) instead of behind a ticket in the launch that finishes the table (k_mse_eval 20.6 -> 11.0 us with six
    // widths, 11.1 -> 7.7 with one, K1 + 1..2 us; FP8Q_SEL_IN_K1=0: tickets, =1: only with the mantissa search)
    static const int sel_in_k1 = getenv("FP8Q_SEL_IN_K1") ? atoi(getenv("FP8Q_SEL_IN_K1")) : 2;
    const bool late = sel_in_k1 && y && C == 1 && (n_m > 1 || sel_in_k1 == 2) && n_m <= kSelMaxM && (int64_t)n_m * n_cand <= 4096 &&
                      (((uintptr_t)x ^ (uintptr_t)y) & 15) == 0 && ((uintptr_t)x & 3) == 0;
    if (late) so.enabled = 0;
    int sel_done = 0;
    if (int rc = mse_grid_impl(x, C, inner, s->grid, n_cand, mbits_host, n_m, n_bits, sign_bits, s->mses, ws_mse, ws_mse_bytes, stream,
                               first != 0, &so, &sel_done, first_in_grid ? s : nullptr))
        return rc;
    if (late) {
        so.enabled = 1;
        return fp8q_quantize_select_f32(x, y, inner, s->mses, s->grid, n_m, n_cand, &so, n_bits, sign_bits, (hipStream_t)stream);
    }
    if (!sel_done)
        if (int rc = fp8q_mse_select_f32(s->mses, s->grid, C, n_cand, mbits_host, n_m, sign_bits, s->mbits, s->vote, s->maxval, s->xmin,
                                         ws_select, ws_select_bytes, stream))
            return rc;
    if (!y) return FP8Q_OK;
    // the batch itself, with the range (and width) just chosen (:119-122: estimate, set the range, then quantize)
    if (n_m == 1) return fp8q_quantize_f32(x, y, C, inner, s->maxval, C, mbits_host[0], n_bits, sign_bits, stream);
    return fp8q_quantize_dm_f32(x, y, C, inner, s->maxval, C, s->mbits, n_bits, sign_bits, stream);
}

}  // extern "C"
