// fp8q_device.h -- per-element device arithmetic shared by every kernel (gfx950 only).
//
// Arithmetic contract (identical, bit for bit, to oracle/fp8q_oracle.c):
//   the reference's fp32 op chain quantize_to_fp8_ste_MM
//   (/root/reference/quantization/quantizers/fp8_quantizer.py:105-133) with log2 and 2^x
//   defined as correctly rounded fp32 functions:
//       xc = min(max(x, lo), maxval)
//       p  = max(floor(fl32(log2(|xc|) + bias)), 1)
//       s  = 2^fl32((p - M) - bias)
//       y  = rint(xc / s) * s            (IEEE division, round half to even)
//
// How each step is exact AND cheap on CDNA4 (~17 VALU ops per element on the fast path):
//   p  v_log_f32 (1 ulp) decides p unless the fp32 sum lands within 2^-15 of an integer; only
//      those lanes (~6e-5 of elements) re-evaluate with a double-precision log2.
//   s  depends on (channel, p) only and p <= 2^E <= 128: a per-channel table {s, 1/s} in LDS,
//      built once per block by scale_exact(); one ds_read_b64 per element.  scale_exact() needs
//      no exp2 per entry: with bias = bi + bf, 2^e = 2^(k-bi) * g * 2^delta, g = 2^-bf (one
//      double exp2 per channel), delta = fl32(k - bias) - (k - bias) (exact in double,
//      |delta| <= 2^-17) and 2^delta = 1 + u + u^2/2, u = delta ln 2 (error < 3e-17).
//   y  q0 = xc * rcp(s) (v_rcp_f32, 1 ulp) differs from fl32(xc / s) by < 2^-22 * q <= 2^(M-21);
//      rint(q0) is therefore rint(xc / s) unless q0 is within 2^(M-20) of a rounding tie; only
//      those lanes (~2e-5) redo the IEEE division.
//   Channels whose bias is outside (-100, 100) (maxval below 1e-28 or non-finite, ...) take the
//   exact path for every element (Chan::pthr = -1), so no range assumption leaks into results.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fp8q_tables.h"

namespace fp8q {

typedef float vf4 __attribute__((ext_vector_type(4)));
// 16 bytes at 4-byte alignment: still one global_load/store_dwordx4 on gfx950
typedef float vf4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int kBlock = 256;
constexpr int kLutMax = 130;   // 2^7 + 1 entries (+1 pad)

// host-computed format constants (kernel argument, by value)
struct QFmt {
    float M;        // clamp(rint(mbits), 1, n_bits - sign_bits)      fp8_quantizer.py:105
    float two_E;    // 2^E, E = n_bits - sign_bits - M                 :106
    float l_c;      // fl32(log2(2 - 2^-M))                            :110
    float qthr;     // 0.5 - 2^(M-20): |q0 - rint(q0)| above this -> redo the division exactly
    int sign_bits;  // 1: clamp lo = -maxval, 0: clamp lo = 0          :112
    int pmax;       // 2^E: largest value p can take (tables have pmax+1 entries)
};

struct __attribute__((aligned(16))) Chan {
    float maxv, minv, bias;
    float pthr;      // 0.5 - 2^-15, or -1: every element takes the exact path
    double g;        // 2^-(bias - floor(bias))  in (0.5, 1]
    double bias_d;   // (double)bias
    float m0;        // fl32(g)
    int bi;          // floor(bias), clamped (non-finite bias: g is NaN anyway)
    float bf;        // bias - floor(bias), exact
    float pad1;
};

// bias, clamp bounds and scale constants of one channel (fp8_quantizer.py:108-113), given the
// correctly rounded fp32 log2(maxval) and a function for g = 2^-frac(bias) in double
__device__ __forceinline__ Chan finish_chan(float maxv, float l_mv, const QFmt &f)
{
    Chan c;
    c.maxv = maxv;
    c.minv = f.sign_bits == 1 ? -maxv : 0.0f;
    float b = f.two_E - l_mv;                       // ((2^E - log2 maxval) + log2(2-2^-M)) - 1,
    b = b + f.l_c;                                  // every step rounded to fp32
    b = b - 1.0f;
    c.bias = b;
    // fast-path preconditions: every scale and its reciprocal are normal fp32 numbers
    c.pthr = (b > -100.0f && b < 100.0f && maxv < 0x1p120f) ? (0.5f - 0x1p-15f) : -1.0f;
    c.bias_d = (double)b;
    const float fb = floorf(b);
    c.bi = (int)fminf(fmaxf(fb, -16384.0f), 16384.0f);
    c.bf = b - fb;                                // frac(bias), exact in fp32
    c.pad1 = 0.0f;
    return c;
}

// Correctly rounded fp32 log2 of a >= 0 (NaN and negative -> NaN, 0 -> -inf, inf -> inf) without
// libm (ocml's double log2 costs ~100 VGPRs wherever it is inlined): table-driven double
// evaluation at < 2^-50 absolute error, so the fp32 rounding is libm's.  `tab` = kFastTab (global,
// or its LDS copy).  Works on the double's bits, so fp32 denormals need no special case.
//   log2(m * 2^k), m in [1,2): i = top 7 mantissa bits, r = m * rc_i - 1 (|r| <= 2^-8),
//                              log2 = k - log2(rc_i) + ln(1+r)/ln2, degree-7 series
__device__ __forceinline__ float log2_tab(float a, const double *tab)
{
    if (__builtin_expect(!(a > 0.0f) || a == __builtin_inff(), 0))
        return a == 0.0f ? -__builtin_inff() : (a > 0.0f ? a : __builtin_nanf(""));
    const uint64_t b = (uint64_t)__double_as_longlong((double)a);
    const int i = (int)(b >> 45) & 0x7f;
    const double m = __longlong_as_double((long long)((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull));
    const double r = fma(m, tab[i], -1.0);
    double p = fma(r, 1.0 / 7.0, -1.0 / 6.0);
    p = fma(r, p, 1.0 / 5.0);
    p = fma(r, p, -1.0 / 4.0);
    p = fma(r, p, 1.0 / 3.0);
    p = fma(r, p, -1.0 / 2.0);
    p = fma(r, p, 1.0);
    p = p * r;                                      // ln(1 + r)
    const double l2 = (double)((int)(b >> 52) - 1023) + fma(p, 1.4426950408889634074, tab[128 + i]);
    return (float)l2;
}

// The f64 lane's log2 (x float64: BASELINE config 1): the same table-driven evaluation returned in DOUBLE -- absolute error
// <= ~2^-52 (1 + |log2 a|), the accuracy class of the 1-ulp double log2 the reference calls -- and the very op sequence of
// oracle/fp8q_oracle.c:orc_log2_d, so the f64 kernels' rare exact path decides binade borders as the oracle does.
__device__ __forceinline__ double log2_tab_d(double a, const double *tab)
{
    if (__builtin_expect(!(a > 0.0) || a == (double)__builtin_inff(), 0))
        return a == 0.0 ? -(double)__builtin_inff() : (a > 0.0 ? a : (double)__builtin_nanf(""));
    int adj = 0;
    if (a < 0x1p-1022) {   // denormal double: renormalise
        a *= 0x1p64;
        adj = -64;
    }
    const uint64_t b = (uint64_t)__double_as_longlong(a);
    const int i = (int)(b >> 45) & 0x7f;
    const double m = __longlong_as_double((long long)((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull));
    const double r = fma(m, tab[i], -1.0);
    double p = fma(r, 1.0 / 7.0, -1.0 / 6.0);
    p = fma(r, p, 1.0 / 5.0);
    p = fma(r, p, -1.0 / 4.0);
    p = fma(r, p, 1.0 / 3.0);
    p = fma(r, p, -1.0 / 2.0);
    p = fma(r, p, 1.0);
    p = p * r;                                      // ln(1 + r)
    return (double)((int)(b >> 52) - 1023 + adj) + fma(p, 1.4426950408889634074, tab[128 + i]);
}

// Channel constants from maxval.  g = 2^-bf, bf in [0,1): j = floor(128 bf),
// 2^-bf = 2^(-j/128) * exp(-(bf - j/128) ln2), degree 6; a non-finite bias has bf = NaN -> g = NaN.
__device__ __forceinline__ Chan make_chan_fast(float maxv, const QFmt &f, const double *tab)
{
    Chan c = finish_chan(maxv, log2_tab(maxv, tab), f);
    if (__builtin_expect(!(c.bf >= 0.0f && c.bf < 1.0f), 0)) {
        c.g = (double)__builtin_nanf("");
    } else {
        const int j = (int)(c.bf * 128.0f);
        const double t = -((double)c.bf - (double)j * (1.0 / 128.0)) * 0.69314718055994530942;
        double q = fma(t, 1.0 / 720.0, 1.0 / 120.0);
        q = fma(t, q, 1.0 / 24.0);
        q = fma(t, q, 1.0 / 6.0);
        q = fma(t, q, 0.5);
        q = fma(t, q, 1.0);
        q = q * t;                                  // exp(t) - 1
        const double ej = tab[256 + j];
        c.g = fma(ej, q, ej);
    }
    c.m0 = (float)c.g;
    return c;
}

// once-per-block callers: tables straight from global memory (L1/L2-resident, 3 KB)
__device__ __forceinline__ Chan make_chan(float maxv, const QFmt &f) { return make_chan_fast(maxv, f, kFastTab); }

// cooperative copy of the tables into LDS (call once per block, then __syncthreads())
__device__ __forceinline__ void stage_fast_tab(double *dst)
{
    for (int i = threadIdx.x; i < kFastTabSize; i += kBlock) dst[i] = kFastTab[i];
}

// correctly rounded fp32 value of 2^(fl32((ls - M) - bias)); ls is an integer-valued float
__device__ __forceinline__ float scale_exact(const Chan &c, float ls, float M)
{
    // degenerate channel (maxval 0 / inf / NaN or negative -> bias +inf / -inf / NaN): 2^(k - bias) as the reference
    // chain evaluates it (0, inf, NaN).  K1 is NaN there either way; the decoder's value of such a channel follows.
    if (__builtin_expect(!(fabsf(c.bias) < __builtin_inff()), 0))
        return c.bias != c.bias ? c.bias : (c.bias > 0.0f ? 0.0f : __builtin_inff());
    const float k = ls - M;                         // exact
    const float e = k - c.bias;                     // the reference's fp32 rounding
    const double d = (double)k - c.bias_d;          // exact k - bias
    const double delta = (double)e - d;             // exact rounding error of e
    const double u = delta * 0.69314718055994530942;
    const double t = fma(u, 0.5 * u, u);            // 2^delta - 1
    const double s = fma(c.g, t, c.g);              // g * 2^delta
    return (float)ldexp(s, (int)k - c.bi);          // one rounding to fp32 (denormals included)
}

// {s, 1/s} table entry for p (entry 0 is never selected by a finite p; it holds NaN)
__device__ __forceinline__ float2 lut_entry(const Chan &c, int p, float M)
{
    if (p == 0) return make_float2(__builtin_nanf(""), __builtin_nanf(""));
    // Shortcut: fl32(k - bias) is EXACT unless |k - bias| falls in a higher binade than bias
    // (k - e == bias  <=>  no rounding happened).  Then 2^e = 2^(k - bi) * g exactly and, in the
    // normal range (pthr >= 0 guarantees it), its fp32 rounding is ldexp(fl32(g), k - bi).
    const float k = (float)p - M;
    const float e = k - c.bias;
    float s;
    if ((k - e) == c.bias && c.pthr >= 0.0f)
        s = ldexpf(c.m0, (int)k - c.bi);
    else
        s = scale_exact(c, (float)p, M);
    return make_float2(s, __builtin_amdgcn_rcpf(s));   // 1-ulp reciprocal: see QFmt::qthr
}

// The whole table of one channel, written by ONE thread (short-row kernels: thread <-> row).
// When fl32(k - bias) is exact at both ends of the table it is exact in between (the rounding
// error grows with |k - bias|), and every entry is ldexp(fl32(g), k - bi): see lut_entry().
__device__ __forceinline__ void lut_row(float2 *lr, const Chan &c, const QFmt &f)
{
    const float k1 = 1.0f - f.M, kp = (float)f.pmax - f.M;
    const bool lin = c.pthr >= 0.0f && (k1 - (k1 - c.bias)) == c.bias && (kp - (kp - c.bias)) == c.bias;
    lr[0] = make_float2(__builtin_nanf(""), __builtin_nanf(""));
    if (lin) {
        const int j0 = (int)k1 - c.bi - 1;
        for (int p = 1; p <= f.pmax; ++p) {
            const float sc = ldexpf(c.m0, j0 + p);
            lr[p] = make_float2(sc, __builtin_amdgcn_rcpf(sc));
        }
    } else {
        for (int p = 1; p <= f.pmax; ++p) lr[p] = lut_entry(c, p, f.M);
    }
}

// lut_row split over `step` lanes that all hold the channel's constants: lane `sub` writes entries sub, sub + step, ...
__device__ __forceinline__ void lut_part(float2 *lr, const Chan &c, const QFmt &f, int sub, int step)
{
    const float k1 = 1.0f - f.M, kp = (float)f.pmax - f.M;
    const bool lin = c.pthr >= 0.0f && (k1 - (k1 - c.bias)) == c.bias && (kp - (kp - c.bias)) == c.bias;
    if (sub == 0) lr[0] = make_float2(__builtin_nanf(""), __builtin_nanf(""));
    if (lin) {
        const int j0 = (int)k1 - c.bi - 1;
        for (int p = sub ? sub : step; p <= f.pmax; p += step) {
            const float sc = ldexpf(c.m0, j0 + p);
            lr[p] = make_float2(sc, __builtin_amdgcn_rcpf(sc));
        }
    } else {
        for (int p = sub ? sub : step; p <= f.pmax; p += step) lr[p] = lut_entry(c, p, f.M);
    }
}

// the three per-element channel constants the table kernels need
struct ChanLite {
    float maxv, minv, bias, pthr;
};

__device__ __forceinline__ ChanLite lite(const Chan &c)
{
    ChanLite l;
    l.maxv = c.maxv;
    l.minv = c.minv;
    l.bias = c.bias;
    l.pthr = c.pthr;
    return l;
}

// exact path of one element (rare lanes only)
__device__ __noinline__ float quant_exact(float x, float maxv, float minv, float bias,
                                          const float2 *lut, float pmaxf)
{
    if (x != x) return x;
    const float xc = __builtin_amdgcn_fmed3f(x, minv, maxv);
    float ls = floorf(log2_tab(fabsf(xc), kFastTab) + bias);
    ls = __builtin_amdgcn_fmed3f(ls, 1.0f, pmaxf);   // NaN -> 1 (then every table entry is NaN)
    const float s = lut[(int)ls].x;
    return rintf(xc / s) * s;
}

// Fast path of one element; sets `risky` when the exact path must redo it.
// class mask 0x93: sNaN | qNaN | -denormal | +denormal (v_log_f32 flushes denormals; a denormal
// xc can only come from a denormal x unless maxval itself is denormal, and then pthr == -1).
// CHECK_X = false: the caller guarantees that a NaN input implies an always-exact channel (the fused min/max+quantize
// kernels: a NaN anywhere in a row makes the row's maxval NaN, hence pthr = -1), so the per-element class test can go;
// denormal inputs need no exact path either: v_log_f32 flushes them to -inf -> p = 1, and with every scale normal
// (pthr >= 0) their quotient rounds to +-0 exactly as the reference's does.
template <bool CHECK_X = true>
__device__ __forceinline__ float quant_fast(float x, const ChanLite &c, const float2 *lut, float pmaxf,
                                            float qthr, bool &risky)
{
    const float xc = __builtin_amdgcn_fmed3f(x, c.minv, c.maxv);
    const float v = __builtin_amdgcn_logf(fabsf(xc)) + c.bias;
    const float fl = floorf(v);
    const float fr = v - fl;
    const float ls = __builtin_amdgcn_fmed3f(fl, 1.0f, pmaxf);
    const float2 t = lut[(int)ls];
    const float q0 = xc * t.y;
    const float r = rintf(q0);
    // (x == 0: v = -inf, fr = NaN, the comparison is false and p = 1 is already exact)
    risky = (CHECK_X && __builtin_amdgcn_classf(x, 0x93)) | (fabsf(fr - 0.5f) > c.pthr) | (fabsf(q0 - r) > qthr) |
            (c.pthr < 0.0f);
    return r * t.x;
}

// N elements of one channel, in place: one branch for the whole group
template <int N, bool CHECK_X = true>
__device__ __forceinline__ void quant_group(float (&v)[N], const ChanLite &c, const float2 *lut,
                                            float pmaxf, float qthr)
{
    float y[N];
    bool rk[N];
    bool any = false;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        y[j] = quant_fast<CHECK_X>(v[j], c, lut, pmaxf, qthr, rk[j]);
        any |= rk[j];
    }
    if (__builtin_expect(any, 0)) {
#pragma unroll
        for (int j = 0; j < N; ++j)
            if (rk[j]) y[j] = quant_exact(v[j], c.maxv, c.minv, c.bias, lut, pmaxf);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = y[j];
}

// one element whose channel varies per element (short-row kernels)
__device__ __forceinline__ float quant_one(float x, const ChanLite &c, const float2 *lut, float pmaxf,
                                           float qthr)
{
    bool risky;
    float y = quant_fast(x, c, lut, pmaxf, qthr, risky);
    if (__builtin_expect(risky, 0)) y = quant_exact(x, c.maxv, c.minv, c.bias, lut, pmaxf);
    return y;
}

// one element, no table (rows too short to amortise one): exact scale computed directly
__device__ __forceinline__ float quant_direct(float x, const Chan &c, float M)
{
    const float xc = __builtin_amdgcn_fmed3f(x, c.minv, c.maxv);
    const float a = fabsf(xc);
    const float v = __builtin_amdgcn_logf(a) + c.bias;
    float fl = floorf(v);
    const float fr = v - fl;
    const bool risky = __builtin_amdgcn_classf(x, 0x90) | (fabsf(fr - 0.5f) > c.pthr) | (c.pthr < 0.0f);
    if (__builtin_expect(risky, 0)) fl = floorf(log2_tab(a, kFastTab) + c.bias);
    const float ls = fmaxf(fl, 1.0f);
    const float s = scale_exact(c, ls, M);
    const float y = rintf(xc / s) * s;
    return (x != x) ? x : y;
}

// ---- storage codes --------------------------------------------------------
// The byte layout is the one the reference's enumerator defines (fp8_quantizer.py:13-41):
// [sign | E exponent bits | M fraction bits]; exponent code 0 is subnormal, the all-ones exponent
// is an ordinary binade (no inf/NaN codes).  In terms of K1's integers (p = binade index >= 1,
// r = rint(|xc| / s_p) <= 2^(M+1)):   r < 2^M (only when p == 1)  ->  exponent code 0, fraction r
//                                     r == 2^(M+1) (rounded up)   ->  exponent code p+1, fraction 0
//                                     otherwise                   ->  exponent code p, fraction r - 2^M
// so decode(code) = +/- (fraction + [exp != 0] * 2^M) * s_max(exp,1), the very product K1 forms.
__device__ __forceinline__ uint32_t encode_one(float x, const ChanLite &c, const float2 *lut, float pmaxf,
                                               float qthr, int M, int sign_shift)
{
    // same decisions as quant_fast / quant_exact, keeping r and the table index
    const float xc = __builtin_amdgcn_fmed3f(x, c.minv, c.maxv);
    const float v = __builtin_amdgcn_logf(fabsf(xc)) + c.bias;
    float fl = floorf(v);
    const float fr = v - fl;
    float ls = __builtin_amdgcn_fmed3f(fl, 1.0f, pmaxf);
    float2 t = lut[(int)ls];
    float q0 = xc * t.y;
    float r = rintf(q0);
    const bool risky = __builtin_amdgcn_classf(x, 0x93) | (fabsf(fr - 0.5f) > c.pthr) |
                       (fabsf(q0 - r) > qthr) | (c.pthr < 0.0f);
    if (__builtin_expect(risky, 0)) {
        if (x != x) return 0u;                      // the format has no NaN code: documented as +0
        fl = floorf(log2_tab(fabsf(xc), kFastTab) + c.bias);
        ls = __builtin_amdgcn_fmed3f(fl, 1.0f, pmaxf);
        t = lut[(int)ls];
        r = rintf(xc / t.x);
    }
    // r is an integer in [0, 2^(M+1)] unless the channel is degenerate (s = 0 / NaN: K1 gives NaN): code 0
    if (__builtin_expect(!(fabsf(r) <= (float)(2u << M)), 0)) return 0u;
    const uint32_t ri = (uint32_t)fabsf(r);
    const uint32_t m2 = 1u << M;
    uint32_t e = (uint32_t)ls, f = ri - m2;
    if (ri < m2) {
        e = 0u;
        f = ri;
    } else if (ri == 2u * m2) {
        e += 1u;
        f = 0u;
    }
    const uint32_t sign = sign_shift >= 0 ? ((__float_as_uint(r) >> 31) << sign_shift) : 0u;
    return sign | (e << M) | f;
}

// Four elements of one channel -> their four codes packed into a dword (byte k = element k): one rare-case branch for
// the group, and the code assembled without a branch -- with p = the binade index (1..2^E) and ri = |r| (0..2^(M+1)),
//     exponent-and-fraction field = (p << M) + ri - 2^M
// covers all three cases of encode_one(): ri < 2^M happens only for p == 1 and gives (0, ri); ri == 2^(M+1) carries into
// the exponent and gives (p + 1, 0); otherwise (p, ri - 2^M).  Degenerate channels (s = 0 / NaN: K1 gives NaN, r is not
// an integer in range) and NaN inputs encode as 0, as documented.
__device__ __forceinline__ uint32_t encode_group4(const float (&v)[4], const ChanLite &c, const float2 *lut, float pmaxf,
                                                  float qthr, int M, int sign_shift)
{
    // The field (p << M) + |r| - 2^M is formed in FLOAT -- the binade index is there as a float (ls), r is a finite
    // integer in [0, 2^(M+1)] unless `any` ends up set -- and v_cvt_pk_u8_f32 converts it and places it in its byte in one
    // instruction (3 slots per element instead of 7 integer ones); the sign of r is the sign of x (the quantizer is odd,
    // -0 keeps its sign: encode_one), so the four sign bits come from the top bytes of the inputs with two v_perm_b32.
    const float m2f = (float)(1u << M);
    float cf[4];
    bool any = c.pthr < 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float xc = __builtin_amdgcn_fmed3f(v[j], c.minv, c.maxv);
        const float w = __builtin_amdgcn_logf(fabsf(xc)) + c.bias;
        const float fl = floorf(w);
        const float fr = w - fl;
        const float ls = __builtin_amdgcn_fmed3f(fl, 1.0f, pmaxf);
        const float2 t = lut[(int)ls];
        const float q0 = xc * t.y;
        const float r = rintf(q0);
        cf[j] = fmaf(ls, m2f, fabsf(r)) - m2f;
        any |= __builtin_amdgcn_classf(v[j], 0x93) | (fabsf(fr - 0.5f) > c.pthr) | (fabsf(q0 - r) > qthr);
    }
    if (__builtin_expect(any, 0))   // rare: the exact path decides every element of the group (NaN inputs, degenerate channels)
        return encode_one(v[0], c, lut, pmaxf, qthr, M, sign_shift) | (encode_one(v[1], c, lut, pmaxf, qthr, M, sign_shift) << 8) |
               (encode_one(v[2], c, lut, pmaxf, qthr, M, sign_shift) << 16) | (encode_one(v[3], c, lut, pmaxf, qthr, M, sign_shift) << 24);
    uint32_t word = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) word = __builtin_amdgcn_cvt_pk_u8_f32(cf[j], j, word);
    if (sign_shift >= 0) {
        const uint32_t t01 = __builtin_amdgcn_perm(__float_as_uint(v[1]), __float_as_uint(v[0]), 0x0c0c0703u);
        const uint32_t t23 = __builtin_amdgcn_perm(__float_as_uint(v[3]), __float_as_uint(v[2]), 0x07030c0cu);
        word |= ((t01 | t23) & 0x80808080u) >> (7 - sign_shift);
    }
    return word;
}

__device__ __forceinline__ float decode_one(uint32_t code, const float2 *lut, int M, int sign_shift)
{
    const uint32_t m2 = 1u << M;
    const uint32_t body = sign_shift >= 0 ? (code & ((1u << sign_shift) - 1u)) : code;
    const uint32_t e = body >> M, f = body & (m2 - 1u);
    const float r = (float)(f + (e ? m2 : 0u));
    const float y = r * lut[e ? e : 1u].x;
    return (sign_shift >= 0 && ((code >> sign_shift) & 1u)) ? -y : y;
}

// ---- streaming memory access: 16 B per lane, optionally nontemporal -------------------------
template <bool NT>
__device__ __forceinline__ vf4 ld16(const vf4 *p)
{
    return NT ? __builtin_nontemporal_load(p) : *p;
}

template <bool NT>
__device__ __forceinline__ void st16(vf4 *p, vf4 v)
{
    if (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}

template <bool NT>
__device__ __forceinline__ vf4 ld16u(const float *p)
{
    const vf4u *q = reinterpret_cast<const vf4u *>(p);
    vf4u v = NT ? __builtin_nontemporal_load(q) : *q;
    return vf4{v.x, v.y, v.z, v.w};
}

template <bool NT>
__device__ __forceinline__ void st16u(float *p, vf4 v)
{
    vf4u *q = reinterpret_cast<vf4u *>(p);
    vf4u w = {v.x, v.y, v.z, v.w};
    if (NT)
        __builtin_nontemporal_store(w, q);
    else
        *q = w;
}

// ---- wave / block reductions (wave = 64 lanes) ---------------------------------------------
struct MinMax {
    float mn, mx;
    int nan;
};

__device__ __forceinline__ void mm_init(MinMax &m)
{
    m.mn = __builtin_inff();
    m.mx = -__builtin_inff();
    m.nan = 0;
}

__device__ __forceinline__ void mm_acc(MinMax &m, float v)
{
    m.nan |= (v != v);
    m.mn = fminf(m.mn, v);   // fminf/fmaxf drop NaN operands; the flag restores torch semantics
    m.mx = fmaxf(m.mx, v);
}

// one butterfly step inside a row of 16 lanes (DPP: no LDS crossbar latency); CTRL = quad_perm / row_half_mirror code
template <int CTRL>
__device__ __forceinline__ void mm_dpp(MinMax &m)
{
    const float mn = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m.mn), CTRL, 0xF, 0xF, true));
    const float mx = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m.mx), CTRL, 0xF, 0xF, true));
    m.nan |= __builtin_amdgcn_update_dpp(0, m.nan, CTRL, 0xF, 0xF, true);
    m.mn = fminf(m.mn, mn);
    m.mx = fmaxf(m.mx, mx);
}

__device__ __forceinline__ void mm_wave_reduce(MinMax &m)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        m.mn = fminf(m.mn, __shfl_xor(m.mn, off, 64));
        m.mx = fmaxf(m.mx, __shfl_xor(m.mx, off, 64));
        m.nan |= __shfl_xor(m.nan, off, 64);
    }
}

}  // namespace fp8q
