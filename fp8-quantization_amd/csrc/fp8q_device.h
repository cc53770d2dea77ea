// fp8q_device.h -- per-element device arithmetic shared by every kernel (gfx950 only).
//
// Arithmetic contract (identical, bit for bit, to oracle/fp8q_oracle.c):
//   the reference's fp32 op chain quantize_to_fp8_ste_MM
//   (/root/reference/quantization/quantizers/fp8_quantizer.py:105-133) with log2 and 2^x
//   defined as correctly rounded fp32 functions.
//
// How each step is made exact AND cheap on CDNA4:
//   p = floor(log2|xc| + bias)   v_log_f32 (1 ulp) decides p unless the sum lands within
//                                2^-13 of an integer (or |xc| is denormal): those rare lanes
//                                (~2.4e-4) re-evaluate with a double-precision log2.
//   s = 2^fl32((p - M) - bias)   only (channel, p) matter, p <= 2^E <= 128:
//                                  * LUT kernels: one table per channel in LDS, built once per
//                                    block with scale_exact(); 1 ds_read per element.
//                                  * direct kernels: scale_exact() per element.
//                                scale_exact() never calls exp2 per element: with bias = bi + bf,
//                                2^e = 2^(k-bi) * g * 2^delta, g = 2^-bf (per channel, double),
//                                delta = fl32(k - bias) - (k - bias) (exact in double, |delta| <=
//                                2^-17) and 2^delta = 1 + u + u^2/2, u = delta*ln2 (error < 3e-17).
//   y = rint(xc / s) * s         IEEE fp32 division (hipcc default: correctly rounded), v_rndne.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fp8q {

constexpr int kBlock = 256;
constexpr int kLutMax = 132;                  // 2^7 + 1 entries, padded
constexpr float kNearEps = 1.220703125e-4f;   // 2^-13, see p_of()

// host-computed format constants (kernel argument, by value)
struct QFmt {
    float M;        // clamp(rint(mbits), 1, n_bits - sign_bits)      fp8_quantizer.py:105
    float two_E;    // 2^E, E = n_bits - sign_bits - M                 :106
    float l_c;      // fl32(log2(2 - 2^-M))                            :110
    int sign_bits;  // 1: clamp lo = -maxval, 0: clamp lo = 0          :112
    int pmax;       // 2^E: largest value p can take (LUT has pmax+1 entries)
};

struct __attribute__((aligned(16))) Chan {
    float maxv, minv, bias;
    int bi;          // floor(bias)
    double g;        // 2^-(bias - floor(bias))  in (0.5, 1]
    double bias_d;   // (double)bias
};

// bias, clamp bounds and scale constants of one channel (fp8_quantizer.py:108-113)
__device__ __forceinline__ Chan make_chan(float maxv, const QFmt &f)
{
    Chan c;
    c.maxv = maxv;
    c.minv = f.sign_bits == 1 ? -maxv : 0.0f;
    const float l_mv = (float)log2((double)maxv);  // correctly rounded fp32 log2
    float b = f.two_E - l_mv;                       // ((2^E - log2 maxval) + log2(2-2^-M)) - 1,
    b = b + f.l_c;                                  // every step rounded to fp32
    b = b - 1.0f;
    c.bias = b;
    const float fb = floorf(b);
    c.bi = (int)fminf(fmaxf(fb, -16384.0f), 16384.0f);  // non-finite bias: g is NaN anyway
    c.g = exp2(-(double)(b - fb));                  // b - fb is exact in fp32
    c.bias_d = (double)b;
    return c;
}

// correctly rounded fp32 value of 2^(fl32((ls - M) - bias)); ls is an integer-valued float
__device__ __forceinline__ float scale_exact(const Chan &c, float ls, float M)
{
    const float k = ls - M;                         // exact
    const float e = k - c.bias;                     // the reference's fp32 rounding
    const double d = (double)k - c.bias_d;          // exact k - bias
    const double delta = (double)e - d;             // exact rounding error of e
    const double u = delta * 0.69314718055994530942;
    const double t = fma(u, 0.5 * u, u);            // 2^delta - 1
    const double s = fma(c.g, t, c.g);              // g * 2^delta
    const int n = (int)k - c.bi;
    return (float)ldexp(s, n);                      // one rounding to fp32 (denormals included)
}

__device__ __forceinline__ float p_exact(float a, float bias)
{
    return floorf((float)log2((double)a) + bias);
}

// floor(fl32(log2_cr(a) + bias)) for a = |xc| >= 0.
// v_log_f32 is accurate to 1 ulp and |log2 a|, |bias| < 512, so the fast sum differs from the
// exact one by < 6e-5: the floors can only disagree if the fast sum is within 2^-13 of an integer.
__device__ __forceinline__ float p_of(float a, float bias)
{
    const float v = __builtin_amdgcn_logf(a) + bias;
    float fl = floorf(v);
    const float fr = v - fl;
    // class mask 0x90 = +/- denormal (v_log_f32 flushes denormal inputs)
    const bool risky = (fabsf(fr - 0.5f) > (0.5f - kNearEps)) | __builtin_amdgcn_classf(a, 0x90);
    if (__builtin_expect(risky, 0)) fl = p_exact(a, bias);
    return fl;   // NaN if a is NaN or (-inf + inf); -inf for a == 0
}

// NaN-propagating clamp: torch.min(torch.max(x, lo), hi).  x == NaN is patched by the caller's
// final select; v_max_f32(-0, +0) = +0 and v_min_f32(-0, +0) = -0 match ATen's CPU kernels.
__device__ __forceinline__ float clamp_ref(float x, const Chan &c)
{
    return fminf(fmaxf(x, c.minv), c.maxv);
}

// one element, scale from a per-channel LUT in LDS (lut[0] unused, lut[1..pmax])
__device__ __forceinline__ float quant_lut(float x, const Chan &c, const float *lut, float pmaxf)
{
    const float xc = clamp_ref(x, c);
    float ls = p_of(fabsf(xc), c.bias);
    ls = fminf(fmaxf(ls, 1.0f), pmaxf);             // clamp(min=1); NaN -> 1 (x NaN patched below,
    const float s = lut[(int)ls];                   // degenerate maxval makes every LUT entry NaN)
    const float y = rintf(xc / s) * s;
    return (x != x) ? x : y;
}

// one element, scale computed directly
__device__ __forceinline__ float quant_direct(float x, const Chan &c, float M)
{
    const float xc = clamp_ref(x, c);
    float ls = p_of(fabsf(xc), c.bias);
    ls = fmaxf(ls, 1.0f);
    const float s = scale_exact(c, ls, M);
    const float y = rintf(xc / s) * s;
    return (x != x) ? x : y;
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
