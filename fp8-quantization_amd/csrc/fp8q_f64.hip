// fp8q_f64.hip -- the float64 lane: K1, row min/max and the candidate search (K4) on float64 tensors.
//
// BASELINE config 1 (compute_quant_error.py:19-20) draws its samples in float64 and the reference evaluates the 1000
// candidates of LineSearchEstimator (range_estimators.py:161-169, 236-256) and the empirical check
// (quant_error_estimator.py:67-73) on them.  ATen's type promotion in quantize_to_fp8_ste_MM (fp8_quantizer.py:105-133)
// with x float64 and maxval / mantissa bits float32 tensors: M, E and bias are float32 (every operand of :105-110 is);
// xc, log2|xc| + bias, floor / clamp, the scale exponent, 2^e, the division, round and the product are float64.
//
// Arithmetic contract: bit-identical to oracle/fp8q_oracle.c:orc_quant1_f64 (log2 / 2^x in double = the table-driven
// evaluations shared with it; against the reference's 1-ulp Sleef routines: <= 2 ulp(double) per element, same grid point).
//   p   = floor(log2|xc| + bias) from the exponent field of t = |xc| * 2^bf (bias = bi + bf): exact unless t is within
//         2^-20 of a binade border; those lanes (~2^-19) evaluate floor(log2_tab_d(|xc|) + bias) as the oracle does.
//   s_p = 2^((p - M) - bias): the subtraction is exact in double (bias is a float32 multiple of 2^-24 below 2^9), so
//         s_p = 2^-bf * 2^(p - M - bi) -- the channel's g with the exponent field advanced: no table, no exp2 per element.
//         A finite float32 bias keeps every s_p a normal double (|e| < 450).
//   y   = rint(xc / s_p) * s_p with the IEEE double division (full-rate fp64 on CDNA4; K1 stays HBM-bound at ~30
//         instructions per 16 bytes of traffic).
// A non-finite bias (maxval 0 / inf / NaN / negative) makes every output NaN, as the reference's chain does.
#include "fp8q_common.h"

namespace {

typedef double vd2u __attribute__((ext_vector_type(2), aligned(8)));   // 16 bytes at 8-byte alignment: one dwordx4 access

struct Chan64 {
    double maxv, minv;   // clamp bounds (float32 values)
    double c1;           // 2^bf = 1 / g
    double bias_d;
    uint32_t ghi, glo;   // bits of g = 2^-bf in (0.5, 1]
    int koff;            // p = exponent_field(t) + koff
    int j0;              // exponent advance of s_p: p + j0 = p - M - bi
    int degenerate;      // non-finite bias: all NaN
};

__device__ __forceinline__ Chan64 make_chan64(float maxv, const QFmt &f)
{
    const Chan c = make_chan(maxv, f);
    Chan64 k;
    k.maxv = (double)c.maxv;
    k.minv = (double)c.minv;
    k.c1 = 1.0 / c.g;
    k.bias_d = c.bias_d;
    const uint64_t gb = (uint64_t)__double_as_longlong(c.g);
    k.ghi = (uint32_t)(gb >> 32);
    k.glo = (uint32_t)gb;
    k.koff = c.bi - 1023;
    k.j0 = -(int)f.M - c.bi;
    k.degenerate = !(fabsf(c.bias) < __builtin_inff());
    return k;
}

__device__ __forceinline__ double quant_f64(double x, const Chan64 &c)
{
    const double xc = fmin(fmax(x, c.minv), c.maxv);
    const double t = fabs(xc) * c.c1;
    const uint32_t hi = (uint32_t)__double2hiint(t);
    const uint32_t ef = hi >> 20;                       // exponent field (t >= 0)
    int p = max((int)ef + c.koff, 1);
    const uint32_t mh = hi & 0xfffffu;
    if (__builtin_expect((((mh + 1u) & 0xfffffu) <= 1u) & (ef != 0u) & (ef != 0x7ffu), 0)) {
        const double ls = floor(log2_tab_d(fabs(xc), kFastTab) + c.bias_d);   // the oracle's own decision at a border
        p = (int)fmax(ls, 1.0);
    }
    const double s = __hiloint2double((int)(c.ghi + ((uint32_t)(p + c.j0) << 20)), (int)c.glo);
    const double y = rint(xc / s) * s;
    return (x != x) ? x : (c.degenerate ? (double)__builtin_nanf("") : y);
}

// ---------------------------------------------------------------------------------------------
// K1 on float64: one aligned-ish 16 KiB piece (2048 doubles) of one row per block, 4 x 16 B in flight per lane.
// 16 B of traffic per element; rows are [C, inner] with the channel constants rebuilt per block (every lane evaluates
// them: ~150 instructions against 8 elements x ~30).
// ---------------------------------------------------------------------------------------------
constexpr int kPiece64 = 2048;

template <bool NT>
__global__ void __launch_bounds__(kBlock)
k_quant_f64(const double *__restrict__ x, double *__restrict__ y, int64_t inner, const float *__restrict__ maxval,
            int per_channel, QFmt f, int64_t ppr, int64_t total)
{
    for (int64_t blk = blockIdx.x; blk < total; blk += gridDim.x) {
        const int64_t row = blk / ppr, piece = blk - row * ppr;
        const Chan64 c = make_chan64(maxval[per_channel ? row : 0], f);
        const int64_t base = row * inner, e0 = piece * kPiece64;
        const int64_t n = min((int64_t)kPiece64, inner - e0);
        const double *xr = x + base + e0;
        double *yr = y + base + e0;
        if (n == kPiece64) {
            vd2u v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const vd2u *q = reinterpret_cast<const vd2u *>(xr + u * 512 + threadIdx.x * 2);
                v[u] = NT ? __builtin_nontemporal_load(q) : *q;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                vd2u w;
                w.x = quant_f64(v[u].x, c);
                w.y = quant_f64(v[u].y, c);
                vd2u *q = reinterpret_cast<vd2u *>(yr + u * 512 + threadIdx.x * 2);
                if (NT)
                    __builtin_nontemporal_store(w, q);
                else
                    *q = w;
            }
        } else {
            for (int64_t i = threadIdx.x; i < n; i += kBlock) yr[i] = quant_f64(xr[i], c);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Row min / max of float64 rows (two launches: plumbing of config 1, once per search).  NaN anywhere -> NaN (torch).
// ws: [C, nsplit, 3] doubles {min, max, nan flag}
// ---------------------------------------------------------------------------------------------
struct MinMax64 {
    double mn, mx;
    int nan;
};

__device__ __forceinline__ bool block_minmax64(MinMax64 &m)
{
    __shared__ double s_mn[4], s_mx[4];
    __shared__ int s_nan[4];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        m.mn = fmin(m.mn, __shfl_xor(m.mn, off, 64));
        m.mx = fmax(m.mx, __shfl_xor(m.mx, off, 64));
        m.nan |= __shfl_xor(m.nan, off, 64);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_mn[wave] = m.mn;
        s_mx[wave] = m.mx;
        s_nan[wave] = m.nan;
    }
    __syncthreads();
    if (threadIdx.x != 0) return false;
    m.mn = fmin(fmin(s_mn[0], s_mn[1]), fmin(s_mn[2], s_mn[3]));
    m.mx = fmax(fmax(s_mx[0], s_mx[1]), fmax(s_mx[2], s_mx[3]));
    m.nan = s_nan[0] | s_nan[1] | s_nan[2] | s_nan[3];
    return true;
}

__global__ void __launch_bounds__(kBlock)
k_minmax_f64_partial(const double *__restrict__ x, int64_t inner, int nsplit, double *__restrict__ ws)
{
    const int64_t row = blockIdx.y;
    const double *xr = x + row * inner;
    MinMax64 m = {(double)__builtin_inff(), -(double)__builtin_inff(), 0};
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < inner; i += (int64_t)nsplit * kBlock) {
        const double v = xr[i];
        m.nan |= (v != v);
        m.mn = fmin(m.mn, v);
        m.mx = fmax(m.mx, v);
    }
    if (block_minmax64(m)) {
        double *o = ws + (row * nsplit + blockIdx.x) * 3;
        o[0] = m.mn;
        o[1] = m.mx;
        o[2] = (double)m.nan;
    }
}

__global__ void __launch_bounds__(kBlock)
k_minmax_f64_final(const double *__restrict__ ws, int nsplit, double *__restrict__ mn, double *__restrict__ mx)
{
    const int64_t row = blockIdx.x;
    MinMax64 m = {(double)__builtin_inff(), -(double)__builtin_inff(), 0};
    for (int s2 = threadIdx.x; s2 < nsplit; s2 += kBlock) {
        const double *o = ws + (row * nsplit + s2) * 3;
        m.mn = fmin(m.mn, o[0]);
        m.mx = fmax(m.mx, o[1]);
        m.nan |= o[2] != 0.0;
    }
    if (block_minmax64(m)) {
        const double nanv = (double)__builtin_nanf("");
        mn[row] = m.nan ? nanv : m.mn;
        mx[row] = m.nan ? nanv : m.mx;
    }
}

// ---------------------------------------------------------------------------------------------
// K4 on float64 (ALU-bound, full-rate fp64): lane = candidate.  A block owns 256 candidates of one mantissa width and one
// row, and walks its share of the row in tiles of 1024 doubles staged in LDS; every lane reads the SAME element
// (broadcast ds_read_b128), runs the K1-f64 arithmetic for its own candidate and accumulates (x - q(x))^2 in a
// register -- no cross-lane reduction.  Squares are rounded and summed as the reference does ((data - y) ** 2, then
// torch.sum), 32 elements into a short accumulator, short accumulators into the block's: a two-level sum whose error
// (~1e-16 relative) is far below what separates two candidates.  Partial sums per (row, width, candidate, split) go to
// the workspace; k_sse_f64_final adds them in a fixed order (deterministic) and accumulates into `out`.
// ---------------------------------------------------------------------------------------------
constexpr int kSseTile = 1024;
constexpr int kSseMaxM = 8;

struct SseArgs {
    QFmt fmt[kSseMaxM];
    int n_m, n_cand, cgroups, nsplit, tpb;
    int64_t inner, C, ntiles;
};

__global__ void __launch_bounds__(kBlock)
k_sse_f64(const double *__restrict__ x, const float *__restrict__ grid, double *__restrict__ ws, SseArgs a)
{
    __shared__ __attribute__((aligned(16))) double xs[kSseTile];
    const int tid = threadIdx.x;
    const int split = blockIdx.x;
    const int m = blockIdx.y / a.cgroups;
    const int cand = (blockIdx.y - m * a.cgroups) * kBlock + tid;
    const int64_t c = blockIdx.z;
    const bool active = cand < a.n_cand;
    // set_quant_range(-g, g): maxval = |max(|-g|, g)|  (fp8_quantizer.py:236)
    const float gv = active ? grid[(int64_t)cand * a.C + c] : 1.0f;
    const Chan64 ch = make_chan64(fabsf(fmaxf(fabsf(-gv), gv)), a.fmt[m]);
    const double *xr = x + c * a.inner;
    double acc = 0.0;
    const int64_t t_begin = (int64_t)split * a.tpb, t_end = min(t_begin + a.tpb, a.ntiles);
    for (int64_t t = t_begin; t < t_end; ++t) {
        const int64_t e0 = t * kSseTile;
        const int n = (int)min((int64_t)kSseTile, a.inner - e0);
        __syncthreads();
        // zero padding: q(0) = 0 exactly, contributes nothing (a degenerate candidate is NaN anyway)
        for (int i = tid; i < kSseTile; i += kBlock) xs[i] = i < n ? xr[e0 + i] : 0.0;
        __syncthreads();
        const int n32 = (n + 31) & ~31;
        for (int j = 0; j < n32; j += 32) {
            double pa = 0.0;
#pragma unroll 4
            for (int u = 0; u < 32; u += 2) {
                const double2 v = *reinterpret_cast<const double2 *>(xs + j + u);
                const double d0 = v.x - quant_f64(v.x, ch);
                pa += d0 * d0;
                const double d1 = v.y - quant_f64(v.y, ch);
                pa += d1 * d1;
            }
            acc += pa;
        }
    }
    if (active) ws[(((c * a.n_m + m) * a.n_cand) + cand) * a.nsplit + split] = acc;
}

// out[m, i, c] += (sum over the splits) [/ inner]: one wave per (c, m, i), fixed summation order
__global__ void __launch_bounds__(kBlock)
k_sse_f64_final(const double *__restrict__ ws, double *__restrict__ out, int64_t C, int n_m, int n_cand, int64_t nsplit,
                double divisor)
{
    const int64_t total = C * n_m * n_cand;
    const int lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= total) return;
    double sum = 0.0;
    for (int64_t s2 = lane; s2 < nsplit; s2 += 64) sum += ws[j * nsplit + s2];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane == 0) {
        const int64_t c = j / ((int64_t)n_m * n_cand);
        const int64_t mi = j - c * n_m * n_cand;
        out[mi * C + c] += sum / divisor;
    }
}

struct SseGeo {
    int64_t ntiles;
    int nsplit, tpb, cgroups;
};

SseGeo sse_geo(int64_t C, int64_t inner, int64_t n_cand, int n_m)
{
    SseGeo g;
    g.ntiles = cdiv(inner, kSseTile);
    g.cgroups = (int)cdiv(n_cand, kBlock);
    int64_t cap = 8192 / (C * n_m * g.cgroups > 0 ? C * n_m * g.cgroups : 1);
    if (cap < 1) cap = 1;
    g.tpb = (int)cdiv(g.ntiles, cap);
    g.nsplit = (int)cdiv(g.ntiles, g.tpb);
    return g;
}

int minmax64_nsplit(int64_t C, int64_t inner)
{
    int64_t ns = cdiv(inner, (int64_t)kBlock * 16);
    int64_t cap = 4096 / (C > 0 ? C : 1);
    if (cap < 1) cap = 1;
    if (ns > cap) ns = cap;
    return (int)(ns < 1 ? 1 : ns);
}

}  // namespace

extern "C" {

int fp8q_quantize_f64(const double *x, double *y, int64_t C, int64_t inner, const float *maxval, int64_t n_maxval,
                      float mbits, int n_bits, int sign_bits, fp8q_stream_t stream)
{
    if (C < 0 || inner < 0 || (n_maxval != 1 && n_maxval != C)) return FP8Q_EINVAL;
    QFmt f;
    if (int rc = make_fmt(mbits, n_bits, sign_bits, &f)) return rc;
    if (C == 0 || inner == 0) return FP8Q_OK;
    if (!x || !y || !maxval || ((uintptr_t)x & 7) || ((uintptr_t)y & 7)) return FP8Q_EINVAL;
    const int64_t ppr = cdiv(inner, kPiece64), total = C * ppr;
    const unsigned grid = (unsigned)(total < (1 << 20) ? total : (1 << 20));
    const int per_channel = n_maxval != 1;
    hipStream_t st = (hipStream_t)stream;
    if (C * inner * 8 >= kNtBytes)
        hipLaunchKernelGGL(k_quant_f64<true>, dim3(grid), dim3(kBlock), 0, st, x, y, inner, maxval, per_channel, f, ppr, total);
    else
        hipLaunchKernelGGL(k_quant_f64<false>, dim3(grid), dim3(kBlock), 0, st, x, y, inner, maxval, per_channel, f, ppr, total);
    return launch_rc();
}

size_t fp8q_minmax_f64_workspace_bytes(int64_t C, int64_t inner)
{
    if (C <= 0 || inner <= 0) return 16;
    return (size_t)C * minmax64_nsplit(C, inner) * 3 * sizeof(double) + 16;
}

int fp8q_minmax_f64(const double *x, int64_t C, int64_t inner, double *row_min, double *row_max, void *ws,
                    size_t ws_bytes, fp8q_stream_t stream)
{
    if (!x || !row_min || !row_max || C <= 0 || inner <= 0 || ((uintptr_t)x & 7)) return FP8Q_EINVAL;
    if (C > 65535) return FP8Q_ETOOMANY;
    if (!ws || ws_bytes < fp8q_minmax_f64_workspace_bytes(C, inner) || ((uintptr_t)ws & 7)) return FP8Q_EWORKSPACE;
    const int ns = minmax64_nsplit(C, inner);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_minmax_f64_partial, dim3((unsigned)ns, (unsigned)C), dim3(kBlock), 0, st, x, inner, ns, (double *)ws);
    if (int rc = launch_rc()) return rc;
    hipLaunchKernelGGL(k_minmax_f64_final, dim3((unsigned)C), dim3(kBlock), 0, st, (const double *)ws, ns, row_min, row_max);
    return launch_rc();
}

size_t fp8q_mse_f64_workspace_bytes(int64_t C, int64_t inner, int64_t n_cand, int n_m)
{
    if (C <= 0 || inner <= 0 || n_cand <= 0 || n_m <= 0) return 16;
    return (size_t)C * n_m * n_cand * sse_geo(C, inner, n_cand, n_m).nsplit * sizeof(double) + 16;
}

int fp8q_mse_grid_f64(const double *x, int64_t C, int64_t inner, const float *grid, int64_t n_cand,
                      const float *mbits_host, int n_m, int n_bits, int sign_bits, double *out, int reduce_sum, void *ws,
                      size_t ws_bytes, fp8q_stream_t stream)
{
    if (!x || !grid || !mbits_host || !out || C <= 0 || inner <= 0 || n_cand <= 0 || n_m <= 0 || n_m > kSseMaxM ||
        n_cand > (1 << 20) || ((uintptr_t)x & 7))
        return FP8Q_EINVAL;
    if (C > 65535) return FP8Q_ETOOMANY;
    if (!ws || ws_bytes < fp8q_mse_f64_workspace_bytes(C, inner, n_cand, n_m) || ((uintptr_t)ws & 7)) return FP8Q_EWORKSPACE;
    SseArgs a;
    for (int m = 0; m < n_m; ++m)
        if (int rc = make_fmt(mbits_host[m], n_bits, sign_bits, &a.fmt[m])) return rc;
    const SseGeo g = sse_geo(C, inner, n_cand, n_m);
    a.n_m = n_m;
    a.n_cand = (int)n_cand;
    a.cgroups = g.cgroups;
    a.nsplit = g.nsplit;
    a.tpb = g.tpb;
    a.inner = inner;
    a.C = C;
    a.ntiles = g.ntiles;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_sse_f64, dim3((unsigned)g.nsplit, (unsigned)(n_m * g.cgroups), (unsigned)C), dim3(kBlock), 0, st, x,
                       grid, (double *)ws, a);
    if (int rc = launch_rc()) return rc;
    const int64_t rows = C * n_m * n_cand;
    hipLaunchKernelGGL(k_sse_f64_final, dim3((unsigned)cdiv(rows, 4)), dim3(kBlock), 0, st, (const double *)ws, out, C, n_m,
                       (int)n_cand, (int64_t)g.nsplit, reduce_sum ? 1.0 : (double)inner);
    return launch_rc();
}

}  // extern "C"
