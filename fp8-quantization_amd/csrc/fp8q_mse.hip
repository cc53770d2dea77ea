// fp8q_mse.hip -- K4: FP-MSE grid search (FP_MSE_Estimator / LineSearchEstimator candidates in one pass over x).
#include "fp8q_common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// K4: FP-MSE grid search (range_estimators.py:337-347), ALU-bound.
// One lane = one candidate maxval; every lane walks the SAME x tile, broadcast out of LDS, so a
// candidate's squared error accumulates in one register and no cross-lane reduction exists.
// Block = 128 lanes (candidates i0..i0+127 of one mantissa width m, one row c, one split of the
// row).  Per-candidate scale LUT (and its reciprocal) in LDS, built with scale_exact(): the
// scales are the same numbers K1 uses; rint(xc * (1/s)) differs from rint(xc / s) only at exact
// ties, where |x - q| is the same either way.
// Dynamic LDS: float xs[kMseTile] | float lut[128 * stride] | float ilut[128 * stride]
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
    int64_t inner;
    int64_t C;
};

__global__ void __launch_bounds__(kMseBlock)
k_mse_grid(const float *__restrict__ x, const float *__restrict__ grid, double *__restrict__ ws,
           MseArgs a)
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

    // set_quant_range(-g, g): maxval = |max(|-g|, g)|  (fp8_quantizer.py:236)
    const float gv = active ? grid[(int64_t)cand * a.C + c] : 1.0f;
    const Chan ch = make_chan(fabsf(fmaxf(fabsf(-gv), gv)), f);
    lut[0] = __builtin_nanf("");
    for (int p = 1; p <= f.pmax; ++p) lut[p] = scale_exact(ch, (float)p, f.M);
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
    double acc = 0.0;

    for (int64_t t0 = (int64_t)split * kMseTile; t0 < a.inner; t0 += (int64_t)a.nsplit * kMseTile) {
        const int n = (int)((a.inner - t0) < kMseTile ? (a.inner - t0) : kMseTile);
        __syncthreads();
        for (int i = tid; i < kMseTile; i += kMseBlock) xs[i] = i < n ? xr[t0 + i] : 0.0f;
        __syncthreads();
        // zero padding: q(0) = 0 exactly, contributes nothing (degenerate maxval -> NaN anyway)
        const int n32 = (n + 31) & ~31;
        for (int j = 0; j < n32; j += 32) {
            float pa = 0.0f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float4 v = *reinterpret_cast<const float4 *>(xs + j + u * 4);
                const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float xv = e[q];
                    const float xc = __builtin_amdgcn_fmed3f(xv, ch.minv, ch.maxv);
                    const float tt = xc * c1;
                    int e8 = (int)__builtin_amdgcn_ubfe(__float_as_uint(tt), 23u, 8u);
                    e8 = max(min(e8, e_hi), e_lo);
                    const float r = rintf(ldexpf(tt, jk - e8));
                    const float d = xv - r * lutk[e8];
                    pa = fmaf(d, d, pa);
                }
            }
            acc += (double)pa;
        }
    }
    if (active) ws[((c * a.n_m + m) * a.n_cand + cand) * a.nsplit + split] = acc;
}

// mses[m, i, c] += sum_over_splits / inner
__global__ void __launch_bounds__(kBlock)
k_mse_final(const double *__restrict__ ws, float *__restrict__ mses, int64_t C, int n_m, int n_cand,
            int nsplit, double inv_inner)
{
    const int64_t total = C * n_m * n_cand;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < total;
         j += (int64_t)gridDim.x * kBlock) {
        // j indexes ws rows: ((c * n_m + m) * n_cand + i)
        const int64_t c = j / ((int64_t)n_m * n_cand);
        const int64_t mi = j - c * n_m * n_cand;   // m * n_cand + i
        double sum = 0.0;
        for (int s2 = 0; s2 < nsplit; ++s2) sum += ws[j * nsplit + s2];
        mses[mi * C + c] += (float)(sum * inv_inner);
    }
}

}  // namespace

extern "C" {

static int mse_nsplit(int64_t C, int64_t inner, int64_t n_cand, int n_m)
{
    const int64_t cg = cdiv(n_cand, kMseBlock);
    int64_t ns = cdiv(inner, kMseTile);
    int64_t cap = (4 * kTargetBlocks) / (C * n_m * cg > 0 ? C * n_m * cg : 1);
    if (cap < 1) cap = 1;
    if (ns > cap) ns = cap;
    if (ns < 1) ns = 1;
    return (int)ns;
}

size_t fp8q_mse_workspace_bytes(int64_t C, int64_t inner, int64_t n_cand, int n_m)
{
    if (C <= 0 || inner <= 0 || n_cand <= 0 || n_m <= 0) return 16;
    return (size_t)C * n_m * n_cand * mse_nsplit(C, inner, n_cand, n_m) * sizeof(double) + 16;
}

int fp8q_mse_grid_f32(const float *x, int64_t C, int64_t inner, const float *grid, int64_t n_cand,
                      const float *mbits_host, int n_m, int n_bits, int sign_bits, float *mses,
                      void *ws, size_t ws_bytes, fp8q_stream_t stream)
{
    if (!x || !grid || !mbits_host || !mses || C <= 0 || inner <= 0 || n_cand <= 0 || n_m <= 0 ||
        n_m > kMseMaxM || n_cand > (1 << 20))
        return FP8Q_EINVAL;
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
    a.inner = inner;
    a.C = C;
    hipStream_t st = (hipStream_t)stream;
    const size_t shmem = (size_t)kMseTile * 4 + (size_t)kMseBlock * ((pmax_all + 1) | 1) * sizeof(float);
    if (shmem > 64 * 1024) {
        static int opted = 0;
        if (!opted) {
            hipError_t e = hipFuncSetAttribute((const void *)k_mse_grid,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
            opted = 1;
        }
    }
    for (int64_t c0 = 0; c0 < C; c0 += 65535) {
        // gridDim.z limit: rows are processed in slabs; ws/grid/mses keep their global indexing
        const int64_t cn = (C - c0) < 65535 ? (C - c0) : 65535;
        if (c0 != 0) return FP8Q_EUNSUPPORTED;  // > 65535 channels: not needed by any model here
        hipLaunchKernelGGL(k_mse_grid, dim3((unsigned)a.nsplit, (unsigned)(n_m * a.cgroups), (unsigned)cn),
                           dim3(kMseBlock), shmem, st, x, grid, (double *)ws, a);
    }
    int64_t fb = cdiv(C * n_m * n_cand, kBlock);
    if (fb > kTargetBlocks) fb = kTargetBlocks;
    hipLaunchKernelGGL(k_mse_final, dim3((unsigned)fb), dim3(kBlock), 0, st, (const double *)ws, mses, C,
                       n_m, (int)n_cand, a.nsplit, 1.0 / (double)inner);
    return launch_rc();
}

}  // extern "C"
