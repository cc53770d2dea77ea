// fp8q_select.h -- the winner selection of FP_MSE_Estimator for ONE row (per-tensor quantizers, C == 1), appended to the
// launch that finishes the MSE table: the last workgroup to finish (a ticket) reads the n_m x n_cand entries and writes the
// voted mantissa width, its index, the winning clipping value and -sign * maxval (quantization/range_estimators.py:350-369
// with a single channel: the plurality vote is that channel's best width).  Saves the separate fp8q_mse_select_f32 launch
// (6.5 us each, 64 per MobileNetV2 calibration pass).  Included by fp8q_mse.hip and fp8q_mse_hist.hip.
#pragma once
#include "fp8q_common.h"

constexpr int kSelMaxM = 8;

// (global scope, the same in every translation unit: it crosses from fp8q_mse.hip into fp8q_mse_hist.hip)
struct SelOne {
    float *mbits_out;      // [1]
    int *vote_out;         // [1] or NULL
    float *maxval_out;     // [1]
    float *xmin_out;       // [1] or NULL
    unsigned *ticket;      // zero between launches (atomicInc wraps it back): word 1 of the selection workspace's header
    float sign;            // -sign_bits
    float M[kSelMaxM];     // the candidate widths as given (the vote returns one of them)
    int enabled;
};

namespace {

// torch.min / torch.argmin over one dimension: the first index of the smallest value, a NaN counting as smaller than
// everything (the first NaN wins).  Key = (isnan desc, value asc, index asc).
struct ArgMin {
    float v;
    int idx;
};

__device__ __forceinline__ bool argmin_less(const ArgMin &a, const ArgMin &b)
{
    const bool an = a.v != a.v, bn = b.v != b.v;
    if (an != bn) return an;
    if (!an && a.v != b.v) return a.v < b.v;
    return a.idx < b.idx;
}

__device__ __forceinline__ ArgMin wave_argmin(ArgMin a)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        ArgMin o;
        o.v = __shfl_xor(a.v, off, 64);
        o.idx = __shfl_xor(a.idx, off, 64);
        if (argmin_less(o, a)) a = o;
    }
    return a;
}

// All threads of the workgroup call this after their last store to the table; true (uniformly) in the workgroup that
// finishes last, which may then read what every other workgroup of the launch wrote.
__device__ __forceinline__ bool last_workgroup(unsigned *ticket, unsigned nwg)
{
    __shared__ int s_last_wg;
    if (nwg <= 1u) return true;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // this workgroup's table entries leave the XCD's L2 ...
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // ... before the ticket can be seen
    __syncthreads();
    if (threadIdx.x == 0) s_last_wg = atomicInc(ticket, nwg - 1u) == nwg - 1u;
    __syncthreads();
    if (!s_last_wg) return false;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return true;
}

// One workgroup (any multiple of 64 threads up to 1024): mses [n_m, n_cand] of the single row, grid [n_cand].
__device__ __forceinline__ void select_one_row(const float *mses, const float *grid, int n_m, int n_cand, const SelOne &so)
{
    __shared__ float s_v[16];
    __shared__ int s_i[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    ArgMin best_m = {__builtin_inff(), 0x7fffffff};
    int best_arg = 0;
    for (int m = 0; m < n_m; ++m) {
        ArgMin am = {__builtin_inff(), 0x7fffffff};
        for (int i = threadIdx.x; i < n_cand; i += blockDim.x) {
            const ArgMin o = {mses[(int64_t)m * n_cand + i], i};
            if (argmin_less(o, am)) am = o;
        }
        am = wave_argmin(am);
        __syncthreads();                       // (the previous width's slots are no longer read)
        if (lane == 0) {
            s_v[wave] = am.v;
            s_i[wave] = am.idx;
        }
        __syncthreads();
        for (int w = 0; w < nw; ++w) {
            const ArgMin o = {s_v[w], s_i[w]};
            if (w == 0 || argmin_less(o, am)) am = o;      // every thread: the block's argmin for this width
        }
        const ArgMin o = {am.v, m};
        if (argmin_less(o, best_m)) {
            best_m = o;
            best_arg = am.idx;
        }
    }
    if (threadIdx.x == 0) {
        const int v = best_m.idx;
        so.mbits_out[0] = so.M[v];
        if (so.vote_out) so.vote_out[0] = v;
        const float mv = grid[best_arg];
        so.maxval_out[0] = mv;
        if (so.xmin_out) so.xmin_out[0] = so.sign * mv;       // sign_bits * -1.0 * maxval (:369)
    }
}

}  // namespace
