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
    unsigned *ticket;      // the selection workspace's ticket block (its first kTicketBytes: zero between launches)
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

// Cross-workgroup hand-over WITHOUT fences: what the last workgroup reads is written with agent-scope atomic stores
// (write-through: performed at the device's coherence point once vmcnt says so) and read with agent-scope atomic loads (served
// past this CU's L1 and the XCD's L2).  An agent-scope RELEASE fence instead writes back the XCD's whole L2: with one per
// workgroup k_mse_eval went from 10.9 to 48.2 us on 666 workgroups (profiles/r06_select_fence_ab.txt).
__device__ __forceinline__ void agent_store(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void agent_store(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float agent_load(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int agent_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The ticket block: zero between launches (every counter wraps back to zero when its last member arrives).  A single
// counter serialises one returning atomic per workgroup on one address (666 workgroups of k_mse_eval: +15 us); here the
// workgroups are dealt round-robin to up to 31 groups whose counters lie 128 bytes apart, and only a group's last member
// goes on to the launch's own counter (word 1): the longest chain on one address is ~nwg / 31 + 31.
constexpr size_t kTicketBytes = 4096;       // word 1: the launch; words 32 g, g = 1 .. 31: the groups
constexpr unsigned kTicketGroups = 31;

// All threads of the workgroup call this after their last agent_store() to the data the last workgroup will read; true
// (uniformly) in the workgroup that finishes last -- which must read the others' data with agent_load().
// `wg`: this workgroup's linear index in the launch.
__device__ __forceinline__ bool last_workgroup(unsigned *tickets, unsigned nwg, unsigned wg)
{
    __shared__ int s_last_wg;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's stores are performed ...
    __syncthreads();                                       // ... and every other thread's of the workgroup
    if (nwg <= 1u) return true;
    if (threadIdx.x == 0) {
        unsigned *top = tickets + 1;
        const unsigned ng = nwg <= 32u ? 1u : (nwg / 16u < kTicketGroups ? nwg / 16u : kTicketGroups);
        int last;
        if (ng <= 1u) {
            last = atomicInc(top, nwg - 1u) == nwg - 1u;
        } else {
            const unsigned g = wg % ng, members = nwg / ng + (g < nwg % ng ? 1u : 0u);
            last = 0;
            if (atomicInc(tickets + 32u * (g + 1u), members - 1u) == members - 1u) last = atomicInc(top, ng - 1u) == ng - 1u;
        }
        s_last_wg = last;
    }
    __syncthreads();
    return s_last_wg != 0;
}

// One workgroup (any multiple of 64 threads up to 1024): mses [n_m, n_cand] of the single row, grid [n_cand].
// The entries of ALL widths are requested before any is used: an agent-scope load is a round trip past the caches (~2 us), and
// a loop over the widths with one load and two barriers each made the selection 10 us of k_mse_eval with the six widths of the
// mantissa search (21.6 us per launch on every MobileNetV2 activation; 8.3 us with one width).
__device__ __forceinline__ void select_one_row(const float *mses, const float *grid, int n_m, int n_cand, const SelOne &so)
{
    __shared__ float s_v[kSelMaxM * 16];
    __shared__ int s_i[kSelMaxM * 16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    ArgMin am[kSelMaxM];
#pragma unroll
    for (int m = 0; m < kSelMaxM; ++m) am[m] = ArgMin{__builtin_inff(), 0x7fffffff};
    for (int i = threadIdx.x; i < n_cand; i += blockDim.x) {
        float t[kSelMaxM];
#pragma unroll
        for (int m = 0; m < kSelMaxM; ++m) t[m] = m < n_m ? agent_load(mses + (int64_t)m * n_cand + i) : 0.0f;
#pragma unroll
        for (int m = 0; m < kSelMaxM; ++m) {
            const ArgMin o = {t[m], i};
            if (m < n_m && argmin_less(o, am[m])) am[m] = o;
        }
    }
#pragma unroll
    for (int m = 0; m < kSelMaxM; ++m) {
        if (m < n_m) {                             // (uniform)
            const ArgMin w = wave_argmin(am[m]);
            if (lane == 0) {
                s_v[m * 16 + wave] = w.v;
                s_i[m * 16 + wave] = w.idx;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        ArgMin best_m = {__builtin_inff(), 0x7fffffff};
        int best_arg = 0;
        for (int m = 0; m < n_m; ++m) {
            ArgMin a = {s_v[m * 16], s_i[m * 16]};
            for (int w = 1; w < nw; ++w) {
                const ArgMin o = {s_v[m * 16 + w], s_i[m * 16 + w]};
                if (argmin_less(o, a)) a = o;      // the block's argmin for this width
            }
            const ArgMin o = {a.v, m};
            if (argmin_less(o, best_m)) {
                best_m = o;
                best_arg = a.idx;
            }
        }
        const int v = best_m.idx;
        so.mbits_out[0] = so.M[v];
        if (so.vote_out) so.vote_out[0] = v;
        const float mv = grid[best_arg];
        so.maxval_out[0] = mv;
        if (so.xmin_out) so.xmin_out[0] = so.sign * mv;       // sign_bits * -1.0 * maxval (:369)
    }
}

}  // namespace
