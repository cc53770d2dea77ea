// fp8q_common.h -- what every translation unit of libfp8q_hip.so shares: launch-geometry helpers, the running-estimate
// fold, the block-level min/max reduction, the second stage of the two-stage min/max, format setup and error mapping.
// Everything has internal linkage (anonymous namespace): each .hip file gets its own copy.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/fp8q.h"
#include "fp8q_device.h"

using namespace fp8q;

// short-row path of the storage codec: lives with k_rows_flat in fp8q_quant.hip, called from fp8q_codec.hip
constexpr int FP8Q_CODEC_NOT_FLAT = -1000;
int fp8q_codec_flat_launch(bool encode, const void *in, void *out, int64_t C, int64_t inner, const float *maxval,
                           const QFmt &f, int n_bits, hipStream_t st);

namespace {

constexpr int kUnroll = 4;           // 16-byte loads in flight per lane
constexpr int kTargetBlocks = 2048;  // 256 CUs x 8 blocks
constexpr int kDirectMaxInner = 16384; // k_rows_direct handles rows up to here (magic division: n*inner < 2^32)
constexpr int kDirectElems = 32768;    // elements per k_rows_direct iteration (tables capped at 40 KiB)
// tensors at least this big stream with nontemporal hints (FP8Q_NT_MB: the threshold in MiB, a tuning knob)
static const int64_t kNtBytes = [] {
    const char *e = getenv("FP8Q_NT_MB");
    const long v = e ? atol(e) : 0;
    return (int64_t)(v >= 1 ? v : 64) << 20;
}();

// Blocks for `pieces` equal pieces of work with at most `cap` blocks: every block gets the same number of
// steps (a persistent grid of exactly `cap` blocks over 6.1 steps' worth of pieces runs 7 steps: -12 %).
inline int64_t balanced_blocks(int64_t pieces, int64_t cap)
{
    if (cap < 1) cap = 1;
    if (pieces <= cap) return pieces < 1 ? 1 : pieces;
    const int64_t steps = (pieces + cap - 1) / cap;
    return (pieces + steps - 1) / steps;
}

constexpr int kMagicMaxDivisor = 60000;   // (4096 + d) * d < 2^32: largest divisor div_small takes for piece-local offsets

// n / d == umulhi(n, magic) for n * d < 2^32; magic == 0 encodes d == 1
inline uint32_t magic_of(int d) { return d <= 1 ? 0u : (uint32_t)((1ull << 32) / (uint64_t)d) + 1u; }

__device__ __forceinline__ int div_small(uint32_t n, uint32_t magic)
{
    return magic == 0u ? (int)n : (int)__umulhi(n, magic);
}

// torch.min / torch.max of two values (NaN from either side wins)
__device__ __forceinline__ float tmin(float a, float b) { return (a != a) ? a : ((b != b) ? b : fminf(a, b)); }
__device__ __forceinline__ float tmax(float a, float b) { return (a != a) ? a : ((b != b) ? b : fmaxf(a, b)); }

struct FoldArgs {
    int mode;     // FP8Q_FOLD_*
    int first;    // no previous estimate
    float om;     // fl32(1 - momentum)   (python double arithmetic, then cast: range_estimators.py:122)
    float mo;     // fl32(momentum)
    float *packed = nullptr;     // optional [C, 4] {-min, max, isnan(min), isnan(max)} of the folded estimate: the operand of
                                 // ONE all-reduce(MAX) in batch-sharded calibration (fp8q_minmax_packed_f32, include/fp8q.h)
    unsigned *status = nullptr;  // workspace header word: the reducer counts its timeouts here (FP8Q_ETIMEDOUT)
    int spin_limit = 1 << 23;    // polls before the reducer gives up (~2 s); FP8Q_K3_SPIN_LIMIT
    int fault = 0;               // FP8Q_TEST_FAULT=drop_publish: split 0 never publishes (exercises the timeout path)
    // optional: the MSE estimator's search grid of the row, written by the thread that stores the row's range
    // (fp8q_minmax_linspace_f32: the first calibration batch needs max|x| and linspace(lo * max|x|, hi * max|x|, steps))
    float *lin_grid = nullptr;   // [lin_steps, lin_C]
    int lin_steps = 0;
    int64_t lin_C = 0;
    double lin_lo = 0.0, lin_hi = 0.0;
};

// grid[i, row] = torch.linspace(lo * mx, hi * mx, steps)[i] bit for bit (range_estimators.py:296-305: the products are
// python-float (double) multiplications of mx.item(), narrowed to float32 by linspace).  ATen's CPU kernel, for fewer steps
// than its parallel grain, evaluates element i as fl32(start + step * i) for i < steps / 2 and fl32(end - step * (steps - 1 - i))
// after, each with one fused multiply-add, step = fl32(fl32(end - start) / (steps - 1)).
__device__ __forceinline__ float linspace_at(float mx, double lo_frac, double hi_frac, int steps, int i)
{
    const double m = (double)mx;
    const float start = (float)(lo_frac * m), end = (float)(hi_frac * m);
    const float step = (end - start) / (float)(steps - 1);
    return i < steps / 2 ? fmaf(step, (float)i, start) : fmaf(-step, (float)(steps - 1 - i), end);
}

// spin limit / fault injection of the single-launch min/max, read once (tests shorten the 2 s and drop a publisher)
inline void fold_debug_env(FoldArgs &fa)
{
    static const int limit = [] {
        const char *e = getenv("FP8Q_K3_SPIN_LIMIT");
        const long v = e ? atol(e) : 0;
        return v >= 1 && v <= (1l << 30) ? (int)v : 1 << 23;
    }();
    static const int fault = [] {
        const char *e = getenv("FP8Q_TEST_FAULT");
        return e && !strcmp(e, "drop_publish") ? 1 : 0;
    }();
    static const bool warned = [] {   // test hooks in a production library: never honoured silently
        if (fault || limit != (1 << 23))
            fprintf(stderr, "fp8q: WARNING: test hooks active (FP8Q_TEST_FAULT=%s, FP8Q_K3_SPIN_LIMIT=%d): split-row min/max "
                            "ranges may time out to NaN (reported by fp8q_minmax_workspace_check); unset them outside the test suite\n",
                    fault ? "drop_publish" : "off", limit);
        return true;
    }();
    (void)warned;
    fa.spin_limit = limit;
    fa.fault = fault;
}

constexpr size_t kMinmaxWsHeader = 16;   // bytes in front of the granules: {timeout count, 3 reserved words}

// (returns the row's max|x| after the fold)
__device__ __forceinline__ float fold_store(float mn, float mx, int64_t row, float *cur_min,
                                            float *cur_max, float *maxval_out, const FoldArgs &fa)
{
    if (!fa.first && fa.mode == FP8Q_FOLD_ALL) {
        mn = tmin(cur_min[row], mn);
        mx = tmax(cur_max[row], mx);
    } else if (!fa.first && fa.mode == FP8Q_FOLD_RUNNING) {
        // (1-m)*new + m*cur as three separately rounded fp32 ops (no FMA: -ffp-contract=off)
        mn = fa.om * mn + fa.mo * cur_min[row];
        mx = fa.om * mx + fa.mo * cur_max[row];
    }
    if (cur_min) cur_min[row] = mn;
    if (cur_max) cur_max[row] = mx;
    const float absmax = fabsf(tmax(fabsf(mn), mx));               // fp8_quantizer.py:236
    if (maxval_out) maxval_out[row] = absmax;
    if (fa.lin_grid)
        for (int i = 0; i < fa.lin_steps; ++i) fa.lin_grid[(int64_t)i * fa.lin_C + row] = linspace_at(absmax, fa.lin_lo, fa.lin_hi, fa.lin_steps, i);
    if (fa.packed) {
        // NaN must win on every rank (torch.min / torch.max): it travels as a flag, the value as -inf, so that the
        // collective itself never sees a NaN (what MAX does with one is the communication library's business)
        const bool nmn = mn != mn, nmx = mx != mx;
        const float ninf = -__builtin_inff();
        reinterpret_cast<float4 *>(fa.packed)[row] =
            make_float4(nmn ? ninf : -mn, nmx ? ninf : mx, nmn ? 1.0f : 0.0f, nmx ? 1.0f : 0.0f);
    }
    return absmax;
}

// ---------------------------------------------------------------------------------------------
// K2/K3 two-stage min/max in ONE launch (a separate stage-2 launch cost ~8 of the 38 us of a [64,64,112,112]
// activation).  Every streaming block publishes {min, max} of its part of a row as two tagged 8-byte granules
// {value, tag} with agent-scope atomic stores (global_store_dwordx2 sc1: write-through, untorn) and exits -- no wait,
// no ticket.  One extra block per row (blockIdx.x == nsplit) is the reducer: it polls the row's granules until every
// tag has arrived, reduces, folds into the running estimate (+ K5) and clears the granules again.
// Measured and rejected: last-block-done tickets -- an agent-scope release fence per block writes back the XCD's whole
// L2 (38 -> 105 us); sc1 partials + a ticket atomic per block serialise ~1600 same-address atomics behind two memory
// round trips at every block's end (45 us).
// Workspace contract (include/fp8q.h): zero before the first use of a buffer, zero again after every call.  `tag` is a
// per-call nonzero value from the host, so a stale or foreign word is not mistaken for an arrival.
// The reducer spins: bounded (~2 s), after which it reports NaN instead of hanging the queue AND counts the event in the
// workspace's header word (fp8q_minmax_workspace_check -> FP8Q_ETIMEDOUT: the enqueue-only entry point itself cannot know).
// Progress does not depend on the dispatch order: streaming blocks never wait for anything, and the reducers (one per
// row; rows are split only when C <= 1024) can hold at most half of the chip's 2048 resident 256-thread workgroup
// slots, so streaming blocks always find a slot even if every reducer were dispatched first.  What in-order dispatch
// (blockIdx.x == nsplit last: what the hardware does, not something HIP promises) buys is only that the reducer spins
// briefly instead of for the whole kernel.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long pack_tagged(float v, unsigned tag)
{
    return (unsigned long long)__float_as_uint(v) | ((unsigned long long)tag << 32);
}

// block-level reduction of m -> thread 0 holds {mn, mx} (NaN-propagating); returns true in thread 0 only
__device__ __forceinline__ bool block_minmax(MinMax m, float &mn, float &mx)
{
    __shared__ float s_mn[4], s_mx[4];
    __shared__ int s_nan[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    mm_wave_reduce(m);
    if (lane == 0) {
        s_mn[wave] = m.mn;
        s_mx[wave] = m.mx;
        s_nan[wave] = m.nan;
    }
    __syncthreads();
    if (tid != 0) return false;
    mn = fminf(fminf(s_mn[0], s_mn[1]), fminf(s_mn[2], s_mn[3]));
    mx = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
    if (s_nan[0] | s_nan[1] | s_nan[2] | s_nan[3]) mn = mx = __builtin_nanf("");
    return true;
}

// a streaming block's exit: publish (or, when the row has a single block, fold directly)
__device__ __forceinline__ void block_minmax_publish(MinMax m, unsigned long long *slots /* [2 * nsplit] of this row */,
                                                     int split, int nsplit, unsigned tag, int64_t row, float *cur_min,
                                                     float *cur_max, float *maxval_out, const FoldArgs &fa)
{
    float mn, mx;
    if (!block_minmax(m, mn, mx)) return;
    if (nsplit == 1) {
        fold_store(mn, mx, row, cur_min, cur_max, maxval_out, fa);
    } else if (!(fa.fault && split == 0)) {
        __hip_atomic_store(slots + 2 * split, pack_tagged(mn, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(slots + 2 * split + 1, pack_tagged(mx, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// the reducer block of a row
constexpr int kCollectBatch = 4;
__device__ __forceinline__ void block_minmax_collect(unsigned long long *slots, int nsplit, unsigned tag, int64_t row,
                                                     float *cur_min, float *cur_max, float *maxval_out, const FoldArgs &fa)
{
    const int tid = threadIdx.x;
    MinMax m;
    mm_init(m);
    int lost = 0;
    // kCollectBatch granule pairs per thread are requested together (round 6): an agent-scope load is a round trip past the
    // caches (~2 us), and a thread that polled its slots one after the other made a row of 2048 partials a chain of eight
    for (int s0 = 0; s0 < nsplit; s0 += kBlock * kCollectBatch) {
        unsigned long long a[kCollectBatch], b[kCollectBatch];
#pragma unroll
        for (int q = 0; q < kCollectBatch; ++q) {
            const int s2 = s0 + q * kBlock + tid;
            a[q] = b[q] = 0ull;
            if (s2 < nsplit) {
                a[q] = __hip_atomic_load(slots + 2 * s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b[q] = __hip_atomic_load(slots + 2 * s2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#pragma unroll
        for (int q = 0; q < kCollectBatch; ++q) {
            const int s2 = s0 + q * kBlock + tid;
            if (s2 >= nsplit) continue;
            int spins = 0;
            // agent-scope atomic loads (sc1): served past this CU's L1, see other XCDs' write-through stores
            while (!(((unsigned)(a[q] >> 32) == tag) & ((unsigned)(b[q] >> 32) == tag))) {
                if (++spins > fa.spin_limit) {   // ~2 s: something upstream died; do not hang the queue
                    lost = 1;
                    if (fa.status) atomicAdd(fa.status, 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
                a[q] = __hip_atomic_load(slots + 2 * s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b[q] = __hip_atomic_load(slots + 2 * s2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const float mn = __uint_as_float((unsigned)a[q]), mx = __uint_as_float((unsigned)b[q]);
            // an EMPTY split (a row a few elements longer than a whole number of steps) holds {+inf, -inf}
            m.nan |= (mn != mn) | (mx != mx) | lost;
            m.mn = fminf(m.mn, mn);
            m.mx = fmaxf(m.mx, mx);
            // consumed: back to zero for the next call on this workspace (ordered behind this kernel: same stream)
            __hip_atomic_store(slots + 2 * s2, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(slots + 2 * s2 + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // the search grid of the row's maximum (first batch of an MSE estimator) is written by the whole workgroup: 111 entries
    // of double arithmetic one after the other in thread 0 were ~2 us at the end of every such launch
    __shared__ float s_absmax;
    float mn, mx;
    if (block_minmax(m, mn, mx)) {
        FoldArgs f2 = fa;
        f2.lin_grid = nullptr;
        s_absmax = fold_store(mn, mx, row, cur_min, cur_max, maxval_out, f2);
    }
    if (fa.lin_grid) {
        __syncthreads();
        const float absmax = s_absmax;
        for (int i = tid; i < fa.lin_steps; i += kBlock)
            fa.lin_grid[(int64_t)i * fa.lin_C + row] = linspace_at(absmax, fa.lin_lo, fa.lin_hi, fa.lin_steps, i);
    }
}

// per-call tag of the granules: nonzero, different from call to call (a multiplicative hash of a counter; this header is
// compiled into several translation units, each with its own counter, so the counter's address salts the hash)
inline unsigned next_minmax_tag()
{
    static std::atomic<unsigned> counter{0};
    const unsigned salt = (unsigned)((uintptr_t)&counter >> 4);
    const unsigned t = ((counter.fetch_add(1, std::memory_order_relaxed) + 1u) ^ salt) * 0x9E3779B1u;
    return t ? t : 0x9E3779B1u;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
int make_fmt(float mbits, int n_bits, int sign_bits, QFmt *f)
{
    if (!(mbits == mbits) || n_bits < 2 || n_bits > 16 || (sign_bits != 0 && sign_bits != 1))
        return FP8Q_EINVAL;
    float M = nearbyintf(mbits);  // round half to even (default rounding mode) = torch.round
    const float hi = (float)(n_bits - sign_bits);
    if (M < 1.0f) M = 1.0f;
    if (M > hi) M = hi;
    const int E = n_bits - sign_bits - (int)M;
    if (E < 0) return FP8Q_EINVAL;
    if (E > 7) return FP8Q_EUNSUPPORTED;
    f->M = M;
    f->two_E = (float)(1 << E);
    f->l_c = (float)log2((double)(2.0f - exp2f(-M)));
    f->qthr = 0.5f - ldexpf(1.0f, (int)M - 20);
    f->sign_bits = sign_bits;
    f->pmax = 1 << E;
    return FP8Q_OK;
}

inline int hip_rc(hipError_t e) { return e == hipSuccess ? FP8Q_OK : (int)e; }
inline int launch_rc() { return hip_rc(hipGetLastError()); }


inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace
