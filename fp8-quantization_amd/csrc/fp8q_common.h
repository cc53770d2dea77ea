// fp8q_common.h -- what every translation unit of libfp8q_hip.so shares: launch-geometry helpers, the running-estimate
// fold, the block-level min/max reduction, the second stage of the two-stage min/max, format setup and error mapping.
// Everything has internal linkage (anonymous namespace): each .hip file gets its own copy.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/fp8q.h"
#include "fp8q_device.h"

using namespace fp8q;

namespace {

constexpr int kUnroll = 4;           // 16-byte loads in flight per lane
constexpr int kTargetBlocks = 2048;  // 256 CUs x 8 blocks
constexpr int kDirectMaxInner = 16384; // k_rows_direct handles rows up to here (magic division: n*inner < 2^32)
constexpr int kDirectElems = 32768;    // elements per k_rows_direct iteration (tables capped at 40 KiB)
constexpr int64_t kNtBytes = 64ll << 20;  // tensors at least this big stream with nontemporal hints

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
};

__device__ __forceinline__ void fold_store(float mn, float mx, int64_t row, float *cur_min,
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
    if (maxval_out) maxval_out[row] = fabsf(tmax(fabsf(mn), mx));  // fp8_quantizer.py:236
}

// ---------------------------------------------------------------------------------------------
// K2/K3 two-stage min/max in ONE launch.  Block (split, row) publishes {min, max} of its part of the row in the
// workspace and draws a ticket of the row; the block that draws the row's LAST ticket reduces the row's partials,
// folds them into the running estimate (+ K5) and returns the ticket counter to zero.  A separate stage-2 launch cost
// 4.6 us of the 38 us of a [64,64,112,112] activation (profiles/r01_*).  Workspace contract (include/fp8q.h): the
// first FP8Q_WS_TICKET_BYTES of ws are the counters -- zero before the first use of a buffer, zero after every call;
// the partials behind them need no initialisation.  nsplit == 1: no workspace traffic at all.
// Visibility across the 8 XCDs (one L2 each): partials are written and read with 8-byte agent-scope atomics (sc1:
// write-through / L1-bypassing), the store is waited for before the ticket atomic; no fences (an agent-scope release
// writes back the whole XCD L2 -- per block, that tripled the kernel's time).
// ---------------------------------------------------------------------------------------------
constexpr int kTicketRows = FP8Q_WS_TICKET_BYTES / 4;   // nsplit > 1 implies C <= kTargetBlocks / 2 rows

__device__ __forceinline__ unsigned long long pack_mm(float mn, float mx)
{
    return (unsigned long long)__float_as_uint(mn) | ((unsigned long long)__float_as_uint(mx) << 32);
}

__device__ __forceinline__ float2 unpack_mm(unsigned long long v)
{
    return make_float2(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
}

__device__ __forceinline__ void block_minmax_fold(MinMax m, unsigned long long *parts /* [nsplit] of this row */, int split, int nsplit,
                                                  unsigned *ticket, int64_t row, float *cur_min, float *cur_max,
                                                  float *maxval_out, const FoldArgs &fa)
{
    __shared__ float s_mn[4], s_mx[4];
    __shared__ int s_nan[4];
    __shared__ int s_last;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    mm_wave_reduce(m);
    if (lane == 0) {
        s_mn[wave] = m.mn;
        s_mx[wave] = m.mx;
        s_nan[wave] = m.nan;
    }
    __syncthreads();
    if (tid == 0) {
        float mn = fminf(fminf(s_mn[0], s_mn[1]), fminf(s_mn[2], s_mn[3]));
        float mx = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
        if (s_nan[0] | s_nan[1] | s_nan[2] | s_nan[3]) mn = mx = __builtin_nanf("");
        if (nsplit == 1) {
            fold_store(mn, mx, row, cur_min, cur_max, maxval_out, fa);
        } else {
            // publish {min, max} as ONE 8-byte agent-scope atomic store (global_store_dwordx2 sc1: write-through, the
            // line leaves this XCD's L2), wait for it, then draw the ticket.  An agent-scope release fence here would
            // write back the whole L2 from every block (measured: 38 -> 105 us on [64,64,112,112]).
            __hip_atomic_store(parts + split, pack_mm(mn, mx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = old == (unsigned)(nsplit - 1);
        }
    }
    if (nsplit == 1) return;
    __syncthreads();
    if (!s_last) return;
    // the row's partials: independent 8-byte agent-scope atomic loads (sc1: bypass this CU's L1; no block of this launch
    // has read these lines before, so this XCD's L2 holds no older copy) -- no acquire fence needed (guide: "sc1 loads
    // may replace the acquire when the producer stored sc1").  An EMPTY split (a row a few elements longer than a whole
    // number of steps) holds {+inf, -inf}, so the two halves must not be mixed.
    mm_init(m);
    float2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int s2 = tid + u * kBlock;
        v[u] = s2 < nsplit ? unpack_mm(__hip_atomic_load(parts + s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                           : make_float2(__builtin_inff(), -__builtin_inff());
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        m.nan |= (v[u].x != v[u].x) | (v[u].y != v[u].y);
        m.mn = fminf(m.mn, v[u].x);
        m.mx = fmaxf(m.mx, v[u].y);
    }
    for (int s2 = tid + 8 * kBlock; s2 < nsplit; s2 += kBlock) {
        const float2 ab = unpack_mm(__hip_atomic_load(parts + s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        m.nan |= (ab.x != ab.x) | (ab.y != ab.y);
        m.mn = fminf(m.mn, ab.x);
        m.mx = fmaxf(m.mx, ab.y);
    }
    mm_wave_reduce(m);
    if (lane == 0) {
        s_mn[wave] = m.mn;
        s_mx[wave] = m.mx;
        s_nan[wave] = m.nan;
    }
    __syncthreads();
    if (tid == 0) {
        float mn = fminf(fminf(s_mn[0], s_mn[1]), fminf(s_mn[2], s_mn[3]));
        float mx = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
        if (s_nan[0] | s_nan[1] | s_nan[2] | s_nan[3]) mn = mx = __builtin_nanf("");
        fold_store(mn, mx, row, cur_min, cur_max, maxval_out, fa);
        // back to zero for the next call on this workspace (ordered behind this kernel: same stream); atomic so that it
        // lands where the ticket atomics operate, not in this XCD's L2
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
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
