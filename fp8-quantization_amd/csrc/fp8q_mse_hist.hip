// fp8q_mse_hist.hip -- K4 for one long row (per-tensor activations): partition once, then every candidate costs ~200 lookups.
//
// The candidate loops of FP_MSE_Estimator.forward (quantization/range_estimators.py:337-347) evaluate 111 (x 6 with the
// mantissa search of the reference CLI's default) quantizers on the same tensor; LineSearchEstimator (:236-256) evaluates
// 1000.  k_mse_row (fp8q_mse.hip) does that at 4-7 VALU issue slots per candidate-element and is VALU-bound.  But a
// quantizer is a STEP FUNCTION of |x|: for one candidate (maxval, M) the keys k = |x| fall into <= (2^E + 1) 2^M + 2 cells,
// each mapped to one grid value q, and
//     sum over a cell of (k - q)^2 = S2 - 2 q S1 + n q^2     with n, S1 = sum k, S2 = sum k^2 of the cell's keys.
// All cells of all candidates are unions of the INTERVALS between consecutive cell borders (~15 K borders for 111
// candidates, ~100 K for 666), so what is needed is n / S1 / S2 per interval -- a histogram with moments, not a sort:
//   1. borders_body   one workgroup per candidate (k_stage1): the exact border of every cell -- the smallest float for which the
//                     reference's own fp32 decisions (floor(fl32(log2 k) + bias) >= p, rint(fl32(k / s_p)) >= r,
//                     k > maxval) flip, located by guess-and-walk on the exact predicates -- and the cell's grid value.
//   2. part_hist / part_scatter   ONE most-significant-digit partition of the nonzero keys by their top 11 bits
//                     (exponent + 3 fraction bits: 2048 coarse buckets): LDS histogram per workgroup -> count table ->
//                     column scan (no global atomics), then a tile-local counting sort in LDS so that every (tile, bucket)
//                     run leaves as contiguous 4-byte stores.  Both passes run at the chip's copy rate.  Zeros (half of a
//                     post-ReLU tensor) contribute nothing to any candidate and are dropped.
//   3. border_sort    the borders, bucketed the same way, are sorted per bucket in LDS: one more radix step on the next 10
//                     key bits (1024 sub-bins, ~2 borders each), ranks inside a sub-bin by counting; every border learns
//                     its global rank, every bucket gets its sub-bin table.
//   4. k_moments      per (bucket, slice of its keys): the bucket's sorted borders and sub-bin table sit in LDS, a key finds
//                     its interval with two table entries + a bisection over the handful of borders between them and adds
//                     {1, d, d^2}, d = its low 20 bits, to the interval's LDS counters with INTEGER atomics: exact,
//                     order-independent -> deterministic.
//   5. k_iv_scan_*    intervals -> S1, S2 as exact double-double numbers (k = (A_bucket + d) * ulp), exclusive prefix.
//   6. k_mse_eval     one workgroup per candidate, a lane per cell: two prefix lookups, S2 - 2 q S1 + n q^2 in
//                     double-double (the three terms cancel: on data that sit on the grid the squared error is 1e-13 of
//                     the signal energy), fixed-tree sum, mses += mean.
// Launches (round 6): k_stage1 (1 || the histogram pass of 2) -> k_tab_scan (column scans, counters cleared) -> k_sort_plan_scatter
// (3 || the unit plan || the scatter pass of 2: all three need only the column totals) -> k_moments -> k_iv_scan_super -> k_mse_eval:
// six dependent launches, ~48 us + 0.05 us per (width, candidate) pair + 4.3 ps per element.
// Every element is classified exactly as K1 / the oracle classify it; what differs from the reference is only that
// (k - q)^2 is summed in (near-)exact arithmetic instead of fp32-rounded per element: ~1e-7 relative, inside K4's stated
// contract (include/fp8q.h).  Cost for a 25.7 M-element activation: 12 B of HBM traffic per key for the partition (histogram pass 4, scatter
// 4 + 4) + 4 B for the moments, independent of the number of candidates.  No library primitive: everything here is hand-written.
#include "fp8q_common.h"
#include "fp8q_select.h"

namespace {

constexpr int kHShift = 20;                         // key bits below the coarse bucket: d = key & (2^20 - 1)
constexpr int kHBuckets = 1 << (31 - kHShift);      // 2048 = 8 exponent bits + 3 fraction bits
constexpr uint32_t kHMask = (1u << kHShift) - 1u;
constexpr int kHSubBits = 10;                       // sub-bin table of k_moments: next 10 key bits
constexpr int kHSub = 1 << kHSubBits;
constexpr int kPartTile = 8192;                     // keys per partition tile (32 per thread)
constexpr int kHistMaxM = 8;
constexpr int kStrideBound = 520;                   // cells of one candidate of a format of <= 8 bits: (2^E + 1) 2^M + 2 <= 514 (unsigned, M = 8)
constexpr int kSuper = 1024;                        // intervals per scan superblock
constexpr int kTopLds = 512;                        // superblock totals that k_mse_eval scans for itself in LDS
constexpr int kSortLds = 4096;                      // borders of one bucket sorted in LDS (more: a bitonic network on global memory)
constexpr int kMomTabWords = (kHSub + 1 + 3) & ~3;  // k_moments' sub-bin table in LDS, padded to 16 bytes
constexpr int kPartLds = 4 * (2 * kHBuckets + kPartTile + 4);   // bytes of LDS of a scatter workgroup: 49168 (3 per CU)

typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(4)));

struct HistArgs {
    QFmt fmt[kHistMaxM];
    int ncells[kHistMaxM];       // cells of width m: 2^(M+1) + 1 + (pmax - 1)(2^M + 1) + 1 (the clamp cell)
    int n_m, n_cand;
    int stride;                  // max ncells: row pitch of the per-candidate border tables
    int uns;                     // unsigned formats (sign_bits == 0): a negative element is clipped to 0 -> its error is x^2
    int overwrite;               // table entries are written, not added to (first batch of fp8q_mse_calibrate_f32)
    int64_t n;                   // elements of the row
};

// ---- double-double: value = hi + lo, |lo| <= ulp(hi) / 2.  Error-free transformations only (no fast-math in this build).
struct DD {
    double hi, lo;
};

__device__ __forceinline__ DD two_sum(double a, double b)
{
    const double s = a + b, bb = s - a;
    return DD{s, (a - (s - bb)) + (b - bb)};
}

__device__ __forceinline__ DD fast_two_sum(double a, double b)   // |a| >= |b| (or a == 0)
{
    const double s = a + b;
    return DD{s, b - (s - a)};
}

__device__ __forceinline__ DD dd_add(DD x, DD y)                  // accurate variant: safe when the high parts cancel
{
    DD s = two_sum(x.hi, y.hi);
    const DD t = two_sum(x.lo, y.lo);
    s.lo += t.hi;
    s = fast_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return fast_two_sum(s.hi, s.lo);
}

__device__ __forceinline__ DD dd_add_d(DD x, double y)
{
    DD s = two_sum(x.hi, y);
    s.lo += x.lo;
    return fast_two_sum(s.hi, s.lo);
}

__device__ __forceinline__ DD dd_neg(DD x) { return DD{-x.hi, -x.lo}; }

__device__ __forceinline__ DD two_prod(double a, double b)
{
    const double p = a * b;
    return DD{p, fma(a, b, -p)};
}

__device__ __forceinline__ DD dd_mul_d(DD x, double y)
{
    DD p = two_prod(x.hi, y);
    p.lo = fma(x.lo, y, p.lo);
    return fast_two_sum(p.hi, p.lo);
}

__device__ __forceinline__ DD dd_shfl_up(DD v, int off) { return DD{__shfl_up(v.hi, off, 64), __shfl_up(v.lo, off, 64)}; }
__device__ __forceinline__ DD dd_shfl(DD v, int lane) { return DD{__shfl(v.hi, lane, 64), __shfl(v.lo, lane, 64)}; }

// an unsigned 64-bit integer as an exact double-double (two 32-bit halves are exact doubles)
__device__ __forceinline__ DD dd_from_u64(uint64_t v)
{
    return fast_two_sum((double)(v >> 32) * 4294967296.0, (double)(uint32_t)v);
}

__device__ __forceinline__ DD dd_scale2(DD x, int e) { return DD{ldexp(x.hi, e), ldexp(x.lo, e)}; }   // exact (no underflow: see k_iv_scan_super)

// ---- block scans ------------------------------------------------------------------------------------------------------
// exclusive scan of one value per thread over a block of NT threads (NT / 64 waves); `total` = the block's sum.
// s_w: NT / 64 words of LDS; ends with a barrier, so it can be called again right away.
template <int NT>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *s_w, uint32_t &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) {
        const uint32_t t = s_w[w];
        if (w < wave) base += t;
        tot += t;
    }
    __syncthreads();
    total = tot;
    return base + incl - v;
}

// ---- 2. partition -----------------------------------------------------------------------------------------------------
// 32 elements of a tile per thread, as bit patterns (sign included: tile_key() makes the key); a tile's last, partial part
// reads element by element
__device__ __forceinline__ void load_tile_keys(const uint32_t *__restrict__ x, int64_t n, int64_t base, uint32_t (&k)[32])
{
    const int tid = threadIdx.x;
    if (base + kPartTile <= n) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const u32x4u v = *reinterpret_cast<const u32x4u *>(x + base + u * 1024 + tid * 4);
            k[4 * u] = v.x;
            k[4 * u + 1] = v.y;
            k[4 * u + 2] = v.z;
            k[4 * u + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t i = base + u * 1024 + tid * 4 + q;
                k[4 * u + q] = i < n ? x[i] : 0u;
            }
    }
}

// the key of an element: |x|; 0 (= dropped, like a zero) for a negative element of an UNSIGNED format -- it is clipped to 0,
// its squared error is x^2 whatever the candidate (summed separately: part_hist_body)
template <bool UNS>
__device__ __forceinline__ uint32_t tile_key(uint32_t raw)
{
    return (UNS && (raw >> 31)) ? 0u : (raw & 0x7fffffffu);
}

// Workgroup w of both partition passes owns the tiles [w * tpw, (w + 1) * tpw): pass A leaves its bucket counts as row w of
// `ktab`, k_tab_scan turns every column into exclusive prefixes over the workgroups, and pass B starts workgroup w's run of
// bucket b at koff[b] + ktab[w][b] -- no global atomics anywhere (a first version reserved space with one returning atomic
// per (tile, bucket) on a 2048-word cursor array: 1.9 M atomics into two memory channels, 481 us for 25.7 M keys).
template <bool UNS>
__device__ __forceinline__ void part_hist_body(int w, uint32_t *s_hist, const uint32_t *__restrict__ x, int64_t n, int64_t ntiles,
                                               int tpw, uint32_t *__restrict__ ktab, uint32_t *__restrict__ kmax, double *__restrict__ kneg)
{
    const int tid = threadIdx.x;
    uint32_t *s_mk = s_hist + kHBuckets;
    double *s_ng = reinterpret_cast<double *>(s_hist + kHBuckets + 8);
    for (int i = tid; i < kHBuckets; i += kBlock) s_hist[i] = 0u;
    __syncthreads();
    uint32_t mk = 0u;
    double neg = 0.0;       // sum of x^2 over the negative elements (unsigned formats); fp32 squares are exact in double
    const int64_t t0 = (int64_t)w * tpw, t1 = min(t0 + tpw, ntiles);
    for (int64_t t = t0; t < t1; ++t) {
        uint32_t k[32];
        load_tile_keys(x, n, t * kPartTile, k);
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            mk = max(mk, k[j] & 0x7fffffffu);
            const uint32_t key = tile_key<UNS>(k[j]);
            if (UNS && (k[j] >> 31)) {
                const double v = (double)__uint_as_float(k[j]);
                neg += v * v;
            }
            if (key) atomicAdd(&s_hist[key >> kHShift], 1u);
        }
    }
    __syncthreads();
    for (int i = tid; i < kHBuckets; i += kBlock) ktab[(int64_t)w * kHBuckets + i] = s_hist[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        mk = max(mk, (uint32_t)__shfl_xor((int)mk, off, 64));
        if (UNS) neg += __shfl_xor(neg, off, 64);               // fixed tree: deterministic
    }
    if ((tid & 63) == 0) {
        s_mk[tid >> 6] = mk;
        if (UNS) s_ng[tid >> 6] = neg;
    }
    __syncthreads();
    if (tid == 0) {
        kmax[w] = max(max(s_mk[0], s_mk[1]), max(s_mk[2], s_mk[3]));
        if (UNS) kneg[w] = (s_ng[0] + s_ng[1]) + (s_ng[2] + s_ng[3]);
    }
}

// scatter: per tile a counting sort by bucket in LDS (ranks from returning LDS atomics), then position p of the sorted
// tile goes to delta[bucket] + p: consecutive lanes write consecutive addresses within a run.  The order of the keys inside
// a (workgroup, bucket) run depends on the LDS atomics' timing; nothing downstream depends on it (integer moments).
// The first key of bucket b in the partitioned array (koff[b]) is the exclusive scan of the column totals `hist`: every
// workgroup makes it for itself (2048 entries, one block scan) -- so the scatter needs nothing from the plan and runs in the
// launch that also sorts the borders (k_sort_plan_scatter).
template <bool UNS>
__device__ __forceinline__ void part_scatter_body(int w, uint32_t *s_raw, const uint32_t *__restrict__ x, int64_t n, int64_t ntiles,
                                                  int tpw, const uint32_t *__restrict__ hist, const uint32_t *__restrict__ ktab,
                                                  uint32_t *__restrict__ out)
{
    uint32_t *s_hist = s_raw;                    // counts, then the bucket's first position in the sorted tile
    uint32_t *s_delta = s_raw + kHBuckets;       // global index of the run minus its first position
    uint32_t *s_keys = s_raw + 2 * kHBuckets;    // kPartTile
    uint32_t *s_w = s_keys + kPartTile;          // 4
    const int tid = threadIdx.x;
    constexpr int kPer = kHBuckets / kBlock;     // 8 consecutive buckets per thread in the scan
    uint32_t cur[kPer];                          // where this workgroup's next key of buckets tid * 8 .. tid * 8 + 7 goes
    {
        uint32_t h[kPer], mine[kPer], sum = 0u;
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            h[q] = hist[tid * kPer + q];
            mine[q] = ktab[(int64_t)w * kHBuckets + tid * kPer + q];
            sum += h[q];
        }
        uint32_t total;
        uint32_t run = block_excl_scan<kBlock>(sum, s_w, total);
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            cur[q] = run + mine[q];
            run += h[q];
        }
    }
    const int64_t t0 = (int64_t)w * tpw, t1 = min(t0 + tpw, ntiles);
    for (int64_t t = t0; t < t1; ++t) {
        for (int i = tid; i < kHBuckets; i += kBlock) s_hist[i] = 0u;
        __syncthreads();
        uint32_t k[32];
        uint16_t rk[32];
        load_tile_keys(x, n, t * kPartTile, k);
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            k[j] = tile_key<UNS>(k[j]);
            rk[j] = k[j] ? (uint16_t)atomicAdd(&s_hist[k[j] >> kHShift], 1u) : (uint16_t)0;
        }
        __syncthreads();
        uint32_t c[kPer], sum = 0u;
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            c[q] = s_hist[tid * kPer + q];
            sum += c[q];
        }
        uint32_t total;
        uint32_t run = block_excl_scan<kBlock>(sum, s_w, total);
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const int b = tid * kPer + q;
            s_delta[b] = cur[q] - run;
            cur[q] += c[q];
            s_hist[b] = run;
            run += c[q];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 32; ++j)
            if (k[j]) s_keys[s_hist[k[j] >> kHShift] + rk[j]] = k[j];
        __syncthreads();
        for (uint32_t p = tid; p < total; p += kBlock) {
            const uint32_t key = s_keys[p];
            out[s_delta[key >> kHShift] + p] = key;
        }
        __syncthreads();
    }
}

// Exclusive prefix down every column of a [rows, 2048] count table (in place) + the column totals.  One workgroup of 1024
// threads scans 32 columns: thread (g, col) walks 1/32 of the rows of its column, the groups are stitched through LDS
// (64 columns x 4 groups in 256 threads left most of the chip idle: 43 us for the 768 x 2048 key table).
// blockIdx.y selects the table: 0 = keys (rows = partition workgroups), 1 = borders (rows = candidates; entries
// {count | first cell << 16}: only the column totals of the counts are needed, the table stays as it is).
// The launch also clears the interval counters of k_moments (`zero`, a multiple of 16 bytes).
__global__ void __launch_bounds__(1024)
k_tab_scan(uint32_t *__restrict__ ktab, int krows, uint32_t *__restrict__ ktot, uint32_t *__restrict__ btab, int brows,
           uint32_t *__restrict__ btot, uint4 *__restrict__ zero, int64_t zero_n16)
{
    __shared__ uint32_t s_q[32][33];
    const int64_t nthreads = (int64_t)gridDim.x * gridDim.y * 1024;
    for (int64_t i = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 1024 + threadIdx.x; i < zero_n16; i += nthreads)
        zero[i] = make_uint4(0u, 0u, 0u, 0u);
    uint32_t *tab = blockIdx.y ? btab : ktab;
    const int rows = blockIdx.y ? brows : krows;
    uint32_t *tot = blockIdx.y ? btot : ktot;
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + c;
    const int per = (rows + 31) / 32, r0 = g * per, r1 = min(r0 + per, rows);
    const uint32_t mask = blockIdx.y ? 0xffffu : 0xffffffffu;
    uint32_t sum = 0u;
#pragma unroll 8
    for (int r = r0; r < r1; ++r) sum += tab[(int64_t)r * kHBuckets + col] & mask;
    s_q[g][c] = sum;
    __syncthreads();
    uint32_t run = 0u;
    for (int w = 0; w < g; ++w) run += s_q[w][c];
    if (blockIdx.y) {
        if (g == 31) tot[col] = run + sum;
        return;
    }
#pragma unroll 8
    for (int r = r0; r < r1; ++r) {
        const int64_t i = (int64_t)r * kHBuckets + col;
        const uint32_t v = tab[i];
        tab[i] = run;
        run += v;
    }
    if (g == 31) tot[col] = run;
}

// ---- 1. borders -------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float next_up(float a) { return __uint_as_float(__float_as_uint(a) + 1u); }     // a >= 0, finite
__device__ __forceinline__ float next_down(float a) { return __uint_as_float(__float_as_uint(a) - 1u); }   // a > 0

// smallest non-negative float k with pred(k), for a predicate that is monotone (false ... false true ... true) on
// [0, +inf]; `guess` should be close.  Walks at most 8 steps, then bisects the bit patterns (always terminates).
template <class Pred>
__device__ __forceinline__ float first_true(float guess, Pred pred)
{
    if (!(guess >= 0.0f)) guess = 0.0f;
    if (!(guess < __builtin_inff())) guess = 0x1.fffffep127f;
    float g = guess;
    if (pred(g)) {
        for (int i = 0; i < 8; ++i) {
            if (g == 0.0f) return 0.0f;
            const float d = next_down(g);
            if (!pred(d)) return g;
            g = d;
        }
        uint32_t lo = 0u, hi = __float_as_uint(g);            // pred(hi) true; find the first true in [lo, hi]
        if (pred(0.0f)) return 0.0f;
        while (hi - lo > 1u) {                                 // invariant: !pred(lo), pred(hi)
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (pred(__uint_as_float(mid))) hi = mid; else lo = mid;
        }
        return __uint_as_float(hi);
    }
    for (int i = 0; i < 8; ++i) {
        g = next_up(g);
        if (!(g < __builtin_inff())) return __builtin_inff();
        if (pred(g)) return g;
    }
    uint32_t lo = __float_as_uint(g), hi = 0x7f800000u;       // !pred(lo); +inf counts as true
    while (hi - lo > 1u) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (pred(__uint_as_float(mid))) hi = mid; else lo = mid;
    }
    return __uint_as_float(hi);
}

constexpr int kFlagCells = 0, kFlagBrute = 1, kFlagNaN = 2;

// The cells of candidate (m, cand) in ascending order: binade p = 1 holds r = 0 .. 2^(M+1), binades p >= 2 hold
// r = 2^M .. 2^(M+1), then the clamp cell (keys above maxval).  Cell c covers [T[c], T[c+1]) (T of the clamp cell's end is
// +inf) and maps to q[c]; T is non-decreasing, an empty cell has T[c] == T[c+1].
// T(p, r) = min(clamp_from, clamp(first k of binade p with rint(fl32(k / s_p)) >= r, L_p, L_(p+1))), L_p = smallest key of
// binade p by K1's exact decision.  A candidate whose scales are not positive normal numbers (E = 7 formats with a tiny
// maxval underflow s_1 to 0: the reference then yields NaN for the elements of that binade) has no cells: it is flagged for
// the element-by-element evaluation; maxval 0 / inf / NaN makes every element NaN.
// Row j of btab = per coarse bucket {this candidate's borders there | the cell index of the first of them << 16}.
__device__ __forceinline__ void borders_body(int j, uint32_t *s_hist, const float *__restrict__ grid, const HistArgs &a, int brute,
                                             float *__restrict__ bt, float *__restrict__ bq, int *__restrict__ cflag,
                                             uint32_t *__restrict__ btab)
{
    float *s_scale = reinterpret_cast<float *>(s_hist + kHBuckets);   // s_p, p = 1 .. pmax (exact: lut_entry)
    float *s_border = s_scale + kLutMax;                               // L_p (L_1 = 0, L_(pmax+1) = +inf)
    uint32_t *s_w = reinterpret_cast<uint32_t *>(s_border + kLutMax);
    // the log2 / exp2 tables in LDS (behind the cells' tables: s_raw of k_stage1): fetched together with the candidate in ONE
    // round trip -- from global memory the channel constants alone were three dependent ones (table entry by table entry)
    double *s_tab = reinterpret_cast<double *>(s_hist + kHBuckets + 2 * kLutMax + 16 + 2 * kStrideBound);
    const int tid = threadIdx.x;
    const int m = j / a.n_cand, cand = j - m * a.n_cand;
    const float gv = grid[cand];
    stage_fast_tab(s_tab);
    for (int i = tid; i < kHBuckets; i += kBlock) s_hist[i] = 0u;
    uint32_t *row = btab + (int64_t)j * kHBuckets;
    const QFmt f = a.fmt[m];
    const float mv = fabsf(fmaxf(fabsf(-gv), gv));              // set_quant_range(-g, g): fp8_quantizer.py:236
    __syncthreads();
    const Chan ch = make_chan_fast(mv, f, s_tab);
    const int M = (int)f.M, pmax = f.pmax;
    if (!(fabsf(ch.bias) < __builtin_inff())) {                 // maxval 0 / inf / NaN: every element quantizes to NaN
        if (tid == 0) cflag[j] = kFlagNaN;
        for (int i = tid; i < kHBuckets; i += kBlock) row[i] = 0u;
        return;
    }
    const float pmaxf = (float)pmax;
    auto p_of = [&](float k) -> float {                         // K1's exact binade decision (quant_exact)
        const float ls = floorf(log2_tab(k, s_tab) + ch.bias);
        return __builtin_amdgcn_fmed3f(ls, 1.0f, pmaxf);
    };
    for (int p = tid + 1; p <= pmax + 1; p += kBlock) {
        if (p <= pmax) s_scale[p] = lut_entry(ch, p, f.M).x;
        float L = 0.0f;
        if (p > pmax) {
            L = __builtin_inff();
        } else if (p >= 2) {
            const float pf = (float)p;
            L = first_true((float)ldexp(ch.g, p - ch.bi), [&](float k) { return p_of(k) >= pf; });   // ~2^(p - bias)
        }
        s_border[p] = L;
    }
    __syncthreads();
    bool odd = brute != 0;
    for (int p = 1; p <= pmax; ++p) odd |= !(s_scale[p] >= 0x1p-126f && s_scale[p] < __builtin_inff());
    if (odd) {
        if (tid == 0) cflag[j] = kFlagBrute;
        for (int i = tid; i < kHBuckets; i += kBlock) row[i] = 0u;
        return;
    }
    if (tid == 0) cflag[j] = kFlagCells;
    const float clamp_from = next_up(mv);                        // keys >= this are clipped to maxval (mv finite here)
    const int r_top = 2 << M, r_norm = 1 << M;
    const int n_first = r_top + 1, n_other = r_norm + 1;
    const int ncells = a.ncells[m];                              // n_first + (pmax - 1) * n_other + 1
    float *T = bt + (int64_t)j * a.stride, *Q = bq + (int64_t)j * a.stride;
    float *s_T = s_border + kLutMax + 8, *s_Q = s_T + kStrideBound;   // the cells' raw lower ends and grid values
    for (int c = tid; c < ncells; c += kBlock) {
        float lo, q;
        if (c == ncells - 1) {                                   // clipped elements: xc = maxval
            const float sc = s_scale[(int)p_of(mv)];
            q = rintf(mv / sc) * sc;
            lo = clamp_from;
        } else {
            int p, r;
            if (c < n_first) {
                p = 1;
                r = c;
            } else {
                const int cc = c - n_first;
                p = 2 + cc / n_other;
                r = r_norm + (cc - (p - 2) * n_other);
            }
            const float s = s_scale[p];
            const float rf = (float)r;
            const int r_lo = p == 1 ? 0 : r_norm;
            lo = s_border[p];
            if (r > r_lo) {
                // smallest k with rint(fl32(k / s)) >= r (IEEE division, as K1 decides), kept inside the binade
                const float ft = first_true((float)(((double)r - 0.5) * (double)s), [&](float k) { return rintf(k / s) >= rf; });
                lo = fmaxf(lo, fminf(ft, s_border[p + 1]));
            }
            lo = fminf(lo, clamp_from);
            q = rf * s;                                          // the fp32 product K1 forms
        }
        s_T[c] = lo;
        s_Q[c] = q;
    }
    __syncthreads();
    // Cells that start at or above clamp_from hold no element (the clamp decides there): the whole top binade p = 2^E -- it
    // begins at maxval * 2 / (2 - 2^-M) -- and the last cells below it.  Left in, they put ~150 equal borders per candidate
    // group into one sub-bin (the rank loop of the border sort is quadratic in that: 118 us per launch).  The FIRST of them
    // becomes the clamp cell [clamp_from, +inf) -> q(maxval); the others get T = +inf: empty, not registered as borders.
    for (int c = tid; c < ncells; c += kBlock) {
        float lo = s_T[c], q = s_Q[c];
        if (c >= 1 && lo >= clamp_from) {
            const bool first = !(s_T[c - 1] >= clamp_from);
            lo = first ? clamp_from : __builtin_inff();
            q = s_Q[ncells - 1];
        }
        T[c] = lo;
        Q[c] = q;
        if (c && lo < __builtin_inff()) atomicAdd(&s_hist[__float_as_uint(lo) >> kHShift], 1u);
    }
    __syncthreads();
    // T is sorted, so the borders of bucket b are the cells 1 + (borders in lower buckets) ...: an exclusive scan of the row
    constexpr int kPer = kHBuckets / kBlock;
    uint32_t cnt[kPer], sum = 0u;
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        cnt[q] = s_hist[tid * kPer + q];
        sum += cnt[q];
    }
    uint32_t total;
    uint32_t run = 1u + block_excl_scan<kBlock>(sum, s_w, total);
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        row[tid * kPer + q] = cnt[q] | (run << 16);      // (both below 2^9: kStrideBound)
        run += cnt[q];
    }
}

// stage 1, one launch: the candidates' borders (workgroups 0 .. n_pairs - 1) next to the key histogram (the rest)
template <bool UNS>
__global__ void __launch_bounds__(kBlock)
k_stage1(const uint32_t *__restrict__ x, int64_t n, int64_t ntiles, int tpw, uint32_t *__restrict__ ktab, uint32_t *__restrict__ kmax,
         double *__restrict__ kneg, const float *__restrict__ grid, HistArgs a, int brute, float *__restrict__ bt, float *__restrict__ bq,
         int *__restrict__ cflag, uint32_t *__restrict__ btab)
{
    static_assert(((kHBuckets + 2 * kLutMax + 16 + 2 * kStrideBound) & 1) == 0, "the tables' LDS copy is 8-byte aligned");
    __shared__ __attribute__((aligned(8))) uint32_t s_raw[kHBuckets + 2 * kLutMax + 16 + 2 * kStrideBound + 2 * kFastTabSize];
    const int n_pairs = a.n_m * a.n_cand;
    if ((int)blockIdx.x < n_pairs)
        borders_body((int)blockIdx.x, s_raw, grid, a, brute, bt, bq, cflag, btab);
    else
        part_hist_body<UNS>((int)blockIdx.x - n_pairs, s_raw, x, n, ntiles, tpw, ktab, kmax, kneg);
}

// ---- plan: offsets of keys and borders per bucket, the work units of k_moments ------------------------------------------
// A unit = (bucket, chunk of <= bcap sorted borders of the bucket, slice of the bucket's keys).  The slice length grows
// with the number of borders so that flushing the LDS counters stays a small part of a unit's work.
struct __attribute__((aligned(32))) Unit {
    uint32_t b_ch;      // bucket | chunk << 16
    uint32_t key0, kn;  // first key (index into the partitioned array) and count
    uint32_t bo, nb;    // the bucket's first sorted border and its border count
    uint32_t li;        // the bucket's position in the list of buckets that have borders (its sub-bin table)
    uint32_t pad[2];
};

__device__ __forceinline__ uint32_t slice_len(uint32_t nb, int slice_min)
{
    uint32_t s = ((16u * (nb + 1u)) + 4095u) & ~4095u;
    // ... but not beyond four times the smallest slice: with the mantissa search most buckets hold > 1024 borders, every
    // slice became 20 K keys whatever the tensor's size, and a 1 M-element activation was ~60 x (2 .. 5 chunks) units on 2048
    // workgroups (k_moments 50 us on every MobileNetV2 activation with 666 pairs)
    if (s > 4u * (uint32_t)slice_min) s = 4u * (uint32_t)slice_min;
    if (s < (uint32_t)slice_min) s = (uint32_t)slice_min;
    if (s > 65536u) s = 65536u;       // counters pack {n, sum d} into 64 bits: n < 2^24, sum d < 2^40
    return s;
}

// (a device function: it runs as one more workgroup of the border-sort launch -- both need only the column totals.  Round 6:
// writing the unit list from eight workgroups instead of one changed nothing -- k_border_sort_plan 24.4 us with 666 pairs either
// way: the sort workgroups of the fullest buckets are the launch's long pole, not the plan)
__device__ __forceinline__ void plan_body(uint32_t *s_raw, const uint32_t *__restrict__ hist, const uint32_t *__restrict__ bhist,
                                          const uint32_t *__restrict__ kmax, const double *__restrict__ kneg, int nkmax, uint32_t *__restrict__ koff,
                                          uint32_t *__restrict__ boff, Unit *__restrict__ units, uint32_t *__restrict__ nunits,
                                          uint32_t *__restrict__ maxkey, uint32_t units_max, int bcap, int slice_min)
{
    uint32_t *s_uoff = s_raw;                      // kHBuckets + 1 (+ 3 pad)
    uint32_t *s_ck = s_uoff + kHBuckets + 4, *s_cb = s_ck + kHBuckets, *s_ko = s_cb + kHBuckets, *s_bo = s_ko + kHBuckets;
    uint32_t *s_li = s_bo + kHBuckets, *s_w = s_li + kHBuckets;       // s_w: 8
    const int tid = threadIdx.x;
    constexpr int kPer = kHBuckets / kBlock;       // 8 consecutive buckets per thread
    uint32_t ck[kPer], cb[kPer], cu[kPer], sk = 0, sb = 0, su = 0, sl_ = 0;
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        const int b = tid * kPer + q;
        ck[q] = hist[b];
        cb[q] = bhist[b];
        const uint32_t nch = cb[q] ? (cb[q] + bcap - 1) / bcap : 1u;
        const uint32_t sl = slice_len(cb[q] < (uint32_t)bcap ? cb[q] : (uint32_t)bcap, slice_min);
        cu[q] = ck[q] ? nch * ((ck[q] + sl - 1) / sl) : 0u;
        sk += ck[q];
        sb += cb[q];
        su += cu[q];
        sl_ += cb[q] ? 1u : 0u;
    }
    uint32_t tk, tb, tu, tl;
    uint32_t ek = block_excl_scan<kBlock>(sk, s_w, tk);
    uint32_t eb = block_excl_scan<kBlock>(sb, s_w, tb);
    uint32_t eu = block_excl_scan<kBlock>(su, s_w, tu);
    uint32_t el = block_excl_scan<kBlock>(sl_, s_w, tl);
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        const int b = tid * kPer + q;
        koff[b] = ek;
        boff[b] = eb;
        s_uoff[b] = eu;
        s_ck[b] = ck[q];
        s_cb[b] = cb[q];
        s_ko[b] = ek;
        s_bo[b] = eb;
        s_li[b] = el;
        el += cb[q] ? 1u : 0u;
        ek += ck[q];
        eb += cb[q];
        eu += cu[q];
    }
    if (tid == kBlock - 1) {
        koff[kHBuckets] = tk;
        boff[kHBuckets] = tb;
        s_uoff[kHBuckets] = tu;
        nunits[0] = tu < units_max ? tu : units_max;   // (units_max is a proven bound: see hist_layout)
        nunits[1] = tl;                                // buckets that have borders
        nunits[2] = tu > units_max ? 1u : 0u;          // never seen; if it happens the call reports NaN instead of dropping work
    }
    __syncthreads();
    // the units, all threads: unit u belongs to the last bucket whose first unit is <= u; inside a bucket the chunks of one
    // key slice are neighbours (they read the same keys)
    const uint32_t nu = tu < units_max ? tu : units_max;
    for (uint32_t u = tid; u < nu; u += kBlock) {
        int lo = 0, hi = kHBuckets;                    // s_uoff[lo] <= u < s_uoff[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_uoff[mid] <= u) lo = mid; else hi = mid;
        }
        const uint32_t b = (uint32_t)lo, nbk = s_cb[b], nk = s_ck[b];
        const uint32_t nch = nbk ? (nbk + bcap - 1) / bcap : 1u;
        const uint32_t sl = slice_len(nbk < (uint32_t)bcap ? nbk : (uint32_t)bcap, slice_min);
        const uint32_t local = u - s_uoff[b], si = local / nch, chn = local - si * nch;
        const uint32_t k0 = si * sl;
        units[u] = Unit{b | (chn << 16), s_ko[b] + k0, min(sl, nk - k0), s_bo[b], nbk, s_li[b], {0u, 0u}};
    }
    // the largest key of the row (non-finite data): max over the partition workgroups
    uint32_t mk = 0u;
    for (int i = tid; i < nkmax; i += kBlock) mk = max(mk, kmax[i]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mk = max(mk, (uint32_t)__shfl_xor((int)mk, off, 64));
    __syncthreads();
    if ((tid & 63) == 0) s_w[tid >> 6] = mk;
    __syncthreads();
    if (tid == 0) maxkey[0] = max(max(s_w[0], s_w[1]), max(s_w[2], s_w[3]));
    if (kneg) {
        // unsigned formats: the negative elements' sum of squares, over the partition workgroups in a fixed order
        double *s_ng = reinterpret_cast<double *>(s_uoff);      // (the unit tables are no longer read: barrier above)
        double v = 0.0;
        for (int i = tid; i < nkmax; i += kBlock) v += kneg[i];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        __syncthreads();
        if ((tid & 63) == 0) s_ng[tid >> 6] = v;
        __syncthreads();
        if (tid == 0) reinterpret_cast<double *>(maxkey)[1] = (s_ng[0] + s_ng[1]) + (s_ng[2] + s_ng[3]);
    }
}

// ---- 3. sort the borders of each bucket ---------------------------------------------------------------------------------
// Bitonic network in its all-ascending form (the first step of every merge compares i with its mirror image in the block,
// the others i with i + j): every comparator moves the smaller element to the lower index, so a tail of "+inf" padding
// never moves and comparators that touch it are simply skipped -- any length works.
template <class Ptr>
__device__ __forceinline__ void bitonic_sort(Ptr d, int n)
{
    int l2 = 0;
    while ((1 << l2) < n) ++l2;
    const int half = (1 << l2) >> 1;
    for (int lk = 1; lk <= l2; ++lk) {
        for (int lj = lk - 1; lj >= 0; --lj) {
            const int j = 1 << lj;
            for (int i = threadIdx.x; i < half; i += kBlock) {
                const int t = i & (j - 1), blk = i >> lj;
                int lo, hi;
                if (lj == lk - 1) {
                    lo = (blk << lk) + t;
                    hi = (blk << lk) + (1 << lk) - 1 - t;
                } else {
                    lo = (blk << (lj + 1)) + t;
                    hi = lo + j;
                }
                if (hi < n) {
                    const uint64_t x = d[lo], y = d[hi];
                    if (x > y) {
                        d[lo] = y;
                        d[hi] = x;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// inclusive scan of tab[0 .. kHSub] in place (tab[s + 1] held the entries of sub-bin s: afterwards tab[s] = entries below
// sub-bin s, tab[kHSub] = all)
__device__ __forceinline__ void subbin_scan(uint32_t *tab, uint32_t *s_w)
{
    constexpr int kTabPer = (kHSub + kBlock) / kBlock;   // 5 entries per thread (1025 entries)
    const int tid = threadIdx.x;
    uint32_t c[kTabPer], sum = 0u;
#pragma unroll
    for (int q = 0; q < kTabPer; ++q) {
        const int i = tid * kTabPer + q;
        c[q] = i <= kHSub ? tab[i] : 0u;
        sum += c[q];
    }
    uint32_t total;
    uint32_t run = block_excl_scan<kBlock>(sum, s_w, total);
#pragma unroll
    for (int q = 0; q < kTabPer; ++q) {
        const int i = tid * kTabPer + q;
        run += c[q];
        if (i <= kHSub) tab[i] = run;
    }
    __syncthreads();
}

__device__ __forceinline__ uint32_t subbin_of(uint32_t v) { return (v >> (kHShift - kHSubBits)) & (kHSub - 1); }

// One bucket: gather its borders from the candidates' sorted tables (candidate j has `count` of them there, starting at cell
// `first`: btab), sort them by (value, owner) and tell every border its prefix index.
// Interval ids: bucket b owns ids boff[b] + b + i, i = 0 .. nb (interval i = keys of the bucket in [border i-1, border i));
// "everything below border i of bucket b" = the intervals with ids < boff[b] + b + i + 1.
// The sort is one more most-significant-digit step: the bucket's borders share their top 11 key bits, the next 10 bits
// (sub-bin) place a border to within ~2 positions -- count, scan, place -- and the rank inside a sub-bin is the number of
// smaller {value, owner} pairs there.  (A 66-stage bitonic network on 2048 pairs took 39 us per launch.)  The scan is also
// the bucket's sub-bin table for k_moments.  More than kSortLds borders (never seen): the network on global memory.
__device__ __forceinline__ void border_sort_body(int b, uint32_t *s_raw, const float *__restrict__ bt, const uint32_t *__restrict__ btab,
                                                 const uint32_t *__restrict__ bhist, int n_pairs, int stride,
                                                 uint32_t *__restrict__ sb, uint32_t *__restrict__ rank,
                                                 uint32_t *__restrict__ gtab, uint64_t *__restrict__ pairs)
{
    uint64_t *e = reinterpret_cast<uint64_t *>(s_raw);                  // kSortLds pairs {value bits << 32 | owner}, as gathered
    uint16_t *perm = reinterpret_cast<uint16_t *>(s_raw + 2 * kSortLds); // kSortLds: position in sub-bin order -> index into e
    uint32_t *tab = s_raw + 2 * kSortLds + kSortLds / 2;               // kHSub + 2
    uint32_t *cur = tab + kHSub + 2;                                   // kHSub + 2
    uint32_t *s_w = cur + kHSub + 2;                                   // 4 + the gather cursor + 2 x 4 for the offsets
    const int tid = threadIdx.x;
    const int nb = (int)bhist[b];
    if (nb == 0) return;
    // this bucket's first border (boff[b]) and its position among the buckets that have borders, from the column totals
    uint32_t o, li;
    {
        uint32_t so = 0u, sl = 0u;
        for (int i = tid; i < b; i += kBlock) {
            const uint32_t c = bhist[i];
            so += c;
            sl += c ? 1u : 0u;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            so += __shfl_xor(so, off, 64);
            sl += __shfl_xor(sl, off, 64);
        }
        if ((tid & 63) == 0) {
            s_w[8 + (tid >> 6)] = so;
            s_w[12 + (tid >> 6)] = sl;
        }
        __syncthreads();
        o = s_w[8] + s_w[9] + s_w[10] + s_w[11];
        li = s_w[12] + s_w[13] + s_w[14] + s_w[15];
    }
    {
        const bool in_lds = nb <= kSortLds;
        uint64_t *dst = in_lds ? e : pairs + o;
        for (int i = tid; i <= kHSub + 1; i += kBlock) tab[i] = 0u;
        if (tid == 0) s_w[4] = 0u;
        __syncthreads();
        // gather (any order) + count per sub-bin: four candidates per thread and trip, their table entries loaded together
        uint32_t gathered = 0u;
        for (int j0 = 0; j0 < n_pairs; j0 += 4 * kBlock) {
            uint32_t ent[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = j0 + q * kBlock + tid;
                ent[q] = j < n_pairs ? btab[(int64_t)j * kHBuckets + b] : 0u;
            }
            // where a candidate's borders go: an exclusive scan of the counts over the workgroup (a returning LDS atomic per
            // candidate on ONE address serialised 666 lanes)
            uint32_t pos[4];
            {
                uint32_t mine = 0u, total;
#pragma unroll
                for (int q = 0; q < 4; ++q) mine += ent[q] & 0xffffu;
                uint32_t run = gathered + block_excl_scan<kBlock>(mine, s_w, total);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    pos[q] = run;
                    run += ent[q] & 0xffffu;
                }
                gathered += total;
            }
            // a candidate's borders in this bucket (<= ~9 with 64 cells per binade): the first ten of all four candidates are
            // requested before any is used -- ONE memory round trip per trip of the loop (one per candidate and batch of
            // eight made the gather 7-8 us of a full bucket's ~18 with 666 pairs: instrumented, round 6)
            constexpr uint32_t kFirst = 10u;
            uint32_t vv[4][kFirst];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t c = ent[q] & 0xffffu, cs = ent[q] >> 16;
                const uint32_t j = (uint32_t)(j0 + q * kBlock + tid);
#pragma unroll
                for (uint32_t u = 0; u < kFirst; ++u) vv[q][u] = u < c ? __float_as_uint(bt[j * (uint32_t)stride + cs + u]) : 0u;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t c = ent[q] & 0xffffu, cs = ent[q] >> 16;
                const uint32_t j = (uint32_t)(j0 + q * kBlock + tid);
#pragma unroll
                for (uint32_t u = 0; u < kFirst; ++u) {
                    if (u < c) {
                        const uint32_t idx = j * (uint32_t)stride + cs + u;
                        dst[pos[q] + u] = ((uint64_t)vv[q][u] << 32) | (uint64_t)idx;
                        atomicAdd(&tab[subbin_of(vv[q][u]) + 1], 1u);
                    }
                }
                for (uint32_t i0 = kFirst; i0 < c; i0 += 8u) {          // (formats with more than 64 cells per binade)
                    uint32_t v8[8];
#pragma unroll
                    for (uint32_t u = 0; u < 8u; ++u) v8[u] = i0 + u < c ? __float_as_uint(bt[j * (uint32_t)stride + cs + i0 + u]) : 0u;
#pragma unroll
                    for (uint32_t u = 0; u < 8u; ++u) {
                        if (i0 + u < c) {
                            const uint32_t idx = j * (uint32_t)stride + cs + i0 + u;
                            dst[pos[q] + i0 + u] = ((uint64_t)v8[u] << 32) | (uint64_t)idx;
                            atomicAdd(&tab[subbin_of(v8[u]) + 1], 1u);
                        }
                    }
                }
            }
        }
        __syncthreads();
        subbin_scan(tab, s_w);
        uint32_t *gt = gtab + (int64_t)li * (kHSub + 1);
        for (int i = tid; i <= kHSub; i += kBlock) {
            const uint32_t v = tab[i];
            gt[i] = v;
            cur[i] = v;
        }
        __syncthreads();
        if (in_lds) {
            for (int i = tid; i < nb; i += kBlock)            // place into the sub-bin's segment (any order inside it)
                perm[atomicAdd(&cur[subbin_of((uint32_t)(e[i] >> 32))], 1u)] = (uint16_t)i;
            __syncthreads();
            if (tid == 0) cur[0] = 0u;        // (the cursors are done with: cur becomes the list of the fuller sub-bins)
            __syncthreads();
            // rank inside the sub-bin = smaller {value, owner} pairs there.  A thread per SUB-BIN (round 6): its <= 8 members
            // (~1.6 on average in a full bucket) are fetched with independent LDS reads -- two dependent levels in all -- and
            // ranked in registers; a thread per border re-read its sub-bin through perm -> e for every member and waited for the
            // fullest sub-bin of its wave in every round: 6-9 us of a full bucket's ~18 with 666 pairs (instrumented).
            for (int sbin = tid; sbin < kHSub; sbin += kBlock) {
                const uint32_t s0 = tab[sbin], s1 = tab[sbin + 1];
                const uint32_t np = s1 - s0;
                if (np == 0u) continue;
                if (np <= 8u) {
                    uint64_t v[8];
#pragma unroll
                    for (uint32_t u = 0; u < 8u; ++u) v[u] = u < np ? e[perm[s0 + u]] : ~0ull;      // (~0: never smaller)
#pragma unroll
                    for (uint32_t u = 0; u < 8u; ++u) {
                        if (u < np) {
                            uint32_t r = s0;
#pragma unroll
                            for (uint32_t w = 0; w < 8u; ++w) r += v[w] < v[u] ? 1u : 0u;
                            sb[o + r] = (uint32_t)(v[u] >> 32);
                            rank[(uint32_t)v[u]] = o + (uint32_t)b + r + 1u;
                        }
                    }
                } else {
                    cur[1u + atomicAdd(&cur[0], 1u)] = (uint32_t)sbin;      // (<= kHSub entries)
                }
            }
            __syncthreads();
            // fuller sub-bins (the six widths of one candidate share its clamp border: up to 27 borders sit around the smallest
            // candidate's; two or three such sub-bins per full bucket): a WAVE per sub-bin, a lane per member -- every member is
            // broadcast in turn (shuffles), a lane counts the smaller ones: no LDS traffic beyond the members themselves.  (A
            // thread per member walking the sub-bin through perm -> e: 4.5-6 us of a full bucket's 16; a thread alone: 108 us.)
            {
                const int nheavy = (int)cur[0];
                const int lane = tid & 63;
                for (int h = tid >> 6; h < nheavy; h += kBlock / 64) {
                    const uint32_t sbin = cur[1 + h];
                    const uint32_t s0 = tab[sbin], s1 = tab[sbin + 1];
                    const uint32_t np = s1 - s0;
                    if (np <= 64u) {
                        const uint64_t me = (uint32_t)lane < np ? e[perm[s0 + (uint32_t)lane]] : ~0ull;
                        const uint32_t mhi = (uint32_t)(me >> 32), mlo = (uint32_t)me;
                        uint32_t r = s0;
                        for (uint32_t k = 0; k < np; ++k) {
                            const uint32_t ohi = (uint32_t)__shfl((int)mhi, (int)k, 64), olo = (uint32_t)__shfl((int)mlo, (int)k, 64);
                            r += (ohi < mhi || (ohi == mhi && olo < mlo)) ? 1u : 0u;
                        }
                        if ((uint32_t)lane < np) {
                            sb[o + r] = mhi;
                            rank[mlo] = o + (uint32_t)b + r + 1u;
                        }
                    } else {
                        for (uint32_t i = s0 + (uint32_t)lane; i < s1; i += 64u) {      // (never seen: > 64 borders in one sub-bin)
                            const uint64_t me = e[perm[i]];
                            uint32_t r = s0;
                            for (uint32_t k = s0; k < s1; ++k) r += e[perm[k]] < me ? 1u : 0u;
                            sb[o + r] = (uint32_t)(me >> 32);
                            rank[(uint32_t)me] = o + (uint32_t)b + r + 1u;
                        }
                    }
                }
            }
        } else {
            bitonic_sort(dst, nb);     // (global memory: a workgroup's own stores are visible to it after the barrier)
            for (int i = tid; i < nb; i += kBlock) {
                const uint64_t v = dst[i];
                sb[o + i] = (uint32_t)(v >> 32);
                rank[(uint32_t)v] = o + (uint32_t)b + (uint32_t)i + 1u;
            }
        }
    }
}

// the border sort: one workgroup per coarse bucket (most have no borders and leave at once); needs only the column totals
// `bhist`, so it runs BEFORE the key scatter and never next to it (there, with the memory system saturated, its chains of
// dependent loads took 35 us per bucket).  The plan -- one more workgroup -- needs the same totals: same launch.
constexpr int kSortRaw = 2 * kSortLds + kSortLds / 2 + 2 * (kHSub + 2) + 16;      // words of LDS
static_assert(kSortRaw >= 6 * kHBuckets + 16, "the plan's tables fit in the sort's LDS");

__global__ void __launch_bounds__(kBlock)
k_border_sort_plan(const float *__restrict__ bt, const uint32_t *__restrict__ btab, const uint32_t *__restrict__ bhist, int n_pairs,
                   int stride, uint32_t *__restrict__ sb, uint32_t *__restrict__ rank, uint32_t *__restrict__ gtab,
                   uint64_t *__restrict__ pairs, const uint32_t *__restrict__ hist, const uint32_t *__restrict__ kmax,
                   const double *__restrict__ kneg, int nkmax,
                   uint32_t *__restrict__ koff, uint32_t *__restrict__ boff, Unit *__restrict__ units, uint32_t *__restrict__ nunits,
                   uint32_t *__restrict__ maxkey, uint32_t units_max, int bcap, int slice_min)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_raw[kSortRaw];
    if (blockIdx.x == 0)
        plan_body(s_raw, hist, bhist, kmax, kneg, nkmax, koff, boff, units, nunits, maxkey, units_max, bcap, slice_min);
    else
        border_sort_body((int)blockIdx.x - 1, s_raw, bt, btab, bhist, n_pairs, stride, sb, rank, gtab, pairs);
}

template <bool UNS>
__global__ void __launch_bounds__(kBlock)
k_part_scatter(const uint32_t *__restrict__ x, int64_t n, int64_t ntiles, int tpw, const uint32_t *__restrict__ hist,
               const uint32_t *__restrict__ ktab, uint32_t *__restrict__ keys)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_raw[kPartLds / 4];
    part_scatter_body<UNS>((int)blockIdx.x, s_raw, x, n, ntiles, tpw, hist, ktab, keys);
}

// Round 6: border sort, plan and key scatter in ONE launch -- all three need only the column totals of k_tab_scan, none needs
// another's result (the scatter makes its bucket offsets itself: part_scatter_body).  Workgroup 0 plans, 1 .. 2048 sort the
// borders of a bucket each (most have none and leave at once; they come first in the grid so that the fullest buckets -- the
// long pole of the sort, a chain of dependent loads -- start before the scatter saturates the memory system), the rest
// scatter.  One dependent launch fewer per call: the sort (9-29 us on MobileNetV2's activations) hides behind the scatter.
template <bool UNS>
__global__ void __launch_bounds__(kBlock)
k_sort_plan_scatter(const float *__restrict__ bt, const uint32_t *__restrict__ btab, const uint32_t *__restrict__ bhist, int n_pairs,
                    int stride, uint32_t *__restrict__ sb, uint32_t *__restrict__ rank, uint32_t *__restrict__ gtab,
                    uint64_t *__restrict__ pairs, const uint32_t *__restrict__ hist, const uint32_t *__restrict__ kmax,
                    const double *__restrict__ kneg, int nkmax,
                    uint32_t *__restrict__ koff, uint32_t *__restrict__ boff, Unit *__restrict__ units, uint32_t *__restrict__ nunits,
                    uint32_t *__restrict__ maxkey, uint32_t units_max, int bcap, int slice_min,
                    const uint32_t *__restrict__ x, int64_t n, int64_t ntiles, int tpw, const uint32_t *__restrict__ ktab,
                    uint32_t *__restrict__ keys)
{
    constexpr int kRaw = kSortRaw > kPartLds / 4 ? kSortRaw : kPartLds / 4;
    __shared__ __attribute__((aligned(16))) uint32_t s_raw[kRaw];
    if (blockIdx.x == 0)
        plan_body(s_raw, hist, bhist, kmax, kneg, nkmax, koff, boff, units, nunits, maxkey, units_max, bcap, slice_min);
    else if (blockIdx.x <= (unsigned)kHBuckets)
        border_sort_body((int)blockIdx.x - 1, s_raw, bt, btab, bhist, n_pairs, stride, sb, rank, gtab, pairs);
    else
        part_scatter_body<UNS>((int)blockIdx.x - 1 - kHBuckets, s_raw, x, n, ntiles, tpw, hist, ktab, keys);
}

// ---- 4. moments of the intervals ----------------------------------------------------------------------------------------
// dynamic LDS: uint32 tab[1025 (+3)] | uint32 bord[bcap (+pad)] | uint64 cnt[bcap + 1][2] = {n << 40 | sum d, sum d^2} -- the table at
// offset 0 (its address is two operations on the key), the two counters of an interval side by side (one address per key)
// tab[s] = borders of the chunk below sub-bin s (the key's next 10 bits), from the bucket's table (border_sort_body): a key's
// interval is found with the two table entries of its sub-bin and a bisection over the handful of borders between them
// (with 64 sub-bins the bisection -- 5-7 dependent LDS reads per key -- was 2/3 of the kernel).  16 keys per lane are in
// flight (four 16-byte loads) and searched side by side, four at a time.
__device__ __forceinline__ void moments_keys4(const uint32_t (&kv)[4], const uint32_t *tab, const uint32_t *bord, unsigned long long *cnt2,
                                              uint32_t prev, int cnt, bool last, int csh1, int cpy2)
{
    int lo[4], hi[4];
    bool any = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t e = tab[subbin_of(kv[q])];           // {borders below the sub-bin | borders below the next one << 16}
        lo[q] = (int)(e & 0xffffu);
        hi[q] = (int)(e >> 16);
        any |= lo[q] < hi[q];
    }
    // borders <= key.  Branch-free per key (third session of round 6): a trip is taken whenever ONE of the wave's 256 keys lies in
    // a sub-bin with a border inside -- practically always -- and as four predicated blocks (an `if` per key) a trip was ~100
    // instructions with one LDS round trip per key; selects instead: the four reads of a trip are in flight together.  A key that
    // is done reads bord[lo] (lo <= cnt <= bcap: inside the allocation, the counters follow) and keeps its bounds.
    while (any) {
        int mid[4];
        uint32_t bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            mid[q] = (lo[q] + hi[q]) >> 1;
            bv[q] = bord[mid[q]];
        }
        any = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool act = lo[q] < hi[q], le = bv[q] <= kv[q];
            lo[q] = (act && le) ? mid[q] + 1 : lo[q];
            hi[q] = (act && !le) ? mid[q] : hi[q];
            any |= lo[q] < hi[q];
        }
    }
    const uint32_t pmin = prev > 1u ? prev : 1u;          // (a real key is never 0: zeros were dropped; 0 = padding of a ragged batch)
    const int cx = last ? -1 : cnt;                       // (lo is never negative)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t key = kv[q];
        // keys below `prev` or behind the chunk's last border: another chunk's
        if ((int)(key < pmin) | (int)(lo[q] == cx)) continue;        // (one predicate, one branch)
        const unsigned long long d = key & kHMask;
        unsigned long long *c = cnt2 + ((lo[q] << csh1) + cpy2);       // csh1 = csh + 1, cpy2 = 2 * copy
        atomicAdd(c, (1ull << 40) + d);
        atomicAdd(c + 1, d * d);
    }
}

// NT threads per workgroup: the kernel is latency-bound (a key is a chain global load -> table entry -> borders -> two LDS
// atomics; 24.6 KB of LDS per workgroup allow six of them per CU), so 512 threads put 32 waves on a CU instead of 24 and the
// next batch of keys is loaded before the current one is searched (round 6).
template <int NT>
__device__ __forceinline__ void moments_load16(const uint32_t *__restrict__ kp, int kn, int r0, uint32_t (&kv)[4][4])
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int i0 = r0 + v * 4 * NT + tid * 4;
        if (i0 + 3 < kn) {
            const u32x4u w = *reinterpret_cast<const u32x4u *>(kp + i0);
            kv[v][0] = w.x, kv[v][1] = w.y, kv[v][2] = w.z, kv[v][3] = w.w;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) kv[v][q] = i0 + q < kn ? kp[i0 + q] : 0u;   // 0: skipped
        }
    }
}

template <int NT>
__global__ void __launch_bounds__(NT)
k_moments(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ sb, const uint32_t *__restrict__ gtab,
          const Unit *__restrict__ units, const uint32_t *__restrict__ nunits, uint32_t *__restrict__ g_n,
          unsigned long long *__restrict__ g_d, unsigned long long *__restrict__ g_d2lo, unsigned long long *__restrict__ g_d2hi,
          int bcap)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *tab = reinterpret_cast<uint32_t *>(smem);            // kHSub + 1 entries (kMomTabWords with the padding)
    uint32_t *bord = tab + kMomTabWords;
    unsigned long long *cnt2 = reinterpret_cast<unsigned long long *>(bord + ((bcap + 3) & ~3));
    const int tid = threadIdx.x;
    const uint32_t nu = nunits[0];
    for (uint32_t u = blockIdx.x; u < nu; u += gridDim.x) {
        const Unit un = units[u];
        const uint32_t b = un.b_ch & 0xffffu, chn = un.b_ch >> 16;
        const uint32_t bo = un.bo;
        const int nb = (int)un.nb;
        const int a0 = (int)chn * bcap;
        const int cnt = min(nb - a0, bcap);                       // borders of this chunk (0 when the bucket has none)
        const bool last = a0 + cnt == nb;                         // the chunk that owns the interval behind the last border
        const uint32_t *kp = keys + un.key0;
        const int kn = (int)un.kn;
        uint32_t kv[4][4];
        moments_load16<NT>(kp, kn, 0, kv);                        // (in flight while the tables are staged)
        const uint32_t prev = a0 > 0 ? sb[bo + a0 - 1] : 0u;      // keys below it belong to an earlier chunk
        const uint32_t *gt = gtab + (int64_t)un.li * (kHSub + 1);
        for (int i = tid; i < kHSub; i += NT) {                   // the bucket's table, relative to the chunk; both ends of a
            const int v = nb ? (int)gt[i] - a0 : 0;               // sub-bin in one word: one LDS read per key (round 6; bcap < 2^16)
            const int w = nb ? (int)gt[i + 1] - a0 : 0;
            tab[i] = (uint32_t)min(max(v, 0), cnt) | ((uint32_t)min(max(w, 0), cnt) << 16);
        }
        for (int i = tid; i < cnt; i += NT) bord[i] = sb[bo + a0 + i];
        // 2^csh copies of every interval's counters, interleaved (copy c of interval i at (i << csh) + c: neighbouring
        // banks), a lane uses copy lane % 2^csh: with ~100 intervals per bucket (111 candidates) two thirds of the kernel's
        // LDS cycles were conflicts of lanes adding to the same counter
        // (round 6: up to 64 copies -- a copy per LANE when the chunk has <= 15 intervals -- changed nothing: 62.1 us and
        // 8.59 M conflict cycles of 15.4 M with 8, 32 or 64 copies on [64,32,112,112] x 111, profiles/r06_moments_copies_ab.txt;
        // neither did one table read per key instead of two: 47.2 us against 48.4)
        int csh = 0;
        while (csh < 3 && ((cnt + 1) << (csh + 1)) <= bcap + 1) ++csh;
        const int ncnt = (cnt + 1) << csh, cpy = tid & ((1 << csh) - 1);
        for (int i = tid; i < 2 * ncnt; i += NT) cnt2[i] = 0ull;
        __syncthreads();
        for (int r0 = 0; r0 < kn; r0 += 16 * NT) {
            uint32_t nx[4][4];
            const bool more = r0 + 16 * NT < kn;
            if (more) moments_load16<NT>(kp, kn, r0 + 16 * NT, nx);
#pragma unroll
            for (int v = 0; v < 4; ++v) moments_keys4(kv[v], tab, bord, cnt2, prev, cnt, last, csh + 1, 2 * cpy);
            if (more) {
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                    for (int q = 0; q < 4; ++q) kv[v][q] = nx[v][q];
            }
        }
        __syncthreads();
        const uint32_t gid0 = bo + b + (uint32_t)a0;
        for (int i = tid; i <= cnt; i += NT) {
            unsigned long long A = 0ull, B = 0ull;
            for (int c = 0; c < (1 << csh); ++c) {
                A += cnt2[2 * ((i << csh) + c)];
                B += cnt2[2 * ((i << csh) + c) + 1];
            }
            if (A == 0ull || (i == cnt && !last)) continue;
            atomicAdd(&g_n[gid0 + i], (uint32_t)(A >> 40));
            atomicAdd(&g_d[gid0 + i], A & ((1ull << 40) - 1ull));
            atomicAdd(&g_d2lo[gid0 + i], B & 0xffffffffull);
            if (B >> 32) atomicAdd(&g_d2hi[gid0 + i], B >> 32);
        }
        __syncthreads();
    }
}

// ---- 5. intervals -> S1, S2 (double-double), exclusive prefix -------------------------------------------------------------
// A key of bucket b = (e << 3) | t is (A + d) * 2^ex with A = [e != 0] 2^23 + t 2^20, ex = max(e, 1) - 150, so over an
// interval   S1 = 2^ex (n A + D),   S2 = 2^(2 ex) (n A^2 + 2 A D + D2)   with D = sum d, D2 = sum d^2: integers below 2^80,
// exact in double-double; the scaling is by powers of two >= 2^-298 on numbers >= 1 (no underflow in either part).
// One workgroup per superblock of 1024 intervals: exclusive prefix WITHIN the superblock + the superblock's totals;
// k_iv_scan_top turns the totals into exclusive prefixes; prefix_at() adds the two levels (fixed association: deterministic).
__global__ void __launch_bounds__(kSuper)
k_iv_scan_super(const uint32_t *__restrict__ boff, const uint32_t *__restrict__ g_n, const unsigned long long *__restrict__ g_d,
                const unsigned long long *__restrict__ g_d2lo, const unsigned long long *__restrict__ g_d2hi, int64_t ni,
                DD *__restrict__ p1, DD *__restrict__ p2, uint32_t *__restrict__ pn, DD *__restrict__ t1, DD *__restrict__ t2,
                uint32_t *__restrict__ tn)
{
    const int sblk = (int)blockIdx.x;
    __shared__ DD s1[2 * (kSuper / 64)], s2[2 * (kSuper / 64)];     // wave totals, then their inclusive scan
    __shared__ uint32_t sn[2 * (kSuper / 64)];
    __shared__ uint32_t s_first[kHBuckets];       // first interval id of every bucket
    const int tid = threadIdx.x;
    const int64_t i = (int64_t)sblk * kSuper + tid;
    for (int b = tid; b < kHBuckets; b += kSuper) s_first[b] = boff[b] + (uint32_t)b;
    const DD zero{0.0, 0.0};
    DD v1 = zero, v2 = zero;
    uint32_t vn = 0u;
    if (i < ni) vn = g_n[i];
    __syncthreads();
    if (vn) {
        // the interval's bucket: the last b whose first interval is <= i
        int lo = 0, hi = kHBuckets - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if ((int64_t)s_first[mid] <= i) lo = mid; else hi = mid - 1;
        }
        const int e = lo >> 3, t = lo & 7;
        const double A = (double)((e ? (1u << 23) : 0u) + ((uint32_t)t << kHShift));
        const int ex = (e ? e : 1) - 150;
        const double nd = (double)vn;
        const unsigned long long D = g_d[i];       // (requested together with g_n, ahead of the `if`: measured, no gain -- 8.9 -> 9.2 us)
        // D2 = hi 2^32 + lo, both below 2^63
        const DD d2 = dd_add(dd_from_u64(g_d2lo[i]), dd_scale2(dd_from_u64(g_d2hi[i]), 32));
        v1 = dd_scale2(dd_add(two_prod(nd, A), dd_from_u64(D)), ex);
        const DD cross = two_prod(2.0 * A, (double)D);             // D < 2^51: exact as a double
        v2 = dd_scale2(dd_add(dd_add(two_prod(nd, A * A), cross), d2), 2 * ex);
    }
    // inclusive scan inside every wave (shuffles), the 16 wave totals scanned by the first wave, stitched: 3 barriers instead
    // of the 20 of a Hillis-Steele scan over LDS (the kernel took 10-12 us for <= 170 workgroups); fixed association
    const int lane = tid & 63, wave = tid >> 6;
    DD i1 = v1, i2 = v2;
    uint32_t in = vn;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const DD a1 = dd_shfl_up(i1, off), a2 = dd_shfl_up(i2, off);   // (every lane takes part in the shuffle)
        const uint32_t an = __shfl_up(in, off, 64);
        if (lane >= off) {
            i1 = dd_add(i1, a1);
            i2 = dd_add(i2, a2);
            in += an;
        }
    }
    if (lane == 63) {
        s1[wave] = i1;
        s2[wave] = i2;
        sn[wave] = in;
    }
    __syncthreads();
    if (wave == 0) {
        constexpr int kW = kSuper / 64;
        DD w1 = lane < kW ? s1[lane] : zero, w2 = lane < kW ? s2[lane] : zero;
        uint32_t wn = lane < kW ? sn[lane] : 0u;
#pragma unroll
        for (int off = 1; off < kW; off <<= 1) {
            const DD a1 = dd_shfl_up(w1, off), a2 = dd_shfl_up(w2, off);
            const uint32_t an = __shfl_up(wn, off, 64);
            if (lane >= off) {
                w1 = dd_add(w1, a1);
                w2 = dd_add(w2, a2);
                wn += an;
            }
        }
        if (lane < kW) {              // inclusive totals up to and including wave `lane`
            s1[kW + lane] = w1;
            s2[kW + lane] = w2;
            sn[kW + lane] = wn;
        }
    }
    __syncthreads();
    {
        constexpr int kW = kSuper / 64;
        // exclusive prefix of this thread = (waves before) + (lanes before in this wave)
        DD e1 = dd_shfl_up(i1, 1), e2 = dd_shfl_up(i2, 1);
        uint32_t en = __shfl_up(in, 1, 64);
        if (lane == 0) {
            e1 = zero;
            e2 = zero;
            en = 0u;
        }
        if (wave > 0) {
            e1 = dd_add(s1[kW + wave - 1], e1);
            e2 = dd_add(s2[kW + wave - 1], e2);
            en += sn[kW + wave - 1];
        }
        if (i <= ni) {                                  // exclusive, within the superblock (entry ni: everything)
            p1[i] = e1;
            p2[i] = e2;
            pn[i] = en;
        }
        if (tid == kSuper - 1) {                        // the superblock's totals
            t1[sblk] = s1[2 * kW - 1];
            t2[sblk] = s2[2 * kW - 1];
            tn[sblk] = sn[2 * kW - 1];
        }
    }
}


__global__ void __launch_bounds__(64)
k_iv_scan_top(DD *__restrict__ t1, DD *__restrict__ t2, uint32_t *__restrict__ tn, int64_t nsb)
{
    // a few hundred entries, one wave: a lane owns a contiguous run, the runs' totals are scanned with shuffles
    const int lane = threadIdx.x;
    const int64_t per = (nsb + 63) / 64, lo = lane * per, hi = lo + per < nsb ? lo + per : nsb;
    const DD zero{0.0, 0.0};
    DD s1 = zero, s2 = zero;
    uint32_t sn = 0u;
    for (int64_t i = lo; i < hi; ++i) {
        s1 = dd_add(s1, t1[i]);
        s2 = dd_add(s2, t2[i]);
        sn += tn[i];
    }
    DD i1 = s1, i2 = s2;                                      // inclusive scan over the lanes
    uint32_t in = sn;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const DD a1 = dd_shfl_up(i1, off), a2 = dd_shfl_up(i2, off);   // (every lane takes part in the shuffle)
        const uint32_t an = __shfl_up(in, off, 64);
        if (lane >= off) {
            i1 = dd_add(i1, a1);
            i2 = dd_add(i2, a2);
            in += an;
        }
    }
    const DD u1 = dd_shfl_up(i1, 1), u2 = dd_shfl_up(i2, 1);
    const uint32_t un = __shfl_up(in, 1, 64);
    DD r1 = lane ? u1 : zero, r2 = lane ? u2 : zero;           // exclusive prefix of this lane's run
    uint32_t rn = lane ? un : 0u;
    for (int64_t i = lo; i < hi; ++i) {
        const DD v1 = t1[i], v2 = t2[i];
        const uint32_t vn = tn[i];
        t1[i] = r1;
        t2[i] = r2;
        tn[i] = rn;
        r1 = dd_add(r1, v1);
        r2 = dd_add(r2, v2);
        rn += vn;
    }
}

// ---- 6. candidates --------------------------------------------------------------------------------------------------------
// one workgroup per (mantissa width, candidate); a lane per cell (looping when a format has more than 256 cells)
// Round 6 measured this kernel fused with the interval scan (scan workgroups count themselves done, 1024-thread evaluating
// workgroups poll the counter and read the prefixes with agent-scope loads) and DROPPED it: 26.7 us against 8.5 + 11.6 as two
// launches with 111 pairs, 71.6 against 8.7 + 22.1 with 666 (profiles/r06_scan_eval_fusion_ab.txt).
__global__ void __launch_bounds__(kBlock)
k_mse_eval(const float *__restrict__ x, const float *__restrict__ grid, const float *__restrict__ bt, const float *__restrict__ bq,
           const uint32_t *__restrict__ rank, const int *__restrict__ cflag, const uint32_t *__restrict__ maxkey,
           const DD *__restrict__ p1, const DD *__restrict__ p2, const uint32_t *__restrict__ pn, const DD *__restrict__ t1,
           const DD *__restrict__ t2, const uint32_t *__restrict__ tn, int64_t ni, int nsb, float *mses, HistArgs a,
           double inv_inner, const uint32_t *__restrict__ nunits, SelOne so)
{
    constexpr int NT = kBlock;
    const int j = (int)blockIdx.x, nwg = (int)gridDim.x;
    __shared__ double s_red[NT];
    __shared__ float s_scale[kLutMax];
    __shared__ DD s_t1[kTopLds], s_t2[kTopLds], s_w1[NT / 64], s_w2[NT / 64];
    __shared__ uint32_t s_tn[kTopLds], s_wn[NT / 64];
    const int tid = threadIdx.x;
    const int m = j / a.n_cand;
    float *out = mses + j;
    // non-finite keys: the reference's mean is NaN (a NaN element) or +inf (an infinite one: (x - xq)^2 = inf)
    const uint32_t last = maxkey[0];
    const int flag = cflag[j];
    const bool is_nan = last > 0x7f800000u || flag == kFlagNaN || nunits[2];
    const bool is_inf = !is_nan && last == 0x7f800000u;
    if (is_nan || is_inf) {
        if (tid == 0) {
            const float v = is_nan ? __builtin_nanf("") : __builtin_inff();
            agent_store(out, a.overwrite ? v : *out + v);
        }
        // (falls through to the selection ticket below: every workgroup of the launch takes part)
    }
    double acc = 0.0;
    if (is_nan || is_inf) {
    } else if (flag == kFlagBrute) {
        // element by element with K1's exact arithmetic (slow: one workgroup walks the whole tensor): candidates whose
        // scales leave the normal range, and every candidate under FP8Q_MSE_HIST=2 (the self-check the tests use)
        const QFmt f = a.fmt[m];
        const float gv = grid[j - m * a.n_cand];
        const float mv = fabsf(fmaxf(fabsf(-gv), gv));
        const Chan ch = make_chan(mv, f);
        const float pmaxf = (float)f.pmax;
        for (int p = tid + 1; p <= f.pmax; p += NT) s_scale[p] = lut_entry(ch, p, f.M).x;
        __syncthreads();
        for (int64_t i = tid; i < a.n; i += NT) {
            const float xv = x[i];
            const float xc = __builtin_amdgcn_fmed3f(xv, ch.minv, ch.maxv);          // (unsigned formats: negative -> 0)
            const float ls = floorf(log2_tab(fabsf(xc), kFastTab) + ch.bias);
            const float sc = s_scale[(int)__builtin_amdgcn_fmed3f(ls, 1.0f, pmaxf)];
            const float d = xv - rintf(xc / sc) * sc;
            acc += (double)(d * d);
        }
    } else {
        // exclusive prefix of the superblock totals, made by every workgroup for itself in LDS (<= kTopLds superblocks: a
        // separate one-wave launch for a few dozen entries cost 4.6 us on the device and a launch on the host; beyond
        // that k_iv_scan_top has run and t1 / t2 / tn already hold the prefixes)
        const DD *q1 = t1, *q2 = t2;
        const uint32_t *qn = tn;
        // this thread's first cell (most formats have <= 256: its only one), requested before the scan of the superblock totals
        // instead of behind its barriers: one round trip less on a launch that is a chain of them
        const int ncells = a.ncells[m];
        const float *T = bt + (int64_t)j * a.stride, *Q = bq + (int64_t)j * a.stride;
        const uint32_t *R = rank + (int64_t)j * a.stride;
        float f_lo = 0.0f, f_hi = 0.0f, f_q = 0.0f;
        uint32_t f_r0 = 0u, f_r1 = 0u;
        if (tid < ncells) {
            f_lo = T[tid];
            f_hi = tid + 1 < ncells ? T[tid + 1] : __builtin_inff();
            f_q = Q[tid];
            f_r0 = R[tid];
            f_r1 = tid + 1 < ncells ? R[tid + 1] : 0u;
        }
        if (nsb <= kTopLds) {
            const DD zero{0.0, 0.0};
            const int lane = tid & 63, wave = tid >> 6;
            DD carry1 = zero, carry2 = zero;
            uint32_t carryn = 0u;
            for (int base = 0; base < nsb; base += NT) {               // (one trip for up to NT superblocks)
                const int i = base + tid;
                DD v1 = i < nsb ? t1[i] : zero, v2 = i < nsb ? t2[i] : zero;
                uint32_t vn = i < nsb ? tn[i] : 0u;
                DD i1 = v1, i2 = v2;
                uint32_t in = vn;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const DD a1 = dd_shfl_up(i1, off), a2 = dd_shfl_up(i2, off);
                    const uint32_t an = __shfl_up(in, off, 64);
                    if (lane >= off) {
                        i1 = dd_add(i1, a1);
                        i2 = dd_add(i2, a2);
                        in += an;
                    }
                }
                if (lane == 63) {
                    s_w1[wave] = i1;
                    s_w2[wave] = i2;
                    s_wn[wave] = in;
                }
                __syncthreads();
                DD e1 = dd_shfl_up(i1, 1), e2 = dd_shfl_up(i2, 1);
                uint32_t en = __shfl_up(in, 1, 64);
                if (lane == 0) {
                    e1 = zero;
                    e2 = zero;
                    en = 0u;
                }
                DD c1 = carry1, c2 = carry2;
                uint32_t cn = carryn;
                for (int w = 0; w < wave; ++w) {                        // fixed association
                    c1 = dd_add(c1, s_w1[w]);
                    c2 = dd_add(c2, s_w2[w]);
                    cn += s_wn[w];
                }
                if (i < nsb) {
                    s_t1[i] = dd_add(c1, e1);
                    s_t2[i] = dd_add(c2, e2);
                    s_tn[i] = cn + en;
                }
                for (int w = 0; w < NT / 64; ++w) {
                    carry1 = dd_add(carry1, s_w1[w]);
                    carry2 = dd_add(carry2, s_w2[w]);
                    carryn += s_wn[w];
                }
                __syncthreads();
            }
            q1 = s_t1;
            q2 = s_t2;
            qn = s_tn;
        }
        for (int c = tid; c < ncells; c += NT) {
            const bool first = c == tid;
            const float lo = first ? f_lo : T[c], hi = first ? f_hi : (c + 1 < ncells ? T[c + 1] : __builtin_inff());
            if (!(lo < hi)) continue;
            const int64_t i0 = c ? (int64_t)(first ? f_r0 : R[c]) : 0;
            const int64_t i1 = hi < __builtin_inff() ? (int64_t)(first ? f_r1 : R[c + 1]) : ni;   // (T = +inf: not a border)
            const int64_t b0 = i0 / kSuper, b1 = i1 / kSuper;
            const uint32_t cnt = (qn[b1] + pn[i1]) - (qn[b0] + pn[i0]);
            if (cnt == 0u) continue;
            const DD m1lo = dd_add(q1[b0], p1[i0]), m1hi = dd_add(q1[b1], p1[i1]);
            const DD m2lo = dd_add(q2[b0], p2[i0]), m2hi = dd_add(q2[b1], p2[i1]);
            // S2 - 2 q S1 + n q^2 in double-double (q^2 of an fp32 q is exact in double); the cell's result is >= 0
            const double qd = (double)(first ? f_q : Q[c]);
            const DD d2 = dd_add(m2hi, dd_neg(m2lo)), d1 = dd_add(m1hi, dd_neg(m1lo));
            const DD e = dd_add(dd_add(d2, dd_mul_d(d1, -2.0 * qd)), two_prod((double)cnt, qd * qd));
            acc += e.hi + e.lo;
        }
    }
    s_red[tid] = acc;
    __syncthreads();
    for (int off = NT / 2; off >= 1; off >>= 1) {                // fixed tree: deterministic
        if (tid < off) s_red[tid] += s_red[tid + off];
        __syncthreads();
    }
    if (tid == 0 && !(is_nan || is_inf)) {
        double tot = s_red[0] < 0.0 ? 0.0 : s_red[0];           // (rounding can leave a tiny negative number for an exact fit)
        if (a.uns && flag != kFlagBrute) tot += reinterpret_cast<const double *>(maxkey)[1];   // negative elements: x^2 each
        agent_store(out, a.overwrite ? (float)(tot * inv_inner) : *out + (float)(tot * inv_inner));
    }
    // the winner of the search: the last workgroup to finish its entry selects (fp8q_select.h; per-tensor quantizers)
    if (so.enabled && last_workgroup(so.ticket, (unsigned)nwg, (unsigned)j)) select_one_row(mses, grid, a.n_m, a.n_cand, so);
}


// ---- host -----------------------------------------------------------------------------------------------------------------
size_t align_up(size_t v, size_t al) { return (v + al - 1) / al * al; }

int env_int(const char *name, int dflt, int lo, int hi)
{
    const char *e = getenv(name);
    if (!e) return dflt;
    const long v = atol(e);
    return v < lo ? lo : (v > hi ? hi : (int)v);
}

// tuning knobs (defaults are the measured best): borders per k_moments chunk, smallest key slice
int hist_bcap() { static const int v = env_int("FP8Q_MSE_BCAP", 1024, 16, 6144); return v; }
// Smallest key slice of a k_moments unit.  The launch has 2048 workgroups; a tensor of a few million keys cut into 16 K-key
// slices gives a few hundred units -- most of the chip idle and k_moments latency-bound at ~29 us whatever the size.  Slices
// that make ~2048-4096 units: [64,24,56,56] 29.2 -> 12.4 us, [64,192,14,14] 28.8 -> 11.1; at 25.7 M elements 16 K stays best
// (profiles/r06_mse_slice_ab.txt).  FP8Q_MSE_SLICE overrides (A/B).
int hist_slice_min(int64_t n)
{
    static const int env = getenv("FP8Q_MSE_SLICE") ? env_int("FP8Q_MSE_SLICE", 16384, 1024, 65536) & ~3 : 0;
    if (env) return env;
    int s = 2048;
    while (s < 16384 && (int64_t)s * 2048 < n) s <<= 1;
    return s;
}


struct HistLayout {
    size_t keys, zero0, zero_bytes;           // [zero0, zero0 + zero_bytes): cleared in every call (k_tab_scan)
    size_t gn, gd, gd2lo, gd2hi;
    size_t hist, bhist, ktab, btab, kmax, kneg, maxkey, nunits, gtab;
    size_t koff, boff, units, bt, bq, rank, cflag, pairs, sb, p1, p2, pn, t1, t2, tn;
    size_t total;
    int64_t nbord, ni, nsb;
    uint32_t units_max;
};

constexpr int kPartWgs = 768;       // partition workgroups (3 per CU: 48 KB of LDS each): rows of the key count table

HistLayout hist_layout(int64_t n, int64_t n_pairs, int stride, int bcap, int slice_min)
{
    HistLayout L;
    L.nbord = n_pairs * stride;                       // borders: at most stride - 1 per candidate
    L.ni = L.nbord + kHBuckets;                       // intervals: one more than its borders per bucket
    L.nsb = cdiv(L.ni + 1, kSuper);
    // units: sum over the buckets of chunks x slices <= (sum of slices) x (largest chunk count).  A candidate has at most
    // 2^M / 4 + 3 borders in one coarse bucket (an eighth of a binade: bucket length / s_p < 2^M / 4), and 2^M <= stride / 3;
    // should a bucket ever hold more, plan_body raises the overflow flag and every table entry of the call becomes NaN
    const int64_t per_bucket = n_pairs * (stride / 12 + 4);
    const int64_t max_chunks = cdiv(per_bucket, bcap) + 1, max_slices = cdiv(n, slice_min) + kHBuckets;
    const int64_t um = max_chunks * max_slices;
    L.units_max = (uint32_t)(um > (1 << 22) ? (1 << 22) : um);
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o += align_up(bytes, 256);
        return at;
    };
    L.keys = take((size_t)n * 4 + 16);
    L.zero0 = o;
    L.gn = take((size_t)(L.ni + 1) * 4);
    L.gd = take((size_t)(L.ni + 1) * 8);
    L.gd2lo = take((size_t)(L.ni + 1) * 8);
    L.gd2hi = take((size_t)(L.ni + 1) * 8);
    L.zero_bytes = o - L.zero0;
    L.hist = take(kHBuckets * 4);
    L.bhist = take(kHBuckets * 4);
    L.ktab = take((size_t)kPartWgs * kHBuckets * 4);
    L.btab = take((size_t)n_pairs * kHBuckets * 4);
    L.kmax = take(kPartWgs * 4);
    L.kneg = take(kPartWgs * 8);
    L.maxkey = take(16);           // {largest key, pad, double: sum of squares of the negative elements (unsigned formats)}
    L.nunits = take(16);
    // sub-bin tables: one per bucket that has borders -- at most min(2048, borders) of them
    L.gtab = take((size_t)(L.nbord < kHBuckets ? L.nbord : kHBuckets) * (kHSub + 1) * 4);
    L.koff = take((kHBuckets + 1) * 4);
    L.boff = take((kHBuckets + 1) * 4);
    L.units = take((size_t)L.units_max * sizeof(Unit));
    L.bt = take((size_t)L.nbord * 4);
    L.bq = take((size_t)L.nbord * 4);
    L.rank = take((size_t)L.nbord * 4);
    L.cflag = take((size_t)n_pairs * 4);
    L.pairs = take((size_t)L.nbord * 8);
    L.sb = take((size_t)L.nbord * 4);
    L.p1 = take((size_t)(L.nsb * kSuper) * sizeof(DD));
    L.p2 = take((size_t)(L.nsb * kSuper) * sizeof(DD));
    L.pn = take((size_t)(L.nsb * kSuper) * 4);
    L.t1 = take((size_t)(L.nsb + 1) * sizeof(DD));
    L.t2 = take((size_t)(L.nsb + 1) * sizeof(DD));
    L.tn = take((size_t)(L.nsb + 1) * 4);
    L.total = o + 256;
    return L;
}

}  // namespace

// (called from fp8q_mse.hip)
size_t fp8q_mse_hist_workspace_bytes(int64_t n, int64_t n_pairs)
{
    return hist_layout(n, n_pairs, kStrideBound, hist_bcap(), hist_slice_min(n)).total;
}

// formats this route takes: at most 8 bits, signed or unsigned (the cell tables are sized for them)
bool fp8q_mse_hist_supported(const QFmt *fmts, int n_m, int n_bits)
{
    if (n_m > kHistMaxM || n_bits > 8) return false;
    for (int m = 0; m < n_m; ++m)
        if (fmts[m].sign_bits != fmts[0].sign_bits || (fmts[m].pmax + 1) * (1 << (int)fmts[m].M) + 2 > kStrideBound) return false;
    return true;
}

int fp8q_mse_hist_launch(const float *x, int64_t n, const float *grid, int64_t n_cand, const QFmt *fmts, int n_m, float *mses,
                         void *ws, size_t ws_bytes, hipStream_t st, int brute, int overwrite, const SelOne *sel)
{
    if (n_m > kHistMaxM || n >= (1ll << 31) || ((uintptr_t)ws & 7)) return FP8Q_EWORKSPACE;
    HistArgs a;
    memset(&a, 0, sizeof(a));
    a.n_m = n_m;
    a.n_cand = (int)n_cand;
    a.n = n;
    a.stride = 0;
    a.uns = fmts[0].sign_bits == 0;
    a.overwrite = overwrite;
    for (int m = 0; m < n_m; ++m) {
        a.fmt[m] = fmts[m];
        const int M = (int)fmts[m].M;
        a.ncells[m] = (2 << M) + 1 + (fmts[m].pmax - 1) * ((1 << M) + 1) + 1;
        if (a.ncells[m] > a.stride) a.stride = a.ncells[m];
    }
    if (a.stride > kStrideBound) return FP8Q_EUNSUPPORTED;
    const int bcap = hist_bcap(), slice_min = hist_slice_min(n);
    const int64_t n_pairs = (int64_t)n_m * n_cand;
    const HistLayout L = hist_layout(n, n_pairs, a.stride, bcap, slice_min);
    if (ws_bytes < L.total) return FP8Q_EWORKSPACE;
    char *w = (char *)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    auto at = [&](size_t off) { return (void *)(w + off); };
    uint32_t *keys = (uint32_t *)at(L.keys), *hist = (uint32_t *)at(L.hist), *bhist = (uint32_t *)at(L.bhist);
    uint32_t *ktab = (uint32_t *)at(L.ktab), *btab = (uint32_t *)at(L.btab), *kmax = (uint32_t *)at(L.kmax);
    double *kneg = a.uns ? (double *)at(L.kneg) : nullptr;
    uint32_t *maxkey = (uint32_t *)at(L.maxkey), *nunits = (uint32_t *)at(L.nunits), *gn = (uint32_t *)at(L.gn);
    uint32_t *gtab = (uint32_t *)at(L.gtab);
    unsigned long long *gd = (unsigned long long *)at(L.gd), *gd2lo = (unsigned long long *)at(L.gd2lo),
                       *gd2hi = (unsigned long long *)at(L.gd2hi);
    uint32_t *koff = (uint32_t *)at(L.koff), *boff = (uint32_t *)at(L.boff);
    Unit *units = (Unit *)at(L.units);
    float *bt = (float *)at(L.bt), *bq = (float *)at(L.bq);
    uint32_t *rank = (uint32_t *)at(L.rank), *sb = (uint32_t *)at(L.sb), *pn = (uint32_t *)at(L.pn), *tn = (uint32_t *)at(L.tn);
    int *cflag = (int *)at(L.cflag);
    uint64_t *pairs = (uint64_t *)at(L.pairs);
    DD *p1 = (DD *)at(L.p1), *p2 = (DD *)at(L.p2), *t1 = (DD *)at(L.t1), *t2 = (DD *)at(L.t2);

    const uint32_t *xb = reinterpret_cast<const uint32_t *>(x);
    const int64_t ntiles = cdiv(n, kPartTile);
    // (round 6: an even split of the tiles over all 768 workgroups -- 4 or 5 tiles each instead of 628 workgroups of 5 on
    // [64,32,112,112] -- was slower: histogram 21.8 -> 23.8 us, scatter 44.4 -> 47.9)
    const int tpw = (int)cdiv(ntiles, kPartWgs);              // tiles per partition workgroup
    const int pwgs = (int)cdiv(ntiles, tpw);                  // <= kPartWgs, none of them empty
    if (a.uns)
        hipLaunchKernelGGL(k_stage1<true>, dim3((unsigned)(n_pairs + pwgs)), dim3(kBlock), 0, st, xb, n, ntiles, tpw, ktab, kmax, kneg,
                           grid, a, brute, bt, bq, cflag, btab);
    else
        hipLaunchKernelGGL(k_stage1<false>, dim3((unsigned)(n_pairs + pwgs)), dim3(kBlock), 0, st, xb, n, ntiles, tpw, ktab, kmax, kneg,
                           grid, a, brute, bt, bq, cflag, btab);
    if (int rc = launch_rc()) return rc;
    hipLaunchKernelGGL(k_tab_scan, dim3(kHBuckets / 32, 2), dim3(1024), 0, st, ktab, pwgs, hist, btab, (int)n_pairs, bhist,
                       (uint4 *)at(L.zero0), (int64_t)(L.zero_bytes / 16));
    if (int rc = launch_rc()) return rc;
    static const int merged = env_int("FP8Q_MSE_MERGE", 1, 0, 1);      // 0: sort + plan and scatter as two launches (A/B)
    if (merged) {
        if (a.uns)
            hipLaunchKernelGGL(k_sort_plan_scatter<true>, dim3((unsigned)(1 + kHBuckets + pwgs)), dim3(kBlock), 0, st, bt, btab, bhist,
                               (int)n_pairs, a.stride, sb, rank, gtab, pairs, hist, kmax, kneg, pwgs, koff, boff, units, nunits, maxkey,
                               L.units_max, bcap, slice_min, xb, n, ntiles, tpw, ktab, keys);
        else
            hipLaunchKernelGGL(k_sort_plan_scatter<false>, dim3((unsigned)(1 + kHBuckets + pwgs)), dim3(kBlock), 0, st, bt, btab, bhist,
                               (int)n_pairs, a.stride, sb, rank, gtab, pairs, hist, kmax, kneg, pwgs, koff, boff, units, nunits, maxkey,
                               L.units_max, bcap, slice_min, xb, n, ntiles, tpw, ktab, keys);
        if (int rc = launch_rc()) return rc;
    } else {
        hipLaunchKernelGGL(k_border_sort_plan, dim3(kHBuckets + 1), dim3(kBlock), 0, st, bt, btab, bhist, (int)n_pairs, a.stride, sb,
                           rank, gtab, pairs, hist, kmax, kneg, pwgs, koff, boff, units, nunits, maxkey, L.units_max, bcap, slice_min);
        if (int rc = launch_rc()) return rc;
        if (a.uns)
            hipLaunchKernelGGL(k_part_scatter<true>, dim3((unsigned)pwgs), dim3(kBlock), 0, st, xb, n, ntiles, tpw, hist, ktab, keys);
        else
            hipLaunchKernelGGL(k_part_scatter<false>, dim3((unsigned)pwgs), dim3(kBlock), 0, st, xb, n, ntiles, tpw, hist, ktab, keys);
        if (int rc = launch_rc()) return rc;
    }
    const size_t shmem = (size_t)kMomTabWords * 4 + (size_t)((bcap + 3) & ~3) * 4 + (size_t)(bcap + 1) * 16;
    // threads per k_moments workgroup.  At 25.7 M keys the kernel is bound by VALU issue (~44 instructions per key) and 256 threads are
    // best (48.6 us against 51.7 with 111 pairs, 88 against 107 with 666); on MobileNetV2's smaller activations a few hundred
    // units of one or two 4096-key batches each leave most CUs with one workgroup, the unit's own latency is the launch's
    // duration, and 512 threads halve it: 25.9 -> 17.7 us at 0.5 M elements, 42.1 -> 27.4 at 7.2 M with 666 pairs
    // (profiles/r06_moments_nt_ab.txt).  Third session, after the branch-free bisection: the same rule holds -- with 111 pairs 256
    // threads win from 4.8 M keys (10.7 us against 14.1), with 666 pairs 512 threads up to 12.8 M (33.2 against 37.5; 19.3 M: 51.7
    // against 47.8) -- the second threshold moved from 12 M to 16 M elements.  FP8Q_MSE_MOM_NT = 256 / 512 overrides.
    static const int mom_env = env_int("FP8Q_MSE_MOM_NT", 0, 0, 512);
    const int mom_nt = mom_env >= 256 ? (mom_env >= 512 ? 512 : 256) : (n <= (n_pairs > 256 ? (16ll << 20) : (3ll << 20)) ? 512 : 256);
    if (shmem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(mom_nt == 512 ? (const void *)k_moments<512> : (const void *)k_moments<256>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return hip_rc(e);
    }
    if (mom_nt == 512)
        hipLaunchKernelGGL(k_moments<512>, dim3(2048), dim3(512), shmem, st, keys, sb, gtab, units, nunits, gn, gd, gd2lo, gd2hi, bcap);
    else
        hipLaunchKernelGGL(k_moments<256>, dim3(2048), dim3(256), shmem, st, keys, sb, gtab, units, nunits, gn, gd, gd2lo, gd2hi, bcap);
    if (int rc = launch_rc()) return rc;
    SelOne so;
    memset(&so, 0, sizeof(so));
    if (sel) so = *sel;
    hipLaunchKernelGGL(k_iv_scan_super, dim3((unsigned)L.nsb), dim3(kSuper), 0, st, boff, gn, gd, gd2lo, gd2hi, L.ni, p1, p2, pn, t1,
                       t2, tn);
    if (int rc = launch_rc()) return rc;
    if (L.nsb > kTopLds) {          // (more than half a million intervals: thousands of candidates)
        hipLaunchKernelGGL(k_iv_scan_top, dim3(1), dim3(64), 0, st, t1, t2, tn, L.nsb);
        if (int rc = launch_rc()) return rc;
    }
    hipLaunchKernelGGL(k_mse_eval, dim3((unsigned)n_pairs), dim3(kBlock), 0, st, x, grid, bt, bq, rank, cflag, maxkey, p1, p2, pn, t1,
                       t2, tn, L.ni, (int)L.nsb, mses, a, 1.0 / (double)n, nunits, so);
    return launch_rc();
}
