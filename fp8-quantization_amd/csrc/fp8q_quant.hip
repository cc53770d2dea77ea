// fp8q_quant.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the quantize / min-max family + their C ABI
// (include/fp8q.h).  The library has four translation units: this one, fp8q_mse.hip (K4), fp8q_epilogue.hip (N2) and
// fp8q_codec.hip (N3); fp8q_common.h holds what they share, fp8q_device.h the per-element arithmetic.
//
// Every kernel here is elementwise or a reduction: the roofline is HBM bandwidth, not MFMA.
// Common shape: 256-thread blocks (4 waves, one per SIMD), 16 B per lane per memory instruction
// (1 KiB per wave-instruction), 4 independent loads in flight per lane, and -- what HBM turned out to
// care about most (a copy-kernel sweep: docs/HISTORY.md) -- every block moves ALIGNED 16 KiB pieces, neighbouring
// blocks neighbouring pieces, one piece (or one short tile) per block rather than a persistent grid.
//
// Kernels of this file:
//   k_quant_rows      K1, one channel per blockIdx.y (per-tensor: one row).  Scale LUT in LDS.
//   k_rows_flat       per-channel tensors with short rows, cut into aligned 4096-element chunks regardless of
//                     the rows (per-row tables in LDS): MODE 0 = K1, MODE 1 = K2+K5+K1 fused (rows <= 256).
//   k_rows_staged     K2+K5+K1 fused for rows <= 256 elements of any length (147): the aligned chunk is fetched once and
//                     parked in LDS; k_rows_staged_mm = its K2 (+fold) twin for rows of 4..256 elements.
//   k_rows_reg        K2+K5+K1 fused, or K2 alone, for rows of 128..8192 elements: the row stays in registers.
//   k_rows_direct     round-1 row-tiled kernel: rows too short for per-row tables, unaligned pointers, the
//                     fused / K2 cases the kernels above do not take.
//   k_multi_flat      multi-tensor K1: one block = one chunk of one of <= 32 tensors.
//   k_quant_scalar    K1 fallback for x / y that are not 16-byte co-aligned.
//   k_minmax_partial  K2/K3 stage 1: per-(row, split) min / max / NaN flag  (stage 2 + K5: fp8q_common.h).
//   k_copy            float4 copy with K1's launch geometry (measured HBM ceiling).
#include <vector>

#include "fp8q_common.h"
#include "fp8q_select.h"

namespace {

// ---------------------------------------------------------------------------------------------
// K1 rows: blockIdx.y = row (channel), blockIdx.x strides over the row.  {s, 1/s} table in LDS.
// ---------------------------------------------------------------------------------------------
// U = 16-byte groups per lane and step: 4 (16 KiB pieces per block), or 1 for cache-sized tensors, whose launches are
// latency-bound and want four times the blocks (as k_affine_act: fp8q_epilogue.hip)
template <bool NT, int U>
__device__ __forceinline__ void quant_rows_body(const float *__restrict__ x, float *__restrict__ y, int64_t inner,
                                                const float *__restrict__ maxval, int per_channel, const QFmt &f)
{
    __shared__ float2 lut[kLutMax];
    const int row = blockIdx.y;
    const int tid = threadIdx.x;
    const Chan cfull = make_chan(maxval[per_channel ? row : 0], f);
    for (int i = tid; i <= f.pmax; i += kBlock) lut[i] = lut_entry(cfull, i, f.M);
    __syncthreads();
    const ChanLite c = lite(cfull);
    const float pmaxf = (float)f.pmax;
    const float qthr = f.qthr;

    const float *xr = x + (int64_t)row * inner;
    float *yr = y + (int64_t)row * inner;
    // peel to 16-byte alignment (x and y are co-aligned: checked on the host)
    int64_t head = ((16 - ((uintptr_t)xr & 15)) & 15) >> 2;
    if (head > inner) head = inner;
    const int64_t nvec = (inner - head) >> 2;
    const int64_t tail0 = head + (nvec << 2);
    if (blockIdx.x == 0) {
        if (tid < head) yr[tid] = quant_one(xr[tid], c, lut, pmaxf, qthr);
        const int64_t t = tail0 + tid;
        if (t < inner) yr[t] = quant_one(xr[t], c, lut, pmaxf, qthr);
    }
    const vf4 *xv = reinterpret_cast<const vf4 *>(xr + head);
    vf4 *yv = reinterpret_cast<vf4 *>(yr + head);

    const int64_t step = (int64_t)gridDim.x * (kBlock * U);
    for (int64_t base = (int64_t)blockIdx.x * (kBlock * U); base < nvec; base += step) {
        if (base + kBlock * U <= nvec) {
            vf4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = ld16<NT>(xv + base + u * kBlock + tid);
            float e[U * 4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                e[4 * u + 0] = v[u].x;
                e[4 * u + 1] = v[u].y;
                e[4 * u + 2] = v[u].z;
                e[4 * u + 3] = v[u].w;
            }
            quant_group<U * 4>(e, c, lut, pmaxf, qthr);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                vf4 w = {e[4 * u + 0], e[4 * u + 1], e[4 * u + 2], e[4 * u + 3]};
                st16<NT>(yv + base + u * kBlock + tid, w);
            }
        } else {
            for (int u = 0; u < U; ++u) {
                const int64_t i = base + u * kBlock + tid;
                if (i < nvec) {
                    const vf4 w = ld16<NT>(xv + i);
                    float e[4] = {w.x, w.y, w.z, w.w};
                    quant_group<4>(e, c, lut, pmaxf, qthr);
                    vf4 o = {e[0], e[1], e[2], e[3]};
                    st16<NT>(yv + i, o);
                }
            }
        }
    }
}

template <bool NT, int U>
__global__ void __launch_bounds__(kBlock)
k_quant_rows(const float *__restrict__ x, float *__restrict__ y, int64_t inner,
             const float *__restrict__ maxval, int per_channel, QFmt f)
{
    quant_rows_body<NT, U>(x, y, inner, maxval, per_channel, f);
}

// arguments of k_rows_direct
struct TileArgs {
    int inner;          // row length
    int rows;           // R: rows per tile
    int lut_stride;     // pmax + 1
    int group;          // G: lanes per row (power of two <= 64, or 256 = whole block)
    int coaligned;      // k_rows_direct: x and y share their 16-byte phase -> aligned vector body
    uint32_t magic;     // n / inner      (see magic_of)
    uint32_t lmagic;    // n / lut_stride
};


__device__ __forceinline__ ChanLite lite_lds(const Chan *c)
{
    const float4 h = *reinterpret_cast<const float4 *>(c);   // one ds_read_b128
    ChanLite l;
    l.maxv = h.x;
    l.minv = h.y;
    l.bias = h.z;
    l.pthr = h.w;
    return l;
}

constexpr int kModeQuant = 0, kModeFused = 1, kModeMinMax = 2;
constexpr int kModeEncode = 3, kModeDecode = 4;   // k_rows_flat only: storage codes (N3) of per-channel short rows

// ---------------------------------------------------------------------------------------------
// Short rows, register-streamed: k_rows_direct (inner <= kDirectMaxInner).
// G lanes own one row: every element a lane touches belongs to ONE channel, so the channel
// constants / table pointer are loaded once per row and the inner loop is the per-tensor one.
// Rows start at arbitrary 4-byte offsets: lanes use 16-byte accesses at 4-byte alignment (one
// dwordx4 each); the G lanes of a row cover G*16 contiguous bytes per instruction.
// A block takes R rows per iteration; LDS holds only the tables of those rows:
//   pass A (MODE 1, 2) row min/max straight from global, G-lane shuffle reduction
//   tables            make_chan (thread j <-> row j), then {s, 1/s} entries over all threads
//   pass B            quantize the R rows as one flat contiguous range (coalesced 16-byte I/O);
//                     in MODE 1 this re-reads the rows, which are L2-resident
// Dynamic LDS: float rowmv[R4] | float4 patch[R] | Chan chans[R] | float2 lut[R * lut_stride]
// ---------------------------------------------------------------------------------------------
template <int MODE, bool LUT, bool NT>
__global__ void __launch_bounds__(kBlock, 4)   // <= 128 VGPRs: 4 blocks of 256 per CU
k_rows_direct(const float *__restrict__ x, float *__restrict__ y, int64_t C,
              const float *__restrict__ maxval, float *row_min, float *row_max, float *maxval_out,
              QFmt f, TileArgs a, FoldArgs fa)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Rmax = a.rows;
    float *rowmv = reinterpret_cast<float *>(smem);
    float4 *patch = reinterpret_cast<float4 *>(rowmv + ((Rmax + 3) & ~3));
    Chan *chans = reinterpret_cast<Chan *>(patch + Rmax);
    float2 *lut = reinterpret_cast<float2 *>(chans + Rmax);
    const int tid = threadIdx.x;
    constexpr int BS = kBlock;
    const int G = a.group, rpp = BS / G;
    const int sub = tid & (G - 1), slot = tid / G;
    const int inner = a.inner, inner4 = inner & ~3;
    const float pmaxf = (float)f.pmax;
    // log2/exp2 tables: staged in LDS for 256-thread blocks; single-wave blocks read them through L1
    __shared__ double ftab_lds[kFastTabSize];
    const double *ftab = ftab_lds;
    if (MODE != kModeMinMax)
        for (int i = threadIdx.x; i < kFastTabSize; i += BS) ftab_lds[i] = kFastTab[i];

    for (int64_t r0 = (int64_t)blockIdx.x * Rmax; r0 < C; r0 += (int64_t)gridDim.x * Rmax) {
        const int R = (int)((C - r0) < Rmax ? (C - r0) : Rmax);
        // geometry of pass B (the R rows as one flat range) -- needed early for the prefetch
        const int n = R * inner;
        const float *xt = x + r0 * inner;
        float *yt = y + r0 * inner;
        int head = a.coaligned ? (int)((4 - (((uintptr_t)xt >> 2) & 3)) & 3) : 0;
        if (head > n || inner < 4) head = n;          // rows shorter than a group: all scalar
        const int nvec = (n - head) >> 2;
        const int bend = head + (nvec << 2);
        constexpr int U = 4;   // four 16-byte loads in flight per lane
        // (prefetching the first U loads before the table phase was measured: +25 VGPRs, one wave
        // per SIMD less, -8 %: not done)
        __syncthreads();   // tables of the previous iteration are no longer read (and ftab is staged)
        if (MODE != kModeQuant) {
            for (int rb = 0; rb < R; rb += rpp) {
                const int r = rb + slot;
                MinMax m;
                mm_init(m);
                if (r < R) {
                    const float *xr = x + (r0 + r) * inner;
                    int i = sub * 4;
                    for (; i + 3 * G * 4 < inner4; i += G * 16) {   // four 16-byte loads in flight
                        vf4 v[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) v[u] = ld16u<false>(xr + i + u * G * 4);   // stay in L2 for pass B
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            mm_acc(m, v[u].x);
                            mm_acc(m, v[u].y);
                            mm_acc(m, v[u].z);
                            mm_acc(m, v[u].w);
                        }
                    }
                    for (; i < inner4; i += G * 4) {
                        const vf4 v = ld16u<false>(xr + i);
                        mm_acc(m, v.x);
                        mm_acc(m, v.y);
                        mm_acc(m, v.z);
                        mm_acc(m, v.w);
                    }
                    for (int j = inner4 + sub; j < inner; j += G) mm_acc(m, xr[j]);
                }
                for (int off = G >> 1; off >= 1; off >>= 1) {
                    m.mn = fminf(m.mn, __shfl_xor(m.mn, off, 64));
                    m.mx = fmaxf(m.mx, __shfl_xor(m.mx, off, 64));
                    m.nan |= __shfl_xor(m.nan, off, 64);
                }
                if (r < R && sub == 0) {
                    if (m.nan) m.mn = m.mx = __builtin_nanf("");
                    if (MODE == kModeMinMax) {
                        fold_store(m.mn, m.mx, r0 + r, row_min, row_max, maxval_out, fa);
                    } else {
                        const float mv = fabsf(tmax(fabsf(m.mn), m.mx));   // fp8_quantizer.py:236
                        if (row_min) row_min[r0 + r] = m.mn;
                        if (row_max) row_max[r0 + r] = m.mx;
                        if (maxval_out) maxval_out[r0 + r] = mv;
                        rowmv[r] = mv;
                    }
                }
            }
            if (MODE == kModeMinMax) continue;
            __syncthreads();
        }
        {
            const float *mvsrc = MODE == kModeQuant ? maxval + r0 : rowmv;
            // thread j <-> row j: channel constants, then the row's whole {s, 1/s} table from registers
            for (int j = tid; j < R; j += BS) {
                const Chan c = make_chan_fast(mvsrc[j], f, ftab);
                chans[j] = c;
                if (LUT) lut_row(lut + j * a.lut_stride, c, f);
            }
            __syncthreads();
        }
        // ---- pass B: the R rows are one contiguous range -> flat, fully coalesced 16-byte I/O.
        // The channel of a 16-byte group comes from one magic division.  A group that straddles
        // a row boundary is completed from an LDS patch: one thread per boundary first quantizes
        // the <= 3 elements that follow it (with the next row's constants), so every store of the
        // body is a full aligned 16 bytes (no partial-line read-modify-write in HBM).
        {
            auto quant_at = [&](int i) -> float {
                const int ch = div_small((uint32_t)i, a.magic);
                if (LUT) return quant_one(xt[i], lite_lds(chans + ch), lut + ch * a.lut_stride, pmaxf, f.qthr);
                return quant_direct(xt[i], chans[ch], f.M);
            };
            for (int i = tid; i < head; i += BS) yt[i] = quant_at(i);
            for (int i = bend + tid; i < n; i += BS) yt[i] = quant_at(i);
            // patches: row c+1 starts at local index (c+1)*inner
            for (int c = tid; c < R - 1; c += BS) {
                const int idx = (c + 1) * inner;
                float pv[3] = {0.0f, 0.0f, 0.0f};
                if (idx > head && idx < bend) {
                    const int end = head + (((idx - head) + 3) & ~3);   // end of the straddling group
                    for (int i = idx, k = 0; i < end; ++i, ++k) pv[k] = quant_at(i);   // 0..3 elements
                }
                patch[c] = make_float4(pv[0], pv[1], pv[2], 0.0f);
            }
            __syncthreads();
            for (int j0 = tid; j0 < nvec; j0 += BS * U) {
                vf4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (j0 + u * BS < nvec) v[u] = ld16u<NT>(xt + head + (j0 + u * BS) * 4);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int j = j0 + u * BS;
                    if (j >= nvec) break;
                    const int o = head + j * 4;
                    const int ch = div_small((uint32_t)o, a.magic);
                    const int b = inner - (o - ch * inner);        // elements left in this row (>= 1)
                    float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                    if (LUT) {
                        quant_group<4>(e, lite_lds(chans + ch), lut + ch * a.lut_stride, pmaxf, f.qthr);
                    } else {
                        const Chan c = chans[ch];
#pragma unroll
                        for (int q = 0; q < 4; ++q) e[q] = quant_direct(e[q], c, f.M);
                    }
                    if (b < 4) {   // e[b..3] belong to the next row: take them from its patch
                        const float4 pt = patch[ch];
                        e[3] = b == 3 ? pt.x : (b == 2 ? pt.y : pt.z);
                        if (b < 3) e[2] = b == 2 ? pt.x : pt.y;
                        if (b < 2) e[1] = pt.x;
                    }
                    st16u<NT>(yt + o, vf4{e[0], e[1], e[2], e[3]});
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Short rows, flat: k_rows_flat (MODE 0 = K1, MODE 1 = fused K2+K5+K1), x and y 16-byte aligned.
// HBM wants what a plain grid-stride copy does: every block moves one ALIGNED 16 KiB chunk per step
// and concurrently running blocks touch neighbouring chunks (copy-kernel sweep, docs/HISTORY.md: 6.3-6.4 TB/s;
// a block that owns 128 contiguous KiB, or row-aligned chunks of 16 464 B: 5.2-5.7).  So the tensor
// is cut by ADDRESS, not by rows: chunk c = elements [4096 c, 4096 (c+1)); a tile = nch chunks
// (t*nch + i) * gridDim + blockIdx, i < nch; rows are whatever overlaps a chunk (a row cut by a chunk
// border gets its table built by both neighbours).  Per tile:
//   geometry  one thread per chunk: first row, phase within it, rows overlapped   (1 64-bit division)
//   pass A    (MODE 1) min/max of every overlapping row, G lanes per row; the chunk in which a row
//             STARTS writes row_min / row_max / maxval_out
//   tables    thread <-> (chunk, row): channel constants + the {s, 1/s} table
//   patches   first elements of a row up to the next 16-byte boundary, quantized with THAT row's
//             constants (so the streaming loop only issues whole aligned 16-byte stores); tail scalars
//   stream    one chunk per step: 4 x 16 B in flight per lane, channel of a group by magic division
// LDS: ChunkInfo[8] | float4 patch[Rt] | float4 chanlite[Rt] | float2 lut[Rt * stride] | float rowmv[Rt]
// ---------------------------------------------------------------------------------------------
constexpr int kChunkElems = 4096, kChunkGroups = 1024, kFlatMaxCh = 8;
constexpr int kFlatFusedMaxInner = 256;    // fused: rows cut by a chunk border are read by both neighbours; longer
                                           // rows do better in the row-tiled kernel (measured at 576: 4.65 vs 4.32 TB/s)

struct FlatArgs {
    int inner;        // row length (>= 4)
    int rpc;          // table rows per chunk: most rows a window of 4096 (+3 tail) elements can overlap
    int nch;          // chunks per tile (<= kFlatMaxCh)
    int lut_stride;   // pmax + 1
    int group;        // pass A: lanes per row (power of two <= 64); k_rows_staged: log2 of it
    int tail;         // n - 4 * nvec: scalars after the last 16-byte group
    uint32_t magic;   // o / inner
    uint32_t rmagic;  // lr / rpc
    int64_t nvec;     // 16-byte groups in the tensor (>= 1)
    int64_t nchunks;  // ceil(nvec / 1024)
    int n_bits;       // encode / decode: position of the sign bit
    int pad0;
};

struct __attribute__((aligned(16))) ChunkInfo {
    int64_t row_lo;   // first row overlapping the chunk
    int phase;        // offset of the chunk's first element within that row
    int nrows;        // rows overlapping the chunk (tail scalars included)
    int len;          // elements in the chunk's aligned body (multiple of 4, <= 4096)
    int tail;         // scalars after the body (last chunk of the tensor only)
    int pad[2];
};

__device__ __forceinline__ ChanLite lite_of(const float4 h)
{
    ChanLite l;
    l.maxv = h.x;
    l.minv = h.y;
    l.bias = h.z;
    l.pthr = h.w;
    return l;
}

template <int MODE, bool NT>
__global__ void __launch_bounds__(kBlock, 4)
k_rows_flat(const float *__restrict__ x, float *__restrict__ y, const float *__restrict__ maxval,
            float *row_min, float *row_max, float *maxval_out, QFmt f, FlatArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double ftab[kFastTabSize];
    const int Rt = a.rpc * a.nch;
    ChunkInfo *cinfo = reinterpret_cast<ChunkInfo *>(smem);
    float4 *patch = reinterpret_cast<float4 *>(cinfo + kFlatMaxCh);
    float4 *chl = patch + Rt;
    float2 *lut = reinterpret_cast<float2 *>(chl + Rt);
    float *rowmv = reinterpret_cast<float *>(lut + Rt * a.lut_stride);
    const int tid = threadIdx.x;
    const int inner = a.inner;
    const int64_t G = gridDim.x;
    const float pmaxf = (float)f.pmax;
    for (int i = tid; i < kFastTabSize; i += kBlock) ftab[i] = kFastTab[i];

    for (int64_t c0 = blockIdx.x; c0 < a.nchunks; c0 += G * a.nch) {
        int nct = 1;
        while (nct < a.nch && c0 + nct * G < a.nchunks) ++nct;
        const int nlr = nct * a.rpc;
        __syncthreads();   // the previous tile's tables are no longer read (and ftab is staged)
        if (tid < nct) {
            const int64_t c = c0 + tid * G;
            const int64_t elo = c * kChunkElems;
            const int64_t rem = a.nvec * 4 - elo;
            ChunkInfo ci;
            ci.len = rem < kChunkElems ? (int)rem : kChunkElems;
            ci.tail = (c == a.nchunks - 1) ? a.tail : 0;
            ci.row_lo = elo / inner;
            ci.phase = (int)(elo - ci.row_lo * inner);
            ci.nrows = (ci.phase + ci.len + ci.tail - 1) / inner + 1;
            ci.pad[0] = ci.pad[1] = 0;
            cinfo[tid] = ci;
        }
        __syncthreads();
        if (MODE == kModeFused) {
            const int Gl = a.group, rpp = kBlock / Gl, sub = tid & (Gl - 1), slot = tid / Gl;
            const int inner4 = inner & ~3;
            for (int lrb = 0; lrb < nlr; lrb += rpp) {
                const int lr = lrb + slot;
                const int i = div_small((uint32_t)lr, a.rmagic), r = lr - i * a.rpc;
                const bool valid = lr < nlr && r < cinfo[i].nrows;
                MinMax m;
                mm_init(m);
                if (valid) {
                    const float *xr = x + (cinfo[i].row_lo + r) * inner;   // rows start at any 4-byte phase
                    int j = sub * 4;
                    for (; j + 3 * Gl * 4 < inner4; j += Gl * 16) {
                        vf4 v[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) v[u] = ld16u<false>(xr + j + u * Gl * 4);   // stay in L2 for pass B
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            mm_acc(m, v[u].x);
                            mm_acc(m, v[u].y);
                            mm_acc(m, v[u].z);
                            mm_acc(m, v[u].w);
                        }
                    }
                    for (; j < inner4; j += Gl * 4) {
                        const vf4 v = ld16u<false>(xr + j);
                        mm_acc(m, v.x);
                        mm_acc(m, v.y);
                        mm_acc(m, v.z);
                        mm_acc(m, v.w);
                    }
                    for (int k = inner4 + sub; k < inner; k += Gl) mm_acc(m, xr[k]);
                }
                for (int off = Gl >> 1; off >= 1; off >>= 1) {
                    m.mn = fminf(m.mn, __shfl_xor(m.mn, off, 64));
                    m.mx = fmaxf(m.mx, __shfl_xor(m.mx, off, 64));
                    m.nan |= __shfl_xor(m.nan, off, 64);
                }
                if (valid && sub == 0) {
                    if (m.nan) m.mn = m.mx = __builtin_nanf("");
                    const float mv = fabsf(tmax(fabsf(m.mn), m.mx));   // fp8_quantizer.py:236
                    rowmv[lr] = mv;
                    if (r > 0 || cinfo[i].phase == 0) {   // the row starts in this chunk: this block reports it
                        const int64_t grow = cinfo[i].row_lo + r;
                        if (row_min) row_min[grow] = m.mn;
                        if (row_max) row_max[grow] = m.mx;
                        if (maxval_out) maxval_out[grow] = mv;
                    }
                }
            }
            __syncthreads();
        }
        for (int lr = tid; lr < nlr; lr += kBlock) {
            const int i = div_small((uint32_t)lr, a.rmagic), r = lr - i * a.rpc;
            if (r < cinfo[i].nrows) {
                const float mv = MODE != kModeFused ? maxval[cinfo[i].row_lo + r] : rowmv[lr];
                const Chan c = make_chan_fast(mv, f, ftab);
                chl[lr] = make_float4(c.maxv, c.minv, c.bias, c.pthr);
                lut_row(lut + lr * a.lut_stride, c, f);
            }
        }
        __syncthreads();
        // storage codes (N3): `x` / `y` are the fp32 side, the other pointer is a byte array of codes
        const int Mi = (int)f.M, sign_shift = f.sign_bits == 1 ? a.n_bits - 1 : -1;
        const uint8_t *codes_in = reinterpret_cast<const uint8_t *>(x);    // kModeDecode
        uint8_t *codes_out = reinterpret_cast<uint8_t *>(y);               // kModeEncode
        if (MODE != kModeDecode) {
            for (int lr = tid; lr < nlr; lr += kBlock) {
                const int i = div_small((uint32_t)lr, a.rmagic), r = lr - i * a.rpc;
                const ChunkInfo ci = cinfo[i];
                float pv[3] = {0.0f, 0.0f, 0.0f};
                if (r + 1 < ci.nrows) {
                    const int idx = (r + 1) * inner - ci.phase;   // chunk-local index of row r+1's first element
                    if (idx < ci.len && (idx & 3)) {
                        const float *xc = x + (c0 + i * G) * kChunkElems;
                        const ChanLite cl = lite_of(chl[lr + 1]);
                        const float2 *lt = lut + (lr + 1) * a.lut_stride;
                        for (int k = 0; k < 4 - (idx & 3); ++k)
                            pv[k] = MODE == kModeEncode
                                        ? __uint_as_float(encode_one(xc[idx + k], cl, lt, pmaxf, f.qthr, Mi, sign_shift))
                                        : quant_one(xc[idx + k], cl, lt, pmaxf, f.qthr);
                    }
                }
                patch[lr] = make_float4(pv[0], pv[1], pv[2], 0.0f);
            }
        }
        if (tid < cinfo[nct - 1].tail) {   // the tensor's last <= 3 elements
            const ChunkInfo ci = cinfo[nct - 1];
            const int e = ci.len + tid;
            const int lr = (nct - 1) * a.rpc + div_small((uint32_t)(ci.phase + e), a.magic);
            const int64_t at = (c0 + (nct - 1) * G) * kChunkElems + e;
            if (MODE == kModeEncode)
                codes_out[at] = (uint8_t)encode_one(x[at], lite_of(chl[lr]), lut + lr * a.lut_stride, pmaxf, f.qthr, Mi, sign_shift);
            else if (MODE == kModeDecode)
                y[at] = decode_one(codes_in[at], lut + lr * a.lut_stride, Mi, sign_shift);
            else
                y[at] = quant_one(x[at], lite_of(chl[lr]), lut + lr * a.lut_stride, pmaxf, f.qthr);
        }
        __syncthreads();
        constexpr int U = 4;
        if (MODE == kModeEncode || MODE == kModeDecode) {
            for (int i = 0; i < nct; ++i) {
                const int phase = cinfo[i].phase, ng = cinfo[i].len >> 2;
                const int64_t base = (c0 + i * G) * kChunkElems;
                const int lr0 = i * a.rpc;
                if (MODE == kModeEncode) {
                    // a lane converts FOUR CONSECUTIVE groups (16 elements: 64 contiguous bytes in, the line's other
                    // quarters hit L1) so that their codes leave as ONE 16-byte store: with lane <-> group and a dword
                    // store per group the kernel ran at 2.5 TB/s of its 5 B/element -- store-instruction bound
                    const vf4 *xv = reinterpret_cast<const vf4 *>(x + base);
                    uint32_t *cw = reinterpret_cast<uint32_t *>(codes_out + base);
                    const bool wide = (reinterpret_cast<uintptr_t>(cw) & 15) == 0;
                    vf4 v[U];
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        if (4 * tid + u < ng) v[u] = ld16<false>(xv + 4 * tid + u);
                    uint32_t word[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int q = 4 * tid + u;
                        word[u] = 0u;
                        if (q >= ng) break;
                        const int o = phase + 4 * q;
                        const int lrow = div_small((uint32_t)o, a.magic);
                        const int b = inner - (o - lrow * inner);   // elements left in this row (>= 1)
                        const int lr = lr0 + lrow;
                        const ChanLite cl = lite_of(chl[lr]);
                        const float2 *lt = lut + lr * a.lut_stride;
                        const float in[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                        uint32_t wd = encode_group4(in, cl, lt, pmaxf, f.qthr, Mi, sign_shift);
                        if (b < 4) {   // elements b..3 belong to the next row: its head patch holds their codes
                            const float4 pt = patch[lr];
                            const uint32_t c3 = __float_as_uint(b == 3 ? pt.x : (b == 2 ? pt.y : pt.z));
                            wd = (wd & 0x00ffffffu) | (c3 << 24);
                            if (b < 3) wd = (wd & 0xff00ffffu) | (__float_as_uint(b == 2 ? pt.x : pt.y) << 16);
                            if (b < 2) wd = (wd & 0xffff00ffu) | (__float_as_uint(pt.x) << 8);
                        }
                        word[u] = wd;
                    }
                    if (wide && 4 * tid + 3 < ng) {
                        *reinterpret_cast<uint4 *>(cw + 4 * tid) = make_uint4(word[0], word[1], word[2], word[3]);
                    } else {
#pragma unroll
                        for (int u = 0; u < U; ++u)
                            if (4 * tid + u < ng) cw[4 * tid + u] = word[u];
                    }
                } else {
                    const uint32_t *cw = reinterpret_cast<const uint32_t *>(codes_in + base);
                    vf4 *yv = reinterpret_cast<vf4 *>(y + base);
                    uint32_t w[U];
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        if (tid + u * kBlock < ng) w[u] = cw[tid + u * kBlock];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int q = tid + u * kBlock;
                        if (q >= ng) break;
                        const int o = phase + 4 * q;
                        const int lrow = div_small((uint32_t)o, a.magic);
                        const int b = inner - (o - lrow * inner);
                        const float2 *la = lut + (lr0 + lrow) * a.lut_stride, *lb = la + a.lut_stride;   // this row / the next
                        st16<NT>(yv + q, vf4{decode_one(w[u] & 255u, la, Mi, sign_shift),
                                             decode_one((w[u] >> 8) & 255u, b > 1 ? la : lb, Mi, sign_shift),
                                             decode_one((w[u] >> 16) & 255u, b > 2 ? la : lb, Mi, sign_shift),
                                             decode_one(w[u] >> 24, b > 3 ? la : lb, Mi, sign_shift)});
                    }
                }
            }
            continue;
        }
        for (int i = 0; i < nct; ++i) {
            const int phase = cinfo[i].phase, ng = cinfo[i].len >> 2;
            const int64_t base = (c0 + i * G) * kChunkElems;
            const vf4 *xv = reinterpret_cast<const vf4 *>(x + base);
            vf4 *yv = reinterpret_cast<vf4 *>(y + base);
            const int lr0 = i * a.rpc;
            vf4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (tid + u * kBlock < ng) v[u] = ld16<NT>(xv + tid + u * kBlock);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = tid + u * kBlock;
                if (q >= ng) break;
                const int o = phase + 4 * q;
                const int lrow = div_small((uint32_t)o, a.magic);
                const int b = inner - (o - lrow * inner);   // elements left in this row (>= 1)
                const int lr = lr0 + lrow;
                float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                quant_group<4>(e, lite_of(chl[lr]), lut + lr * a.lut_stride, pmaxf, f.qthr);
                if (b < 4) {   // e[b..3] belong to the next row: take them from its patch
                    const float4 pt = patch[lr];
                    e[3] = b == 3 ? pt.x : (b == 2 ? pt.y : pt.z);
                    if (b < 3) e[2] = b == 2 ? pt.x : pt.y;
                    if (b < 2) e[1] = pt.x;
                }
                st16<NT>(yv + q, vf4{e[0], e[1], e[2], e[3]});
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Short rows, fused, staged: k_rows_staged does what k_rows_flat<1> does but fetches every element ONCE.
// k_rows_flat<1> finds the row ranges with a first pass over global memory and streams the chunk a
// second time; with ~1000 tiles in flight that second read has left L2 (PMC: 1.86x the tensor fetched,
// profiles/r01_pmc_other_kernels.json).  Here a block loads its aligned 4096-element chunk once
// (16 B per lane, coalesced, nontemporal) and parks it in LDS together with the head of the first and
// the tail of the last overlapping row (<= 255 scalars each, the neighbouring chunks' data); the row
// min/max, the boundary patches and the quantize pass all read LDS.  The loop is software-pipelined:
// the next chunk's loads are issued right after the current chunk is parked, so they fly during the
// four barrier phases.
// LDS: float win[kStagePad | 4096 | kStagePad] | float4 patch[rpc] | float4 chanlite[rpc] |
//      float2 lut[rpc * stride]
// patch[r] = the first (4 - start % 4) % 4 elements of row r, quantized: what the 16-byte group shared with row r-1 stores
// ---------------------------------------------------------------------------------------------
constexpr int kStagePad = 256;                                  // >= kFlatFusedMaxInner - 1, multiple of 4
constexpr int kStageWin = kStagePad + kChunkElems + kStagePad;  // floats
constexpr size_t kStageMaxLds = 36 * 1024;                      // dynamic LDS per block: 4 blocks per CU with the 3 KiB of statics
constexpr int kStageGrid = 2048;                                // persistent blocks (FP8Q_STAGED_GRID)
static_assert(kStagePad >= kFlatFusedMaxInner - 1 && kStagePad % 4 == 0, "border rows must fit the pads");

// elo / inner for 0 <= elo < 2^52, 1 <= inner < 2^31 without the ~200-instruction software 64-bit division: the double
// quotient is within 1 of the integer one; an exact integer remainder fixes it up
__device__ __forceinline__ int64_t div_rows(int64_t elo, int inner)
{
    int64_t q = (int64_t)((double)elo / (double)inner);
    int64_t r = elo - q * inner;
    if (r < 0) --q, r += inner;
    if (r >= inner) ++q;
    return q;
}

__device__ __forceinline__ ChunkInfo stage_geometry(int64_t c, const FlatArgs &a)
{
    ChunkInfo ci;
    const int64_t elo = c * kChunkElems;
    const int64_t rem = a.nvec * 4 - elo;
    ci.len = rem < kChunkElems ? (int)rem : kChunkElems;
    ci.tail = (c == a.nchunks - 1) ? a.tail : 0;
    ci.row_lo = div_rows(elo, a.inner);
    ci.phase = (int)(elo - ci.row_lo * a.inner);
    ci.nrows = div_small((uint32_t)(ci.phase + ci.len + ci.tail - 1), a.magic) + 1;
    ci.pad[0] = ci.nrows * a.inner - ci.phase - ci.len;   // elements of the last row behind the body (tail scalars included)
    ci.pad[1] = 0;
    return ci;
}

// The same geometry advanced from chunk c to chunk c + G WITHOUT a 64-bit division: the division of stage_geometry()
// is ~200 instructions of software long division, and with one thread computing it per chunk, ahead of a barrier, it sat
// on every chunk's critical path.  phase + G * 4096 < 2^32 / inner (checked by the caller), so the 32-bit magic division
// is exact; every thread computes the (wave-uniform) result itself: no LDS hand-off, no single-thread section.
__device__ __forceinline__ ChunkInfo stage_geometry_next(const ChunkInfo &cur, int64_t cn, uint32_t adv, const FlatArgs &a)
{
    ChunkInfo ci;
    const int64_t elo = cn * kChunkElems;
    const int64_t rem = a.nvec * 4 - elo;
    ci.len = rem < kChunkElems ? (int)rem : kChunkElems;
    ci.tail = (cn == a.nchunks - 1) ? a.tail : 0;
    const uint32_t t = (uint32_t)cur.phase + adv;
    const uint32_t q = (uint32_t)div_small(t, a.magic);
    ci.row_lo = cur.row_lo + q;
    ci.phase = (int)(t - q * (uint32_t)a.inner);
    ci.nrows = div_small((uint32_t)(ci.phase + ci.len + ci.tail - 1), a.magic) + 1;
    ci.pad[0] = ci.nrows * a.inner - ci.phase - ci.len;
    ci.pad[1] = 0;
    return ci;
}

// ---- pieces shared by k_rows_staged and k_rows_staged_mm (one 4096-element chunk per step, 256 threads) --------------
constexpr int kStageU = 4;   // 16-byte groups per lane and chunk

// the chunk's aligned body: 4 x 16 B per lane, coalesced (needs only the chunk index)
template <bool NT>
__device__ __forceinline__ void stage_load_body(const float *x, int64_t c, const FlatArgs &a, vf4 (&v)[kStageU])
{
    const int64_t elo = c * kChunkElems;
    const int64_t rem = a.nvec * 4 - elo;
    const int ng = (rem < kChunkElems ? (int)rem : kChunkElems) >> 2;
    const vf4 *xv = reinterpret_cast<const vf4 *>(x + elo);
#pragma unroll
    for (int u = 0; u < kStageU; ++u)
        if ((int)threadIdx.x + u * kBlock < ng) v[u] = ld16<NT>(xv + threadIdx.x + u * kBlock);
}

// head of the first and tail of the last overlapping row: <= 255 scalars each, one per thread (needs the geometry)
__device__ __forceinline__ void stage_load_borders(const float *x, int64_t c, const ChunkInfo &ci, float &bh, float &bt)
{
    const int64_t elo = c * kChunkElems;
    const int tid = threadIdx.x;
    if (tid < ci.phase) bh = x[elo - ci.phase + tid];
    if (tid < ci.pad[0]) bt = x[elo + ci.len + tid];
}

// registers -> LDS window: body at [kStagePad, kStagePad + len), the border pieces right before / behind it
__device__ __forceinline__ void stage_park(float *win, const ChunkInfo &ci, const vf4 (&v)[kStageU], float bh, float bt)
{
    const int tid = threadIdx.x, ng = ci.len >> 2;
#pragma unroll
    for (int u = 0; u < kStageU; ++u)
        if (tid + u * kBlock < ng) *reinterpret_cast<vf4 *>(win + kStagePad + 4 * (tid + u * kBlock)) = v[u];
    if (tid < ci.phase) win[kStagePad - ci.phase + tid] = bh;
    if (tid < ci.pad[0]) win[kStagePad + ci.len + tid] = bt;
}

// min / max / NaN of row `wr[0, inner)` in the window over Gl = 2^gs (<= 8) adjacent lanes; every lane gets the result.
// Clamped indices re-read the last element: no remainder loop.  DPP butterflies: no LDS crossbar latency.
__device__ __forceinline__ MinMax stage_row_range(const float *wr, bool valid, int inner, int sub, int gs)
{
    const int Gl = 1 << gs, last = inner - 1;
    MinMax m;
    mm_init(m);
    if (valid) {
        for (int j = sub; j < inner; j += 4 * Gl) {
            const float t0 = wr[j], t1 = wr[min(j + Gl, last)], t2 = wr[min(j + 2 * Gl, last)], t3 = wr[min(j + 3 * Gl, last)];
            mm_acc(m, t0);
            mm_acc(m, t1);
            mm_acc(m, t2);
            mm_acc(m, t3);
        }
    }
    if (gs >= 1) mm_dpp<0xB1>(m);    // quad_perm [1,0,3,2]
    if (gs >= 2) mm_dpp<0x4E>(m);    // quad_perm [2,3,0,1]
    if (gs >= 3) mm_dpp<0x141>(m);   // row_half_mirror
    if (m.nan) m.mn = m.mx = __builtin_nanf("");
    return m;
}

template <bool NT>
__global__ void __launch_bounds__(kBlock, 4)
k_rows_staged(const float *__restrict__ x, float *__restrict__ y, float *row_min, float *row_max,
              float *maxval_out, QFmt f, FlatArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double ftab[kFastTabSize];
    float *win = reinterpret_cast<float *>(smem);
    float4 *patch = reinterpret_cast<float4 *>(win + kStageWin);
    float4 *chl = patch + a.rpc;
    float2 *lut = reinterpret_cast<float2 *>(chl + a.rpc);
    const int tid = threadIdx.x;
    const int inner = a.inner;
    const int64_t G = gridDim.x;
    const float pmaxf = (float)f.pmax;
    constexpr int U = kStageU;
    for (int i = tid; i < kFastTabSize; i += kBlock) ftab[i] = kFastTab[i];

    int64_t c = blockIdx.x;   // gridDim.x <= nchunks
    // chunk geometry lives in registers (wave-uniform), advanced incrementally: see stage_geometry_next()
    const uint32_t adv = (uint32_t)G * (uint32_t)kChunkElems;
    const bool inc_ok = (uint64_t)(G * kChunkElems + 256) * (uint64_t)inner < (1ull << 32);
    ChunkInfo cur = stage_geometry(c, a);
    vf4 v[U];
    float bh = 0.0f, bt = 0.0f;
    stage_load_body<NT>(x, c, a, v);   // prologue: the first chunk's loads
    stage_load_borders(x, c, cur, bh, bt);
    for (;;) {
        const int64_t elo = c * kChunkElems;
        const int phase = cur.phase, nrows = cur.nrows, len = cur.len;
        const int ng = len >> 2;
        stage_park(win, cur, v, bh, bt);
        const int64_t cn = c + G;
        const bool more = cn < a.nchunks;
        ChunkInfo nxt = cur;
        if (more) nxt = inc_ok ? stage_geometry_next(cur, cn, adv, a) : stage_geometry(cn, a);
        __syncthreads();
        if (more) {   // next chunk: in flight during the phases below
            stage_load_body<NT>(x, cn, a, v);
            stage_load_borders(x, cn, nxt, bh, bt);
        }
        {   // per row, Gl (<= 8) lanes: range from LDS -> channel constants -> table -> the row's head patch
            const int gs = a.group, Gl = 1 << gs, rpp = kBlock >> gs, sub = tid & (Gl - 1), rs = tid >> gs;
            const float *w0 = win + (kStagePad - phase);
            for (int rb = 0; rb < nrows; rb += rpp) {
                const int r = rb + rs;
                const bool valid = r < nrows;
                const MinMax m = stage_row_range(w0 + r * inner, valid, inner, sub, gs);   // in every lane of the row
                if (valid) {
                    const float mv = fabsf(tmax(fabsf(m.mn), m.mx));   // fp8_quantizer.py:236
                    if (sub == 0 && (r > 0 || phase == 0)) {   // the row starts in this chunk: this block reports it
                        const int64_t grow = cur.row_lo + r;
                        if (row_min) row_min[grow] = m.mn;
                        if (row_max) row_max[grow] = m.mx;
                        if (maxval_out) maxval_out[grow] = mv;
                    }
                    const Chan ch = make_chan_fast(mv, f, ftab);   // the same in all Gl lanes (lockstep: no extra issue slots)
                    if (sub == 0) chl[r] = make_float4(ch.maxv, ch.minv, ch.bias, ch.pthr);
                    lut_part(lut + r * a.lut_stride, ch, f, sub, Gl);
                }
                // the table was written by this wave's own lanes: DS operations of a wave complete in order
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (valid) {
                    const int idx = r * inner - phase;   // chunk-local index of the row's first element
                    if (idx > 0 && idx < len && (idx & 3)) {   // it shares a 16-byte group with the previous row
                        const ChanLite cl = lite_of(chl[r]);
                        for (int k = sub; k < 4 - (idx & 3); k += Gl)
                            reinterpret_cast<float *>(patch)[4 * r + k] =
                                quant_one(win[kStagePad + idx + k], cl, lut + r * a.lut_stride, pmaxf, f.qthr);
                    }
                }
            }
        }
        __syncthreads();
        if (tid < cur.tail) {   // the tensor's last <= 3 elements
            const int e = len + tid;
            const int r = div_small((uint32_t)(phase + e), a.magic);
            y[elo + e] = quant_one(win[kStagePad + e], lite_of(chl[r]), lut + r * a.lut_stride, pmaxf, f.qthr);
        }
        {
            vf4 *yv = reinterpret_cast<vf4 *>(y + elo);
            vf4 w[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (tid + u * kBlock < ng) w[u] = *reinterpret_cast<const vf4 *>(win + kStagePad + 4 * (tid + u * kBlock));
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = tid + u * kBlock;
                if (q >= ng) break;
                const int o = phase + 4 * q;
                const int lrow = div_small((uint32_t)o, a.magic);
                const int b = inner - (o - lrow * inner);   // elements left in this row (>= 1)
                float e[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
                quant_group<4, false>(e, lite_of(chl[lrow]), lut + lrow * a.lut_stride, pmaxf, f.qthr);   // fused: NaN rows are all-exact
                if (b < 4) {   // e[b..3] belong to the next row: its head patch
                    const float4 pt = patch[lrow + 1];
                    e[3] = b == 3 ? pt.x : (b == 2 ? pt.y : pt.z);
                    if (b < 3) e[2] = b == 2 ? pt.x : pt.y;
                    if (b < 2) e[1] = pt.x;
                }
                st16<NT>(yv + q, vf4{e[0], e[1], e[2], e[3]});
            }
        }
        if (!more) break;
        __syncthreads();   // the window and the tables are rewritten by the next chunk
        c = cn;
        cur = nxt;
    }
}

// K2 twin of k_rows_staged: per-row min/max (+ fold into the running estimate) of rows <= 256 elements at any row
// length and phase.  Loads are the aligned, coalesced 16 KiB chunks of a plain copy (the row-tiled kernel reads
// row-aligned tiles: 5.2-5.4 TB/s); LDS only transposes them for the G-lanes-per-row reduction.  No tables: 18.4 KiB of
// LDS and < 64 VGPRs, 8 blocks per CU.
template <bool NT>
__global__ void __launch_bounds__(kBlock, 8)
k_rows_staged_mm(const float *__restrict__ x, float *row_min, float *row_max, float *maxval_out, FoldArgs fa, FlatArgs a)
{
    __shared__ __attribute__((aligned(16))) float win[kStageWin];
    const int tid = threadIdx.x;
    const int inner = a.inner;
    const int64_t G = gridDim.x;
    int64_t c = blockIdx.x;   // gridDim.x <= nchunks
    const uint32_t adv = (uint32_t)G * (uint32_t)kChunkElems;
    const bool inc_ok = (uint64_t)(G * kChunkElems + 256) * (uint64_t)inner < (1ull << 32);
    ChunkInfo cur = stage_geometry(c, a);   // registers, wave-uniform (no single-thread section, no LDS hand-off)
    vf4 v[kStageU];
    float bh = 0.0f, bt = 0.0f;
    stage_load_body<NT>(x, c, a, v);
    stage_load_borders(x, c, cur, bh, bt);
    for (;;) {
        const int phase = cur.phase, nrows = cur.nrows;
        stage_park(win, cur, v, bh, bt);
        const int64_t cn = c + G;
        const bool more = cn < a.nchunks;
        ChunkInfo nxt = cur;
        if (more) nxt = inc_ok ? stage_geometry_next(cur, cn, adv, a) : stage_geometry(cn, a);
        __syncthreads();
        if (more) {
            stage_load_body<NT>(x, cn, a, v);
            stage_load_borders(x, cn, nxt, bh, bt);
        }
        {
            const int gs = a.group, Gl = 1 << gs, rpp = kBlock >> gs, sub = tid & (Gl - 1), rs = tid >> gs;
            const float *w0 = win + (kStagePad - phase);
            const int64_t row_lo = cur.row_lo;
            for (int rb = 0; rb < nrows; rb += rpp) {
                const int r = rb + rs;
                const bool valid = r < nrows;
                const MinMax m = stage_row_range(w0 + r * inner, valid, inner, sub, gs);
                if (valid && sub == 0 && (r > 0 || phase == 0))   // the chunk in which a row starts reports it
                    fold_store(m.mn, m.mx, row_lo + r, row_min, row_max, maxval_out, fa);
            }
        }
        if (!more) break;
        __syncthreads();   // the window is rewritten by the next chunk
        c = cn;
        cur = nxt;
    }
}

// ---------------------------------------------------------------------------------------------
// Multi-tensor K1: every weight tensor of a model in ONE launch (21 launches of ~7 us each for
// ResNet-18's 11.7 M weights are launch-bound; the data is 47 MB).  One block = one aligned 4096-element
// chunk of one tensor, processed exactly like a k_rows_flat<0> tile with a single chunk; the tensor of a
// block comes from a <= 32-entry table passed by value in the kernel arguments (no device-side table,
// no workspace, no host-to-device copy).
// ---------------------------------------------------------------------------------------------
constexpr int kMultiMax = 32;

struct MultiDesc {
    const float *x;
    float *y;
    const float *maxval;
    int64_t nvec;        // 16-byte groups
    int inner;           // row length; per-tensor entries: the whole tensor is one row (single_row)
    int rpc;
    int tail;
    int single_row;
    uint32_t magic;
    uint32_t chunk0;     // first global chunk id of this tensor
    int n_bits;          // (storage codes: the sign bit's position)
    QFmt f;
};

struct MultiArgs {
    int n;
    int rpc_max;         // most table rows any chunk of any tensor needs
    uint32_t total_chunks;
    MultiDesc d[kMultiMax];
};

// One launch, every block resident at once (<= 1024 blocks: 4 per CU), chunks handed out grid-stride (neighbouring
// blocks on neighbouring chunks, a block's few chunks software-ordered: loads first, tables while they fly).  Round 2's
// version -- one chunk per block, 2850 blocks for ResNet-18 = 2.8 rounds -- spent most of its 24 us in per-block serial
// latency: a single thread's 64-bit software division, one thread per row building a whole table, a head-patch phase
// with dependent global loads, and only then the chunk's own loads.  Here: the chunk's loads are issued before anything
// else; the geometry is computed by every thread (double-precision quotient + fix-up: no LDS hand-off); a row's table
// is built by up to 32 lanes; the <= 3 elements of a 16-byte group that belong to the NEXT row are quantized in place
// with that row's table (a rare divergent branch) instead of a patch phase; the 3 KiB of log2 / exp2 tables are staged
// once per block, not once per chunk.
// Round 4 ablations on ResNet-18's 21 tensors (tools/ab.py multi, 93 MB of traffic; the plain copy of the same bytes:
// 14.2 us): this kernel 20.4 us; with the arithmetic removed (loads, tables, stores only) 16.2; with the tables of a
// block's first chunk reused for its other chunks 19.6 -- i.e. the per-chunk table phase costs ~1 us and the
// quantizer arithmetic ~4.5 us, which adds to the memory time instead of hiding under it: all ~1000 resident blocks
// start together and stay in phase (everybody loads, then everybody computes; 11.7 M elements x ~22 issue slots are
// ~6.5 us of a busy VALU), and at 3 chunks per block the kernel ends before the phases drift apart.  Requesting a
// block's next chunk right before the current chunk's arithmetic (16 more VGPRs: 100) did not change that (20.4 vs
// 20.1 us), nor did 950 / 1280 / 1425 / 2850 blocks (21.8 / 20.4 / 19.7 / 21.5 us).  One table phase for all of a
// block's chunks would remove at most the ~1 us the tables cost.  A fully software-pipelined variant was then written
// and measured (k_multi_flat_pipe, removed again): arithmetic into registers first, the next chunk's data AND the
// maxvals of its table rows requested before that arithmetic (unpredicated, fenced: the scheduler otherwise sinks the
// requests below the stores), descriptor index made provably uniform (readfirstlane) and the ballot key hoisted so
// that no vector load from the kernel-argument segment is left in the loop, one explicit s_waitcnt vmcnt(0) in front
// of the stores -- i.e. NO wait in the loop ever covers a store (gfx950's single in-order vmcnt would otherwise drain a
// chunk's stores before the next table phase; checked in the ISA) -- bit-exact, 117 VGPRs: 19.7 us by rocprofv3 against
// 19.9.  Counters of the plain kernel (rocprofv3 --pmc, per launch): 4.76 M VALU wave-instructions (418 per wave and
// chunk) = ~39 % of the VALU issue slots of a 19.9 us launch, 2.6 M SALU, LDS bank conflicts 1 % of LDS instructions,
// waves waiting 61 % of their cycles.  Neither memory latency, store drains, the table phase nor occupancy (1...3
// chunks per block measured equal) is THE limit; the launch is short enough (3 chunks per block) that its fixed phases
// (launch ramp, table staging, first load round trip, last compute + store drain) make up the gap to the copy.
// Round 5, the last structural attempt (profiles/r05_multi_stagger_ab.txt): blocks started out of phase -- block b waits
// (b % 4) x s x 0.9 us before its first load, so that a quarter of the chip computes while another quarter loads -- measured
// 19.96 / 21.7 / 23.6 / 25.8 / 28.3 us for s = 0 / 1 / 2 / 3 / 4 (rocprofv3, 170 launches each): every step of stagger is
// simply added to the launch, nothing overlaps better.  The phases are not what separates this launch from the copy; with 3
// chunks per block its fixed parts are (launch ramp, table staging, first round trip, last compute + store drain).  Closed.
// The launch replaces 21 launches (130 us from Python).
// MODE 0: K1 (fp32 -> fp32).  MODE 3 / 4 (round 5): the storage codes of N3 for many tensors at once -- encode (fp32 -> 1 byte,
// x = values, y = codes) / decode (1 byte -> fp32, x = codes, y = values): what the bucketed all-gather of channel-sharded
// weights packs into / unpacks from its send buffer in one launch each (fp8q_multi_minmax_encode_u8, fp8q_multi_decode_u8).
// Same chunks, tables and row bookkeeping; a group is 4 elements = one 16-byte load and one 4-byte store or vice versa.
template <int MODE>
__global__ void __launch_bounds__(kBlock, 4)
k_multi_flat(MultiArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double ftab[kFastTabSize];
    const int tid = threadIdx.x;
    float4 *chl = reinterpret_cast<float4 *>(smem);
    float2 *lut = reinterpret_cast<float2 *>(chl + a.rpc_max);
    for (int i = tid; i < kFastTabSize; i += kBlock) ftab[i] = kFastTab[i];
    constexpr int U = 4;
    // which tensor chunk g belongs to: lane l looks at descriptor l's first chunk, one ballot -- a scan over the
    // descriptors is up to 31 DEPENDENT scalar loads from the kernel-argument segment (~2 us for a model's last tensors)
    auto tensor_of = [&](uint32_t g) -> int {
        const int l = tid & 63;
        const uint32_t c0 = l < a.n ? a.d[l].chunk0 : 0xffffffffu;
        return __popcll(__ballot(c0 <= g)) - 1;   // chunk0 ascends from 0: uniform, >= 0
    };
    auto issue = [&](uint32_t g, int t, vf4 (&w)[U], uint32_t (&wc)[U]) {   // the chunk's groups of 4 elements: 4 per lane
        const MultiDesc &d = a.d[t];
        const int64_t elo = (int64_t)(g - d.chunk0) * kChunkElems;
        const int64_t rem = d.nvec * 4 - elo;
        const int ng = (rem < kChunkElems ? (int)rem : kChunkElems) >> 2;
        if (MODE == 4) {
            const uint32_t *xc = reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(d.x) + elo);
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (tid + u * kBlock < ng) wc[u] = xc[tid + u * kBlock];
        } else {
            const vf4 *xv = reinterpret_cast<const vf4 *>(d.x + elo);
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (tid + u * kBlock < ng) w[u] = ld16<false>(xv + tid + u * kBlock);
        }
    };
    for (uint32_t g = blockIdx.x; g < a.total_chunks; g += gridDim.x) {
        const int t = tensor_of(g);
        vf4 v[U];
        uint32_t vc[U];
        issue(g, t, v, vc);   // in flight during the table phase (requesting the block's NEXT chunk here as well measured slower:
                          // 21.1 vs 19.1 us for ResNet-18's 21 tensors)
        const MultiDesc &d = a.d[t];
        const QFmt f = d.f;
        const int inner = d.inner, lut_stride = f.pmax + 1;
        const float pmaxf = (float)f.pmax;
        const int64_t elo = (int64_t)(g - d.chunk0) * kChunkElems;
        const float *x = d.x + elo;                                                       // (MODE 4: codes, see xb)
        float *y = d.y + elo;                                                             // (MODE 3: codes, see yb)
        const uint8_t *xb = reinterpret_cast<const uint8_t *>(d.x) + elo;
        uint8_t *yb = reinterpret_cast<uint8_t *>(d.y) + elo;
        const int Mi = (int)f.M, sign_shift = f.sign_bits == 1 ? d.n_bits - 1 : -1;
        const int64_t rem = d.nvec * 4 - elo;
        const int len = rem < kChunkElems ? (int)rem : kChunkElems;
        const int ng = len >> 2;
        const int tail = (rem <= kChunkElems) ? d.tail : 0;
        const int64_t row_lo = d.single_row ? 0 : div_rows(elo, inner);
        const int phase = d.single_row ? 0 : (int)(elo - row_lo * inner);
        const int nrows = d.single_row ? 1 : div_small((uint32_t)(phase + len + tail - 1), d.magic) + 1;
        __syncthreads();   // the previous chunk's tables are no longer read (first chunk: ftab is staged)
        {
            int gs = 0;   // log2(lanes per row): as many as hold all rows in one pass, at most 32
            while (gs < 5 && (nrows << (gs + 1)) <= kBlock) ++gs;
            const int L = 1 << gs, sub = tid & (L - 1);
            for (int r = tid >> gs; r < nrows; r += kBlock >> gs) {
                const Chan ch = make_chan_fast(d.maxval[d.single_row ? 0 : row_lo + r], f, ftab);
                if (sub == 0) chl[r] = make_float4(ch.maxv, ch.minv, ch.bias, ch.pthr);
                lut_part(lut + r * lut_stride, ch, f, sub, L);
            }
        }
        __syncthreads();
        if (tid < tail) {   // the tensor's last <= 3 elements
            const int e = len + tid;
            const int r = d.single_row ? 0 : div_small((uint32_t)(phase + e), d.magic);
            if (MODE == 3)
                yb[e] = (uint8_t)encode_one(x[e], lite_of(chl[r]), lut + r * lut_stride, pmaxf, f.qthr, Mi, sign_shift);
            else if (MODE == 4)
                y[e] = decode_one(xb[e], lut + r * lut_stride, Mi, sign_shift);
            else
                y[e] = quant_one(x[e], lite_of(chl[r]), lut + r * lut_stride, pmaxf, f.qthr);
        }
        vf4 *yv = reinterpret_cast<vf4 *>(y);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = tid + u * kBlock;
            if (q >= ng) break;
            const int o = phase + 4 * q;
            const int lrow = d.single_row ? 0 : div_small((uint32_t)o, d.magic);
            const int b = d.single_row ? 4 : inner - (o - lrow * inner);   // elements left in this row (>= 1)
            if (MODE == 4) {   // b..3 of the group are the next row's: its table
                const float2 *la = lut + lrow * lut_stride, *lb = la + lut_stride;
                const uint32_t w = vc[u];
                st16<false>(yv + q, vf4{decode_one(w & 255u, la, Mi, sign_shift), decode_one((w >> 8) & 255u, b > 1 ? la : lb, Mi, sign_shift),
                                        decode_one((w >> 16) & 255u, b > 2 ? la : lb, Mi, sign_shift),
                                        decode_one(w >> 24, b > 3 ? la : lb, Mi, sign_shift)});
                continue;
            }
            const float in[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            if (MODE == 3) {
                uint32_t wd = encode_group4(in, lite_of(chl[lrow]), lut + lrow * lut_stride, pmaxf, f.qthr, Mi, sign_shift);
                if (b < 4) {
                    const ChanLite cl = lite_of(chl[lrow + 1]);
                    const float2 *lt = lut + (lrow + 1) * lut_stride;
#pragma unroll
                    for (int k = 1; k < 4; ++k)
                        if (k >= b) wd = (wd & ~(255u << (8 * k))) | (encode_one(in[k], cl, lt, pmaxf, f.qthr, Mi, sign_shift) << (8 * k));
                }
                reinterpret_cast<uint32_t *>(yb)[q] = wd;
                continue;
            }
            float e[4] = {in[0], in[1], in[2], in[3]};
            quant_group<4>(e, lite_of(chl[lrow]), lut + lrow * lut_stride, pmaxf, f.qthr);
            if (b < 4) {   // e[b..3] belong to the next row (rows are >= 4 long: one boundary per group at most)
                const ChanLite cl = lite_of(chl[lrow + 1]);
                const float2 *lt = lut + (lrow + 1) * lut_stride;
#pragma unroll
                for (int k = 1; k < 4; ++k)
                    if (k >= b) e[k] = quant_one(in[k], cl, lt, pmaxf, f.qthr);
            }
            st16<false>(yv + q, vf4{e[0], e[1], e[2], e[3]});
        }
    }
}

// Per-channel ranges of MANY tensors in one launch: the estimate-state twin of k_multi_flat.  A model's weight tensors in
// estimate_ranges state (current_minmax, set_maxval: quantization_manager.py:114-122 per layer, i.e. one fused launch per
// layer = 21 launches of ~4 us for ResNet-18, or ~23 us each when driven from Python -- launch-bound either way) need
// every row's min / max before anything can be quantized.  Here one wave owns one row (rows of these tensors are
// 4 ... 16384 elements: 64 B ... 64 KiB), rows of all tensors are numbered consecutively, and the result
// maxval[c] = |max(|min_c|, max_c)| (fp8_quantizer.py:236) goes where k_multi_flat will read it: the two launches
// together are fp8q_multi_minmax_quantize_f32.  Dword loads (rows start at any 4-byte phase), coalesced per wave.
struct RowsDesc {
    const float *x;
    float *maxval;     // [C] output
    float *row_min;    // [C] output or nullptr
    float *row_max;
    int inner;
    uint32_t row0;     // first global row id of this tensor
};

struct RowsArgs {
    int n;
    uint32_t total_rows;
    RowsDesc d[kMultiMax];
};

__global__ void __launch_bounds__(kBlock)
k_multi_rowmax(RowsArgs a)
{
    const int lane = threadIdx.x & 63;
    const uint32_t row = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (row >= a.total_rows) return;   // whole wave
    const uint32_t r0 = lane < a.n ? a.d[lane].row0 : 0xffffffffu;
    const int t = __popcll(__ballot(r0 <= row)) - 1;   // row0 ascends from 0
    const RowsDesc &d = a.d[t];
    const int64_t c = row - d.row0;
    const float *xr = d.x + c * d.inner;
    MinMax m;
    mm_init(m);
    int i = lane;
    for (; i + 192 < d.inner; i += 256) {   // four loads in flight per lane
        const float v0 = xr[i], v1 = xr[i + 64], v2 = xr[i + 128], v3 = xr[i + 192];
        mm_acc(m, v0);
        mm_acc(m, v1);
        mm_acc(m, v2);
        mm_acc(m, v3);
    }
    for (; i < d.inner; i += 64) mm_acc(m, xr[i]);
    mm_wave_reduce(m);
    if (lane == 0) {
        float mn = m.mn, mx = m.mx;
        if (m.nan) mn = mx = __builtin_nanf("");
        if (d.row_min) d.row_min[c] = mn;
        if (d.row_max) d.row_max[c] = mx;
        d.maxval[c] = fabsf(tmax(fabsf(mn), mx));
    }
}

// ---------------------------------------------------------------------------------------------
// Fused K2+K5+K1 for rows of 257..8192 elements (a multiple of 4, 16-byte aligned): the row stays in
// REGISTERS between the min/max pass and the quantize pass, so the tensor is read once (8 B/element of
// HBM traffic for real).  L lanes per row: 16 or 32 (16 / 8 rows per block), 64 (one wave per row) or 256
// (the whole block per row); the host picks the L whose lanes are best filled.  Rows are handed out grid-stride, so concurrently running blocks
// work on neighbouring rows -- the access pattern of a copy.  Per row: EPT x 16 B per lane in flight,
// wave (+ LDS) min/max reduction, the row's {s, 1/s} table written by its own lanes, quantize, store.
// ---------------------------------------------------------------------------------------------
template <int L, int EPT, bool NT, bool QUANT>
__global__ void __launch_bounds__(kBlock)
k_rows_reg(const float *__restrict__ x, float *__restrict__ y, int64_t C, int inner, float *row_min,
           float *row_max, float *maxval_out, QFmt f, FoldArgs fa)
{
    constexpr int RPB = kBlock / L;      // rows per block and step (L = lanes per row: 16, 32, 64 or 256)
    __shared__ float2 lut[RPB][kLutMax];
    __shared__ double ftab[kFastTabSize];
    __shared__ float s_mn[4], s_mx[4];
    __shared__ int s_nan[4];
    const int tid = threadIdx.x, sub = tid % L, rslot = tid / L, wave = tid >> 6;
    const int nvec = inner >> 2, rem = inner & 3;   // rem != 0 only for K2
    const float pmaxf = (float)f.pmax;
    if (QUANT) {
        for (int i = tid; i < kFastTabSize; i += kBlock) ftab[i] = kFastTab[i];
        __syncthreads();
    }
    for (int64_t r0 = (int64_t)blockIdx.x * RPB; r0 < C; r0 += (int64_t)gridDim.x * RPB) {
        const int64_t row = r0 + rslot;
        const bool valid = row < C;
        const float *xr = x + (valid ? row : 0) * inner;
        vf4 v[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            v[k] = vf4{0.0f, 0.0f, 0.0f, 0.0f};   // a group beyond the row: quantizes to 0 on the fast path, never stored
            const int idx = k * L + sub;
            if (valid && idx < nvec) {
                v[k] = ld16u<NT>(xr + 4 * idx);   // rows may start at any 4-byte phase (K2); one dwordx4 either way
            } else if (!QUANT && valid && idx == nvec && rem) {   // K2 only: the row's last 1..3 elements
                v[k].x = xr[4 * idx];
                if (rem > 1) v[k].y = xr[4 * idx + 1];
                if (rem > 2) v[k].z = xr[4 * idx + 2];
            }
        }
        MinMax m;
        mm_init(m);
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int idx = k * L + sub;
            if (valid && idx < nvec) {
                mm_acc(m, v[k].x);
                mm_acc(m, v[k].y);
                mm_acc(m, v[k].z);
                mm_acc(m, v[k].w);
            } else if (!QUANT && valid && idx == nvec && rem) {
                mm_acc(m, v[k].x);
                if (rem > 1) mm_acc(m, v[k].y);
                if (rem > 2) mm_acc(m, v[k].z);
            }
        }
#pragma unroll
        for (int off = (L < 64 ? L : 64) >> 1; off >= 1; off >>= 1) {
            m.mn = fminf(m.mn, __shfl_xor(m.mn, off, 64));
            m.mx = fmaxf(m.mx, __shfl_xor(m.mx, off, 64));
            m.nan |= __shfl_xor(m.nan, off, 64);
        }
        if (L == 256) {
            if ((tid & 63) == 0) {
                s_mn[wave] = m.mn;
                s_mx[wave] = m.mx;
                s_nan[wave] = m.nan;
            }
            __syncthreads();
            m.mn = fminf(fminf(s_mn[0], s_mn[1]), fminf(s_mn[2], s_mn[3]));
            m.mx = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
            m.nan = s_nan[0] | s_nan[1] | s_nan[2] | s_nan[3];
        }
        if (m.nan) m.mn = m.mx = __builtin_nanf("");
        if (!QUANT) {   // K2: fold into the running estimate and go on (no tables, no stores)
            if (valid && sub == 0) fold_store(m.mn, m.mx, row, row_min, row_max, maxval_out, fa);
            if (L == 256) __syncthreads();   // s_mn / s_mx are rewritten by the next step
            continue;
        }
        const float mv = fabsf(tmax(fabsf(m.mn), m.mx));   // fp8_quantizer.py:236
        if (valid && sub == 0) {
            if (row_min) row_min[row] = m.mn;
            if (row_max) row_max[row] = m.mx;
            if (maxval_out) maxval_out[row] = mv;
        }
        const Chan c = make_chan_fast(mv, f, ftab);
        for (int p = sub; p <= f.pmax; p += L) lut[rslot][p] = lut_entry(c, p, f.M);
        __syncthreads();
        {
            // one branch for all groups of the lane (missing groups hold zeros: no rare-case work)
            const ChanLite cl = lite(c);
            vf4 *yv = reinterpret_cast<vf4 *>(y + (valid ? row : 0) * inner);
            float e[EPT * 4];
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                e[4 * k] = v[k].x;
                e[4 * k + 1] = v[k].y;
                e[4 * k + 2] = v[k].z;
                e[4 * k + 3] = v[k].w;
            }
            quant_group<EPT * 4, false>(e, cl, lut[rslot], pmaxf, f.qthr);   // fused: a NaN makes the row's range NaN -> all-exact
#pragma unroll
            for (int k = 0; k < EPT; ++k)
                if (valid && k * L + sub < nvec)
                    st16<NT>(yv + k * L + sub, vf4{e[4 * k], e[4 * k + 1], e[4 * k + 2], e[4 * k + 3]});
        }
        __syncthreads();   // the tables are rewritten by the next step
    }
}

// K1 scalar fallback (x / y not 16-byte co-aligned): one row per blockIdx.y, dword accesses
__device__ __forceinline__ void quant_scalar_body(const float *__restrict__ x, float *__restrict__ y, int64_t inner,
                                                  const float *__restrict__ maxval, int per_channel, const QFmt &f)
{
    __shared__ float2 lut[kLutMax];
    const int row = blockIdx.y;
    const Chan cfull = make_chan(maxval[per_channel ? row : 0], f);
    for (int i = threadIdx.x; i <= f.pmax; i += kBlock) lut[i] = lut_entry(cfull, i, f.M);
    __syncthreads();
    const ChanLite c = lite(cfull);
    const float *xr = x + (int64_t)row * inner;
    float *yr = y + (int64_t)row * inner;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < inner;
         i += (int64_t)gridDim.x * kBlock)
        yr[i] = quant_one(xr[i], c, lut, (float)f.pmax, f.qthr);
}

__global__ void __launch_bounds__(kBlock)
k_quant_scalar(const float *__restrict__ x, float *__restrict__ y, int64_t inner,
               const float *__restrict__ maxval, int per_channel, QFmt f)
{
    quant_scalar_body(x, y, inner, maxval, per_channel, f);
}

// BASELINE config 2 at its literal size (conv1 [64, 3, 7, 7]: 37 KB) and every other weight tensor that small: the launch is
// all latency.  k_rows_direct makes two passes (row min/max from global, tables, then the rows again as one flat range) around
// two workgroup barriers and builds R tables per workgroup in one thread each: 8.5 us for a tensor whose launch floor is ~4.
// Here a WAVE owns a row for the whole kernel: the row sits in registers (EPL elements per lane), min / max by wave shuffles,
// the channel constants once per wave, the {s, 1/s} table by the wave's 64 lanes into its own slice of LDS, the quantized row
// straight from the registers -- one pass, no workgroup barrier.  Same per-element arithmetic (quant_one) and the same min / max
// semantics (mm_acc, NaN flag) as the other fused routes: bit-identical results.
template <int EPL>
__global__ void __launch_bounds__(kBlock)
k_small_rows_fused(const float *__restrict__ x, float *__restrict__ y, int64_t C, int inner, float *row_min, float *row_max,
                   float *maxval_out, QFmt f)
{
    __shared__ float2 lut[kBlock / 64][kLutMax];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * (kBlock / 64) + wave;
    if (row >= C) return;                       // (no workgroup barrier below)
    const float *xr = x + row * inner;
    float *yr = y + row * inner;
    float v[EPL];
    MinMax m;
    mm_init(m);
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int i = lane + 64 * e;
        v[e] = 0.0f;
        if (i < inner) {
            v[e] = xr[i];
            mm_acc(m, v[e]);
        }
    }
    mm_wave_reduce(m);
    if (m.nan) m.mn = m.mx = __builtin_nanf("");
    const float mv = fabsf(tmax(fabsf(m.mn), m.mx));   // fp8_quantizer.py:236
    if (lane == 0) {
        if (row_min) row_min[row] = m.mn;
        if (row_max) row_max[row] = m.mx;
        if (maxval_out) maxval_out[row] = mv;
    }
    const Chan cfull = make_chan(mv, f);
    lut_part(lut[wave], cfull, f, lane, 64);
    __builtin_amdgcn_wave_barrier();
    const ChanLite c = lite(cfull);
    const float pmaxf = (float)f.pmax;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int i = lane + 64 * e;
        if (i < inner) yr[i] = quant_one(v[e], c, lut[wave], pmaxf, f.qthr);
    }
}

constexpr int64_t kSmallFusedElems = 16384;    // tensors up to 64 KB ...
constexpr int kSmallFusedInner = 512;          // ... of rows up to 8 elements per lane

// K1 with the mantissa width read from DEVICE memory (fp8q_quantize_dm_f32): the MSE estimator's plurality vote on the
// mantissa bits (range_estimators.py:350-354) stays on the GPU, and the batch that follows it in the same calibration
// forward is quantized with the winner without a host round trip.  The host cannot know the format, so it passes the
// constants of every width the call admits (M = 1 .. n_bits - sign_bits); calibration-only path: one row per
// blockIdx.y whatever the row length (the by-value kernels keep their tuned routes).
struct FmtSel {
    const float *mbits_dev;
    const unsigned char *signed_dev;   // fp8q_quantize_ds_f32: the SIGN in device memory instead (1: tab[0], 0: tab[1]), else null
    int hi;           // n_bits - sign_bits
    QFmt tab[8];      // tab[M - 1]
};

__device__ __forceinline__ QFmt pick_fmt(const FmtSel &s)
{
    if (s.signed_dev) return s.tab[*s.signed_dev ? 0 : 1];
    float M = rintf(*s.mbits_dev);                       // torch.round: half to even (fp8_quantizer.py:105)
    M = fminf(fmaxf(M, 1.0f), (float)s.hi);              // NaN -> 1 (the by-value entry point refuses NaN on the host)
    return s.tab[(int)M - 1];
}

// width AND sign in device memory (fp8q_quantize_dms_f32: an MSE estimator's vote next to the flag of fp8q_sign_fold_u8)
struct FmtSel2 {
    const float *mbits_dev;
    const unsigned char *signed_dev;
    int hi[2];          // n_bits - sign_bits for sign_bits = 1, 0
    QFmt tab[2][8];     // tab[1 - sign_bits][M - 1]
};

__device__ __forceinline__ QFmt pick_fmt(const FmtSel2 &s)
{
    const int u = *s.signed_dev ? 0 : 1;
    float M = rintf(*s.mbits_dev);
    M = fminf(fmaxf(M, 1.0f), (float)s.hi[u]);
    return s.tab[u][(int)M - 1];
}

template <bool NT, int U, class SEL>
__global__ void __launch_bounds__(kBlock)
k_quant_rows_dm(const float *__restrict__ x, float *__restrict__ y, int64_t inner, const float *__restrict__ maxval,
                int per_channel, SEL sel)
{
    const QFmt f = pick_fmt(sel);
    quant_rows_body<NT, U>(x, y, inner, maxval, per_channel, f);
}

// K1 of a per-tensor quantizer whose winner -- mantissa width and clipping value -- is still the MSE table of the search that ran
// just before it (fp8q_mse_calibrate_f32 with the mantissa search): every workgroup takes the selection for itself -- a few
// loads per thread from L2 (n_m x n_cand <= 8 x 128 entries), the reference's two-level vote (range_estimators.py:350-369 with
// one channel) as ONE arg-min over the table in (width, candidate) order: the smallest entry, the first of equals, a NaN before
// everything -- and workgroup 0 writes the estimator's outputs.  The kernel boundary makes the table visible: no tickets, no
// agent-scope traffic in the launch that finishes the table (k_mse_eval with the selection appended: 20.4 us per MobileNetV2
// activation with six widths, 12.5 without).
struct SelIn {
    const float *mses, *grid;
    int n_m, n_cand;
    SelOne so;
};

template <bool NT, int U>
__global__ void __launch_bounds__(kBlock)
k_quant_rows_sel(const float *__restrict__ x, float *__restrict__ y, int64_t inner, FmtSel sel, SelIn si)
{
    __shared__ float s_v[kBlock / 64], s_g[kBlock / 64];
    __shared__ int s_i[kBlock / 64];
    __shared__ float s_mv;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int total = si.n_m * si.n_cand;
    ArgMin a = {__builtin_inff(), 0x7fffffff};
    float ag = 0.0f;
    for (int i0 = 0; i0 < total; i0 += 4 * kBlock) {
        float t[4], g[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {                 // (all loads of a trip before the first use)
            const int i = i0 + q * kBlock + tid;
            t[q] = i < total ? si.mses[i] : 0.0f;
            g[q] = i < total ? si.grid[i % si.n_cand] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + q * kBlock + tid;
            const ArgMin o = {t[q], i};
            if (i < total && argmin_less(o, a)) {
                a = o;
                ag = g[q];
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        ArgMin o;
        o.v = __shfl_xor(a.v, off, 64);
        o.idx = __shfl_xor(a.idx, off, 64);
        const float og = __shfl_xor(ag, off, 64);
        if (argmin_less(o, a)) {
            a = o;
            ag = og;
        }
    }
    if (lane == 0) {
        s_v[wave] = a.v;
        s_i[wave] = a.idx;
        s_g[wave] = ag;
    }
    __syncthreads();
    a = ArgMin{s_v[0], s_i[0]};
    ag = s_g[0];
#pragma unroll
    for (int w = 1; w < kBlock / 64; ++w) {
        const ArgMin o = {s_v[w], s_i[w]};
        if (argmin_less(o, a)) {
            a = o;
            ag = s_g[w];
        }
    }
    const int vote = a.idx / si.n_cand;
    const float mb = si.so.M[vote];
    if (tid == 0) {
        s_mv = ag;
        if (blockIdx.x == 0 && blockIdx.y == 0) {
            si.so.mbits_out[0] = mb;
            if (si.so.vote_out) si.so.vote_out[0] = vote;
            si.so.maxval_out[0] = ag;
            if (si.so.xmin_out) si.so.xmin_out[0] = si.so.sign * ag;      // sign_bits * -1.0 * maxval (:369)
        }
    }
    __syncthreads();
    float M = rintf(mb);
    M = fminf(fmaxf(M, 1.0f), (float)sel.hi);
    const QFmt f = sel.tab[(int)M - 1];
    quant_rows_body<NT, U>(x, y, inner, &s_mv, 0, f);
}

// Short per-channel rows with the width in device memory (MobileNetV2's weights in the mantissa search: [1280, 320],
// [96, 1, 3, 3] ...): a WAVE per row -- its channel constants and {s, 1/s} table built once per wave in the wave's own slice of
// LDS, the row streamed by its 64 lanes -- instead of a 256-thread workgroup (and a ~50-operation double-precision set-up) per
// row of a few hundred elements.  Same arithmetic as quant_rows_body (quant_one), bit for bit.
template <class SEL>
__global__ void __launch_bounds__(kBlock)
k_quant_short_rows_dm(const float *__restrict__ x, float *__restrict__ y, int64_t C, int inner, const float *__restrict__ maxval,
                      SEL sel)
{
    __shared__ float2 lut[kBlock / 64][kLutMax];
    const QFmt f = pick_fmt(sel);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float pmaxf = (float)f.pmax;
    for (int64_t row = (int64_t)blockIdx.x * (kBlock / 64) + wave; row < C; row += (int64_t)gridDim.x * (kBlock / 64)) {
        const Chan cfull = make_chan(maxval[row], f);
        for (int i = lane; i <= f.pmax; i += 64) lut[wave][i] = lut_entry(cfull, i, f.M);
        __builtin_amdgcn_wave_barrier();
        const ChanLite c = lite(cfull);
        const float *xr = x + row * inner;
        float *yr = y + row * inner;
        for (int i = lane; i < inner; i += 64) yr[i] = quant_one(xr[i], c, lut[wave], pmaxf, f.qthr);
        __builtin_amdgcn_wave_barrier();       // (the next row's table overwrites this one)
    }
}

template <class SEL>
__global__ void __launch_bounds__(kBlock)
k_quant_scalar_dm(const float *__restrict__ x, float *__restrict__ y, int64_t inner, const float *__restrict__ maxval,
                  int per_channel, SEL sel)
{
    const QFmt f = pick_fmt(sel);
    quant_scalar_body(x, y, inner, maxval, per_channel, f);
}

// ---------------------------------------------------------------------------------------------
// K2/K3: min / max / NaN of x[row, split range], published as tagged granules; block nsplit of a row is the row's
// reducer (block_minmax_publish / block_minmax_collect, fp8q_common.h): one launch.
// ---------------------------------------------------------------------------------------------
template <bool NT>
__global__ void __launch_bounds__(kBlock)
k_minmax_partial(const float *__restrict__ x, int64_t inner, int nsplit, unsigned long long *slots, unsigned tag,
                 float *cur_min, float *cur_max, float *maxval_out, FoldArgs fa)
{
    if ((int)blockIdx.x == nsplit) {   // only launched when nsplit > 1
        block_minmax_collect(slots + (int64_t)blockIdx.y * nsplit * 2, nsplit, tag, blockIdx.y, cur_min, cur_max,
                             maxval_out, fa);
        return;
    }
    const int row = blockIdx.y, split = blockIdx.x, tid = threadIdx.x;
    const float *xr = x + (int64_t)row * inner;
    MinMax m;
    mm_init(m);
    int64_t head = ((16 - ((uintptr_t)xr & 15)) & 15) >> 2;
    if (head > inner) head = inner;
    const int64_t nvec = (inner - head) >> 2;
    const int64_t tail0 = head + (nvec << 2);
    if (split == 0) {
        if (tid < head) mm_acc(m, xr[tid]);
        if (tail0 + tid < inner) mm_acc(m, xr[tail0 + tid]);
    }
    const vf4 *xv = reinterpret_cast<const vf4 *>(xr + head);
    constexpr int U = 8;
    const int64_t step = (int64_t)nsplit * (kBlock * U);
    for (int64_t base = (int64_t)split * (kBlock * U); base < nvec; base += step) {
        if (base + kBlock * U <= nvec) {
            vf4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = ld16<NT>(xv + base + u * kBlock + tid);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                mm_acc(m, v[u].x);
                mm_acc(m, v[u].y);
                mm_acc(m, v[u].z);
                mm_acc(m, v[u].w);
            }
        } else {
            for (int u = 0; u < U; ++u) {
                const int64_t i = base + u * kBlock + tid;
                if (i < nvec) {
                    const vf4 w = ld16<NT>(xv + i);
                    mm_acc(m, w.x);
                    mm_acc(m, w.y);
                    mm_acc(m, w.z);
                    mm_acc(m, w.w);
                }
            }
        }
    }
    block_minmax_publish(m, slots + (int64_t)row * nsplit * 2, split, nsplit, tag, row, cur_min, cur_max, maxval_out, fa);
}

// After the all-reduce(MAX) of the packed ranges of batch-sharded calibration (FoldArgs::packed, fold_store): back to
// {min, max} (+ K5: maxval = |max(|min|, max)|, fp8_quantizer.py:236) -- one launch instead of ~8 tiny tensor ops.
__global__ void __launch_bounds__(kBlock)
k_ranges_unpack(const float *__restrict__ packed, int64_t n, float *cur_min, float *cur_max, float *maxval_out)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 p = reinterpret_cast<const float4 *>(packed)[i];
    const float nan = __builtin_nanf("");
    const float mn = p.z > 0.0f ? nan : -p.x, mx = p.w > 0.0f ? nan : p.y;
    if (cur_min) cur_min[i] = mn;
    if (cur_max) cur_max[i] = mx;
    if (maxval_out) maxval_out[i] = fabsf(tmax(fabsf(mn), mx));
}

// 16-byte-per-lane copy with K1's launch shape: the achievable-HBM yardstick
template <bool NT>
__global__ void __launch_bounds__(kBlock)
k_copy(const vf4 *__restrict__ x, vf4 *__restrict__ y, int64_t nvec)
{
    const int tid = threadIdx.x;
    const int64_t step = (int64_t)gridDim.x * (kBlock * kUnroll);
    for (int64_t base = (int64_t)blockIdx.x * (kBlock * kUnroll); base < nvec; base += step) {
        if (base + kBlock * kUnroll <= nvec) {
            vf4 v[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) v[u] = ld16<NT>(x + base + u * kBlock + tid);
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) st16<NT>(y + base + u * kBlock + tid, v[u]);
        } else {
            for (int u = 0; u < kUnroll; ++u) {
                const int64_t i = base + u * kBlock + tid;
                if (i < nvec) st16<NT>(y + i, ld16<NT>(x + i));
            }
        }
    }
}

// K1 / K2: rows up to this length take k_rows_direct, longer ones the 2-D row kernels (measured
// cross-over: [58254,4608] K1 6.3 TB/s with k_quant_rows vs 5.5 with k_rows_direct).  The fused
// K2+K5+K1 path uses k_rows_direct up to kDirectMaxInner (5.1 TB/s at 4608 vs 3.9 two-pass).
int64_t direct_max_inner()
{
    static const int v = [] {
        const char *e = getenv("FP8Q_DIRECT_MAX_INNER");   // tuning knob for experiments
        const int n = e ? atoi(e) : 0;
        return n >= 4 && n <= kDirectMaxInner ? n : 2047;
    }();
    return v;
}

// Launch k_rows_flat if the problem fits it; returns -1000 when the caller must use k_rows_direct
// (pointers not 16-byte aligned, rows too short for per-row tables in LDS, in-place fused, ...).
constexpr int kNotFlat = -1000;
int launch_rows_flat(int mode, const float *x, float *y, int64_t C, int64_t inner, const float *maxval,
                     float *row_min, float *row_max, float *maxval_out, const QFmt &f, hipStream_t st)
{
    static const int flat_env = [] {   // FP8Q_FLAT=0: round-1 row-tiled kernel everywhere (A/B)
        const char *e = getenv("FP8Q_FLAT");
        return e ? atoi(e) : 1;
    }();
    if (!flat_env || mode == kModeMinMax || inner < 4) return kNotFlat;
    if ((((uintptr_t)x | (uintptr_t)y) & 15) != 0) return kNotFlat;
    if (mode == kModeFused && (inner > kFlatFusedMaxInner || x == y)) return kNotFlat;
    FlatArgs a = {};
    a.inner = (int)inner;
    a.lut_stride = f.pmax + 1;
    a.magic = magic_of((int)inner);
    const int64_t n = C * inner;
    a.nvec = n >> 2;
    a.tail = (int)(n & 3);
    a.nchunks = cdiv(a.nvec, kChunkGroups);
    a.rpc = (int)((inner + (kChunkElems + 3) - 2) / inner) + 1;
    a.rmagic = magic_of(a.rpc);
    const int64_t per_row = 16 + 16 + (int64_t)a.lut_stride * 8 + (mode == kModeFused ? 4 : 0);
    static const int lds_kb_env = [] {
        const char *e = getenv("FP8Q_FLAT_LDS_KB");
        const int v = e ? atoi(e) : 0;
        return v >= 4 && v <= 120 ? v : 36;
    }();
    const int64_t cap = (int64_t)lds_kb_env * 1024 - (int64_t)kFlatMaxCh * sizeof(ChunkInfo);
    int64_t nch = cap / (a.rpc * per_row);
    if (nch < 1) return kNotFlat;
    static const int nch_env = [] {
        const char *e = getenv("FP8Q_FLAT_NCH");
        const int v = e ? atoi(e) : 0;
        // default 4: measured on K1 (tools/mb_flat_nch.py, one box, rows of 147 / 288 / 576 / 1152 / 2047 elements):
        // 4 chunks per tile 5.69 / 5.86 / 5.88 / 5.84 / 5.76 TB/s, 8 chunks 5.66 / 5.66 / 5.75 / 5.74 / 5.66, 2 chunks
        // 5.22 / 5.90 / 5.92 / 5.91 / 5.63, 1 chunk 4.27 / 5.44 / 5.44 / 5.41 / 4.62
        return v >= 1 && v <= kFlatMaxCh ? v : 4;
    }();
    if (nch > nch_env) nch = nch_env;
    const bool nt = n * 4 >= kNtBytes;
    if (mode == kModeFused) {   // one fetch per element: k_rows_staged, if window + tables leave room for 4 blocks per CU
        static const int staged_env = [] {   // FP8Q_STAGED=0: two-pass k_rows_flat<1> (A/B)
            const char *e = getenv("FP8Q_STAGED");
            return e ? atoi(e) : 1;
        }();
        static const int staged_grid = [] {   // persistent grid cap; 0 = one chunk per block
            const char *e = getenv("FP8Q_STAGED_GRID");
            const int v = e ? atoi(e) : -1;
            return v >= 0 ? v : kStageGrid;
        }();
        const size_t sh = (size_t)kStageWin * sizeof(float) + (size_t)a.rpc * per_row;
        if (staged_env && sh <= kStageMaxLds) {
            int gs = 0;
            while (gs < 6 && (2 << gs) * a.rpc <= kBlock) ++gs;
            a.group = gs;   // log2(lanes per row) here
            a.nch = 1;
            const int64_t blocks = staged_grid ? balanced_blocks(a.nchunks, staged_grid) : a.nchunks;
            const dim3 g((unsigned)blocks), b(kBlock);
            if (nt) hipLaunchKernelGGL((k_rows_staged<true>), g, b, sh, st, x, y, row_min, row_max, maxval_out, f, a);
            else hipLaunchKernelGGL((k_rows_staged<false>), g, b, sh, st, x, y, row_min, row_max, maxval_out, f, a);
            return launch_rc();
        }
    }
    while (nch > 1 && cdiv(a.nchunks, nch) < 1024) --nch;   // small tensors: more blocks, not longer tiles
    a.nch = (int)nch;
    int G = 1;
    while (G < 64 && (int64_t)G * 24 < inner) G <<= 1;
    a.group = G;
    static const int grid_env = [] {
        const char *e = getenv("FP8Q_FLAT_GRID");
        const int v = e ? atoi(e) : 0;
        return v >= 1 ? v : 0;
    }();
    // Tensors beyond the caches: one tile per block (a grid of tens of thousands of short blocks streams 5-10 % faster
    // than a persistent one).  Cache-sized tensors (K1, <= 16384 chunks = 64 MiB): one tile per block means 1..16 ROUNDS of
    // the 1024 resident blocks, and a fractional last round is lost time (2352 tiles = 2.3 rounds pay for 3) -- a
    // resident grid striding over the chunks hands every block the same number +- 1 instead: [2^17,147] 40.2 -> 35.1 us,
    // [2^18,147] 66.2 -> 61.2, [30000,1152] 58.7 -> 51.4, [100000,576] 87.0 -> 83.1; at 36864 chunks it already loses
    // on 576 / 1152-element rows (207 -> 216..241 us), on the headline (75264 chunks) 408 -> 437..477.
    int64_t blocks = cdiv(a.nchunks, nch);
    const int64_t grid_cap = grid_env ? grid_env : ((mode == kModeQuant && a.nchunks <= 16384) ? 1024 : 32768);
    if (blocks > grid_cap) blocks = grid_cap;
    const size_t shmem = (size_t)kFlatMaxCh * sizeof(ChunkInfo) + (size_t)a.rpc * nch * per_row;
    const dim3 g((unsigned)blocks), b(kBlock);
    if (mode == kModeQuant) {
        if (nt) hipLaunchKernelGGL((k_rows_flat<kModeQuant, true>), g, b, shmem, st, x, y, maxval, row_min, row_max, maxval_out, f, a);
        else hipLaunchKernelGGL((k_rows_flat<kModeQuant, false>), g, b, shmem, st, x, y, maxval, row_min, row_max, maxval_out, f, a);
    } else {
        if (nt) hipLaunchKernelGGL((k_rows_flat<kModeFused, true>), g, b, shmem, st, x, y, maxval, row_min, row_max, maxval_out, f, a);
        else hipLaunchKernelGGL((k_rows_flat<kModeFused, false>), g, b, shmem, st, x, y, maxval, row_min, row_max, maxval_out, f, a);
    }
    return launch_rc();
}

// Storage codes of per-channel tensors with short rows through k_rows_flat (aligned 16 KiB chunks of the fp32 side,
// per-row tables in LDS): the row-per-block codec kernel spends a 256-thread block, a double-precision constant
// evaluation and a 33-entry table on every 147-element filter.  kNotFlat when the shape does not fit.
int launch_codec_flat(bool encode, const void *in, void *out, int64_t C, int64_t inner, const float *maxval, const QFmt &f,
                      int n_bits, hipStream_t st)
{
    const void *fp = encode ? in : (const void *)out;       // the fp32 side
    const void *cp = encode ? (const void *)out : in;       // the code side
    if (inner < 4 || inner > direct_max_inner() || ((uintptr_t)fp & 15) != 0 || ((uintptr_t)cp & 3) != 0) return kNotFlat;
    FlatArgs a = {};
    a.inner = (int)inner;
    a.lut_stride = f.pmax + 1;
    a.magic = magic_of((int)inner);
    a.n_bits = n_bits;
    const int64_t n = C * inner;
    a.nvec = n >> 2;
    a.tail = (int)(n & 3);
    a.nchunks = cdiv(a.nvec, kChunkGroups);
    a.rpc = (int)((inner + (kChunkElems + 3) - 2) / inner) + 1;
    a.rmagic = magic_of(a.rpc);
    const int64_t per_row = 16 + 16 + (int64_t)a.lut_stride * 8;
    const int64_t cap = 36 * 1024 - (int64_t)kFlatMaxCh * sizeof(ChunkInfo);
    int64_t nch = cap / (a.rpc * per_row);
    if (nch < 1) return kNotFlat;
    if (nch > 4) nch = 4;   // as K1's tiles (launch_rows_flat)
    while (nch > 1 && cdiv(a.nchunks, nch) < 1024) --nch;
    a.nch = (int)nch;
    a.group = 1;
    int64_t blocks = cdiv(a.nchunks, nch);
    if (blocks > 32768) blocks = 32768;
    const size_t shmem = (size_t)kFlatMaxCh * sizeof(ChunkInfo) + (size_t)a.rpc * nch * per_row;
    const bool nt = n * 4 >= kNtBytes;
    const dim3 g((unsigned)blocks), b(kBlock);
    const float *xf = (const float *)in;    // encode: fp32 in; decode: the codes, reinterpreted inside the kernel
    float *yf = (float *)out;               // decode: fp32 out; encode: the codes
    if (encode && nt) hipLaunchKernelGGL((k_rows_flat<kModeEncode, true>), g, b, shmem, st, xf, yf, maxval, nullptr, nullptr, nullptr, f, a);
    else if (encode) hipLaunchKernelGGL((k_rows_flat<kModeEncode, false>), g, b, shmem, st, xf, yf, maxval, nullptr, nullptr, nullptr, f, a);
    else if (nt) hipLaunchKernelGGL((k_rows_flat<kModeDecode, true>), g, b, shmem, st, xf, yf, maxval, nullptr, nullptr, nullptr, f, a);
    else hipLaunchKernelGGL((k_rows_flat<kModeDecode, false>), g, b, shmem, st, xf, yf, maxval, nullptr, nullptr, nullptr, f, a);
    return launch_rc();
}

// k_rows_staged_mm for [C, inner]: rows of 4..256 elements, x 16-byte aligned; kNotFlat otherwise
int launch_rows_staged_mm(const float *x, int64_t C, int64_t inner, float *row_min, float *row_max, float *maxval_out,
                          const FoldArgs &fa, hipStream_t st)
{
    static const int staged_env = [] {   // FP8Q_STAGED=0: row-tiled k_rows_direct<2> (A/B)
        const char *e = getenv("FP8Q_STAGED");
        return e ? atoi(e) : 1;
    }();
    if (!staged_env || inner < 4 || inner > kFlatFusedMaxInner || ((uintptr_t)x & 15) != 0) return kNotFlat;
    FlatArgs a = {};
    a.inner = (int)inner;
    a.magic = magic_of((int)inner);
    const int64_t n = C * inner;
    a.nvec = n >> 2;
    a.tail = (int)(n & 3);
    a.nchunks = cdiv(a.nvec, kChunkGroups);
    a.rpc = (int)((inner + (kChunkElems + 3) - 2) / inner) + 1;
    int gs = 0;
    while (gs < 3 && (2 << gs) * a.rpc <= kBlock) ++gs;
    a.group = gs;
    static const int grid_env = [] {
        const char *e = getenv("FP8Q_STAGED_MM_GRID");   // persistent-grid cap; 0 = one chunk per block (measured best: no stores to wait for)
        const int v = e ? atoi(e) : -1;
        return v >= 0 ? v : 0;
    }();
    const int64_t blocks = grid_env ? balanced_blocks(a.nchunks, grid_env) : a.nchunks;
    const dim3 g((unsigned)blocks), b(kBlock);
    if (n * 4 >= kNtBytes) hipLaunchKernelGGL((k_rows_staged_mm<true>), g, b, 0, st, x, row_min, row_max, maxval_out, fa, a);
    else hipLaunchKernelGGL((k_rows_staged_mm<false>), g, b, 0, st, x, row_min, row_max, maxval_out, fa, a);
    return launch_rc();
}

// k_rows_reg for [C, inner] if the rows suit it (128..8192 elements, a multiple of 4, 16-byte aligned, lanes well
// filled); kNotFlat otherwise.  quant: fused min/max + quantize; else K2 (min/max + fold).
int launch_rows_reg(bool quant, const float *x, float *y, int64_t C, int64_t inner, float *row_min, float *row_max,
                    float *maxval_out, const QFmt &f, const FoldArgs &fa, hipStream_t st)
{
    static const int reg_env = [] {   // FP8Q_FUSED_REG=0: never (A/B against the row-tiled kernels)
        const char *e = getenv("FP8Q_FUSED_REG");
        return e ? atoi(e) : 1;
    }();
    if (!reg_env || inner > 8192) return kNotFlat;
    if (quant && (inner < 128 || (inner & 3) != 0 || (((uintptr_t)x | (uintptr_t)y) & 15) != 0)) return kNotFlat;
    if (!quant && (inner < 68 || ((uintptr_t)x & 3) != 0)) return kNotFlat;   // K2 reads rows at any 4-byte phase
    // lanes per row and 16-byte slots per lane (EPT, instantiated for 2..8): the best-filled combination --
    // rows of 576 elements run 8 per block on 32 lanes x 5 slots (90 % filled)
    int reg_lanes = 0, reg_ept = 0;
    const int64_t nvec = (inner + 3) >> 2;
    int64_t best = 0;
    for (int lanes : {16, 32, 64, 256}) {
        const int64_t ept = cdiv(nvec, lanes);
        if (ept < 2 || ept > 8) continue;
        const int64_t fill = nvec * 1000 / (ept * lanes);
        if (fill > best) {
            best = fill;
            reg_lanes = lanes;
            reg_ept = (int)ept;
        }
    }
    if (best < 800) return kNotFlat;   // (147-element rows, 77 % filled: 4.8 TB/s here against 5.4 in k_rows_direct<2>)
    const bool nt = C * inner * 4 >= kNtBytes;
    const int64_t steps = cdiv(C, kBlock / reg_lanes);
    const int64_t grid = balanced_blocks(steps, 65536);
    const dim3 g((unsigned)grid), b(kBlock);
#define FP8Q_LAUNCH_REG(LN, E)                                                                                       \
    do {                                                                                                             \
        if (quant && nt) hipLaunchKernelGGL((k_rows_reg<LN, E, true, true>), g, b, 0, st, x, y, C, (int)inner,      \
                                            row_min, row_max, maxval_out, f, fa);                                    \
        else if (quant) hipLaunchKernelGGL((k_rows_reg<LN, E, false, true>), g, b, 0, st, x, y, C, (int)inner,      \
                                           row_min, row_max, maxval_out, f, fa);                                     \
        else if (nt) hipLaunchKernelGGL((k_rows_reg<LN, E, true, false>), g, b, 0, st, x, y, C, (int)inner,         \
                                        row_min, row_max, maxval_out, f, fa);                                        \
        else hipLaunchKernelGGL((k_rows_reg<LN, E, false, false>), g, b, 0, st, x, y, C, (int)inner, row_min,        \
                                row_max, maxval_out, f, fa);                                                         \
    } while (0)
#define FP8Q_LAUNCH_REG_E(LN)                                  \
    switch (reg_ept) {                                         \
        case 2: FP8Q_LAUNCH_REG(LN, 2); break;                 \
        case 3: FP8Q_LAUNCH_REG(LN, 3); break;                 \
        case 4: FP8Q_LAUNCH_REG(LN, 4); break;                 \
        case 5: FP8Q_LAUNCH_REG(LN, 5); break;                 \
        case 6: FP8Q_LAUNCH_REG(LN, 6); break;                 \
        case 7: FP8Q_LAUNCH_REG(LN, 7); break;                 \
        default: FP8Q_LAUNCH_REG(LN, 8); break;                \
    }
    if (reg_lanes == 16) { FP8Q_LAUNCH_REG_E(16) }
    else if (reg_lanes == 32) { FP8Q_LAUNCH_REG_E(32) }
    else if (reg_lanes == 64) { FP8Q_LAUNCH_REG_E(64) }
    else { FP8Q_LAUNCH_REG_E(256) }
#undef FP8Q_LAUNCH_REG_E
#undef FP8Q_LAUNCH_REG
    return launch_rc();
}

// Launch k_rows_direct for [C, inner], inner <= kDirectMaxInner (any 4-byte aligned pointers).
int launch_rows_direct(int mode, const float *x, float *y, int64_t C, int64_t inner, const float *maxval,
                       float *row_min, float *row_max, float *maxval_out, const QFmt &f,
                       const FoldArgs &fa, hipStream_t st)
{
    {
        const int rc = launch_rows_flat(mode, x, y, C, inner, maxval, row_min, row_max, maxval_out, f, st);
        if (rc != kNotFlat) return rc;
    }
    TileArgs a = {};
    a.inner = (int)inner;
    a.lut_stride = f.pmax + 1;
    a.lmagic = magic_of(a.lut_stride);
    a.magic = magic_of((int)inner);
    const bool lut = mode != kModeMinMax && inner >= 2 * (int64_t)a.lut_stride;
    // lanes per row: ~16-32 elements (4-8 dwordx4) per lane, power of two <= 64
    int G = 1;
    while (G < 64 && (int64_t)G * 24 < inner) G <<= 1;
    static const int elems_env = [] {   // tuning knob for experiments
        const char *e = getenv("FP8Q_DIRECT_ELEMS");
        const int v = e ? atoi(e) : 0;
        return v >= 256 && v <= (1 << 20) ? v : kDirectElems;
    }();
    const int BSZ = kBlock;
    const int rpp = BSZ / G;
    int64_t R = (elems_env * BSZ / 256) / inner;
    static const int mm_passes_env = [] { const char *e = getenv("FP8Q_K2_PASSES"); return e ? atoi(e) : 2; }();
    if (mode == kModeMinMax) R = mm_passes_env * rpp;   // no tables: a few passes per iteration
    if (R < rpp) R = rpp;                         // at least one full pass
    if (R > 256) R = 256;                         // one make_chan pass
    // tables + the 3 KiB of staged log2/exp2 tables must fit in 40 KiB of LDS (4 blocks per CU)
    const int64_t per_row = (int64_t)sizeof(Chan) + 4 + 16 + (lut ? (int64_t)a.lut_stride * 8 : 0);
    static const int lds_kb_env = [] {   // tuning knob: LDS budget of the per-row tables
        const char *e = getenv("FP8Q_DIRECT_LDS_KB");
        const int v = e ? atoi(e) : 0;
        return v >= 4 && v <= 120 ? v : 36;
    }();
    const int64_t lds_cap = (int64_t)lds_kb_env * 1024;
    if (R * per_row > lds_cap) R = lds_cap / per_row;
    const int64_t want = cdiv(C, 1024);           // small tensors: spread over >= ~1024 blocks
    if (R > want) R = want;
    if (R >= 4) R &= ~(int64_t)3;                 // keeps tile starts 16-byte aligned for any inner
    if (R < 1) R = 1;
    a.rows = (int)R;
    a.group = G;
    a.coaligned = (y == nullptr) || ((((uintptr_t)x ^ (uintptr_t)y) & 15) == 0);
    const size_t shmem = (size_t)((R + 3) & ~(int64_t)3) * 4 + (size_t)R * 16 + (size_t)R * sizeof(Chan) +
                         (lut ? (size_t)R * a.lut_stride * sizeof(float2) : 0);
    int64_t blocks = cdiv(C, R);
    // K2: many short blocks (two passes of rows each) measured best: 5.4 TB/s against 4.9 with 4096 x 4 passes
    static const int mm_blocks_env = [] { const char *e = getenv("FP8Q_K2_BLOCKS"); return e ? atoi(e) : 65536; }();
    const int64_t bcap = mode == kModeMinMax ? mm_blocks_env : 2 * kTargetBlocks;
    if (blocks > bcap) blocks = balanced_blocks(blocks, bcap);
    const bool nt = C * inner * 4 >= kNtBytes;
    const dim3 g((unsigned)blocks), b(BSZ);
#define FP8Q_LAUNCH_DIRECT(M, L, N)                                                                    \
    hipLaunchKernelGGL((k_rows_direct<M, L, N>), g, b, shmem, st, x, y, C, maxval, row_min, row_max,    \
                       maxval_out, f, a, fa)
    if (mode == kModeMinMax) {
        FP8Q_LAUNCH_DIRECT(kModeMinMax, false, false);
    } else if (mode == kModeQuant) {
        if (lut && nt) FP8Q_LAUNCH_DIRECT(kModeQuant, true, true);
        else if (lut) FP8Q_LAUNCH_DIRECT(kModeQuant, true, false);
        else if (nt) FP8Q_LAUNCH_DIRECT(kModeQuant, false, true);
        else FP8Q_LAUNCH_DIRECT(kModeQuant, false, false);
    } else {
        if (lut && nt) FP8Q_LAUNCH_DIRECT(kModeFused, true, true);
        else if (lut) FP8Q_LAUNCH_DIRECT(kModeFused, true, false);
        else if (nt) FP8Q_LAUNCH_DIRECT(kModeFused, false, true);
        else FP8Q_LAUNCH_DIRECT(kModeFused, false, false);
    }
#undef FP8Q_LAUNCH_DIRECT
    return launch_rc();
}

}  // namespace

// used by fp8q_codec.hip (same library, not part of the C ABI)
__attribute__((visibility("hidden"))) int fp8q_codec_flat_launch(bool encode, const void *in, void *out, int64_t C,
                                                                 int64_t inner, const float *maxval, const QFmt &f,
                                                                 int n_bits, hipStream_t st)
{
    const int rc = launch_codec_flat(encode, in, out, C, inner, maxval, f, n_bits, st);
    return rc == kNotFlat ? FP8Q_CODEC_NOT_FLAT : rc;
}

extern "C" {

int fp8q_version(void) { return FP8Q_VERSION; }

const char *fp8q_strerror(int code)
{
    switch (code) {
        case FP8Q_OK: return "ok";
        case FP8Q_EINVAL: return "invalid argument";
        case FP8Q_EUNSUPPORTED: return "unsupported format (more than 7 exponent bits)";
        case FP8Q_EWORKSPACE: return "workspace too small or misaligned";
        case FP8Q_ETOOLONG: return "rows longer than fp8q_fused_max_inner(): use fp8q_minmax_f32 + fp8q_quantize_f32";
        case FP8Q_ETOOMANY: return "more than 65535 channels in one MSE grid-search call";
        case FP8Q_ETIMEDOUT: return "a min/max reducer block timed out waiting for its streaming blocks (that call's range is NaN)";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

int fp8q_quantize_f32(const float *x, float *y, int64_t C, int64_t inner, const float *maxval,
                      int64_t n_maxval, float mbits, int n_bits, int sign_bits, fp8q_stream_t stream)
{
    if (C < 0 || inner < 0 || (n_maxval != 1 && n_maxval != C)) return FP8Q_EINVAL;
    QFmt f;
    if (int rc = make_fmt(mbits, n_bits, sign_bits, &f)) return rc;
    if (C == 0 || inner == 0) return FP8Q_OK;   // empty tensor: nothing to do (pointers may be null)
    if (!x || !y || !maxval) return FP8Q_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int per_channel = n_maxval != 1;
    if (!per_channel) {  // one row
        inner *= C;
        C = 1;
    }
    const bool aligned = (((uintptr_t)x ^ (uintptr_t)y) & 15) == 0 && ((uintptr_t)x & 3) == 0;
    const bool nt = C * inner * 4 >= kNtBytes;

    if (per_channel && inner <= direct_max_inner() && ((uintptr_t)x & 3) == 0 && ((uintptr_t)y & 3) == 0) {
        // short rows: G lanes per row, tables in LDS
        const FoldArgs nofold = {0, 1, 0.0f, 0.0f};
        return launch_rows_direct(kModeQuant, x, y, C, inner, maxval, nullptr, nullptr, nullptr, f, nofold,
                                  st);
    }
    if (C > 65535) {
        // very many long rows: one launch per 65535 rows (gridDim.y limit)
        for (int64_t c0 = 0; c0 < C; c0 += 65535) {
            const int64_t cn = (C - c0) < 65535 ? (C - c0) : 65535;
            int rc = fp8q_quantize_f32(x + c0 * inner, y + c0 * inner, cn, inner, maxval + c0, cn,
                                       mbits, n_bits, sign_bits, stream);
            if (rc) return rc;
        }
        return FP8Q_OK;
    }
    static const int k1_blocks_env = [] {   // tuning knob
        const char *e = getenv("FP8Q_K1_BLOCKS");
        const int v = e ? atoi(e) : 0;
        return v >= 1 ? v : 0;
    }();
    // One 16 KiB piece per block for big tensors (measured: 6.3 TB/s against 5.8 with a persistent grid of 2048
    // blocks at 1 GiB; small tensors prefer the smaller grid); a partial last piece of a row gets no block of its own.
    static const int64_t small_elems = [] {   // FP8Q_K1_SMALL_M: tensors below this many Mi elements run 4 KiB pieces per block
        const char *e = getenv("FP8Q_K1_SMALL_M");
        const long v = e ? atol(e) : -1;
        return (int64_t)(v >= 0 ? v : 8) << 20;
    }();
    const bool small = C == 1 && inner < small_elems;   // per tensor only: a per-channel block also builds its row's table
    const int U = small ? 1 : kUnroll;
    const int64_t pieces = inner / (4 * kBlock * U) > 0 ? inner / (4 * kBlock * U) : 1;
    // cache-sized tensors (< 64 MiB): a resident grid of 2048 blocks with the same number of pieces each, not several
    // ragged rounds of one-piece blocks ([64,128,28,28] 12.2 -> 10.9 us, [64,144,28,28] 13.0 -> 11.6, [64,24,56,56]
    // 10.0 -> 9.0 by rocprofv3); tensors beyond the caches: one piece per block
    const int64_t total_cap = k1_blocks_env > 0 ? k1_blocks_env : ((!nt || pieces * C <= 4096) ? kTargetBlocks : 65536);
    const int64_t cap = total_cap / C > 0 ? total_cap / C : 1;
    const int64_t bx = balanced_blocks(pieces, cap);
    if (aligned) {
        const dim3 g((unsigned)bx, (unsigned)C), b(kBlock);
        if (nt)
            hipLaunchKernelGGL((k_quant_rows<true, kUnroll>), g, b, 0, st, x, y, inner, maxval, per_channel, f);
        else if (small)
            hipLaunchKernelGGL((k_quant_rows<false, 1>), g, b, 0, st, x, y, inner, maxval, per_channel, f);
        else
            hipLaunchKernelGGL((k_quant_rows<false, kUnroll>), g, b, 0, st, x, y, inner, maxval, per_channel, f);
    } else {
        int64_t bs = cdiv(inner, kBlock);
        if (bs > cap * 4) bs = cap * 4;
        hipLaunchKernelGGL(k_quant_scalar, dim3((unsigned)bs, (unsigned)C), dim3(kBlock), 0, st, x, y,
                           inner, maxval, per_channel, f);
    }
    return launch_rc();
}

}  // extern "C"

// the launch geometry shared by the entry points whose format is chosen on the device (width: fp8q_quantize_dm_f32, sign:
// fp8q_quantize_ds_f32)
template <class SEL>
static int quantize_sel_launch(const float *x, float *y, int64_t C, int64_t inner, const float *maxval, int64_t n_maxval,
                               const SEL &sel, fp8q_stream_t stream)
{
    if (C == 0 || inner == 0) return FP8Q_OK;
    if (!x || !y || !maxval) return FP8Q_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int per_channel = n_maxval != 1;
    if (!per_channel) {
        inner *= C;
        C = 1;
    }
    if (per_channel && inner <= 2048 && inner < (1ll << 31) / 4) {
        // short rows: a wave per row
        const int64_t blocks = cdiv(C, kBlock / 64);
        hipLaunchKernelGGL(k_quant_short_rows_dm<SEL>, dim3((unsigned)(blocks < 8 * kTargetBlocks ? blocks : 8 * kTargetBlocks)), dim3(kBlock), 0, st,
                           x, y, C, (int)inner, maxval, sel);
        return launch_rc();
    }
    for (int64_t c0 = 0; c0 < C; c0 += 65535) {   // gridDim.y limit
        const int64_t cn = (C - c0) < 65535 ? (C - c0) : 65535;
        const float *xs = x + c0 * inner;
        float *ys = y + c0 * inner;
        // k_quant_rows peels every row to 16-byte alignment assuming x and y rows are co-aligned
        const bool aligned = (((uintptr_t)xs ^ (uintptr_t)ys) & 15) == 0 && ((uintptr_t)xs & 3) == 0;
        // the geometry of fp8q_quantize_f32 (nontemporal beyond the caches, 4 KiB pieces on a resident grid for cache-sized
        // per-tensor rows): the format is the only thing this entry point does not know on the host
        const bool nt = cn * inner * 4 >= kNtBytes;
        const bool small = cn == 1 && inner < ((int64_t)8 << 20);
        const int U = small ? 1 : kUnroll;
        const int64_t pieces = inner / (4 * kBlock * U) > 0 ? inner / (4 * kBlock * U) : 1;
        const int64_t total_cap = (!nt || pieces * cn <= 4096) ? kTargetBlocks : 65536;
        const int64_t cap = total_cap / cn > 0 ? total_cap / cn : 1;
        const int64_t bx = balanced_blocks(pieces, cap);
        if (aligned) {
            const dim3 g((unsigned)bx, (unsigned)cn), b(kBlock);
            const float *mvp = maxval + (per_channel ? c0 : 0);
            if (nt)
                hipLaunchKernelGGL((k_quant_rows_dm<true, kUnroll, SEL>), g, b, 0, st, xs, ys, inner, mvp, per_channel, sel);
            else if (small)
                hipLaunchKernelGGL((k_quant_rows_dm<false, 1, SEL>), g, b, 0, st, xs, ys, inner, mvp, per_channel, sel);
            else
                hipLaunchKernelGGL((k_quant_rows_dm<false, kUnroll, SEL>), g, b, 0, st, xs, ys, inner, mvp, per_channel, sel);
        } else {
            int64_t bs = cdiv(inner, kBlock);
            if (bs > cap * 4) bs = cap * 4;
            hipLaunchKernelGGL(k_quant_scalar_dm<SEL>, dim3((unsigned)bs, (unsigned)cn), dim3(kBlock), 0, st, xs, ys, inner,
                               maxval + (per_channel ? c0 : 0), per_channel, sel);
        }
        if (int rc = launch_rc()) return rc;
    }
    return FP8Q_OK;
}

extern "C" {

int fp8q_quantize_dm_f32(const float *x, float *y, int64_t C, int64_t inner, const float *maxval, int64_t n_maxval,
                         const float *mbits_dev, int n_bits, int sign_bits, fp8q_stream_t stream)
{
    if (C < 0 || inner < 0 || (n_maxval != 1 && n_maxval != C) || !mbits_dev) return FP8Q_EINVAL;
    FmtSel sel;
    sel.mbits_dev = mbits_dev;
    sel.signed_dev = nullptr;
    sel.hi = n_bits - sign_bits;
    if (sel.hi < 1 || sel.hi > 8) return FP8Q_EINVAL;
    for (int M = 1; M <= 8; ++M)
        if (int rc = make_fmt((float)(M <= sel.hi ? M : sel.hi), n_bits, sign_bits, &sel.tab[M - 1])) return rc;
    return quantize_sel_launch(x, y, C, inner, maxval, n_maxval, sel, stream);
}

// K1 of a quantizer with allow_unsigned whose sign_bits is still a flag in device memory (fp8q_sign_fold_u8): both formats
// of the width travel by value, the kernel reads the flag (fp8_quantizer.py:216-225 decides it with a host round trip).
int fp8q_quantize_ds_f32(const float *x, float *y, int64_t C, int64_t inner, const float *maxval, int64_t n_maxval,
                         float mbits, int n_bits, const unsigned char *signed_flag, fp8q_stream_t stream)
{
    if (C < 0 || inner < 0 || (n_maxval != 1 && n_maxval != C) || !signed_flag) return FP8Q_EINVAL;
    FmtSel sel;
    sel.mbits_dev = nullptr;
    sel.signed_dev = signed_flag;
    sel.hi = 2;
    if (int rc = make_fmt(mbits, n_bits, 1, &sel.tab[0])) return rc;
    if (int rc = make_fmt(mbits, n_bits, 0, &sel.tab[1])) return rc;
    for (int i = 2; i < 8; ++i) sel.tab[i] = sel.tab[1];
    return quantize_sel_launch(x, y, C, inner, maxval, n_maxval, sel, stream);
}

// ... and with the mantissa width in device memory as well (the MSE estimator's vote for a quantizer whose sign is still
// pending): 2 x 8 formats by value.
int fp8q_quantize_dms_f32(const float *x, float *y, int64_t C, int64_t inner, const float *maxval, int64_t n_maxval,
                          const float *mbits_dev, int n_bits, const unsigned char *signed_flag, fp8q_stream_t stream)
{
    if (C < 0 || inner < 0 || (n_maxval != 1 && n_maxval != C) || !mbits_dev || !signed_flag) return FP8Q_EINVAL;
    FmtSel2 sel;
    sel.mbits_dev = mbits_dev;
    sel.signed_dev = signed_flag;
    for (int u = 0; u < 2; ++u) {
        sel.hi[u] = n_bits - (1 - u);
        if (sel.hi[u] < 1 || sel.hi[u] > 8) return FP8Q_EINVAL;
        for (int M = 1; M <= 8; ++M)
            if (int rc = make_fmt((float)(M <= sel.hi[u] ? M : sel.hi[u]), n_bits, 1 - u, &sel.tab[u][M - 1])) return rc;
    }
    return quantize_sel_launch(x, y, C, inner, maxval, n_maxval, sel, stream);
}

// FPQuantizer.set_quant_range's sign decision (fp8_quantizer.py:216-225: `allow_unsigned and torch.all(x_min >= 0)` ->
// sign_bits = 0, never back) on the device: signed_flag[0] stays 1 only while some range minimum is not >= 0 (NaN: signed).
__global__ void __launch_bounds__(kBlock) k_sign_fold(const float *__restrict__ x_min, int64_t C, unsigned char *flag)
{
    __shared__ int s_any;
    if (threadIdx.x == 0) s_any = 0;
    __syncthreads();
    int any = 0;
    for (int64_t i = threadIdx.x; i < C; i += kBlock) any |= !(x_min[i] >= 0.0f);
    if (any) s_any = 1;
    __syncthreads();
    if (threadIdx.x == 0 && !s_any) flag[0] = 0;
}

int fp8q_sign_fold_u8(const float *x_min, int64_t C, unsigned char *signed_flag, fp8q_stream_t stream)
{
    if (C < 0 || !signed_flag || (C > 0 && !x_min)) return FP8Q_EINVAL;
    hipLaunchKernelGGL(k_sign_fold, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, x_min, C, signed_flag);
    return launch_rc();
}

}  // extern "C"

// (called from fp8q_mse.hip: the last step of fp8q_mse_calibrate_f32 for a per-tensor quantizer with the mantissa search)
// Returns FP8Q_EUNSUPPORTED when the tensor does not take the aligned vector kernel: the caller then selects in its own launch.
int fp8q_quantize_select_f32(const float *x, float *y, int64_t n, const float *mses, const float *grid, int n_m, int n_cand,
                             const SelOne *so, int n_bits, int sign_bits, hipStream_t st)
{
    FmtSel sel;
    sel.mbits_dev = nullptr;
    sel.signed_dev = nullptr;
    sel.hi = n_bits - sign_bits;
    if (sel.hi < 1 || sel.hi > 8 || n_m < 1 || n_m > kSelMaxM || n_cand < 1 || !so || !x || !y || n <= 0) return FP8Q_EINVAL;
    for (int M = 1; M <= 8; ++M)
        if (int rc = make_fmt((float)(M <= sel.hi ? M : sel.hi), n_bits, sign_bits, &sel.tab[M - 1])) return rc;
    const bool aligned = (((uintptr_t)x ^ (uintptr_t)y) & 15) == 0 && ((uintptr_t)x & 3) == 0;
    if (!aligned) return FP8Q_EUNSUPPORTED;
    SelIn si;
    si.mses = mses;
    si.grid = grid;
    si.n_m = n_m;
    si.n_cand = n_cand;
    si.so = *so;
    // the geometry of fp8q_quantize_dm_f32 for one row
    const bool nt = n * 4 >= kNtBytes;
    const bool small = n < ((int64_t)8 << 20);
    const int U = small ? 1 : kUnroll;
    const int64_t pieces = n / (4 * kBlock * U) > 0 ? n / (4 * kBlock * U) : 1;
    // (a resident grid also beyond the caches -- fp8q_quantize_dm_f32 gives every 16 KiB piece its own workgroup there --: the
    // selection prologue is paid once per workgroup: 39.8 us with 6272 workgroups on [64,32,112,112] against 32.7 without it)
    static const int64_t sel_cap = getenv("FP8Q_SEL_CAP") ? atoll(getenv("FP8Q_SEL_CAP")) : kTargetBlocks;
    const int64_t cap = (!nt || pieces <= 4096) ? kTargetBlocks : sel_cap;
    const dim3 g((unsigned)balanced_blocks(pieces, cap), 1u), b(kBlock);
    if (nt)
        hipLaunchKernelGGL((k_quant_rows_sel<true, kUnroll>), g, b, 0, st, x, y, n, sel, si);
    else if (small)
        hipLaunchKernelGGL((k_quant_rows_sel<false, 1>), g, b, 0, st, x, y, n, sel, si);
    else
        hipLaunchKernelGGL((k_quant_rows_sel<false, kUnroll>), g, b, 0, st, x, y, n, sel, si);
    return launch_rc();
}

extern "C" {

static int minmax_nsplit(int64_t C, int64_t inner)
{
    static const int cap_env = [] {   // FP8Q_K3_BLOCKS: streaming blocks of the two-stage min/max (tuning knob)
        const char *e = getenv("FP8Q_K3_BLOCKS");
        const int v = e ? atoi(e) : 0;
        return v >= 1 && v <= kTargetBlocks ? v : kTargetBlocks;   // <= 2048: split rows keep their reducers <= 1024 (progress argument, fp8q_common.h)
    }();
    return (int)balanced_blocks(cdiv(cdiv(inner, 4), kBlock * 8), cap_env / (C > 0 ? C : 1));
}

size_t fp8q_minmax_workspace_bytes(int64_t C, int64_t inner)
{
    if (C <= 0 || inner <= 0) return 16;
    if (inner <= direct_max_inner() && C > 1) return 16;  // short-row path needs none
    const int ns = minmax_nsplit(C, inner);
    // the 16-byte header {timeout count, reserved} + two tagged granules per part
    return ns > 1 ? kMinmaxWsHeader + (size_t)C * (size_t)ns * 2 * sizeof(unsigned long long) : kMinmaxWsHeader;
}

// Synchronising check of a min/max workspace (fp8q_minmax_workspace_check; also run on entry by every min/max call
// under FP8Q_DEBUG_WS=1): the header's timeout count and the "all granules zero between calls" contract.
static int minmax_ws_check(void *ws, size_t ws_bytes, int clear, hipStream_t st)
{
    if (!ws || ws_bytes < kMinmaxWsHeader || ((uintptr_t)ws & 7)) return FP8Q_EWORKSPACE;
    if (hipError_t e = hipStreamSynchronize(st)) return (int)e;
    std::vector<unsigned long long> host(ws_bytes / 8);
    if (hipError_t e = hipMemcpy(host.data(), ws, host.size() * 8, hipMemcpyDeviceToHost)) return (int)e;
    const unsigned timeouts = (unsigned)host[0];
    bool dirty = false;
    for (size_t i = kMinmaxWsHeader / 8; i < host.size(); ++i) dirty |= host[i] != 0ull;
    if (clear && (timeouts || dirty)) {
        if (hipError_t e = hipMemsetAsync(ws, 0, ws_bytes, st)) return (int)e;
        if (hipError_t e = hipStreamSynchronize(st)) return (int)e;
    }
    if (timeouts) return FP8Q_ETIMEDOUT;
    return dirty ? FP8Q_EWORKSPACE : FP8Q_OK;
}

static bool minmax_debug_ws()
{
    static const bool on = [] {
        const char *e = getenv("FP8Q_DEBUG_WS");
        return e && atoi(e) != 0;
    }();
    return on;
}

struct LinArgs {
    float *grid = nullptr;
    int steps = 0;
    double lo = 0.0, hi = 0.0;
};

static int minmax_impl(const float *x, int64_t C, int64_t inner, float *cur_min, float *cur_max, float *maxval_out,
                       float *packed, int fold_mode, double momentum, int first, void *ws, size_t ws_bytes,
                       fp8q_stream_t stream, LinArgs lin = LinArgs())
{
    if (!x || !cur_min || !cur_max || C <= 0 || inner <= 0 || fold_mode < 0 || fold_mode > 2)
        return FP8Q_EINVAL;
    if (packed && ((uintptr_t)packed & 15)) return FP8Q_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    FoldArgs fa;
    fa.mode = fold_mode;
    fa.first = first != 0;
    fa.om = (float)(1.0 - momentum);
    fa.mo = (float)momentum;
    fa.packed = packed;
    fa.lin_grid = lin.grid;
    fa.lin_steps = lin.steps;
    fa.lin_C = C;
    fa.lin_lo = lin.lo;
    fa.lin_hi = lin.hi;
    // (the row-kernel launchers below cut C into slabs themselves and pass slab-local rows to fold_store together with
    // slab-offset range pointers, while the grid / table base stays that of row 0: tables per row need C <= 65535 there)
    if (lin.grid && C > 65535) return FP8Q_ETOOMANY;
    if (C > 1) {   // per-channel rows of 128..8192 elements: one launch, the row in registers
        QFmt f = {};
        const int rc = launch_rows_reg(false, x, nullptr, C, inner, cur_min, cur_max, maxval_out, f, fa, st);
        if (rc != kNotFlat) return rc;
    }
    if (C > 1) {   // short rows: aligned chunks through LDS
        const int rc = launch_rows_staged_mm(x, C, inner, cur_min, cur_max, maxval_out, fa, st);
        if (rc != kNotFlat) return rc;
    }
    if (inner <= direct_max_inner() && C > 1 && ((uintptr_t)x & 3) == 0) {
        QFmt f = {};
        return launch_rows_direct(kModeMinMax, x, nullptr, C, inner, nullptr, cur_min, cur_max, maxval_out,
                                  f, fa, st);
    }
    if (ws_bytes < fp8q_minmax_workspace_bytes(C, inner) || !ws || ((uintptr_t)ws & 7)) return FP8Q_EWORKSPACE;
    if (minmax_debug_ws())
        if (int rc = minmax_ws_check(ws, fp8q_minmax_workspace_bytes(C, inner), 0, st)) return rc;
    const int ns = minmax_nsplit(C, inner);
    const unsigned tag = next_minmax_tag();
    const unsigned gx = ns > 1 ? (unsigned)ns + 1u : 1u;   // + the row's reducer block
    fa.status = (unsigned *)ws;
    fold_debug_env(fa);
    unsigned long long *slots = (unsigned long long *)((char *)ws + kMinmaxWsHeader);
    for (int64_t c0 = 0; c0 < C; c0 += 65535) {   // ns > 1 implies C <= kTargetBlocks / 2: a single slab
        const int64_t cn = (C - c0) < 65535 ? (C - c0) : 65535;
        fa.packed = packed ? packed + 4 * c0 : nullptr;
        if (C * inner * 4 >= kNtBytes)
            hipLaunchKernelGGL(k_minmax_partial<true>, dim3(gx, (unsigned)cn), dim3(kBlock), 0, st, x + c0 * inner, inner,
                               ns, slots, tag, cur_min + c0, cur_max + c0, maxval_out ? maxval_out + c0 : nullptr, fa);
        else
            hipLaunchKernelGGL(k_minmax_partial<false>, dim3(gx, (unsigned)cn), dim3(kBlock), 0, st, x + c0 * inner, inner,
                               ns, slots, tag, cur_min + c0, cur_max + c0, maxval_out ? maxval_out + c0 : nullptr, fa);
    }
    return launch_rc();
}

int fp8q_minmax_f32(const float *x, int64_t C, int64_t inner, float *cur_min, float *cur_max,
                    float *maxval_out, int fold_mode, double momentum, int first, void *ws,
                    size_t ws_bytes, fp8q_stream_t stream)
{
    return minmax_impl(x, C, inner, cur_min, cur_max, maxval_out, nullptr, fold_mode, momentum, first, ws, ws_bytes, stream);
}

int fp8q_minmax_packed_f32(const float *x, int64_t C, int64_t inner, float *cur_min, float *cur_max,
                           float *maxval_out, float *packed, int fold_mode, double momentum, int first, void *ws,
                           size_t ws_bytes, fp8q_stream_t stream)
{
    if (!packed) return FP8Q_EINVAL;
    return minmax_impl(x, C, inner, cur_min, cur_max, maxval_out, packed, fold_mode, momentum, first, ws, ws_bytes, stream);
}

int fp8q_minmax_linspace_f32(const float *x, int64_t C, int64_t inner, float *cur_min, float *cur_max, float *maxval_out,
                             float *grid, int n_cand, double lo_frac, double hi_frac, void *ws, size_t ws_bytes,
                             fp8q_stream_t stream)
{
    if (!grid || !maxval_out || n_cand < 2 || n_cand > (1 << 20)) return FP8Q_EINVAL;
    LinArgs lin;
    lin.grid = grid;
    lin.steps = n_cand;
    lin.lo = lo_frac;
    lin.hi = hi_frac;
    return minmax_impl(x, C, inner, cur_min, cur_max, maxval_out, nullptr, FP8Q_FOLD_CURRENT, 0.0, 1, ws, ws_bytes, stream, lin);
}

int fp8q_minmax_workspace_check(void *ws, size_t ws_bytes, int clear, fp8q_stream_t stream)
{
    try {
        return minmax_ws_check(ws, ws_bytes, clear, (hipStream_t)stream);
    } catch (...) {
        return (int)hipErrorOutOfMemory;
    }
}

int fp8q_ranges_unpack_f32(const float *packed, int64_t n, float *cur_min, float *cur_max, float *maxval_out,
                           fp8q_stream_t stream)
{
    if (n < 0 || (n > 0 && (!packed || ((uintptr_t)packed & 15)))) return FP8Q_EINVAL;
    if (n == 0) return FP8Q_OK;
    hipLaunchKernelGGL(k_ranges_unpack, dim3((unsigned)cdiv(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, packed, n,
                       cur_min, cur_max, maxval_out);
    return launch_rc();
}

int64_t fp8q_fused_max_inner(void) { return kDirectMaxInner; }

int fp8q_minmax_quantize_f32(const float *x, float *y, int64_t C, int64_t inner, float *row_min,
                             float *row_max, float *maxval_out, float mbits, int n_bits,
                             int sign_bits, fp8q_stream_t stream)
{
    if (C < 0 || inner < 0) return FP8Q_EINVAL;
    QFmt f;
    if (int rc = make_fmt(mbits, n_bits, sign_bits, &f)) return rc;
    if (C == 0 || inner == 0) return FP8Q_OK;
    if (!x || !y) return FP8Q_EINVAL;
    if (inner > kDirectMaxInner) return FP8Q_ETOOLONG;
    if (((uintptr_t)x & 3) != 0 || ((uintptr_t)y & 3) != 0) return FP8Q_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    static const bool small_fused = [] {   // FP8Q_SMALL_FUSED=0: the general routes for small tensors too (A/B)
        const char *e = getenv("FP8Q_SMALL_FUSED");
        return !e || atoi(e) != 0;
    }();
    if (small_fused && C * inner <= kSmallFusedElems && inner <= kSmallFusedInner) {
        const dim3 g((unsigned)cdiv(C, kBlock / 64)), b(kBlock);
        const int epl = (int)cdiv(inner, 64);
#define FP8Q_LAUNCH_SMALL(E) hipLaunchKernelGGL(k_small_rows_fused<E>, g, b, 0, st, x, y, C, (int)inner, row_min, row_max, maxval_out, f)
        if (epl <= 1) FP8Q_LAUNCH_SMALL(1);
        else if (epl <= 2) FP8Q_LAUNCH_SMALL(2);
        else if (epl <= 3) FP8Q_LAUNCH_SMALL(3);
        else if (epl <= 4) FP8Q_LAUNCH_SMALL(4);
        else FP8Q_LAUNCH_SMALL(8);
#undef FP8Q_LAUNCH_SMALL
        return launch_rc();
    }
    {
        const FoldArgs nofold = {0, 1, 0.0f, 0.0f};
        const int rc = launch_rows_reg(true, x, y, C, inner, row_min, row_max, maxval_out, f, nofold, st);
        if (rc != kNotFlat) return rc;
    }
    const FoldArgs nofold = {0, 1, 0.0f, 0.0f};
    return launch_rows_direct(kModeFused, x, y, C, inner, nullptr, row_min, row_max, maxval_out, f, nofold, st);
}

// A prepared multi-tensor launch: the descriptors validated, classified and packed into kernel arguments once.
struct PlanStep {
    int mode = 0;              // 0: K1, 3: encode to storage codes, 4: decode (k_multi_flat<MODE>)
    bool batched;              // true: one k_multi_flat launch of `args`; false: one single-tensor call of `single`
    MultiArgs args;
    size_t shmem;
    fp8q_tensor_desc single;
};

}  // extern "C"

struct fp8q_multi_plan {
    std::vector<PlanStep> steps;
};

static int plan_build(const fp8q_tensor_desc *descs, int n, fp8q_multi_plan &plan, int mode = 0)
{
    if (n < 0 || (n > 0 && !descs)) return FP8Q_EINVAL;
    // validate everything first: nothing is built (or enqueued) if any descriptor is bad
    for (int i = 0; i < n; ++i) {
        const fp8q_tensor_desc &t = descs[i];
        if (t.C < 0 || t.inner < 0 || (t.n_maxval != 1 && t.n_maxval != t.C)) return FP8Q_EINVAL;
        QFmt f;
        if (int rc = make_fmt(t.mbits, t.n_bits, t.sign_bits, &f)) return rc;
        if (t.C > 0 && t.inner > 0 && (!t.x || !t.y || !t.maxval)) return FP8Q_EINVAL;
        // storage codes: one byte, at least one exponent bit (include/fp8q.h: FP8Q_EUNSUPPORTED otherwise)
        if (mode != 0 && (t.n_bits > 8 || t.n_bits - t.sign_bits - (int)f.M < 1)) return FP8Q_EUNSUPPORTED;
    }
    PlanStep cur;
    cur.mode = mode;
    cur.batched = true;
    cur.args.n = 0;
    cur.args.rpc_max = 0;
    cur.args.total_chunks = 0;
    cur.shmem = 0;
    size_t lut_max = 0;   // largest table area (rows x entries) any tensor of the current launch needs
    auto flush = [&]() {
        if (cur.args.n == 0) return;
        cur.shmem = (size_t)cur.args.rpc_max * 16 + lut_max;   // [chanlite[rpc_max] | tables]
        plan.steps.push_back(cur);
        cur.args.n = 0;
        cur.args.rpc_max = 0;
        cur.args.total_chunks = 0;
        cur.shmem = 0;
        lut_max = 0;
    };
    for (int i = 0; i < n; ++i) {
        const fp8q_tensor_desc &t = descs[i];
        if (t.C == 0 || t.inner == 0) continue;
        QFmt f;
        make_fmt(t.mbits, t.n_bits, t.sign_bits, &f);
        const bool per_channel = t.n_maxval != 1;
        const int64_t nelem = t.C * t.inner;
        const int64_t inner = per_channel ? t.inner : nelem;
        const int64_t rpc = per_channel ? (inner + (kChunkElems + 3) - 2) / inner + 1 : 1;
        const int64_t per_row = 16 + 16 + (int64_t)(f.pmax + 1) * 8;
        // 16-byte groups of fp32 on the value side, 4-byte groups of codes on the other
        const uintptr_t mis = mode == 3 ? (((uintptr_t)t.x & 15) | ((uintptr_t)t.y & 3))
                            : mode == 4 ? (((uintptr_t)t.x & 3) | ((uintptr_t)t.y & 15)) : (((uintptr_t)t.x | (uintptr_t)t.y) & 15);
        const bool batchable = mis == 0 && nelem >= 4 && nelem < (1ll << 31) &&
                               (!per_channel || (inner >= 4 && inner <= kMagicMaxDivisor)) &&
                               rpc * per_row <= 36 * 1024 && nelem * 4 < kNtBytes;
        if (!batchable) {   // unaligned, very short rows, or a tensor big enough to deserve its own launch
            flush();
            PlanStep one;
            one.mode = mode;
            one.batched = false;
            one.args.n = 0;
            one.args.rpc_max = 0;
            one.args.total_chunks = 0;
            one.shmem = 0;
            one.single = t;
            plan.steps.push_back(one);
            continue;
        }
        if (cur.args.n == kMultiMax) flush();
        MultiDesc &d = cur.args.d[cur.args.n++];
        d.x = t.x;
        d.y = t.y;
        d.maxval = t.maxval;
        d.nvec = nelem >> 2;
        d.tail = (int)(nelem & 3);
        d.single_row = per_channel ? 0 : 1;
        d.inner = per_channel ? (int)inner : 0;
        d.rpc = (int)rpc;
        d.magic = per_channel ? magic_of((int)inner) : 0u;
        d.chunk0 = cur.args.total_chunks;
        d.n_bits = t.n_bits;
        d.f = f;
        cur.args.total_chunks += (uint32_t)cdiv(d.nvec, kChunkGroups);
        // per table row: the channel constants (16 B) + pmax + 1 entries {s, 1/s}; every tensor of the launch uses the
        // same layout [chanlite[rpc_max] | tables], so size it for the largest of each
        if ((int)rpc > cur.args.rpc_max) cur.args.rpc_max = (int)rpc;
        const size_t need = (size_t)rpc * (size_t)(f.pmax + 1) * 8;
        if (need > lut_max) lut_max = need;
    }
    flush();
    return FP8Q_OK;
}

static int plan_launch(const fp8q_multi_plan &plan, hipStream_t st)
{
    for (const PlanStep &s : plan.steps) {
        if (s.batched) {
            // every block resident at once (4 per CU at 128 VGPRs): chunks grid-stride, a few per block
            static const int grid_env = [] {   // FP8Q_MULTI_GRID: block cap of the multi-tensor launch (tuning knob)
                const char *e = getenv("FP8Q_MULTI_GRID");
                const int v = e ? atoi(e) : 0;
                return v >= 1 ? v : 1024;
            }();
            const dim3 grid((unsigned)balanced_blocks(s.args.total_chunks, grid_env));
            if (s.mode == 3)
                hipLaunchKernelGGL(k_multi_flat<3>, grid, dim3(kBlock), s.shmem, st, s.args);
            else if (s.mode == 4)
                hipLaunchKernelGGL(k_multi_flat<4>, grid, dim3(kBlock), s.shmem, st, s.args);
            else
                hipLaunchKernelGGL(k_multi_flat<0>, grid, dim3(kBlock), s.shmem, st, s.args);
            if (int rc = launch_rc()) return rc;
        } else {
            const fp8q_tensor_desc &t = s.single;
            int rc;
            if (s.mode == 3)
                rc = fp8q_encode_u8(t.x, reinterpret_cast<uint8_t *>(t.y), t.C, t.inner, t.maxval, t.n_maxval, t.mbits, t.n_bits,
                                    t.sign_bits, (fp8q_stream_t)st);
            else if (s.mode == 4)
                rc = fp8q_decode_u8(reinterpret_cast<const uint8_t *>(t.x), t.y, t.C, t.inner, t.maxval, t.n_maxval, t.mbits, t.n_bits,
                                    t.sign_bits, (fp8q_stream_t)st);
            else
                rc = fp8q_quantize_f32(t.x, t.y, t.C, t.inner, t.maxval, t.n_maxval, t.mbits, t.n_bits, t.sign_bits, (fp8q_stream_t)st);
            if (rc) return rc;
        }
    }
    return FP8Q_OK;
}

extern "C" {

int fp8q_multi_quantize_f32(const fp8q_tensor_desc *descs, int n, fp8q_stream_t stream)
{
    try {
        fp8q_multi_plan plan;
        if (int rc = plan_build(descs, n, plan)) return rc;
        return plan_launch(plan, (hipStream_t)stream);
    } catch (...) {
        return (int)hipErrorOutOfMemory;
    }
}

static int multi_codec(const fp8q_tensor_desc *descs, int n, fp8q_stream_t stream, int mode)
{
    try {
        fp8q_multi_plan plan;
        if (int rc = plan_build(descs, n, plan, mode)) return rc;
        return plan_launch(plan, (hipStream_t)stream);
    } catch (...) {
        return (int)hipErrorOutOfMemory;
    }
}

int fp8q_multi_encode_u8(const fp8q_tensor_desc *descs, int n, fp8q_stream_t stream) { return multi_codec(descs, n, stream, 3); }
int fp8q_multi_decode_u8(const fp8q_tensor_desc *descs, int n, fp8q_stream_t stream) { return multi_codec(descs, n, stream, 4); }

static int multi_minmax_then(const fp8q_tensor_desc *descs, float *const *maxval_out, int n, fp8q_stream_t stream, int mode);

int fp8q_multi_minmax_quantize_f32(const fp8q_tensor_desc *descs, float *const *maxval_out, int n, fp8q_stream_t stream)
{
    return multi_minmax_then(descs, maxval_out, n, stream, 0);
}

int fp8q_multi_minmax_encode_u8(const fp8q_tensor_desc *descs, float *const *maxval_out, int n, fp8q_stream_t stream)
{
    return multi_minmax_then(descs, maxval_out, n, stream, 3);
}

static int multi_minmax_then(const fp8q_tensor_desc *descs, float *const *maxval_out, int n, fp8q_stream_t stream, int mode)
{
    if (n < 0 || (n > 0 && (!descs || !maxval_out))) return FP8Q_EINVAL;
    for (int i = 0; i < n; ++i) {   // per-channel ranges only; nothing is enqueued if a descriptor is bad
        const fp8q_tensor_desc &t = descs[i];
        if (t.C < 0 || t.inner < 0 || t.n_maxval != t.C || t.C >= (1ll << 31) || t.inner >= (1ll << 31)) return FP8Q_EINVAL;
        QFmt f;
        if (int rc = make_fmt(t.mbits, t.n_bits, t.sign_bits, &f)) return rc;
        if (t.C > 0 && t.inner > 0 && (!t.x || !t.y || !maxval_out[i] || ((uintptr_t)t.x & 3))) return FP8Q_EINVAL;
        // the quantize launch reads the ranges where the range launch wrote them: descs[i].maxval names that buffer too
        // (or is NULL); an input range buffer elsewhere would be silently ignored -- refuse it
        if (t.maxval && t.maxval != maxval_out[i]) return FP8Q_EINVAL;
        // the codec's own constraints, checked HERE: the encode launch below would refuse the descriptor only after the
        // range launch had been enqueued (and had overwritten maxval_out)
        if (mode == 3 && (t.n_bits > 8 || t.n_bits - t.sign_bits - (int)f.M < 1)) return FP8Q_EUNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    RowsArgs ra;
    ra.n = 0;
    ra.total_rows = 0;
    auto flush = [&]() -> int {
        if (ra.n == 0) return FP8Q_OK;
        hipLaunchKernelGGL(k_multi_rowmax, dim3((unsigned)cdiv(ra.total_rows, kBlock / 64)), dim3(kBlock), 0, st, ra);
        ra.n = 0;
        ra.total_rows = 0;
        return launch_rc();
    };
    for (int i = 0; i < n; ++i) {
        const fp8q_tensor_desc &t = descs[i];
        if (t.C == 0 || t.inner == 0) continue;
        if (ra.n == kMultiMax || (uint64_t)ra.total_rows + (uint64_t)t.C >= (1ull << 31))
            if (int rc = flush()) return rc;
        RowsDesc &d = ra.d[ra.n++];
        d.x = t.x;
        d.maxval = maxval_out[i];
        d.row_min = d.row_max = nullptr;
        d.inner = (int)t.inner;
        d.row0 = ra.total_rows;
        ra.total_rows += (uint32_t)t.C;
    }
    if (int rc = flush()) return rc;
    std::vector<fp8q_tensor_desc> q(descs, descs + n);
    for (int i = 0; i < n; ++i) q[i].maxval = maxval_out[i];
    // same stream: reads the ranges just written
    return mode == 3 ? fp8q_multi_encode_u8(q.data(), n, stream) : fp8q_multi_quantize_f32(q.data(), n, stream);
}

int fp8q_multi_plan_create(const fp8q_tensor_desc *descs, int n, fp8q_multi_plan **plan_out)
{
    if (!plan_out) return FP8Q_EINVAL;
    *plan_out = nullptr;
    try {
        fp8q_multi_plan *plan = new fp8q_multi_plan();
        if (int rc = plan_build(descs, n, *plan)) {
            delete plan;
            return rc;
        }
        *plan_out = plan;
        return FP8Q_OK;
    } catch (...) {
        return (int)hipErrorOutOfMemory;
    }
}

int fp8q_multi_plan_launch(const fp8q_multi_plan *plan, fp8q_stream_t stream)
{
    if (!plan) return FP8Q_EINVAL;
    return plan_launch(*plan, (hipStream_t)stream);
}

int fp8q_multi_plan_launches(const fp8q_multi_plan *plan) { return plan ? (int)plan->steps.size() : FP8Q_EINVAL; }

void fp8q_multi_plan_destroy(fp8q_multi_plan *plan) { delete plan; }

int fp8q_copy_f32(const float *x, float *y, int64_t n, fp8q_stream_t stream)
{
    if (!x || !y || n < 0 || (n & 3) || ((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return FP8Q_EINVAL;
    if (n == 0) return FP8Q_OK;
    const int64_t pieces = cdiv(n / 4, kBlock * kUnroll);
    const int64_t bx = balanced_blocks(pieces, pieces > 4096 ? 65536 : kTargetBlocks);   // K1's grid rule
    if (n * 4 >= kNtBytes)
        hipLaunchKernelGGL(k_copy<true>, dim3((unsigned)bx), dim3(kBlock), 0, (hipStream_t)stream,
                           (const vf4 *)x, (vf4 *)y, n / 4);
    else
        hipLaunchKernelGGL(k_copy<false>, dim3((unsigned)bx), dim3(kBlock), 0, (hipStream_t)stream,
                           (const vf4 *)x, (vf4 *)y, n / 4);
    return launch_rc();
}

}  // extern "C"
