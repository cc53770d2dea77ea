// fp8q_epilogue.hip -- N2: eval-BN + residual + ReLU/ReLU6 fused with the per-tensor activation quantizer, and its range twin.
#include "fp8q_common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// N2: producer epilogue fused with the activation quantizer:
//   t = fma(x, alpha_c, fma(-mean_c, alpha_c, beta_c)), alpha_c = invstd_c * gamma_c   (eval-mode BN, NCHW)
//   t = t + residual                                          (optional)
//   t = relu(t) / relu6(t)                                    (optional)
//   QUANT:  y = quantize_to_fp8(t; per-tensor maxval)        MINMAX: min/max of t -> partials
// i.e. quantized_folded_bn.py:39-55 and models/resnet_quantized.py:43-46 in one pass (8 B/element
// instead of three 8 B passes).  Flat over N*C*HW; channel of a 16-byte group by two magic
// divisions, per element only when a group crosses a plane boundary (HW % 4 != 0).
// ---------------------------------------------------------------------------------------------
struct AffineArgs {
    int64_t image;      // C * HW elements per image (multiple of 4, < 2^31)
    int C, HW;
    int act;            // 0 none, 1 relu, 2 relu6
    int has_bn, has_res;
    int cpp;            // most channels one 4096-element piece can overlap (size of the LDS constants)
    uint32_t magic;     // o / HW for o < 4096 + HW (magic_of); unused when HW > kAffineMagicMaxHW
    uint64_t magic48;   // floor(2^48 / HW) + 1:  n / HW == (n * magic48) >> 48  for n * HW < 2^48 (calibration twin)
    int passthrough;    // y = t itself, no quantizer (fp8q_affine_act_f32: what an MSE estimator searches on; maxval unused)
};
constexpr int kAffinePiece = kBlock * 4 * 4;   // elements per block and step (16 KiB)
constexpr int kAffineMagicMaxHW = kMagicMaxDivisor;   // above: a piece spans <= 2 planes (compare instead of divide)

// {alpha, beta'} of one channel: alpha = invstd * gamma, beta' = fma(-mean, alpha, beta) (ATen's eval-mode batch norm) --
// or, with has_bn == 2, read from the folded [C, 2] vector that fp8q_bn_fold_f32 wrote (`mean` then points to it): one
// 8-byte load instead of four 4-byte ones per plane, the same two numbers.
__device__ __forceinline__ float2 bn_ab(const float *__restrict__ mean, const float *__restrict__ invstd,
                                        const float *__restrict__ gamma, const float *__restrict__ beta, uint32_t ch, int has_bn)
{
    if (has_bn == 2) return reinterpret_cast<const float2 *>(mean)[ch];
    const float alpha = invstd[ch] * gamma[ch];
    return make_float2(alpha, fmaf(-mean[ch], alpha, beta[ch]));
}

__global__ void __launch_bounds__(kBlock)
k_bn_fold(const float *__restrict__ mean, const float *__restrict__ invstd, const float *__restrict__ gamma,
          const float *__restrict__ beta, int64_t C, float2 *__restrict__ ab)
{
    const int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (c < C) ab[c] = bn_ab(mean, invstd, gamma, beta, (uint32_t)c, 1);
}

// {channel constants, table} of one per-tensor quantizer, laid out as k_affine_act_small copies them: float4 {maxval, lo,
// bias, pthr} followed by float2 {s_p, 1 / s_p}, p = 0 .. pmax
__global__ void __launch_bounds__(kBlock)
k_quantizer_prepare(const float *__restrict__ maxval, QFmt f, float4 *__restrict__ prep)
{
    const Chan cfull = make_chan(maxval[0], f);
    if (threadIdx.x == 0) prep[0] = make_float4(cfull.maxv, cfull.minv, cfull.bias, cfull.pthr);
    for (int i = threadIdx.x; i <= f.pmax; i += kBlock) reinterpret_cast<float2 *>(prep + 1)[i] = lut_entry(cfull, i, f.M);
}

// act(t + r) -- the non-affine part of the epilogue
__device__ __forceinline__ float res_act(float t, float r, const AffineArgs &a)
{
    if (a.has_res) t = t + r;
    if (a.act >= 1) t = t < 0.0f ? 0.0f : t;         // NaN stays NaN (torch.relu)
    if (a.act == 2) t = t > 6.0f ? 6.0f : t;
    return t;
}

// blockIdx.y = image n; blockIdx.x strides over the image's C*HW elements in aligned 16 KiB pieces
// (one piece per block when the grid allows it: measured 6.1-6.3 TB/s against 5.0 for a persistent
// grid of 2048 blocks).  Eval-mode batch norm exactly as ATen's CPU kernel evaluates it (probed:
// bit-identical on 100 % of elements): alpha = invstd * gamma, beta' = fma(-mean, alpha, beta),
// out = fma(x, alpha, beta'); {alpha, beta'} of the planes a piece overlaps are staged in LDS per
// step, the plane of a 16-byte group comes from one 32-bit magic division of its piece-local offset.
template <bool NT, int U>
__global__ void __launch_bounds__(kBlock)
k_affine_act(const float *__restrict__ x, const float *__restrict__ res, float *__restrict__ y,
             const float *__restrict__ mean, const float *__restrict__ invstd,
             const float *__restrict__ gamma, const float *__restrict__ beta,
             const float *__restrict__ maxval, QFmt f, AffineArgs a)
{
    __shared__ float2 lut[kLutMax];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *cst = reinterpret_cast<float2 *>(smem);   // [cpp] {alpha, beta'} (has_bn only)
    const int tid = threadIdx.x;
    ChanLite c = {};
    if (!a.passthrough) {
        const Chan cfull = make_chan(maxval[0], f);
        for (int i = tid; i <= f.pmax; i += kBlock) lut[i] = lut_entry(cfull, i, f.M);
        c = lite(cfull);
    }
    const float pmaxf = (float)f.pmax;
    const int64_t base0 = (int64_t)blockIdx.y * a.image;
    const int nvec = (int)(a.image >> 2);
    const vf4 *xv = reinterpret_cast<const vf4 *>(x + base0);
    const vf4 *rv = reinterpret_cast<const vf4 *>(res + (a.has_res ? base0 : 0));
    vf4 *yv = reinterpret_cast<vf4 *>(y + base0);
    const uint32_t HW = (uint32_t)a.HW;
    // planes of >= 4096 elements: a piece overlaps at most two, whose constants every thread keeps in registers
    // (uniform addresses: scalar loads) -- no LDS staging and no barrier per step, which is what separates the BN
    // variant from the plain one on the large early layers (75 -> 70 us at [64,64,112,112]; the same with up to
    // four planes in registers, for 56x56 planes, was measured and rejected: 94 us)
    const bool direct = a.has_bn && a.cpp <= 2;
    __syncthreads();       // lut is complete
    for (int base = blockIdx.x * (kBlock * U); base < nvec; base += gridDim.x * (kBlock * U)) {
        uint32_t phase = 0;
        float2 p0 = make_float2(1.0f, 0.0f), p1 = p0;
        if (direct) {
            const uint32_t e0 = (uint32_t)base * 4u;
            const uint32_t ch_lo = e0 / HW;
            phase = e0 - ch_lo * HW;
            const uint32_t ch_hi = ch_lo + 1u < (uint32_t)a.C ? ch_lo + 1u : ch_lo;
            p0 = bn_ab(mean, invstd, gamma, beta, ch_lo, a.has_bn);
            p1 = bn_ab(mean, invstd, gamma, beta, ch_hi, a.has_bn);
        }
        if (a.has_bn && !direct) {
            __syncthreads();   // the previous step's constants are no longer read
            const uint32_t e0 = (uint32_t)base * 4u;
            const uint32_t ch_lo = e0 / HW;
            phase = e0 - ch_lo * HW;
            for (int k = tid; k < a.cpp; k += kBlock) {
                const uint32_t ch = ch_lo + (uint32_t)k;
                if (ch < (uint32_t)a.C) cst[k] = bn_ab(mean, invstd, gamma, beta, ch, a.has_bn);
            }
            __syncthreads();
        }
        auto transform = [&](int q, float (&e)[4], const vf4 &r) {   // q: piece-local group index
            const float rr[4] = {r.x, r.y, r.z, r.w};
            if (direct) {
                const uint32_t o = phase + 4u * (uint32_t)q;
                const bool hi = o >= HW;                    // second plane of the piece
                uint32_t off = hi ? o - HW : o;
                float2 p = hi ? p1 : p0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    e[k] = res_act(fmaf(e[k], p.x, p.y), rr[k], a);
                    if (++off == HW && k < 3) {
                        off = 0;
                        p = p1;
                    }
                }
            } else if (a.has_bn) {
                const uint32_t o = phase + 4u * (uint32_t)q;
                uint32_t lch = HW > (uint32_t)kAffineMagicMaxHW ? (o >= HW ? 1u : 0u) : (uint32_t)div_small(o, a.magic);
                uint32_t off = o - lch * HW;
                float2 p = cst[lch];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    e[k] = res_act(fmaf(e[k], p.x, p.y), rr[k], a);
                    if (++off == HW && k < 3) {      // the group crosses into the next plane
                        off = 0;                     // (still inside the image: groups never straddle images)
                        p = cst[++lch];
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) e[k] = res_act(e[k], rr[k], a);
            }
        };
        if (base + kBlock * U <= nvec) {   // whole piece: unpredicated loads, one branch for all its elements
            vf4 vx[U], vr[U];
#pragma unroll
            for (int u = 0; u < U; ++u) vx[u] = ld16<NT>(xv + base + u * kBlock + tid);
#pragma unroll
            for (int u = 0; u < U; ++u)
                vr[u] = a.has_res ? ld16<NT>(rv + base + u * kBlock + tid) : vf4{0.0f, 0.0f, 0.0f, 0.0f};
            float e[U * 4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float t[4] = {vx[u].x, vx[u].y, vx[u].z, vx[u].w};
                transform(u * kBlock + tid, t, vr[u]);
                e[4 * u] = t[0];
                e[4 * u + 1] = t[1];
                e[4 * u + 2] = t[2];
                e[4 * u + 3] = t[3];
            }
            if (!a.passthrough) quant_group<U * 4>(e, c, lut, pmaxf, f.qthr);
#pragma unroll
            for (int u = 0; u < U; ++u)
                st16<NT>(yv + base + u * kBlock + tid, vf4{e[4 * u], e[4 * u + 1], e[4 * u + 2], e[4 * u + 3]});
        } else {
            for (int u = 0; u < U; ++u) {
                const int j = base + u * kBlock + tid;
                if (j >= nvec) break;
                const vf4 v = xv[j];
                const vf4 r = a.has_res ? ld16<NT>(rv + j) : vf4{0.0f, 0.0f, 0.0f, 0.0f};
                float e[4] = {v.x, v.y, v.z, v.w};
                transform(u * kBlock + tid, e, r);
                if (!a.passthrough) quant_group<4>(e, c, lut, pmaxf, f.qthr);
                st16<NT>(yv + j, vf4{e[0], e[1], e[2], e[3]});
            }
        }
    }
}

// Small (cache-sized) activations: the launch is latency-bound -- what counts is the length of the dependent chain from
// block start to the first store, not bandwidth.  k_affine_act's chain is {maxval -> channel constants -> table} ->
// barrier -> {BN vectors -> LDS} -> barrier -> {x} -> arithmetic -> store: three memory round trips behind each other
// where the plain K1 has two.  Here a lane handles ONE 16-byte group per step and fetches the BN vectors of its own
// plane(s) straight from global memory (L2 hits: <= 20 KB per model layer) next to x itself, so x, residual and the BN
// vectors are one round trip, issued BEFORE the table is built (EARLY) so that the table's arithmetic hides it.
// rocprofv3 kernel durations at batch 64 (round 3's staged kernel -> this one): [64,160,7,7] 5.4 -> 4.4 us, [64,96,14,14]
// 6.7 -> 5.3, [64,576,7,7] 6.7 -> 5.5, [64,384,14,14] 12.2 -> 9.5, [64,144,28,28] 16.0 -> 13.1 (plain K1 on the same
// tensors: 3.9 / 5.1 / 5.1 / 7.3 / 10.4); it wins up to the nontemporal threshold ([64,192,28,28] 19.7 -> 14.8 us,
// [64,64,56,56] 21.2 -> 18.5) and loses beyond it, where the staged kernel's LDS constants cost less than eight extra
// dword loads per group ([64,96,56,56] 31.3 vs 35.6 us with NT = true, U = 4) -- so it serves every tensor below 64 MiB.
// Measured and rejected on top of it: 2 or 4 groups per lane and step (U = 2: +8...12 %, U = 4: +15...35 % time),
// nontemporal accesses below 64 MiB (+10 %), grids of 1024...16384 blocks (2048...4096 equal, the rest worse), requesting
// the next step's group before the arithmetic of the current one (no change).  With the folded BN vector (has_bn == 2:
// two 8-byte loads per group instead of eight 4-byte ones) another 5-8 %: [64,64,56,56] 19.9 -> 18.5 us (K1: 17.9).
template <bool EARLY, bool NT, int U>
__global__ void __launch_bounds__(kBlock)
k_affine_act_small(const float *__restrict__ x, const float *__restrict__ res, float *__restrict__ y,
                   const float *__restrict__ mean, const float *__restrict__ invstd, const float *__restrict__ gamma,
                   const float *__restrict__ beta, const float *__restrict__ maxval, QFmt f, AffineArgs a,
                   const float4 *__restrict__ prep)
{
    __shared__ float2 lut[kLutMax];
    const int tid = threadIdx.x;
    const int64_t base0 = (int64_t)blockIdx.y * a.image;
    const int nvec = (int)(a.image >> 2);
    const vf4 *xv = reinterpret_cast<const vf4 *>(x + base0);
    const vf4 *rv = reinterpret_cast<const vf4 *>(res + (a.has_res ? base0 : 0));
    vf4 *yv = reinterpret_cast<vf4 *>(y + base0);
    const uint32_t HW = (uint32_t)a.HW;
    const int step = gridDim.x * (kBlock * U);
    int base = blockIdx.x * (kBlock * U);

    struct Grp {
        vf4 v, r;
        float al0, bp0, al1, bp1;   // {alpha, beta'} of the group's plane and of the next one (HW >= 4: a group spans <= 2)
        uint32_t off;               // offset of the group's first element inside its plane
    };
    auto fetch = [&](int jj, Grp &g) {
        g.v = ld16<NT>(xv + jj);
        g.r = a.has_res ? ld16<NT>(rv + jj) : vf4{0.0f, 0.0f, 0.0f, 0.0f};
        g.off = 0;
        g.al0 = g.al1 = 1.0f;
        g.bp0 = g.bp1 = 0.0f;
        if (a.has_bn) {
            const uint32_t i0 = (uint32_t)jj * 4u;
            const uint32_t ch = (uint32_t)(((uint64_t)i0 * a.magic48) >> 48);      // plane == channel
            g.off = i0 - ch * HW;
            const uint32_t ch1 = ch + 1u < (uint32_t)a.C ? ch + 1u : ch;
            const float2 q0 = bn_ab(mean, invstd, gamma, beta, ch, a.has_bn);
            const float2 q1 = bn_ab(mean, invstd, gamma, beta, ch1, a.has_bn);
            g.al0 = q0.x;
            g.bp0 = q0.y;
            g.al1 = q1.x;
            g.bp1 = q1.y;
        }
    };
    Grp g[U];
    if (EARLY) {
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (base + u * kBlock + tid < nvec) fetch(base + u * kBlock + tid, g[u]);
    }
    // The quantizer's channel constants and {s, 1/s} table: rebuilt from maxval (a load, ~50 dependent double-precision
    // operations, the table entries) -- or, with fixed ranges, copied from the block fp8q_quantizer_prepare_f32 wrote once:
    // the same numbers, one 8-byte load per entry, 0.5-1 us less on the critical path of a 4-10 us launch.
    ChanLite c = {};
    if (a.passthrough) {
    } else if (prep) {
        const float4 h = prep[0];
        c.maxv = h.x;
        c.minv = h.y;
        c.bias = h.z;
        c.pthr = h.w;
        for (int i = tid; i <= f.pmax; i += kBlock) lut[i] = reinterpret_cast<const float2 *>(prep + 1)[i];
    } else {
        const Chan cfull = make_chan(maxval[0], f);
        for (int i = tid; i <= f.pmax; i += kBlock) lut[i] = lut_entry(cfull, i, f.M);
        c = lite(cfull);
    }
    const float pmaxf = (float)f.pmax;
    __syncthreads();
    bool first = true;
    for (; base < nvec; base += step) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = base + u * kBlock + tid;
            if (j < nvec && !(EARLY && first)) fetch(j, g[u]);
        }
        first = false;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = base + u * kBlock + tid;
            if (j >= nvec) break;
            float e[4] = {g[u].v.x, g[u].v.y, g[u].v.z, g[u].v.w};
            const float rr[4] = {g[u].r.x, g[u].r.y, g[u].r.z, g[u].r.w};
            if (a.has_bn) {
                // elements at group-local index >= cross belong to the next plane (HW >= 4: at most one crossing)
                const uint32_t cross = HW - g[u].off;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool nxt = (uint32_t)k >= cross;
                    e[k] = res_act(fmaf(e[k], nxt ? g[u].al1 : g[u].al0, nxt ? g[u].bp1 : g[u].bp0), rr[k], a);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) e[k] = res_act(e[k], rr[k], a);
            }
            if (!a.passthrough) quant_group<4>(e, c, lut, pmaxf, f.qthr);
            st16<NT>(yv + j, vf4{e[0], e[1], e[2], e[3]});
        }
    }
}

// Calibration twin of k_affine_act: min / max of act(bn(x) + residual); the block that finishes last folds all blocks'
// partials into the running estimate (block_minmax_fold: one launch).  Read-only, so the
// trade-offs differ from the quantizing kernel: a persistent grid of <= 2048 blocks with 4 KiB steps, plain
// loads and the constants straight from global measured best (36 us at [64,64,112,112]; 16 KiB steps, LDS-staged
// constants, nontemporal loads or one piece per block: 42-55 us).
// STORE: the transformed tensor t is also written (fp8q_affine_act_minmax_linspace_f32: the first calibration batch of an MSE
// estimator behind a BN + activation needs t itself, its abs-max and the search grid of that maximum -- one pass instead of
// torch's batch_norm + activation + this library's abs-max pass).
template <bool STORE>
__global__ void __launch_bounds__(kBlock)
k_affine_minmax(const float *__restrict__ x, const float *__restrict__ res, const float *__restrict__ mean,
                const float *__restrict__ invstd, const float *__restrict__ gamma,
                const float *__restrict__ beta, AffineArgs a, int64_t N, unsigned long long *slots, int nparts,
                unsigned tag, float *cur_min, float *cur_max, float *maxval_out, FoldArgs fa, float *__restrict__ t_out)
{
    const int nby = nparts / (int)gridDim.x;   // block rows that stream; one more row holds the reducer (nparts > 1)
    if ((int)blockIdx.y == nby) {
        if (blockIdx.x == 0) block_minmax_collect(slots, nparts, tag, 0, cur_min, cur_max, maxval_out, fa);
        return;
    }
    const int tid = threadIdx.x;
    MinMax mm;
    mm_init(mm);
    const int nvec = (int)(a.image >> 2);
    const uint32_t HW = (uint32_t)a.HW;
    for (int64_t img = blockIdx.y; img < N; img += nby) {   // more than 65535 images: several per block row
    const int64_t base = img * a.image;
    const vf4 *xv = reinterpret_cast<const vf4 *>(x + base);
    const vf4 *rv = reinterpret_cast<const vf4 *>(res + (a.has_res ? base : 0));
    for (int j = blockIdx.x * kBlock + tid; j < nvec; j += gridDim.x * kBlock) {
        const vf4 v = xv[j];
        vf4 r = {0.0f, 0.0f, 0.0f, 0.0f};
        if (a.has_res) r = rv[j];
        float e[4] = {v.x, v.y, v.z, v.w};
        const float rr[4] = {r.x, r.y, r.z, r.w};
        if (a.has_bn) {
            const uint32_t i0 = (uint32_t)j * 4u;
            uint32_t ch = (uint32_t)(((uint64_t)i0 * a.magic48) >> 48);      // plane == channel
            uint32_t off = i0 - ch * HW;
            float2 p = bn_ab(mean, invstd, gamma, beta, ch, a.has_bn);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                e[q] = res_act(fmaf(e[q], p.x, p.y), rr[q], a);
                if (++off == HW && q < 3) {      // the group crosses into the next plane
                    off = 0;
                    ++ch;                         // still < C: the group ends inside the image
                    p = bn_ab(mean, invstd, gamma, beta, ch, a.has_bn);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) e[q] = res_act(e[q], rr[q], a);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) mm_acc(mm, e[q]);
        if (STORE) reinterpret_cast<vf4 *>(t_out + base)[j] = vf4{e[0], e[1], e[2], e[3]};
    }
    }
    block_minmax_publish(mm, slots, (int)(blockIdx.y * gridDim.x + blockIdx.x), nparts, tag, 0, cur_min, cur_max,
                         maxval_out, fa);
}

}  // namespace

extern "C" {

static int affine_args(int64_t N, int64_t C, int64_t HW, int act, int has_bn, bool has_res, AffineArgs *a)
{
    if (N < 0 || C <= 0 || HW <= 0 || act < 0 || act > 2) return FP8Q_EINVAL;
    // 16-byte groups must not straddle images; 32-bit element indices within an image; the 48-bit magic
    // division of the calibration twin needs C*HW*HW < 2^48
    if (((C * HW) & 3) != 0 || C * HW >= (1ll << 31) || HW >= (1 << 24) ||
        (double)C * (double)HW * (double)HW >= 281474976710656.0)
        return FP8Q_EUNSUPPORTED;
    a->image = C * HW;
    a->C = (int)C;
    a->HW = (int)HW;
    a->act = act;
    a->has_bn = has_bn;
    a->has_res = has_res;
    const int64_t cpp = (HW + kAffinePiece - 2) / HW + 1;   // planes a 4096-element window can overlap
    a->cpp = (int)(cpp < C ? cpp : C);
    a->magic = HW <= kAffineMagicMaxHW ? magic_of((int)HW) : 0u;
    a->magic48 = (1ull << 48) / (uint64_t)HW + 1ull;
    a->passthrough = 0;
    return FP8Q_OK;
}

static void affine_grid(int64_t N, const AffineArgs &a, bool quant, int64_t *bx, int64_t *by, int U = 4)
{
    const int64_t ymax = quant ? 65535 : 65534;   // the calibration twin adds one block row for its reducer
    *by = N < ymax ? N : ymax;
    const int64_t nvec = a.image >> 2;
    if (quant) {
        // one 16 KiB piece per block while the grid stays below 64 K blocks; a partial last piece of the
        // image gets no block of its own
        const int64_t pieces = nvec / (kBlock * U) > 0 ? nvec / (kBlock * U) : 1;
        static const int64_t cap_env = [] {   // FP8Q_EPI_GRID: total block cap of the quantizing epilogue kernel (experiments)
            const char *e = getenv("FP8Q_EPI_GRID");
            const long v = e ? atol(e) : 0;
            return (int64_t)(v > 0 ? v : 0);
        }();
        // tensors beyond the caches: one piece per block; cache-sized ones (< 64 MiB): a resident grid of 2048 blocks with
        // the same number of pieces each -- one piece per block is then a handful of ROUNDS of the resident blocks, and a
        // fractional last round is lost time ([64,144,28,28] 20.2 -> 17.0 us, [64,24,56,56] 15.8 -> 13.8, [64,64,56,56]
        // 25.2 -> 23.7, [64,128,28,28] 18.5 -> 16.9 by rocprofv3)
        const bool cache_sized = N * a.image * 4 < kNtBytes;
        const int64_t total = cap_env ? cap_env : ((cache_sized || pieces * *by <= 4096) ? kTargetBlocks : 65536);
        *bx = balanced_blocks(pieces, total / *by > 0 ? total / *by : 1);   // K1's grid rule
    } else {
        // read-only twin: a persistent grid of <= 2048 blocks with 4 KiB steps measured best (36 us against
        // 43-55 us for 16 KiB steps or one piece per block at [64,64,112,112])
        int64_t b = cdiv(nvec, kBlock * 4);
        const int64_t cap = kTargetBlocks / *by > 0 ? kTargetBlocks / *by : 1;
        *bx = b > cap ? cap : (b < 1 ? 1 : b);
    }
}

// k_affine_minmax<true> (the epilogue that also WRITES t: first calibration batch of an MSE estimator): the read-only twin's
// persistent grid of <= 2048 blocks with four 16-byte groups per thread and step leaves a small activation on a fraction of
// the chip ([64,160,7,7]: 128 blocks, 10.5 us against 4.9 for the plain epilogue on the same tensor) -- a group per thread
// while the grid stays below FP8Q_EPI_MM_GRID (2048) blocks (more blocks = more partials for the reducer to poll: 4096 and up were slower on
// every shape; four groups per thread and step requested together, or nontemporal accesses beyond the caches: no gain / slower).
static void affine_grid_store(int64_t N, const AffineArgs &a, int64_t *bx, int64_t *by)
{
    static const int64_t cap_env = [] {
        const char *e = getenv("FP8Q_EPI_MM_GRID");
        const long v = e ? atol(e) : 0;
        return (int64_t)(v > 0 ? v : 2048);
    }();
    *by = N < 65534 ? N : 65534;
    const int64_t nvec = a.image >> 2;
    const int64_t pieces = cdiv(nvec, kBlock);
    *bx = balanced_blocks(pieces, cap_env / *by > 0 ? cap_env / *by : 1);
}

// FP8Q_EPI_SMALL_KIND: 0 = the staged kernel for small tensors too (round 3), 1 = k_affine_act_small, 2 = with early loads
static int small_kind()
{
    static const int v = [] {
        const char *e = getenv("FP8Q_EPI_SMALL_KIND");
        return e ? atoi(e) : 2;
    }();
    return v;
}

int fp8q_bn_fold_f32(const float *mean, const float *invstd, const float *gamma, const float *beta, int64_t C,
                     float *alpha_beta, fp8q_stream_t stream)
{
    if (C < 0 || (C > 0 && (!mean || !invstd || !gamma || !beta || !alpha_beta || ((uintptr_t)alpha_beta & 7)))) return FP8Q_EINVAL;
    if (C == 0) return FP8Q_OK;
    hipLaunchKernelGGL(k_bn_fold, dim3((unsigned)cdiv(C, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, mean, invstd, gamma,
                       beta, C, reinterpret_cast<float2 *>(alpha_beta));
    return launch_rc();
}

static int affine_quantize_impl(const float *x, const float *residual, float *y, int64_t N, int64_t C, int64_t HW,
                                const float *mean, const float *invstd, const float *gamma, const float *beta, int has_bn,
                                int act, const float *maxval, const float *prep, float mbits, int n_bits, int sign_bits,
                                fp8q_stream_t stream, int passthrough = 0);

int fp8q_affine_act_quantize_f32(const float *x, const float *residual, float *y, int64_t N, int64_t C,
                                 int64_t HW, const float *mean, const float *invstd, const float *gamma,
                                 const float *beta, int act, const float *maxval, float mbits, int n_bits,
                                 int sign_bits, fp8q_stream_t stream)
{
    const int has_bn = mean != nullptr;
    if (has_bn && (!invstd || !gamma || !beta)) return FP8Q_EINVAL;
    return affine_quantize_impl(x, residual, y, N, C, HW, mean, invstd, gamma, beta, has_bn, act, maxval, nullptr, mbits, n_bits,
                                sign_bits, stream);
}

int fp8q_quantizer_prepare_f32(const float *maxval, float mbits, int n_bits, int sign_bits, float *prep, fp8q_stream_t stream)
{
    if (!maxval || !prep || ((uintptr_t)prep & 15)) return FP8Q_EINVAL;
    QFmt f;
    if (int rc = make_fmt(mbits, n_bits, sign_bits, &f)) return rc;
    hipLaunchKernelGGL(k_quantizer_prepare, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, maxval, f, reinterpret_cast<float4 *>(prep));
    return launch_rc();
}

int fp8q_affine_act_quantize_ab_f32(const float *x, const float *residual, float *y, int64_t N, int64_t C, int64_t HW,
                                    const float *alpha_beta, int act, const float *maxval, const float *prep, float mbits,
                                    int n_bits, int sign_bits, fp8q_stream_t stream)
{
    if (((uintptr_t)alpha_beta & 7) || ((uintptr_t)prep & 15)) return FP8Q_EINVAL;
    return affine_quantize_impl(x, residual, y, N, C, HW, alpha_beta, nullptr, nullptr, nullptr, alpha_beta ? 2 : 0, act, maxval,
                                prep, mbits, n_bits, sign_bits, stream);
}

int fp8q_affine_act_f32(const float *x, const float *residual, float *y, int64_t N, int64_t C, int64_t HW, const float *alpha_beta,
                        int act, fp8q_stream_t stream)
{
    if ((uintptr_t)alpha_beta & 7) return FP8Q_EINVAL;
    // the quantizing kernels with the quantizer switched off: same geometry, same BN arithmetic (what the quantizer will see)
    return affine_quantize_impl(x, residual, y, N, C, HW, alpha_beta, nullptr, nullptr, nullptr, alpha_beta ? 2 : 0, act, nullptr, nullptr,
                                3.0f, 8, 1, stream, 1);
}

static int affine_quantize_impl(const float *x, const float *residual, float *y, int64_t N, int64_t C, int64_t HW,
                                const float *mean, const float *invstd, const float *gamma, const float *beta, int has_bn,
                                int act, const float *maxval, const float *prep, float mbits, int n_bits, int sign_bits,
                                fp8q_stream_t stream, int passthrough)
{
    AffineArgs a;
    if (int rc = affine_args(N, C, HW, act, has_bn, residual != nullptr, &a)) return rc;
    a.passthrough = passthrough != 0;
    QFmt f;
    if (int rc = make_fmt(mbits, n_bits, sign_bits, &f)) return rc;
    if (N == 0) return FP8Q_OK;
    if (!x || !y || (!maxval && !passthrough)) return FP8Q_EINVAL;
    if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) & 15) return FP8Q_EINVAL;
    static const int64_t small_elems = [] {   // FP8Q_EPI_SMALL_M: tensors below this many M elements run 4 KiB pieces per block
        const char *e = getenv("FP8Q_EPI_SMALL_M");
        const long v = e ? atol(e) : -1;
        return (int64_t)(v >= 0 ? v : 16) << 20;
    }();
    // cache-sized tensors are latency-bound: four times as many blocks with a quarter of the piece each
    const bool small = N * a.image < small_elems;
    for (int64_t n0 = 0; n0 < N; n0 += 65535) {
        int64_t bx, by;
        affine_grid(N - n0, a, true, &bx, &by, small ? 1 : 4);
        const size_t shm = has_bn ? (size_t)a.cpp * sizeof(float2) : 0;
        const dim3 g((unsigned)bx, (unsigned)by), b(kBlock);
        const float *rp = residual ? residual + n0 * a.image : nullptr;
        if (N * a.image * 4 >= kNtBytes)
            hipLaunchKernelGGL((k_affine_act<true, 4>), g, b, shm, (hipStream_t)stream, x + n0 * a.image, rp, y + n0 * a.image,
                               mean, invstd, gamma, beta, maxval, f, a);
        else if (small && (a.HW >= 4 || !has_bn) && small_kind() != 0) {
            if (small_kind() == 2)
                hipLaunchKernelGGL((k_affine_act_small<true, false, 1>), g, b, 0, (hipStream_t)stream, x + n0 * a.image, rp, y + n0 * a.image,
                                   mean, invstd, gamma, beta, maxval, f, a, reinterpret_cast<const float4 *>(prep));
            else
                hipLaunchKernelGGL((k_affine_act_small<false, false, 1>), g, b, 0, (hipStream_t)stream, x + n0 * a.image, rp, y + n0 * a.image,
                                   mean, invstd, gamma, beta, maxval, f, a, reinterpret_cast<const float4 *>(prep));
        } else if (small)
            hipLaunchKernelGGL((k_affine_act<false, 1>), g, b, shm, (hipStream_t)stream, x + n0 * a.image, rp, y + n0 * a.image,
                               mean, invstd, gamma, beta, maxval, f, a);
        else
            hipLaunchKernelGGL((k_affine_act<false, 4>), g, b, shm, (hipStream_t)stream, x + n0 * a.image, rp, y + n0 * a.image,
                               mean, invstd, gamma, beta, maxval, f, a);
    }
    return launch_rc();
}

size_t fp8q_affine_act_minmax_workspace_bytes(int64_t N, int64_t C, int64_t HW)
{
    AffineArgs a;
    if (N <= 0 || affine_args(N, C, HW, 0, false, false, &a) != FP8Q_OK) return 16;
    int64_t bx, by, sx, sy;
    affine_grid(N, a, false, &bx, &by);
    affine_grid_store(N, a, &sx, &sy);           // (the variant that writes t has its own grid: the larger of the two)
    const int64_t parts = bx * by > sx * sy ? bx * by : sx * sy;
    return kMinmaxWsHeader + (size_t)parts * 2 * sizeof(unsigned long long);   // header + two tagged granules per streaming block
}

static int affine_minmax_impl(const float *x, const float *residual, int64_t N, int64_t C, int64_t HW,
                              const float *mean, const float *invstd, const float *gamma, const float *beta,
                              int act, float *cur_min, float *cur_max, float *maxval_out, float *packed, int fold_mode,
                              double momentum, int first, void *ws, size_t ws_bytes, fp8q_stream_t stream,
                              int has_bn = -1, float *t_out = nullptr, float *lin_grid = nullptr, int lin_steps = 0,
                              double lin_lo = 0.0, double lin_hi = 0.0)
{
    if (has_bn < 0) has_bn = mean != nullptr;
    if (has_bn == 1 && (!invstd || !gamma || !beta)) return FP8Q_EINVAL;
    AffineArgs a;
    if (int rc = affine_args(N, C, HW, act, has_bn, residual != nullptr, &a)) return rc;
    if (N <= 0 || !x || !cur_min || !cur_max || fold_mode < 0 || fold_mode > 2) return FP8Q_EINVAL;
    if (!ws || ws_bytes < fp8q_affine_act_minmax_workspace_bytes(N, C, HW) || ((uintptr_t)ws & 7)) return FP8Q_EWORKSPACE;
    if ((((uintptr_t)x | (uintptr_t)residual) & 15) || ((uintptr_t)packed & 15)) return FP8Q_EINVAL;
    {
        static const bool debug_ws = [] {
            const char *e = getenv("FP8Q_DEBUG_WS");
            return e && atoi(e) != 0;
        }();
        if (debug_ws)
            if (int rc = fp8q_minmax_workspace_check(ws, fp8q_affine_act_minmax_workspace_bytes(N, C, HW), 0, stream)) return rc;
    }
    int64_t bx, by;
    if (t_out)
        affine_grid_store(N, a, &bx, &by);
    else
        affine_grid(N, a, false, &bx, &by);
    hipStream_t st = (hipStream_t)stream;
    FoldArgs fa;
    fa.mode = fold_mode;
    fa.first = first != 0;
    fa.om = (float)(1.0 - momentum);
    fa.mo = (float)momentum;
    fa.packed = packed;
    fa.status = (unsigned *)ws;
    fa.lin_grid = lin_grid;
    fa.lin_steps = lin_steps;
    fa.lin_C = 1;
    fa.lin_lo = lin_lo;
    fa.lin_hi = lin_hi;
    fold_debug_env(fa);
    const int nparts = (int)(bx * by);
    // one more block row for the reducer when there is more than one streaming block (by <= 65535 - 1: affine_grid)
    if (t_out)
        hipLaunchKernelGGL(k_affine_minmax<true>, dim3((unsigned)bx, (unsigned)(nparts > 1 ? by + 1 : by)), dim3(kBlock), 0, st, x,
                           residual, mean, invstd, gamma, beta, a, N, (unsigned long long *)((char *)ws + kMinmaxWsHeader), nparts,
                           next_minmax_tag(), cur_min, cur_max, maxval_out, fa, t_out);
    else
        hipLaunchKernelGGL(k_affine_minmax<false>, dim3((unsigned)bx, (unsigned)(nparts > 1 ? by + 1 : by)), dim3(kBlock), 0, st, x,
                           residual, mean, invstd, gamma, beta, a, N, (unsigned long long *)((char *)ws + kMinmaxWsHeader), nparts,
                           next_minmax_tag(), cur_min, cur_max, maxval_out, fa, nullptr);
    return launch_rc();
}

int fp8q_affine_act_minmax_f32(const float *x, const float *residual, int64_t N, int64_t C, int64_t HW,
                               const float *mean, const float *invstd, const float *gamma, const float *beta,
                               int act, float *cur_min, float *cur_max, float *maxval_out, int fold_mode,
                               double momentum, int first, void *ws, size_t ws_bytes, fp8q_stream_t stream)
{
    return affine_minmax_impl(x, residual, N, C, HW, mean, invstd, gamma, beta, act, cur_min, cur_max, maxval_out, nullptr,
                              fold_mode, momentum, first, ws, ws_bytes, stream);
}

int fp8q_affine_act_minmax_packed_f32(const float *x, const float *residual, int64_t N, int64_t C, int64_t HW,
                                      const float *mean, const float *invstd, const float *gamma, const float *beta,
                                      int act, float *cur_min, float *cur_max, float *maxval_out, float *packed,
                                      int fold_mode, double momentum, int first, void *ws, size_t ws_bytes,
                                      fp8q_stream_t stream)
{
    if (!packed) return FP8Q_EINVAL;
    return affine_minmax_impl(x, residual, N, C, HW, mean, invstd, gamma, beta, act, cur_min, cur_max, maxval_out, packed,
                              fold_mode, momentum, first, ws, ws_bytes, stream);
}

int fp8q_affine_act_minmax_linspace_f32(const float *x, const float *residual, float *t, int64_t N, int64_t C, int64_t HW,
                                        const float *alpha_beta, int act, float *cur_min, float *cur_max, float *maxval_out,
                                        float *grid, int n_cand, double lo_frac, double hi_frac, void *ws, size_t ws_bytes,
                                        fp8q_stream_t stream)
{
    if (!t || !grid || !maxval_out || n_cand < 2 || n_cand > (1 << 20) || ((uintptr_t)alpha_beta & 7) || ((uintptr_t)t & 15)) return FP8Q_EINVAL;
    return affine_minmax_impl(x, residual, N, C, HW, alpha_beta, nullptr, nullptr, nullptr, act, cur_min, cur_max, maxval_out, nullptr,
                              FP8Q_FOLD_CURRENT, 0.0, 1, ws, ws_bytes, stream, alpha_beta ? 2 : 0, t, grid, n_cand, lo_frac, hi_frac);
}

}  // extern "C"
