// fp8q_codec.hip -- N3: FP8 storage codes (encode / decode).
#include "fp8q_common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// N3: storage codes.  One row per blockIdx.y like k_quant_rows; a lane converts 16 consecutive
// elements per step (4 x 16-byte fp32 accesses <-> one 16-byte access of codes).
// encode: 4 B read + 1 B written per element; decode: 1 B read + 4 B written.
// ---------------------------------------------------------------------------------------------
template <bool ENCODE, bool NT>
__global__ void __launch_bounds__(kBlock)
k_codec_rows(const float *__restrict__ x, uint8_t *__restrict__ codes, float *__restrict__ y, int64_t inner,
             const float *__restrict__ maxval, int per_channel, QFmt f, int n_bits)
{
    __shared__ float2 lut[kLutMax];
    const int row = blockIdx.y, tid = threadIdx.x;
    const Chan cfull = make_chan(maxval[per_channel ? row : 0], f);
    for (int i = tid; i <= f.pmax; i += kBlock) lut[i] = lut_entry(cfull, i, f.M);
    __syncthreads();
    const ChanLite c = lite(cfull);
    const float pmaxf = (float)f.pmax;
    const int M = (int)f.M, sign_shift = f.sign_bits == 1 ? n_bits - 1 : -1;
    const float *xr = ENCODE ? x + (int64_t)row * inner : nullptr;
    float *yr = ENCODE ? nullptr : y + (int64_t)row * inner;
    uint8_t *cr = codes + (int64_t)row * inner;
    // Vector paths need the row's fp32 side 16-byte aligned (and the code side 4 / 16-byte); otherwise scalar.
    const uintptr_t fa = (uintptr_t)(ENCODE ? (const void *)xr : (const void *)yr);
    const bool vec = (fa & 15) == 0 && ((uintptr_t)cr & 3) == 0;
    const int64_t ngrp = vec ? inner >> 2 : 0;
    constexpr int U = 4;
    const vf4 *xv = reinterpret_cast<const vf4 *>(xr);
    vf4 *yv = reinterpret_cast<vf4 *>(yr);
    uint32_t *cw = reinterpret_cast<uint32_t *>(cr);
    if (ENCODE) {
        // encode: the wide side is the LOAD (strided 16-byte loads of 64 consecutive bytes per lane are absorbed
        // by L1); a lane converts 16 consecutive elements and stores their codes as one 16-byte word
        const bool vec16 = vec && ((uintptr_t)cr & 15) == 0;
        const int64_t ng16 = vec16 ? inner >> 4 : 0;
        for (int64_t g = (int64_t)blockIdx.x * kBlock + tid; g < ng16; g += (int64_t)gridDim.x * kBlock) {
            vf4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = xv[g * 4 + k];   // not nontemporal: the line's other quarters hit L1
            uint32_t w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float in[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
                w[k] = encode_group4(in, c, lut, pmaxf, f.qthr, M, sign_shift);
            }
            *reinterpret_cast<uint4 *>(cr + g * 16) = make_uint4(w[0], w[1], w[2], w[3]);
        }
        for (int64_t i = (ng16 << 4) + (int64_t)blockIdx.x * kBlock + tid; i < inner; i += (int64_t)gridDim.x * kBlock)
            cr[i] = (uint8_t)encode_one(xr[i], c, lut, pmaxf, f.qthr, M, sign_shift);
        return;
    }
    // decode: the wide side is the STORE: lane <-> 4-element group, dword code loads (1 KiB per block and
    // instruction), whole aligned 16-byte fp32 stores (4 KiB contiguous)
    for (int64_t base = (int64_t)blockIdx.x * (kBlock * U); base < ngrp; base += (int64_t)gridDim.x * (kBlock * U)) {
        {
            uint32_t w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t q = base + u * kBlock + tid;
                if (q < ngrp) w[u] = cw[q];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t q = base + u * kBlock + tid;
                if (q < ngrp)
                    st16<NT>(yv + q, vf4{decode_one(w[u] & 255u, lut, M, sign_shift),
                                         decode_one((w[u] >> 8) & 255u, lut, M, sign_shift),
                                         decode_one((w[u] >> 16) & 255u, lut, M, sign_shift),
                                         decode_one(w[u] >> 24, lut, M, sign_shift)});
            }
        }
    }
    for (int64_t i = (ngrp << 2) + (int64_t)blockIdx.x * kBlock + tid; i < inner; i += (int64_t)gridDim.x * kBlock)
        yr[i] = decode_one(cr[i], lut, M, sign_shift);
}


}  // namespace

extern "C" {

static int codec_launch(bool encode, const float *x, uint8_t *codes, float *y, int64_t C, int64_t inner,
                        const float *maxval, int64_t n_maxval, float mbits, int n_bits, int sign_bits,
                        fp8q_stream_t stream)
{
    if (C < 0 || inner < 0 || (n_maxval != 1 && n_maxval != C)) return FP8Q_EINVAL;
    if (n_bits > 8) return FP8Q_EUNSUPPORTED;   // a code is one byte (include/fp8q.h)
    QFmt f;
    if (int rc = make_fmt(mbits, n_bits, sign_bits, &f)) return rc;
    if (n_bits - sign_bits - (int)f.M < 1) return FP8Q_EUNSUPPORTED;   // no exponent bit: 2^(M+1) steps do not fit M bits
    if (C == 0 || inner == 0) return FP8Q_OK;
    if (!codes || !maxval || (encode ? !x : !y)) return FP8Q_EINVAL;
    const int per_channel = n_maxval != 1;
    if (!per_channel) {
        inner *= C;
        C = 1;
    }
    if (per_channel && inner < 2048) {   // short rows (weights): aligned chunks with per-row tables, not a block per row
        const int rc = fp8q_codec_flat_launch(encode, encode ? (const void *)x : (const void *)codes,
                                              encode ? (void *)codes : (void *)y, C, inner, maxval, f, n_bits,
                                              (hipStream_t)stream);
        if (rc != FP8Q_CODEC_NOT_FLAT) return rc;
    }
    for (int64_t c0 = 0; c0 < C; c0 += 65535) {
        const int64_t cn = (C - c0) < 65535 ? (C - c0) : 65535;
        const int64_t bx = balanced_blocks(cdiv(cdiv(inner, 16), kBlock), kTargetBlocks / cn);   // 4096 elements per block and step
        const dim3 g((unsigned)bx, (unsigned)cn), b(kBlock);
        const bool nt = C * inner * 4 >= kNtBytes;
        const float *mvp = maxval + (per_channel ? c0 : 0);
        if (encode && nt)
            hipLaunchKernelGGL((k_codec_rows<true, true>), g, b, 0, (hipStream_t)stream, x + c0 * inner,
                               codes + c0 * inner, (float *)nullptr, inner, mvp, per_channel, f, n_bits);
        else if (encode)
            hipLaunchKernelGGL((k_codec_rows<true, false>), g, b, 0, (hipStream_t)stream, x + c0 * inner,
                               codes + c0 * inner, (float *)nullptr, inner, mvp, per_channel, f, n_bits);
        else if (nt)
            hipLaunchKernelGGL((k_codec_rows<false, true>), g, b, 0, (hipStream_t)stream, (const float *)nullptr,
                               codes + c0 * inner, y + c0 * inner, inner, mvp, per_channel, f, n_bits);
        else
            hipLaunchKernelGGL((k_codec_rows<false, false>), g, b, 0, (hipStream_t)stream, (const float *)nullptr,
                               codes + c0 * inner, y + c0 * inner, inner, mvp, per_channel, f, n_bits);
    }
    return launch_rc();
}

int fp8q_encode_u8(const float *x, uint8_t *codes, int64_t C, int64_t inner, const float *maxval,
                   int64_t n_maxval, float mbits, int n_bits, int sign_bits, fp8q_stream_t stream)
{
    return codec_launch(true, x, codes, nullptr, C, inner, maxval, n_maxval, mbits, n_bits, sign_bits, stream);
}

int fp8q_decode_u8(const uint8_t *codes, float *y, int64_t C, int64_t inner, const float *maxval,
                   int64_t n_maxval, float mbits, int n_bits, int sign_bits, fp8q_stream_t stream)
{
    return codec_launch(false, nullptr, const_cast<uint8_t *>(codes), y, C, inner, maxval, n_maxval, mbits,
                        n_bits, sign_bits, stream);
}

}  // extern "C"
