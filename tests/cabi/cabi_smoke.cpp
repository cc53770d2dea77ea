// cabi_smoke.cpp -- the drop-in boundary used the way a non-Python host would use it: plain HIP runtime
// calls for memory, plain pointers and sizes into libfp8q_hip.so (include/fp8q.h), no torch anywhere.
// Checks K1 (by value and with a device-resident mantissa width), the float64 lane, the device-side MSE grid + winner selection, the one-call MSE calibration step (with and without the BN + activation pre-stage), the folded-BN epilogue, the fused min/max+quantize, the folding min/max (zeroed workspace, packed ranges, workspace check), the multi-tensor call and its prepared
// plan, the storage codec and the FP-MSE grid search against the CPU
// oracle (libfp8q_oracle.so, test infrastructure) bit for bit.  Built by tests/test_cabi_and_host.py
// (hipcc cross-compiles it on the CPU box); run by the -m gpu test of the same file.
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/fp8q.h"

extern "C" {
int orc_quantize_f32(const float *x, float *y, int64_t C, int64_t inner, const float *maxval, int64_t n_maxval,
                     float mbits, int n_bits, int sign_bits);
int orc_minmax_f32(const float *x, int64_t C, int64_t inner, float *mn, float *mx);
int orc_absmax_f32(const float *mn, const float *mx, int64_t C, float *maxval);
int orc_encode_u8(const float *x, uint8_t *codes, int64_t C, int64_t inner, const float *maxval, int64_t n_maxval,
                  float mbits, int n_bits, int sign_bits);
int orc_decode_u8(const uint8_t *codes, float *y, int64_t C, int64_t inner, const float *maxval, int64_t n_maxval,
                  float mbits, int n_bits, int sign_bits);
int orc_mse_grid_f32(const float *x, int64_t C, int64_t inner, const float *grid, int64_t n_cand, const float *mbits,
                     int n_m, int n_bits, int sign_bits, float *mses);
int orc_quantize_f64(const double *x, double *y, int64_t C, int64_t inner, const float *maxval, int64_t n_maxval,
                     float mbits, int n_bits, int sign_bits);
int orc_minmax_f64(const double *x, int64_t C, int64_t inner, double *mn, double *mx);
int orc_sse_grid_f64(const double *x, int64_t C, int64_t inner, const float *grid, int64_t n_cand, const float *mbits,
                     int n_m, int n_bits, int sign_bits, double *out, int reduce_sum);
int orc_affine_act_f32(const float *x, const float *res, float *y, int64_t N, int64_t C, int64_t HW, const float *mean,
                       const float *invstd, const float *gamma, const float *beta, int act);
}

#define CK(x) do { int e_ = (int)(x); if (e_ != 0) { printf("FAIL %s -> %d (%s) at line %d\n", #x, e_, fp8q_strerror(e_), __LINE__); return 1; } } while (0)

static int same_bits(const float *a, const float *b, size_t n, const char *what)
{
    for (size_t i = 0; i < n; ++i) {
        uint32_t ua, ub;
        memcpy(&ua, a + i, 4);
        memcpy(&ub, b + i, 4);
        const int nan_a = a[i] != a[i], nan_b = b[i] != b[i];
        if (nan_a != nan_b || (!nan_a && ua != ub)) {
            printf("FAIL %s: element %zu: %a vs %a\n", what, i, a[i], b[i]);
            return 0;
        }
    }
    printf("ok   %s (%zu elements bit-identical)\n", what, n);
    return 1;
}

int main()
{
    const int64_t C = 300, inner = 147, n = C * inner;
    float *x = (float *)malloc(n * 4), *y = (float *)malloc(n * 4), *ref = (float *)malloc(n * 4);
    float mn[300], mx[300], mv[300], gmn[300], gmx[300], gmv[300];
    uint32_t s = 12345u;
    for (int64_t i = 0; i < n; ++i) {   // LCG, two uniforms -> roughly bell-shaped, scaled per row
        s = s * 1664525u + 1013904223u;
        const float u1 = (float)(s >> 8) / 16777216.0f;
        s = s * 1664525u + 1013904223u;
        const float u2 = (float)(s >> 8) / 16777216.0f;
        x[i] = (u1 + u2 - 1.0f) * (0.05f + 0.01f * (float)(i / inner % 17));
    }
    x[5] = 0.0f;
    x[6] = -0.0f;
    orc_minmax_f32(x, C, inner, mn, mx);
    orc_absmax_f32(mn, mx, C, mv);

    float *dx, *dy, *dmv, *dmn, *dmx, *dmvo;
    void *ws;
    CK(hipMalloc((void **)&dx, n * 4));
    CK(hipMalloc((void **)&dy, n * 4));
    CK(hipMalloc((void **)&dmv, C * 4));
    CK(hipMalloc((void **)&dmn, C * 4));
    CK(hipMalloc((void **)&dmx, C * 4));
    CK(hipMalloc((void **)&dmvo, C * 4));
    size_t wsb = fp8q_minmax_workspace_bytes(C, inner);
    if (fp8q_minmax_workspace_bytes(1, n) > wsb) wsb = fp8q_minmax_workspace_bytes(1, n);
    CK(hipMalloc(&ws, wsb));
    CK(hipMemset(ws, 0, wsb));   // min/max workspace: zero once, every call leaves it zero (fp8q.h)
    CK(hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dmv, mv, C * 4, hipMemcpyHostToDevice));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    int ok = 1;
    printf("libfp8q version %d\n", fp8q_version());

    for (int M = 2; M <= 4; ++M) {   // K1, per-channel fixed ranges
        CK(fp8q_quantize_f32(dx, dy, C, inner, dmv, C, (float)M, 8, 1, st));
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(y, dy, n * 4, hipMemcpyDeviceToHost));
        orc_quantize_f32(x, ref, C, inner, mv, C, (float)M, 8, 1);
        char what[64];
        snprintf(what, sizeof what, "fp8q_quantize_f32 per-channel M=%d", M);
        ok &= same_bits(y, ref, n, what);
    }
    // fused min/max + quantize
    CK(fp8q_minmax_quantize_f32(dx, dy, C, inner, dmn, dmx, dmvo, 2.0f, 8, 1, st));
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(y, dy, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gmn, dmn, C * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gmx, dmx, C * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gmv, dmvo, C * 4, hipMemcpyDeviceToHost));
    orc_quantize_f32(x, ref, C, inner, mv, C, 2.0f, 8, 1);
    ok &= same_bits(y, ref, n, "fp8q_minmax_quantize_f32 values");
    ok &= same_bits(gmn, mn, C, "fp8q_minmax_quantize_f32 row_min");
    ok &= same_bits(gmx, mx, C, "fp8q_minmax_quantize_f32 row_max");
    ok &= same_bits(gmv, mv, C, "fp8q_minmax_quantize_f32 maxval");
    // per-tensor min/max with the split kernels + workspace, then per-tensor K1
    float tmn, tmx, tmv, rmn, rmx, rmv;
    CK(fp8q_minmax_f32(dx, 1, n, dmn, dmx, dmvo, FP8Q_FOLD_CURRENT, 0.9, 1, ws, fp8q_minmax_workspace_bytes(1, n), st));
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(&tmn, dmn, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&tmx, dmx, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&tmv, dmvo, 4, hipMemcpyDeviceToHost));
    orc_minmax_f32(x, 1, n, &rmn, &rmx);
    orc_absmax_f32(&rmn, &rmx, 1, &rmv);
    ok &= same_bits(&tmn, &rmn, 1, "fp8q_minmax_f32 per-tensor min");
    ok &= same_bits(&tmx, &rmx, 1, "fp8q_minmax_f32 per-tensor max");
    ok &= same_bits(&tmv, &rmv, 1, "fp8q_minmax_f32 per-tensor maxval");
    CK(fp8q_quantize_f32(dx, dy, C, inner, dmvo, 1, 3.0f, 8, 1, st));
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(y, dy, n * 4, hipMemcpyDeviceToHost));
    orc_quantize_f32(x, ref, C, inner, &rmv, 1, 3.0f, 8, 1);
    ok &= same_bits(y, ref, n, "fp8q_quantize_f32 per-tensor M=3");
    // batch-sharded calibration as a non-Python host would drive it: the min/max kernel also writes the packed
    // {-min, max, nan flags} record (the operand of ONE all-reduce(MAX) between ranks); here "rank 2" is a second call
    // on the negated tensor's statistics emulated on the host; the unpack kernel restores min / max / maxval
    {
        float *dpk;
        CK(hipMalloc((void **)&dpk, 16));
        CK(fp8q_minmax_packed_f32(dx, 1, n, dmn, dmx, dmvo, dpk, FP8Q_FOLD_CURRENT, 0.9, 1, ws, fp8q_minmax_workspace_bytes(1, n), st));
        CK(hipStreamSynchronize(st));
        float pk[4];
        CK(hipMemcpy(pk, dpk, 16, hipMemcpyDeviceToHost));
        const float want[4] = {-rmn, rmx, 0.0f, 0.0f};
        ok &= same_bits(pk, want, 4, "fp8q_minmax_packed_f32 record");
        const float other[4] = {0.5f * -rmn, 2.0f * rmx, 0.0f, 0.0f};   // another rank's record: max() per component
        for (int k = 0; k < 4; ++k) pk[k] = pk[k] > other[k] ? pk[k] : other[k];
        CK(hipMemcpy(dpk, pk, 16, hipMemcpyHostToDevice));
        CK(fp8q_ranges_unpack_f32(dpk, 1, dmn, dmx, dmvo, st));
        CK(hipStreamSynchronize(st));
        float umn, umx, umv, wmx = 2.0f * rmx, wmv;
        CK(hipMemcpy(&umn, dmn, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&umx, dmx, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&umv, dmvo, 4, hipMemcpyDeviceToHost));
        orc_absmax_f32(&rmn, &wmx, 1, &wmv);
        ok &= same_bits(&umn, &rmn, 1, "fp8q_ranges_unpack_f32 min");
        ok &= same_bits(&umx, &wmx, 1, "fp8q_ranges_unpack_f32 max");
        ok &= same_bits(&umv, &wmv, 1, "fp8q_ranges_unpack_f32 maxval");
        CK(hipFree(dpk));
        // the workspace is clean between calls; a timeout count left by a reducer surfaces as FP8Q_ETIMEDOUT
        CK(fp8q_minmax_workspace_check(ws, wsb, 0, st));
        const unsigned one = 1u;
        CK(hipMemcpy(ws, &one, 4, hipMemcpyHostToDevice));
        const int rc = fp8q_minmax_workspace_check(ws, wsb, 1, st);
        if (rc != FP8Q_ETIMEDOUT) { printf("FAIL workspace check: %d\n", rc); ok = 0; }
        CK(fp8q_minmax_workspace_check(ws, wsb, 0, st));
        printf("ok   fp8q_minmax_workspace_check (clean, FP8Q_ETIMEDOUT reported and cleared)\n");
    }
    // multi-tensor: the same buffer as two tensors with different formats
    fp8q_tensor_desc d[2];
    const int64_t C0 = 100, C1 = C - C0;
    d[0] = fp8q_tensor_desc{dx, dy, dmv, C0, inner, C0, 2.0f, 8, 1};
    d[1] = fp8q_tensor_desc{dx + C0 * inner, dy + C0 * inner, dmv + C0, C1, inner, C1, 4.0f, 8, 1};
    CK(fp8q_multi_quantize_f32(d, 2, st));
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(y, dy, n * 4, hipMemcpyDeviceToHost));
    orc_quantize_f32(x, ref, C0, inner, mv, C0, 2.0f, 8, 1);
    orc_quantize_f32(x + C0 * inner, ref + C0 * inner, C1, inner, mv + C0, C1, 4.0f, 8, 1);
    ok &= same_bits(y, ref, n, "fp8q_multi_quantize_f32 (2 tensors, E5M2 + E3M4)");
    // prepared multi-tensor plan: built once, launched twice (the second time after the input changed in place)
    {
        fp8q_multi_plan *plan = nullptr;
        CK(fp8q_multi_plan_create(d, 2, &plan));
        if (fp8q_multi_plan_launches(plan) != 1) { printf("FAIL: one launch expected for two batchable tensors\n"); ok = 0; }
        CK(hipMemset(dy, 0, n * 4));
        CK(fp8q_multi_plan_launch(plan, st));
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(y, dy, n * 4, hipMemcpyDeviceToHost));
        ok &= same_bits(y, ref, n, "fp8q_multi_plan_launch");
        for (int64_t i = 0; i < n; ++i) x[i] *= 0.5f;                 // "an optimizer step": same storage, new contents
        CK(hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice));
        CK(fp8q_multi_plan_launch(plan, st));
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(y, dy, n * 4, hipMemcpyDeviceToHost));
        orc_quantize_f32(x, ref, C0, inner, mv, C0, 2.0f, 8, 1);
        orc_quantize_f32(x + C0 * inner, ref + C0 * inner, C1, inner, mv + C0, C1, 4.0f, 8, 1);
        ok &= same_bits(y, ref, n, "fp8q_multi_plan_launch after an in-place update");
        fp8q_multi_plan_destroy(plan);
    }
    // storage codes (per-channel short rows: the chunked kernel) against the oracle's bytes and decoded values
    {
        uint8_t *codes = (uint8_t *)malloc(n), *rcodes = (uint8_t *)malloc(n), *dcodes;
        CK(hipMalloc((void **)&dcodes, n));
        CK(fp8q_encode_u8(dx, dcodes, C, inner, dmv, C, 2.0f, 8, 1, st));
        CK(fp8q_decode_u8(dcodes, dy, C, inner, dmv, C, 2.0f, 8, 1, st));
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(codes, dcodes, n, hipMemcpyDeviceToHost));
        CK(hipMemcpy(y, dy, n * 4, hipMemcpyDeviceToHost));
        orc_encode_u8(x, rcodes, C, inner, mv, C, 2.0f, 8, 1);
        orc_decode_u8(rcodes, ref, C, inner, mv, C, 2.0f, 8, 1);
        if (memcmp(codes, rcodes, n) != 0) { printf("FAIL fp8q_encode_u8: codes differ\n"); ok = 0; }
        else printf("ok   fp8q_encode_u8 (%lld codes identical)\n", (long long)n);
        ok &= same_bits(y, ref, n, "fp8q_decode_u8");
    }
    // FP-MSE grid search: 7 candidate ranges x 2 mantissa widths, per tensor (the long-row kernel) and per channel
    {
        const int n_cand = 7, n_m = 2;
        const float mb[2] = {2.0f, 3.0f};
        float grid1[7], mses[14], rm[14];
        for (int i = 0; i < n_cand; ++i) grid1[i] = 0.02f * (float)(i + 1);
        float *dgrid, *dmses;
        void *mws;
        const size_t mwsb = fp8q_mse_workspace_bytes(1, n, n_cand, n_m);
        CK(hipMalloc((void **)&dgrid, sizeof(grid1)));
        CK(hipMalloc((void **)&dmses, sizeof(mses)));
        CK(hipMalloc(&mws, mwsb));
        CK(hipMemcpy(dgrid, grid1, sizeof(grid1), hipMemcpyHostToDevice));
        CK(hipMemset(dmses, 0, sizeof(mses)));
        CK(fp8q_mse_grid_f32(dx, 1, n, dgrid, n_cand, mb, n_m, 8, 1, dmses, mws, mwsb, st));
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(mses, dmses, sizeof(mses), hipMemcpyDeviceToHost));
        memset(rm, 0, sizeof(rm));
        orc_mse_grid_f32(x, 1, n, grid1, n_cand, mb, n_m, 8, 1, rm);
        int good = 1;
        for (int i = 0; i < 14; ++i)
            if (!(mses[i] >= rm[i] * (1.0f - 1e-5f) && mses[i] <= rm[i] * (1.0f + 1e-5f))) { printf("FAIL fp8q_mse_grid_f32[%d]: %g vs %g\n", i, mses[i], rm[i]); good = 0; }
        if (good) printf("ok   fp8q_mse_grid_f32 (14 mean squared errors within 1e-5 of the oracle)\n");
        ok &= good;
    }
    // K1 with the mantissa width in device memory (the MSE estimator's vote never leaves the GPU)
    {
        float *dmb;
        CK(hipMalloc((void **)&dmb, 4));
        for (int M = 1; M <= 5; M += 2) {
            const float mbv = (float)M + 0.3f;      // rounded half-to-even on the device, as the host does by value
            CK(hipMemcpy(dmb, &mbv, 4, hipMemcpyHostToDevice));
            CK(fp8q_quantize_dm_f32(dx, dy, C, inner, dmv, C, dmb, 8, 1, st));
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(y, dy, n * 4, hipMemcpyDeviceToHost));
            orc_quantize_f32(x, ref, C, inner, mv, C, mbv, 8, 1);
            char what[64];
            snprintf(what, sizeof what, "fp8q_quantize_dm_f32 (device mantissa bits %.1f)", mbv);
            ok &= same_bits(y, ref, n, what);
        }
        CK(hipFree(dmb));
    }
    // sign_bits decided on the device (allow_unsigned without the host round trip of fp8_quantizer.py:216-225)
    {
        unsigned char *dflag, hflag;
        float *dmin;
        CK(hipMalloc((void **)&dflag, 1));
        CK(hipMalloc((void **)&dmin, 3 * sizeof(float)));
        const float mins[3][3] = {{-0.5f, 0.0f, 2.0f}, {0.0f, 1.0f, 2.0f}, {-1.0f, -2.0f, 3.0f}};
        const int want[3] = {1, 0, 0};             // signed; every minimum >= 0: unsigned; unsigned for good
        hflag = 1;
        CK(hipMemcpy(dflag, &hflag, 1, hipMemcpyHostToDevice));
        for (int k = 0; k < 3; ++k) {
            CK(hipMemcpy(dmin, mins[k], sizeof(mins[k]), hipMemcpyHostToDevice));
            CK(fp8q_sign_fold_u8(dmin, 3, dflag, st));
            CK(fp8q_quantize_ds_f32(dx, dy, C, inner, dmv, C, 3.0f, 8, dflag, st));
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(&hflag, dflag, 1, hipMemcpyDeviceToHost));
            CK(hipMemcpy(y, dy, n * 4, hipMemcpyDeviceToHost));
            orc_quantize_f32(x, ref, C, inner, mv, C, 3.0f, 8, want[k]);
            char what[80];
            snprintf(what, sizeof what, "fp8q_sign_fold_u8 + fp8q_quantize_ds_f32 (step %d: sign_bits %d)", k, want[k]);
            if ((int)hflag != want[k]) { printf("FAIL %s: flag %d\n", what, (int)hflag); ok = 0; }
            ok &= same_bits(y, ref, n, what);
            // ... and with the mantissa width in device memory as well
            float *dmb2;
            const float mbv = 2.0f + (float)k;
            CK(hipMalloc((void **)&dmb2, 4));
            CK(hipMemcpy(dmb2, &mbv, 4, hipMemcpyHostToDevice));
            CK(fp8q_quantize_dms_f32(dx, dy, C, inner, dmv, C, dmb2, 8, dflag, st));
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(y, dy, n * 4, hipMemcpyDeviceToHost));
            orc_quantize_f32(x, ref, C, inner, mv, C, mbv, 8, want[k]);
            snprintf(what, sizeof what, "fp8q_quantize_dms_f32 (device width %.0f, device sign %d)", mbv, want[k]);
            ok &= same_bits(y, ref, n, what);
            CK(hipFree(dmb2));
        }
        CK(hipFree(dflag));
        CK(hipFree(dmin));
    }
    // device-side search grid + winner selection (sync-free MSE calibration)
    {
        const int n_cand = 111, n_m = 3;
        const int64_t Cs = 5;
        const float mb[3] = {2.0f, 3.0f, 4.0f};
        float mxs[5] = {0.5f, 1.0f, 0.7361f, 2.5f, 0.01f};
        float *dmxs, *dgrid, *dm, *dmbo, *dmvs, *dxm;
        int *dvote;
        void *sws;
        float *hm = (float *)malloc(sizeof(float) * n_m * n_cand * Cs), *hgrid = (float *)malloc(sizeof(float) * n_cand * Cs);
        CK(hipMalloc((void **)&dmxs, sizeof(mxs)));
        CK(hipMalloc((void **)&dgrid, sizeof(float) * n_cand * Cs));
        CK(hipMalloc((void **)&dm, sizeof(float) * n_m * n_cand * Cs));
        CK(hipMalloc((void **)&dmbo, 4));
        CK(hipMalloc((void **)&dvote, 4));
        CK(hipMalloc((void **)&dmvs, sizeof(float) * Cs));
        CK(hipMalloc((void **)&dxm, sizeof(float) * Cs));
        CK(hipMalloc(&sws, fp8q_mse_select_workspace_bytes(Cs, n_m)));
        CK(hipMemset(sws, 0, fp8q_mse_select_workspace_bytes(Cs, n_m)));   // header contract: zero before the first call
        CK(hipMemcpy(dmxs, mxs, sizeof(mxs), hipMemcpyHostToDevice));
        CK(fp8q_mse_linspace_f32(dmxs, Cs, n_cand, 0.1, 1.2, dgrid, st));
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(hgrid, dgrid, sizeof(float) * n_cand * Cs, hipMemcpyDeviceToHost));
        int good = 1;
        for (int64_t c = 0; c < Cs; ++c) {     // torch.linspace endpoints: fl32(0.1 * mx), fl32(1.2 * mx), ascending in between
            const float lo = (float)(0.1 * (double)mxs[c]), hi = (float)(1.2 * (double)mxs[c]);
            if (hgrid[c] != lo || hgrid[(n_cand - 1) * Cs + c] != hi) good = 0;
            for (int i = 1; i < n_cand; ++i)
                if (!(hgrid[i * Cs + c] > hgrid[(i - 1) * Cs + c])) good = 0;
        }
        // a table with a known winner: width 1 wins in channels 0..2, width 2 in 3..4 -> vote = 1; argmin index 10 + c
        for (int m = 0; m < n_m; ++m)
            for (int i = 0; i < n_cand; ++i)
                for (int64_t c = 0; c < Cs; ++c)
                    hm[(m * n_cand + i) * Cs + c] = 1.0f + 0.001f * (float)i + ((m == (c < 3 ? 1 : 2)) && i == 10 + c ? -0.5f : 0.0f);
        CK(hipMemcpy(dm, hm, sizeof(float) * n_m * n_cand * Cs, hipMemcpyHostToDevice));
        CK(fp8q_mse_select_f32(dm, dgrid, Cs, n_cand, mb, n_m, 1, dmbo, dvote, dmvs, dxm, sws, fp8q_mse_select_workspace_bytes(Cs, n_m), st));
        CK(hipStreamSynchronize(st));
        float mbo, gmvs[5], gxm[5];
        int vote;
        CK(hipMemcpy(&mbo, dmbo, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&vote, dvote, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(gmvs, dmvs, sizeof(gmvs), hipMemcpyDeviceToHost));
        CK(hipMemcpy(gxm, dxm, sizeof(gxm), hipMemcpyDeviceToHost));
        if (vote != 1 || mbo != 3.0f) good = 0;
        for (int64_t c = 0; c < Cs; ++c) {     // channels 3, 4 voted for another width: THEIR argmin under the winning width is index 0
            const int want = c < 3 ? 10 + (int)c : 0;
            if (gmvs[c] != hgrid[want * Cs + c] || gxm[c] != -gmvs[c]) good = 0;
        }
        printf(good ? "ok   fp8q_mse_linspace_f32 + fp8q_mse_select_f32 (grid endpoints, vote, per-channel argmin)\n"
                    : "FAIL fp8q_mse_linspace_f32 / fp8q_mse_select_f32\n");
        ok &= good;
    }
    // one-call MSE calibration step (fp8q_mse_calibrate_f32): per-channel rows and a per-tensor row behind a folded BN + ReLU6,
    // against the entry points it replaces called one by one (bit for bit) -- state block and workspaces caller-owned
    {
        const int n_cand = 111, n_m = 2;
        const float mb[2] = {2.0f, 3.0f};
        int good = 1;
        for (int variant = 0; variant < 2; ++variant) {
            const int64_t Cc = variant == 0 ? C : 1, in_c = variant == 0 ? inner : n;
            const int64_t N_ = 3, Cb = 100, HW = n / (N_ * Cb);               // variant 1: x as [3, 100, 147] (C * HW a multiple of 4)
            const size_t nblk = (size_t)(5 * Cc + n_cand * Cc + n_m * n_cand * Cc + 2);
            float *blkA, *blkB, *dt, *dyA, *dyB, *dmbA, *dmbB, *dab = nullptr;
            void *w0, *w2, *w3;
            size_t b0 = 0, b2 = 0;
            const size_t b3 = fp8q_mse_calibrate_workspace_bytes(Cc, in_c, n_cand, n_m, &b0, &b2);
            if (variant == 1) {
                const size_t ba = fp8q_affine_act_minmax_workspace_bytes(N_, Cb, HW);
                if (ba > b0) b0 = ba;
            }
            CK(hipMalloc((void **)&blkA, nblk * 4));
            CK(hipMalloc((void **)&blkB, nblk * 4));
            CK(hipMalloc((void **)&dt, n * 4));
            CK(hipMalloc((void **)&dyA, n * 4));
            CK(hipMalloc((void **)&dyB, n * 4));
            CK(hipMalloc((void **)&dmbA, 4));
            CK(hipMalloc((void **)&dmbB, 4));
            CK(hipMalloc(&w0, b0 + 16));
            CK(hipMalloc(&w2, b2));
            CK(hipMalloc(&w3, b3));
            CK(hipMemset(w0, 0, b0 + 16));           // zero once: the contract of the min/max workspace ...
            CK(hipMemset(w2, 0, b2));                // ... and of the selection workspace's ticket block
            auto state = [&](float *b, float *mbits) {
                fp8q_mse_state s_;
                s_.cur_min = b;
                s_.cur_max = b + Cc;
                s_.absmax = b + 2 * Cc;
                s_.maxval = b + 3 * Cc;
                s_.xmin = b + 4 * Cc;
                s_.grid = b + 5 * Cc;
                s_.mses = b + 5 * Cc + n_cand * Cc;
                s_.mbits = mbits;
                s_.vote = (int *)(b + 5 * Cc + n_cand * Cc + n_m * n_cand * Cc);
                return s_;
            };
            const fp8q_mse_state sa = state(blkA, dmbA), sb = state(blkB, dmbB);
            fp8q_affine_pre pre;
            float hab[200];
            if (variant == 1) {
                for (int c = 0; c < 100; ++c) {
                    hab[2 * c] = 0.5f + 0.01f * (float)c;
                    hab[2 * c + 1] = 0.1f - 0.003f * (float)c;
                }
                CK(hipMalloc((void **)&dab, sizeof(hab)));
                CK(hipMemcpy(dab, hab, sizeof(hab), hipMemcpyHostToDevice));
                pre.x = dx;
                pre.residual = nullptr;
                pre.alpha_beta = dab;
                pre.N = N_;
                pre.C = Cb;
                pre.HW = HW;
                pre.act = 2;
            }
            for (int batch = 0; batch < 2; ++batch) {        // the second batch accumulates into the same tables
                // A: one call
                CK(fp8q_mse_calibrate_f32(variant == 1 ? dt : dx, dyA, Cc, in_c, &sa, batch == 0, n_cand, mb, n_m, 8, 1, variant == 1 ? &pre : nullptr,
                                          w0, b0, w2, b2, w3, b3, st));
                // B: the entry points it enqueues, one by one
                const float *src = dx;
                if (variant == 1) {
                    if (batch == 0)
                        CK(fp8q_affine_act_minmax_linspace_f32(dx, nullptr, dt, N_, Cb, HW, dab, 2, sb.cur_min, sb.cur_max, sb.absmax, sb.grid, n_cand, 0.1,
                                                               1.2, w0, b0, st));
                    else
                        CK(fp8q_affine_act_f32(dx, nullptr, dt, N_, Cb, HW, dab, 2, st));
                    src = dt;
                } else if (batch == 0) {
                    CK(fp8q_minmax_linspace_f32(dx, Cc, in_c, sb.cur_min, sb.cur_max, sb.absmax, sb.grid, n_cand, 0.1, 1.2, w0, b0, st));
                }
                if (batch == 0) CK(hipMemsetAsync(sb.mses, 0, sizeof(float) * n_m * n_cand * Cc, st));
                CK(fp8q_mse_grid_f32(src, Cc, in_c, sb.grid, n_cand, mb, n_m, 8, 1, sb.mses, w3, b3, st));
                CK(fp8q_mse_select_f32(sb.mses, sb.grid, Cc, n_cand, mb, n_m, 1, sb.mbits, sb.vote, sb.maxval, sb.xmin, w2, b2, st));
                CK(fp8q_quantize_dm_f32(src, dyB, Cc, in_c, sb.maxval, Cc, sb.mbits, 8, 1, st));
                CK(hipStreamSynchronize(st));
                float *ha = (float *)malloc(nblk * 4), *hb = (float *)malloc(nblk * 4);
                CK(hipMemcpy(ha, blkA, nblk * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hb, blkB, nblk * 4, hipMemcpyDeviceToHost));
                good &= memcmp(ha, hb, (nblk - 2) * 4) == 0 && memcmp(ha + nblk - 2, hb + nblk - 2, 4) == 0;   // tables, ranges, vote
                float ma, mbv;
                CK(hipMemcpy(&ma, dmbA, 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(&mbv, dmbB, 4, hipMemcpyDeviceToHost));
                good &= ma == mbv && (ma == 2.0f || ma == 3.0f);
                CK(hipMemcpy(y, dyA, n * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(ref, dyB, n * 4, hipMemcpyDeviceToHost));
                good &= memcmp(y, ref, n * 4) == 0;
                if (variant == 0 && batch == 0) {            // and the table itself against the oracle (1e-5, as K4 promises)
                    float *rm = (float *)calloc((size_t)n_m * n_cand * Cc, 4);
                    orc_mse_grid_f32(x, Cc, in_c, ha + 5 * Cc, n_cand, mb, n_m, 8, 1, rm);
                    const float *tab = ha + 5 * Cc + n_cand * Cc;
                    for (int64_t i = 0; i < n_m * n_cand * Cc; ++i)
                        if (rm[i] == rm[i] && !(tab[i] >= rm[i] * (1.0f - 1e-4f) && tab[i] <= rm[i] * (1.0f + 1e-4f))) {
                            if (good) printf("FAIL fp8q_mse_calibrate_f32 table[%lld]: %g vs %g\n", (long long)i, tab[i], rm[i]);
                            good = 0;
                        }
                    free(rm);
                }
                free(ha);
                free(hb);
            }
            hipFree(blkA), hipFree(blkB), hipFree(dt), hipFree(dyA), hipFree(dyB), hipFree(dmbA), hipFree(dmbB), hipFree(w0), hipFree(w2), hipFree(w3);
            if (dab) hipFree(dab);
        }
        printf(good ? "ok   fp8q_mse_calibrate_f32 (per-channel rows; per-tensor behind folded BN + ReLU6: two batches each == the separate entry points, bit for bit)\n"
                    : "FAIL fp8q_mse_calibrate_f32 vs the separate entry points\n");
        ok &= good;
    }
    // the float64 lane: K1, row min/max and the candidate search on doubles, bit for bit / to 1e-13 against the oracle
    {
        double *xd = (double *)malloc(n * 8), *yd = (double *)malloc(n * 8), *rd = (double *)malloc(n * 8), *dxd, *dyd, *dmm, *dsse;
        for (int64_t i = 0; i < n; ++i) xd[i] = (double)x[i] * 1.000000123;
        CK(hipMalloc((void **)&dxd, n * 8));
        CK(hipMalloc((void **)&dyd, n * 8));
        CK(hipMalloc((void **)&dmm, 16));
        CK(hipMemcpy(dxd, xd, n * 8, hipMemcpyHostToDevice));
        CK(fp8q_quantize_f64(dxd, dyd, C, inner, dmv, C, 3.0f, 8, 1, st));
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(yd, dyd, n * 8, hipMemcpyDeviceToHost));
        orc_quantize_f64(xd, rd, C, inner, mv, C, 3.0f, 8, 1);
        int good = memcmp(yd, rd, n * 8) == 0;
        printf(good ? "ok   fp8q_quantize_f64 (%lld doubles bit-identical)\n" : "FAIL fp8q_quantize_f64\n", (long long)n);
        ok &= good;
        void *w64;
        const size_t w64b = fp8q_minmax_f64_workspace_bytes(1, n);
        CK(hipMalloc(&w64, w64b));
        CK(fp8q_minmax_f64(dxd, 1, n, dmm, dmm + 1, w64, w64b, st));
        CK(hipStreamSynchronize(st));
        double gmm[2], rmn64, rmx64;
        CK(hipMemcpy(gmm, dmm, 16, hipMemcpyDeviceToHost));
        orc_minmax_f64(xd, 1, n, &rmn64, &rmx64);
        good = gmm[0] == rmn64 && gmm[1] == rmx64;
        printf(good ? "ok   fp8q_minmax_f64\n" : "FAIL fp8q_minmax_f64\n");
        ok &= good;
        const int n_cand = 50;
        const float mb1[1] = {3.0f};
        float g50[50];
        for (int i = 0; i < n_cand; ++i) g50[i] = 0.01f * (float)(i + 1);
        float *dg50;
        void *sw;
        const size_t swb = fp8q_mse_f64_workspace_bytes(1, n, n_cand, 1);
        CK(hipMalloc((void **)&dg50, sizeof(g50)));
        CK(hipMalloc((void **)&dsse, n_cand * 8));
        CK(hipMalloc(&sw, swb));
        CK(hipMemcpy(dg50, g50, sizeof(g50), hipMemcpyHostToDevice));
        CK(hipMemset(dsse, 0, n_cand * 8));
        CK(fp8q_mse_grid_f64(dxd, 1, n, dg50, n_cand, mb1, 1, 8, 1, dsse, 1, sw, swb, st));
        CK(hipStreamSynchronize(st));
        double sse[50], rsse[50];
        CK(hipMemcpy(sse, dsse, sizeof(sse), hipMemcpyDeviceToHost));
        memset(rsse, 0, sizeof(rsse));
        orc_sse_grid_f64(xd, 1, n, g50, n_cand, mb1, 1, 8, 1, rsse, 1);
        good = 1;
        for (int i = 0; i < n_cand; ++i)
            if (!(sse[i] >= rsse[i] * (1 - 1e-13) && sse[i] <= rsse[i] * (1 + 1e-13))) good = 0;
        printf(good ? "ok   fp8q_mse_grid_f64 (50 sums of squares within 1e-13 of the oracle)\n" : "FAIL fp8q_mse_grid_f64\n");
        ok &= good;
    }
    // fused epilogue with the folded batch-norm vector (fp8q_bn_fold_f32 + fp8q_affine_act_quantize_ab_f32)
    {
        const int64_t N = 3, Cc = 100, HW = 147;        // the same buffer viewed as [3, 100, 147]
        float bnv[4][100], *dbn, *dab, *t = (float *)malloc(n * 4);
        for (int k = 0; k < 4; ++k)
            for (int c = 0; c < 100; ++c) bnv[k][c] = (k == 0 ? -0.01f : 0.5f) + 0.003f * (float)((c * 7 + k * 13) % 31);
        CK(hipMalloc((void **)&dbn, sizeof(bnv)));
        CK(hipMalloc((void **)&dab, 100 * 8));
        CK(hipMemcpy(dbn, bnv, sizeof(bnv), hipMemcpyHostToDevice));
        CK(fp8q_bn_fold_f32(dbn, dbn + 100, dbn + 200, dbn + 300, Cc, dab, st));
        const float one_mv = 0.2f;
        CK(hipMemcpy(dmvo, &one_mv, 4, hipMemcpyHostToDevice));
        CK(fp8q_affine_act_quantize_ab_f32(dx, nullptr, dy, N, Cc, HW, dab, 1, dmvo, nullptr, 3.0f, 8, 1, st));
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(y, dy, n * 4, hipMemcpyDeviceToHost));
        orc_affine_act_f32(x, nullptr, t, N, Cc, HW, bnv[0], bnv[1], bnv[2], bnv[3], 1);
        orc_quantize_f32(t, ref, 1, n, &one_mv, 1, 3.0f, 8, 1);
        ok &= same_bits(y, ref, n, "fp8q_bn_fold_f32 + fp8q_affine_act_quantize_ab_f32 (BN + ReLU + E4M3)");
        // the same launch with the quantizer's constants and table prepared once (fixed ranges)
        float *dprep;
        CK(hipMalloc((void **)&dprep, FP8Q_PREP_BYTES));
        CK(fp8q_quantizer_prepare_f32(dmvo, 3.0f, 8, 1, dprep, st));
        CK(hipMemset(dy, 0, n * 4));
        CK(fp8q_affine_act_quantize_ab_f32(dx, nullptr, dy, N, Cc, HW, dab, 1, dmvo, dprep, 3.0f, 8, 1, st));
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(y, dy, n * 4, hipMemcpyDeviceToHost));
        ok &= same_bits(y, ref, n, "fp8q_quantizer_prepare_f32 + fp8q_affine_act_quantize_ab_f32 (prepared table)");
    }
    // error behaviour: bad arguments are reported, nothing throws
    if (fp8q_quantize_f32(dx, dy, C, inner, dmv, C - 1, 2.0f, 8, 1, st) != FP8Q_EINVAL) { printf("FAIL: EINVAL expected\n"); ok = 0; }
    if (fp8q_quantize_f32(dx, dy, C, inner, dmv, C, 0.0f, 16, 1, st) != FP8Q_EUNSUPPORTED) { printf("FAIL: EUNSUPPORTED expected\n"); ok = 0; }
    printf(ok ? "CABI SMOKE PASSED\n" : "CABI SMOKE FAILED\n");
    return ok ? 0 : 1;
}
