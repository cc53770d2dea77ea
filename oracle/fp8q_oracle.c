/*
 * fp8q_oracle.c -- CPU restatement of the reference's FP8 fake-quantization hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (fp8-quantization_amd/) may call,
 * link or import this file; it is the checker that tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py compare the HIP kernels against.
 *
 * Parity pin: every function here is checked against golden vectors produced by importing
 * the reference itself (tests/golden/make_golden.py -> the .npz fixtures in tests/golden, tests/test_oracle_golden.py).
 *
 * Arithmetic contract.  The reference evaluates the chain below with fp32 ATen CPU ops.
 * Two of those ops (log2, pow) are 1-ULP vector routines whose last bit depends on the
 * library build (measured: torch 2.10 CPU pow(2,e) differs from the correctly rounded
 * result on 1.8 % of inputs, numpy's differs from torch's on 20 %).  This restatement -- and
 * the HIP kernels, bit for bit -- define them as CORRECTLY ROUNDED fp32 functions
 * (computed through double precision), and keep every other step (the order of the fp32
 * additions that form `bias`, floor, clamp, the fp32 subtraction that forms the scale
 * exponent, IEEE division, round-half-even, the final multiply) exactly as the reference
 * orders them.  Against the reference's own output this is identical on ~98 % of elements,
 * within 2 fp32 ULP on the rest except at exact rounding ties (<= 1 step of the FP8 grid,
 * a few per million elements).
 *
 * Reference (paths relative to /root/reference):
 *   quantization/quantizers/fp8_quantizer.py:91-133   quantize_to_fp8_ste_MM
 *   quantization/quantizers/fp8_quantizer.py:13-50    generate_all_values_fp(_scaled)
 *   quantization/range_estimators.py:56-125           current/all/running min-max
 *   quantization/range_estimators.py:285-369          FP_MSE_Estimator
 *   quantization/range_estimators.py:133-282          LineSearchEstimator (fp64 lane, see below)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* correctly rounded fp32 log2 / 2^e (double libm is < 1 ulp(double); the second rounding
 * to fp32 is wrong only if the double result sits within ~1e-16 relative of an fp32
 * rounding boundary) */
static inline float cr_log2f(float a) { return (float)log2((double)a); }
static inline float cr_exp2f(float e) { return (float)exp2((double)e); }

/* torch.max / torch.min propagate NaN from either operand */
static inline float t_max(float a, float b) { return (a != a) ? a : ((b != b) ? b : (a > b ? a : b)); }
static inline float t_min(float a, float b) { return (a != a) ? a : ((b != b) ? b : (a < b ? a : b)); }
/* the range folds' variant: -0.0 < +0.0 (IEEE 754-2019 minimum / maximum; see orc_minmax_f32) */
static inline float z_max(float a, float b) { return (a == b) ? (signbit(a) ? b : a) : t_max(a, b); }
static inline float z_min(float a, float b) { return (a == b) ? (signbit(a) ? a : b) : t_min(a, b); }

typedef struct {
    float maxval, minval, bias, M;
} orc_chan_t;

/* fp8_quantizer.py:105-113: M, E, bias, minval for one channel */
static orc_chan_t orc_chan(float maxval, float mbits, int n_bits, int sign_bits)
{
    orc_chan_t p;
    float M = rintf(mbits); /* round_ste_func == torch.round: half to even */
    float hi = (float)(n_bits - sign_bits);
    if (M < 1.0f) M = 1.0f;
    if (M > hi) M = hi;
    float E = (float)(n_bits - sign_bits) - M;
    float two_E = cr_exp2f(E);                      /* 2**E, exact */
    float c = 2.0f - cr_exp2f(-M);                  /* 2 - 2**(-M), exact */
    /* ((2**E - log2(maxval)) + log2(2 - 2**-M)) - 1, each step rounded to fp32 */
    float b = two_E - cr_log2f(maxval);
    b = b + cr_log2f(c);
    b = b - 1.0f;
    p.maxval = maxval;
    p.minval = sign_bits == 1 ? -maxval : 0.0f;     /* -maxval, or zeros_like(maxval) */
    p.bias = b;
    p.M = M;
    return p;
}

/* fp8_quantizer.py:112-133 for one element */
static inline float orc_quant1(float x, const orc_chan_t *p)
{
    float xc = t_min(t_max(x, p->minval), p->maxval);
    float v = cr_log2f(fabsf(xc)) + p->bias;
    float ls = floorf(v);
    if (ls < 1.0f) ls = 1.0f;                       /* torch.clamp(min=1.0); NaN stays NaN */
    float e = (ls - p->M) - p->bias;
    float s = cr_exp2f(e);
    return rintf(xc / s) * s;                       /* torch.round = half to even */
}

/* K1.  x,y: [C, inner] contiguous fp32.  maxval: n_maxval == 1 (per tensor) or == C. */
int orc_quantize_f32(const float *x, float *y, int64_t C, int64_t inner, const float *maxval,
                     int64_t n_maxval, float mbits, int n_bits, int sign_bits)
{
    if (n_maxval != 1 && n_maxval != C) return -1;
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < C; ++c) {
        orc_chan_t p = orc_chan(maxval[n_maxval == 1 ? 0 : c], mbits, n_bits, sign_bits);
        const float *xr = x + c * inner;
        float *yr = y + c * inner;
        for (int64_t i = 0; i < inner; ++i) yr[i] = orc_quant1(xr[i], &p);
    }
    return 0;
}

/* flat per-tensor variant that parallelises over elements (cpu_baseline on big tensors) */
int orc_quantize_flat_f32(const float *x, float *y, int64_t n, float maxval, float mbits,
                          int n_bits, int sign_bits)
{
    orc_chan_t p = orc_chan(maxval, mbits, n_bits, sign_bits);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) y[i] = orc_quant1(x[i], &p);
    return 0;
}

/* K2/K3: min and max over the last dim of [C, inner]; NaN anywhere in a row -> NaN (torch.min/max).
 * range_estimators.py:62-74, 84-98
 * Signed zeros: when a row's minimum (maximum) is zero and the row holds both -0.0 and +0.0, ATen returns whichever its
 * vectorised reduction met first -- the reference does not pin the sign.  The contract here is the order-independent
 * IEEE 754-2019 minimum / maximum: -0.0 < +0.0, so min -> -0.0 and max -> +0.0 (what v_min_f32 / v_max_f32 compute).
 * Pinned by tests/golden/g3b_signed_zero.npz (rows of mixed zeros in both orders, run through the reference: its signs
 * follow the element order, its values, maxval and quantized rows are what this contract gives):
 * tests/test_oracle_golden.py::test_signed_zero_contract_of_minmax_is_pinned_by_a_fixture. */
#define ORC_ZERO_FLAGS(v, nz, pz) do { if ((v) == 0) { if (signbit(v)) nz = 1; else pz = 1; } } while (0)
#define ORC_ZERO_FIX(lo, hi, nz, pz) do { if ((lo) == 0) lo = nz ? -0.0f : 0.0f; if ((hi) == 0) hi = pz ? 0.0f : -0.0f; } while (0)

int orc_minmax_f32(const float *x, int64_t C, int64_t inner, float *mn, float *mx)
{
    if (C == 1 && inner >= (1 << 20)) {   /* one long row (per-tensor activations): split it over the threads */
        float lo = INFINITY, hi = -INFINITY;
        int nan = 0, nz = 0, pz = 0;
#pragma omp parallel for schedule(static) reduction(min : lo) reduction(max : hi) reduction(| : nan, nz, pz)
        for (int64_t i = 0; i < inner; ++i) {
            float v = x[i];
            nan |= (v != v);
            ORC_ZERO_FLAGS(v, nz, pz);
            lo = v < lo ? v : lo;
            hi = v > hi ? v : hi;
        }
        ORC_ZERO_FIX(lo, hi, nz, pz);
        mn[0] = nan ? NAN : lo;
        mx[0] = nan ? NAN : hi;
        return 0;
    }
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < C; ++c) {
        const float *xr = x + c * inner;
        float lo = INFINITY, hi = -INFINITY;
        int nan = 0, nz = 0, pz = 0;
        for (int64_t i = 0; i < inner; ++i) {
            float v = xr[i];
            nan |= (v != v);
            ORC_ZERO_FLAGS(v, nz, pz);
            lo = v < lo ? v : lo;
            hi = v > hi ? v : hi;
        }
        ORC_ZERO_FIX(lo, hi, nz, pz);
        mn[c] = nan ? NAN : lo;
        mx[c] = nan ? NAN : hi;
    }
    return 0;
}

/* fold a new batch estimate into the running one.
 * mode 0: overwrite (current_minmax :72-73); 1: min/max (allminmax :97-98);
 * 2: EMA (running_minmax :122-123: (1-m)*new + m*cur).  `first` = no previous estimate. */
int orc_fold_f32(float *cur_mn, float *cur_mx, const float *mn, const float *mx, int64_t C,
                 int mode, double momentum, int first)
{
    for (int64_t c = 0; c < C; ++c) {
        if (first || mode == 0) {
            cur_mn[c] = mn[c];
            cur_mx[c] = mx[c];
        } else if (mode == 1) {
            cur_mn[c] = z_min(cur_mn[c], mn[c]);
            cur_mx[c] = z_max(cur_mx[c], mx[c]);
        } else {
            /* python: (1 - momentum) in double, then each scalar is cast to fp32 by ATen */
            float om = (float)(1.0 - momentum), mo = (float)momentum;
            cur_mn[c] = om * mn[c] + mo * cur_mn[c];
            cur_mx[c] = om * mx[c] + mo * cur_mx[c];
        }
    }
    return 0;
}

/* K5: fp8_quantizer.py:236  maxval = |max(|x_min|, x_max)| */
int orc_absmax_f32(const float *mn, const float *mx, int64_t C, float *maxval)
{
    for (int64_t c = 0; c < C; ++c) maxval[c] = fabsf(t_max(fabsf(mn[c]), mx[c]));
    return 0;
}

/* K4: FP_MSE_Estimator inner loops (range_estimators.py:337-347).
 * x: [C, inner] (C == 1 for per-tensor); grid: [n_cand, C]; mbits: [n_m];
 * mses: [n_m, n_cand, C], accumulated (+=) with the per-channel mean of (x - xq)^2.
 * The mean itself is accumulated in double (the reference sums fp32 with ATen's
 * vectorised cascade; both are within ~1e-6 relative of the exact value). */
int orc_mse_grid_f32(const float *x, int64_t C, int64_t inner, const float *grid, int64_t n_cand,
                     const float *mbits, int n_m, int n_bits, int sign_bits, float *mses)
{
    int64_t jobs = (int64_t)n_m * n_cand * C;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t j = 0; j < jobs; ++j) {
        int64_t c = j % C;
        int64_t i = (j / C) % n_cand;
        int m = (int)(j / (C * n_cand));
        orc_chan_t p = orc_chan(grid[i * C + c], mbits[m], n_bits, sign_bits);
        const float *xr = x + c * inner;
        double acc = 0.0;
        for (int64_t k = 0; k < inner; ++k) {
            float d = xr[k] - orc_quant1(xr[k], &p);
            acc += (double)(d * d);
        }
        mses[j] += (float)(acc / (double)inner);
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------------------------
 * fp64 lane -- the reference's chain on a float64 tensor (BASELINE config 1: compute_quant_error.py:19-20 samples in
 * float64, range_estimators.py:161-169 / 236-256 evaluate 1000 candidates on them, quant_error_estimator.py:67-73 calls
 * quant.forward on float64).  Type promotion in quantize_to_fp8_ste_MM (fp8_quantizer.py:105-133) with x float64 and
 * maxval / mantissa bits float32 TENSORS: M, E and bias stay float32 (:105-110, every operand is float32); xc, log2|xc|,
 * the sum with bias, floor / clamp, the scale exponent, 2^e, the division, round and the product are float64.
 *
 * log2 and 2^x in double are 1-ulp library routines in the reference (Sleef, through ATen's Vectorized<double>) whose
 * last bit no independent implementation reproduces; here they are the table-driven evaluations below, the SAME IEEE op
 * sequence the HIP kernels run (csrc/fp8q_device.h: log2_tab_d, make_chan_fast), so HIP and oracle agree bit for bit.
 * tests/test_oracle_f64.py pins both against libm (log2: <= 2^-45 (1 + |v|) absolute; 2^x: <= 2 ulp) and the f64 chain
 * against reference-generated goldens (g1c_quantize_f64.npz: <= 2 ulp(double) per element, no grid-step flips).
 * --------------------------------------------------------------------------------------------------------------- */
#include "fp8q_oracle_tables.h"

static double orc_log2_d(double a)
{
    if (!(a > 0.0) || a == INFINITY) return a == 0.0 ? -INFINITY : (a > 0.0 ? a : NAN);
    int adj = 0;
    if (a < 0x1p-1022) { /* denormal double: renormalise */
        a *= 0x1p64;
        adj = -64;
    }
    uint64_t b;
    memcpy(&b, &a, 8);
    const int i = (int)(b >> 45) & 0x7f;
    const uint64_t mb = (b & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m;
    memcpy(&m, &mb, 8);
    const double r = fma(m, orc_tab[i], -1.0);
    double p = fma(r, 1.0 / 7.0, -1.0 / 6.0);
    p = fma(r, p, 1.0 / 5.0);
    p = fma(r, p, -1.0 / 4.0);
    p = fma(r, p, 1.0 / 3.0);
    p = fma(r, p, -1.0 / 2.0);
    p = fma(r, p, 1.0);
    p = p * r; /* ln(1 + r) */
    return (double)((int)(b >> 52) - 1023 + adj) + fma(p, 1.4426950408889634074, orc_tab[128 + i]);
}

/* 2^e = ldexp(2^-bf, n) with n = ceil(e), bf = n - e in [0, 1) (exact); 2^-bf = 2^(-j/128) exp(-(bf - j/128) ln 2) */
static double orc_exp2_d(double e)
{
    if (!(fabs(e) < INFINITY)) return e != e ? e : (e > 0.0 ? INFINITY : 0.0);
    const double n = ceil(e);
    const double bf = n - e;
    const int j = (int)(bf * 128.0);
    const double t = -(bf - (double)j * (1.0 / 128.0)) * 0.69314718055994530942;
    double q = fma(t, 1.0 / 720.0, 1.0 / 120.0);
    q = fma(t, q, 1.0 / 24.0);
    q = fma(t, q, 1.0 / 6.0);
    q = fma(t, q, 0.5);
    q = fma(t, q, 1.0);
    q = q * t; /* exp(t) - 1 */
    const double ej = orc_tab[256 + j];
    const double g = fma(ej, q, ej);
    if (n > 4000.0) return INFINITY;
    if (n < -4000.0) return 0.0;
    return ldexp(g, (int)n);
}

/* exported for tests/test_oracle_f64.py (checked against libm there) */
double orc_log2_f64(double a) { return orc_log2_d(a); }
double orc_exp2_f64(double e) { return orc_exp2_d(e); }

static inline double t_max_d(double a, double b) { return (a != a) ? a : ((b != b) ? b : (a > b ? a : b)); }
static inline double t_min_d(double a, double b) { return (a != a) ? a : ((b != b) ? b : (a < b ? a : b)); }

/* fp8_quantizer.py:112-133 for one float64 element; the channel constants are the float32 ones (orc_chan) */
static inline double orc_quant1_f64(double x, const orc_chan_t *p)
{
    double xc = t_min_d(t_max_d(x, (double)p->minval), (double)p->maxval);
    double v = orc_log2_d(fabs(xc)) + (double)p->bias;
    double ls = floor(v);
    if (ls < 1.0) ls = 1.0;
    double e = (ls - (double)p->M) - (double)p->bias;
    double s = orc_exp2_d(e);
    return rint(xc / s) * s;
}

/* K1 on float64.  x, y: [C, inner] doubles; maxval: fp32, n_maxval == 1 or C. */
int orc_quantize_f64(const double *x, double *y, int64_t C, int64_t inner, const float *maxval, int64_t n_maxval,
                     float mbits, int n_bits, int sign_bits)
{
    if (n_maxval != 1 && n_maxval != C) return -1;
    for (int64_t c = 0; c < C; ++c) {
        const orc_chan_t p = orc_chan(maxval[n_maxval == 1 ? 0 : c], mbits, n_bits, sign_bits);
#pragma omp parallel for schedule(static) if (inner >= 4096)
        for (int64_t i = 0; i < inner; ++i) y[c * inner + i] = orc_quant1_f64(x[c * inner + i], &p);
    }
    return 0;
}

/* min / max of float64 rows (LineSearchEstimator._define_search_range, range_estimators.py:205-222: data.min(), data.max()) */
int orc_minmax_f64(const double *x, int64_t C, int64_t inner, double *mn, double *mx)
{
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < C; ++c) {
        double lo = INFINITY, hi = -INFINITY;
        int nan = 0, nz = 0, pz = 0;
        for (int64_t i = 0; i < inner; ++i) {
            double v = x[c * inner + i];
            nan |= (v != v);
            ORC_ZERO_FLAGS(v, nz, pz);
            lo = v < lo ? v : lo;
            hi = v > hi ? v : hi;
        }
        if (lo == 0) lo = nz ? -0.0 : 0.0;       /* signed zeros: as orc_minmax_f32 */
        if (hi == 0) hi = pz ? 0.0 : -0.0;
        mn[c] = nan ? NAN : lo;
        mx[c] = nan ? NAN : hi;
    }
    return 0;
}

/* K4 on float64: the candidate loop of LineSearchEstimator._perform_1D_search (range_estimators.py:236-256) with
 * loss_fx (:161-169: torch.sum((data - y) ** 2), per row when per_channel_loss) -- reduce_sum != 0 -- or the mean of
 * FP_MSE_Estimator.forward (:337-347) -- reduce_sum == 0.  out: [n_m, n_cand, C] doubles, accumulated (+=).
 * The sum is compensated (Neumaier): the checker's value is the correctly rounded sum to ~1 ulp, whatever the order
 * the reference's ATen cascade or the HIP kernel's tree use (both within ~1e-15 relative of it). */
int orc_sse_grid_f64(const double *x, int64_t C, int64_t inner, const float *grid, int64_t n_cand, const float *mbits,
                     int n_m, int n_bits, int sign_bits, double *out, int reduce_sum)
{
    int64_t jobs = (int64_t)n_m * n_cand * C;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t j = 0; j < jobs; ++j) {
        int64_t c = j % C;
        int64_t i = (j / C) % n_cand;
        int m = (int)(j / (C * n_cand));
        float g = grid[i * C + c];
        orc_chan_t p = orc_chan(fabsf(t_max(fabsf(-g), g)), mbits[m], n_bits, sign_bits); /* set_quant_range(-g, g) */
        const double *xr = x + c * inner;
        double sum = 0.0, comp = 0.0;
        for (int64_t k = 0; k < inner; ++k) {
            double d = xr[k] - orc_quant1_f64(xr[k], &p);
            double v = d * d;
            double t = sum + v;
            comp += fabs(sum) >= fabs(v) ? (sum - t) + v : (v - t) + sum;
            sum = t;
        }
        sum += comp;
        out[j] += reduce_sum ? sum : sum / (double)inner;
    }
    return 0;
}

/* N2: what precedes the activation quantizer in a fused layer -- eval-mode batch norm (+ residual) (+ ReLU / ReLU6):
 * BNFusedHijacker.forward, quantization/quantized_folded_bn.py:39-55 (F.batch_norm(..., training=False) then the
 * activation function) and the residual tail of models/resnet_quantized.py:43-46.  The batch norm is ATen's CPU eval
 * kernel (probed bit for bit, tests/test_epilogue.py::test_fused_bn_matches_aten_cpu_batch_norm):
 *   alpha = invstd * gamma;  beta' = fma(-mean, alpha, beta);  out = fma(x, alpha, beta').
 * x, res, y: [N, C, HW]; mean/invstd/gamma/beta: [C] or all NULL (no batch norm); res may be NULL.
 * act: 0 none, 1 ReLU, 2 ReLU6 (NaN passes through, as torch.relu / hardtanh do). */
int orc_affine_act_f32(const float *x, const float *res, float *y, int64_t N, int64_t C, int64_t HW,
                       const float *mean, const float *invstd, const float *gamma, const float *beta, int act)
{
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < N * C; ++p) {
        const int64_t c = p % C;
        float alpha = 1.0f, bp = 0.0f;
        if (mean) {
            alpha = invstd[c] * gamma[c];
            bp = fmaf(-mean[c], alpha, beta[c]);
        }
        for (int64_t i = p * HW; i < (p + 1) * HW; ++i) {
            float t = mean ? fmaf(x[i], alpha, bp) : x[i];
            if (res) t = t + res[i];
            if (act >= 1) t = t < 0.0f ? 0.0f : t;
            if (act == 2) t = t > 6.0f ? 6.0f : t;
            y[i] = t;
        }
    }
    return 0;
}

/* N3: storage codes [sign | E exponent bits | M fraction bits] (layout of fp8_quantizer.py:13-41).
 * code of x = the byte whose value is orc_quant1(x): from K1's integers p (binade, >= 1) and
 * r = rint(|xc| / s_p):  r < 2^M -> exponent code 0, fraction r;  r == 2^(M+1) -> (p+1, 0);
 * else (p, r - 2^M).  NaN results (NaN input, degenerate maxval) encode as 0. */
static inline float orc_scale(const orc_chan_t *p, float ls) { return cr_exp2f((ls - p->M) - p->bias); }

int orc_encode_u8(const float *x, uint8_t *codes, int64_t C, int64_t inner, const float *maxval,
                  int64_t n_maxval, float mbits, int n_bits, int sign_bits)
{
    if ((n_maxval != 1 && n_maxval != C) || n_bits > 8) return -1;
    {   /* a format without an exponent bit has no code for a value that rounds up to 2^(M+1) steps */
        orc_chan_t p0 = orc_chan(1.0f, mbits, n_bits, sign_bits);
        if (n_bits - sign_bits - (int)p0.M < 1) return -2;
    }
    for (int64_t c = 0; c < C; ++c) {
        orc_chan_t p = orc_chan(maxval[n_maxval == 1 ? 0 : c], mbits, n_bits, sign_bits);
        const int M = (int)p.M;
        for (int64_t i = 0; i < inner; ++i) {
            float xv = x[c * inner + i];
            float xc = t_min(t_max(xv, p.minval), p.maxval);
            float ls = floorf(cr_log2f(fabsf(xc)) + p.bias);
            if (ls < 1.0f) ls = 1.0f;
            float r = rintf(xc / orc_scale(&p, ls));
            uint32_t code = 0;
            /* r is an integer in [0, 2^(M+1)] unless the channel is degenerate (s = 0 or NaN: K1 gives NaN there):
               those, like NaN inputs, take code 0 */
            if (ls == ls && fabsf(r) <= (float)(2u << M)) {
                uint32_t ri = (uint32_t)fabsf(r), m2 = 1u << M, e = (uint32_t)ls, f = ri - m2;
                if (ri < m2) { e = 0; f = ri; }
                else if (ri == 2 * m2) { e += 1; f = 0; }
                code = (e << M) | f;
                if (sign_bits == 1 && signbit(r)) code |= 1u << (n_bits - 1);
            }
            codes[c * inner + i] = (uint8_t)code;
        }
    }
    return 0;
}

int orc_decode_u8(const uint8_t *codes, float *y, int64_t C, int64_t inner, const float *maxval,
                  int64_t n_maxval, float mbits, int n_bits, int sign_bits)
{
    if ((n_maxval != 1 && n_maxval != C) || n_bits > 8) return -1;
    for (int64_t c = 0; c < C; ++c) {
        orc_chan_t p = orc_chan(maxval[n_maxval == 1 ? 0 : c], mbits, n_bits, sign_bits);
        const int M = (int)p.M;
        const uint32_t m2 = 1u << M;
        for (int64_t i = 0; i < inner; ++i) {
            uint32_t code = codes[c * inner + i];
            uint32_t body = sign_bits == 1 ? (code & ((1u << (n_bits - 1)) - 1u)) : code;
            uint32_t e = body >> M, f = body & (m2 - 1u);
            float v = (float)(f + (e ? m2 : 0u)) * orc_scale(&p, (float)(e ? e : 1u));
            y[c * inner + i] = (sign_bits == 1 && ((code >> (n_bits - 1)) & 1u)) ? -v : v;
        }
    }
    return 0;
}

/* a9: every value of an (n_bits, ebits, bias) format, ascending (fp8_quantizer.py:13-41).
 * out must hold 2^n_bits doubles.  Codes: sign | exponent | fraction; exponent code 0 is
 * subnormal; the all-ones exponent is an ordinary binade (no inf/NaN). */
static int cmp_double(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

int orc_fp_grid(int n_bits, int ebits, int bias, double *out)
{
    int fbits = n_bits - 1 - ebits;
    if (fbits < 0 || ebits < 0) return -1;
    int64_t n = 0;
    for (int s = 0; s < 2; ++s)
        for (int e = 0; e < (1 << ebits); ++e)
            for (int f = 0; f < (1 << fbits); ++f) {
                int sub = (e == 0);
                double frac = (double)f / (double)(1 << fbits) + 1.0 - sub;
                out[n++] = (s ? 1.0 : -1.0) * ldexp(frac, e - bias + sub);
            }
    qsort(out, (size_t)n, sizeof(double), cmp_double);
    return 0;
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
