/*
 * fp8q.h -- C ABI of the MI355X-native FP8 fake-quantization engine (libfp8q_hip.so).
 *
 * The reference (Qualcomm-AI-research/FP8-quantization) has no FFI: its "device boundary"
 * is the set of eager ATen op chains listed in SURVEY.md section 2.1.  Each entry point
 * below replaces one of those chains with one (or two) hand-written gfx950 kernels; the
 * Python classes that keep the reference's operator API (fp8-quantization_amd/quantization)
 * call them through ctypes with tensor.data_ptr() and the current HIP stream.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer owned by the caller; nothing is allocated or freed here
 *   - tensors are contiguous fp32, viewed as [C, inner] (per-channel = dim 0, as the
 *     reference does with x.view(x.shape[0], -1)); per-tensor calls pass C == 1
 *   - enqueue-only on `stream` (a hipStream_t, NULL = default stream); no host sync
 *   - return 0 on success, a positive hipError_t on a HIP failure, a negative FP8Q_E* on a
 *     bad argument; never throws; stateless and thread-safe
 *   - arithmetic contract: bit-identical to oracle/fp8q_oracle.c (reference op order in
 *     fp32, correctly rounded log2 / 2^x) -- see DESIGN.md "Arithmetic contract".  ONE exception:
 *     fp8q_mse_grid_f32 (K4) -- see its comment
 */
#ifndef FP8Q_H
#define FP8Q_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FP8Q_VERSION 601 /* 0.6.1: 0.6.0 (one-call MSE calibration step fp8q_mse_calibrate_f32, fused small-tensor routes) + sign_bits decided on the device (fp8q_sign_fold_u8, fp8q_quantize_ds_f32) */

#define FP8Q_OK 0
#define FP8Q_EINVAL (-1)       /* null pointer, negative size, n_maxval not in {1, C}, ... */
#define FP8Q_EUNSUPPORTED (-2) /* n_bits - sign_bits - M > 7 (more than 7 exponent bits) */
#define FP8Q_EWORKSPACE (-3)   /* workspace too small or misaligned */
#define FP8Q_ETOOLONG (-4)     /* fused min/max+quantize: rows longer than fp8q_fused_max_inner() */
#define FP8Q_ETOOMANY (-5)     /* MSE grid search: more than 65535 channels in one call */
#define FP8Q_ETIMEDOUT (-6)    /* fp8q_minmax_workspace_check: a min/max reducer gave up waiting (that range is NaN) */

/* range-estimator fold modes (how a new batch estimate is merged into the running one) */
#define FP8Q_FOLD_CURRENT 0 /* overwrite          range_estimators.py:72-73  CurrentMinMaxEstimator */
#define FP8Q_FOLD_ALL 1     /* min / max          range_estimators.py:97-98  AllMinMaxEstimator     */
#define FP8Q_FOLD_RUNNING 2 /* EMA (1-m)*new+m*cur range_estimators.py:122-123 RunningMinMaxEstimator */

/*
 * Workspaces of the min/max entry points (fp8q_minmax_f32, fp8q_affine_act_minmax_f32).  The two-stage reduction
 * runs in ONE launch: streaming blocks publish tagged partial results in the workspace, a reducer block collects them
 * and clears them again.  Contract: the workspace is ZERO before the first call that uses it (hipMemset once after
 * allocating it) and every call leaves it zero, so a buffer that is only ever handed to these two entry points, by one
 * stream at a time, never needs clearing again.  Do not share one buffer between launches that may run concurrently,
 * and do not let other kernels scribble over it.
 * Layout: a 16-byte header {uint32 timeout count, 12 reserved bytes} followed by the 8-byte granules.
 *
 * Failure modes and how they surface (the entry points only enqueue, so they cannot report what happens on the device):
 *   - the reducer block waits a bounded time (~2 s) for its streaming blocks; if one never arrives it writes NaN as
 *     that row's range AND increments the header's timeout count.  fp8q_minmax_workspace_check() -- which synchronises
 *     the stream -- then returns FP8Q_ETIMEDOUT; call it wherever a sync is acceptable (the Python layer does at
 *     fix_ranges(), i.e. once after calibration).
 *   - a workspace that was not zero on entry, or that two streams used at once, gives wrong ranges silently;
 *     fp8q_minmax_workspace_check() returns FP8Q_EWORKSPACE when it finds non-zero granules between calls, and with
 *     FP8Q_DEBUG_WS=1 in the environment every min/max call runs that check (sync + copy) on entry and refuses to
 *     launch on a dirty workspace.
 * Progress does not rely on the order in which workgroups are dispatched (streaming blocks never wait; reducers are at
 * most half of the resident workgroup slots); in-order dispatch -- what the hardware does, not a HIP guarantee -- only
 * keeps the reducer's wait short.
 */

typedef void *fp8q_stream_t; /* hipStream_t */

int fp8q_version(void);
const char *fp8q_strerror(int code);

/*
 * K1 -- FP8 quantize + dequantize.
 * Replaces quantize_to_fp8_ste_MM(x_float, n_bits, maxval, num_mantissa_bits, sign_bits),
 * quantization/quantizers/fp8_quantizer.py:91-133 (13 eager ATen kernels -> 1 launch).
 *   x, y     [C, inner] fp32 (y may alias x)
 *   maxval   [n_maxval] fp32, n_maxval == 1 (per tensor) or == C (per channel)
 *   mbits    number of mantissa bits; rounded half-to-even and clamped to [1, n_bits - sign_bits]
 *   sign_bits 1 (clamp to [-maxval, maxval]) or 0 (clamp to [0, maxval])
 * HBM traffic: 8 B / element.
 */
int fp8q_quantize_f32(const float *x, float *y, int64_t C, int64_t inner, const float *maxval,
                      int64_t n_maxval, float mbits, int n_bits, int sign_bits,
                      fp8q_stream_t stream);

/*
 * K2/K3/K5 -- min/max range estimation with the running-estimate fold.
 * Replaces {Current,All,Running}MinMaxEstimator.forward, quantization/range_estimators.py:61-125
 * (x.min(), x.max(), torch.min/max fold) and the abs-max of FPQuantizer.set_quant_range,
 * fp8_quantizer.py:236.
 *   x         [C, inner] fp32 (C == 1: per tensor)
 *   cur_min, cur_max [C] running estimate, updated in place;  `first` != 0: no previous estimate
 *   maxval_out [C] or NULL: |max(|cur_min|, cur_max)| after the fold
 *   ws        8-byte aligned scratch of at least fp8q_minmax_workspace_bytes(C, inner) bytes, zero on first use (see above)
 * NaN anywhere in a row makes that row's min and max NaN (torch semantics).
 * Signed zeros: a zero minimum is -0.0 when the row holds a -0.0, a zero maximum is +0.0 when it holds a +0.0 (IEEE 754-2019
 * minimum / maximum, order-independent; ATen returns whichever zero its reduction met first, so the reference does not pin it).
 * HBM traffic: 4 B / element.
 */
size_t fp8q_minmax_workspace_bytes(int64_t C, int64_t inner);
int fp8q_minmax_f32(const float *x, int64_t C, int64_t inner, float *cur_min, float *cur_max,
                    float *maxval_out, int fold_mode, double momentum, int first, void *ws,
                    size_t ws_bytes, fp8q_stream_t stream);

/*
 * K2+K5+K1 fused -- per-channel weight quantization in estimate_ranges state:
 * QuantizationManager.forward (quantization_manager.py:114-122) with CurrentMinMaxEstimator
 * and set_maxval=True: row min/max -> maxval = |max(|min|, max)| -> quantize, rows staged in LDS.
 *   x, y [C, inner]; row_min,row_max,maxval_out [C] outputs (each may be NULL)
 * Requires inner <= fp8q_fused_max_inner(); larger rows: call fp8q_minmax_f32 + fp8q_quantize_f32.
 * HBM traffic: 8 B / element.
 */
/*
 * Batch-sharded (data-parallel) calibration: the _packed variants additionally write, per row, the 16-byte record
 * {-min, max, isnan(min), isnan(max)} of the FOLDED estimate ([C, 4] fp32, 16-byte aligned; a NaN travels as its
 * flag with -inf as the value).  The ranks then need ONE all-reduce(MAX) over that buffer -- min/max commute with the
 * union of the shards -- and fp8q_ranges_unpack_f32 turns the reduced records back into cur_min / cur_max (+ K5
 * maxval_out; each may be NULL).  The reference is single-process (utils/qat_utils.py:27-28); this is north_star's
 * "activation calibration is batch-sharded with an RCCL all-reduce of the running min/max" with no tensor glue
 * around the collective.
 */
int fp8q_minmax_packed_f32(const float *x, int64_t C, int64_t inner, float *cur_min, float *cur_max,
                           float *maxval_out, float *packed, int fold_mode, double momentum, int first, void *ws,
                           size_t ws_bytes, fp8q_stream_t stream);
int fp8q_ranges_unpack_f32(const float *packed, int64_t n, float *cur_min, float *cur_max, float *maxval_out,
                           fp8q_stream_t stream);

/*
 * SYNCHRONISES `stream`, then inspects a min/max workspace: FP8Q_ETIMEDOUT if a reducer timed out since the last
 * clear, FP8Q_EWORKSPACE if granules are non-zero between calls (contract violated), FP8Q_OK otherwise.
 * clear != 0: the workspace is zeroed again after a failure was reported (so the buffer stays usable).
 */
int fp8q_minmax_workspace_check(void *ws, size_t ws_bytes, int clear, fp8q_stream_t stream);

int64_t fp8q_fused_max_inner(void);
int fp8q_minmax_quantize_f32(const float *x, float *y, int64_t C, int64_t inner, float *row_min,
                             float *row_max, float *maxval_out, float mbits, int n_bits,
                             int sign_bits, fp8q_stream_t stream);

/*
 * K4 -- FP-MSE grid search accumulation.
 * Replaces the double loop of FP_MSE_Estimator.forward, quantization/range_estimators.py:337-347
 * (111 * |mbits| full quantizer passes -> one pass over x).
 *   x     [C, inner];  grid [n_cand, C] candidate maxvals;  mbits [n_m] (host array)
 *   mses  [n_m, n_cand, C] fp32, accumulated:  += mean over the row of (x - q(x))^2
 *   ws    scratch of at least fp8q_mse_workspace_bytes(C, inner, n_cand, n_m) bytes
 * Table entries agree with the oracle's to ~1e-7 relative (tests and tests/soak.py: <= 1e-5 on EVERY entry; the CHOSEN
 * (mantissa bits, maxval) equal to the oracle's choice or its oracle-MSE within 1e-6 relative of the oracle's minimum --
 * SURVEY.md 8c).  Every element takes the grid point the reference gives it; what differs is the summation: fp32 squares
 * summed in another order (double partials) or, on the interval-histogram route, exact sums.  Three routes, chosen by the
 * SHAPE of the call only (a tensor is evaluated the same way every time):
 *   rows < 2048 elements      lane = candidate, x broadcast from LDS: x * 2^frac(bias) scaled by the binade's power of two and
 *                             rounded with v_rndne; elements within 5 ulps of a rounding tie (the product carries up to 4 ulps
 *                             of error against the reference's fl32(x / s)) take the reference's own division;
 *   longer rows               lane = element, candidates walked with wave-uniform constants: the same product rounded on its
 *                             bits / by a magic-number add; every lane watches how close its elements come to a tie and the
 *                             elements within 5 ulps are re-evaluated with the reference's division (round 5: before, such an
 *                             element could take the other neighbour, which moved entries carried by a few elements -- the
 *                             search grid's last candidate 1.2 max|x| puts the largest element on a tie for M = 1, 3, 5 -- by up
 *                             to 4e-5);
 *   per-tensor rows of a format of <= 8 bits (signed or unsigned), long enough to pay for ~60 us of fixed cost (>= ~1 M elements
 *   for 111 (width, candidate) pairs, >= ~0.5 M for 666: FP_MSE_Estimator with the mantissa search, LineSearchEstimator's
 *   1000 candidates)               the interval histogram (csrc/fp8q_mse_hist.hip): a quantizer is a step function of |x|, so the
 *                             exact borders of all cells of all candidates (the smallest float at which the reference's own
 *                             fp32 decisions flip) cut |x| into intervals; the nonzero keys are partitioned ONCE by their top
 *                             11 bits, integer moments {n, sum d, sum d^2} of every interval are accumulated with LDS atomics
 *                             (exact, order-independent: the result is deterministic), and a candidate's cells are summed as
 *                             S2 - 2 q S1 + n q^2 in double-double (the terms cancel to 1e-13 of S2 on already-quantized
 *                             data).  Cost independent of the number of candidates: ~4 ps per element.
 * fp8q_mse_workspace_bytes() accounts for the partitioned keys (4 B / element) and the border tables when the shape can take
 * the third route.  FP8Q_MSE_HIST=0 keeps the lane-per-element kernel; =2 evaluates every candidate of that route element
 * by element (the tests' cross-check of the cell logic).
 */
size_t fp8q_mse_workspace_bytes(int64_t C, int64_t inner, int64_t n_cand, int n_m);
int fp8q_mse_grid_f32(const float *x, int64_t C, int64_t inner, const float *grid, int64_t n_cand,
                      const float *mbits_host, int n_m, int n_bits, int sign_bits, float *mses,
                      void *ws, size_t ws_bytes, fp8q_stream_t stream);

/*
 * Sync-free MSE calibration.  The reference's FP_MSE_Estimator goes back to the host three times per call
 * (range_estimators.py:305 mx.item() for the search grid, :353 torch.mode(...).item() for the mantissa vote, :360 one
 * gather per channel); these entry points keep all of it on the device:
 *   fp8q_mse_linspace_f32  grid[i, c] = torch.linspace(lo_frac * mx[c], hi_frac * mx[c], n_cand)[i], bit for bit
 *                          ([n_cand, C]; the reference uses 0.1, 1.2 and 111: :296-305)
 *   fp8q_mse_select_f32    :350-369 on mses [n_m, n_cand, C] / grid [n_cand, C]: per channel the width with the smallest
 *                          minimum, the plurality vote over the channels (torch.mode: smallest value on a tie) ->
 *                          mbits_out[0] = mbits_host[vote] (DEVICE scalar), vote_out[0] = its index (may be NULL), per
 *                          channel maxval_out[c] = grid[argmin_i mses[vote, i, c], c] and xmin_out[c] = -sign_bits *
 *                          maxval (may be NULL).  torch.min / argmin semantics: first index of the minimum, NaN first.
 *                          ONE launch (round 5).  ws: at least fp8q_mse_select_workspace_bytes(C, n_m) bytes, 4-byte
 *                          aligned; its first 4096 bytes (the ticket block) follow the min/max workspace contract: zero
 *                          before the first call that uses the buffer, zero again after every call (counters by which the
 *                          last workgroup of a launch finds out that it is the last; fp8q_mse_calibrate_f32 uses the same
 *                          block for the per-tensor selection it appends to the table-finishing launch).
 *   fp8q_quantize_dm_f32   K1 (fp8q_quantize_f32) with the mantissa width read from a DEVICE scalar, so that the batch
 *                          that follows the vote in the same calibration forward needs no host round trip.  One row
 *                          per workgroup column whatever the row length: a calibration path, not the tuned K1 routes.
 */
int fp8q_mse_linspace_f32(const float *mx, int64_t C, int n_cand, double lo_frac, double hi_frac, float *grid,
                          fp8q_stream_t stream);
/* The first calibration batch of FP_MSE_Estimator in one launch: the row min / max (fp8q_minmax_f32, current fold), K5's
 * maxval_out[c] = |max(|min|, max)| = max|x| of the row, and the search grid of that maximum (what fp8q_mse_linspace_f32
 * would make of maxval_out), written by the thread that stores the row's range.  grid [n_cand, C]; workspace as fp8q_minmax_f32.
 * C <= 65535 (FP8Q_ETOOMANY, as fp8q_mse_grid_f32). */
int fp8q_minmax_linspace_f32(const float *x, int64_t C, int64_t inner, float *cur_min, float *cur_max, float *maxval_out,
                             float *grid, int n_cand, double lo_frac, double hi_frac, void *ws, size_t ws_bytes,
                             fp8q_stream_t stream);
size_t fp8q_mse_select_workspace_bytes(int64_t C, int n_m);
int fp8q_mse_select_f32(const float *mses, const float *grid, int64_t C, int64_t n_cand, const float *mbits_host, int n_m,
                        int sign_bits, float *mbits_out, int *vote_out, float *maxval_out, float *xmin_out, void *ws,
                        size_t ws_bytes, fp8q_stream_t stream);
int fp8q_quantize_dm_f32(const float *x, float *y, int64_t C, int64_t inner, const float *maxval, int64_t n_maxval,
                         const float *mbits_dev, int n_bits, int sign_bits, fp8q_stream_t stream);

/*
 * FPQuantizer with allow_unsigned=True (quantization/quantizers/fp8_quantizer.py:216-225): set_quant_range() switches the
 * quantizer to an unsigned format (sign_bits = 0, for good) when every range minimum is >= 0 -- in the reference a host
 * round trip per call (`torch.all(x_min >= 0)` inside an `if`).  Here the decision stays on the device:
 *   fp8q_sign_fold_u8      signed_flag[0] (1 = signed; the caller initialises it to 1) is cleared when all C values of x_min
 *                          are >= 0 (a NaN minimum keeps the sign bit, as the reference's comparison does); never set again.
 *   fp8q_quantize_ds_f32   K1 (fp8q_quantize_f32) with sign_bits read from that flag: both formats of the width travel by
 *                          value, every workgroup picks one.  Same launch geometry as fp8q_quantize_dm_f32.
 *   fp8q_quantize_dms_f32  the same with the width read from device memory as well.
 */
int fp8q_sign_fold_u8(const float *x_min, int64_t C, unsigned char *signed_flag, fp8q_stream_t stream);
int fp8q_quantize_ds_f32(const float *x, float *y, int64_t C, int64_t inner, const float *maxval, int64_t n_maxval,
                         float mbits, int n_bits, const unsigned char *signed_flag, fp8q_stream_t stream);
/* ... with the mantissa width in device memory too (fp8q_quantize_dm_f32's mbits_dev): an MSE estimator's vote for a quantizer
 * whose sign is still that flag.  n_bits <= 8 (both signs' widths 1 .. n_bits - sign_bits travel by value). */
int fp8q_quantize_dms_f32(const float *x, float *y, int64_t C, int64_t inner, const float *maxval, int64_t n_maxval,
                          const float *mbits_dev, int n_bits, const unsigned char *signed_flag, fp8q_stream_t stream);

/*
 * One calibration step of a quantizer whose range comes from FP_MSE_Estimator, in ONE call:
 * QuantizationManager.forward in estimate_ranges state (quantization/quantization_manager.py:114-122) around
 * FP_MSE_Estimator.forward (quantization/range_estimators.py:318-369) --
 *   first != 0   row max|x| -> search grid linspace(0.1 max|x|, 1.2 max|x|, n_cand) per row; this batch's table entries are
 *                written instead of added (the table need not be cleared beforehand)                            (:295-316)
 *   always       mses[n_m, n_cand, C] += row-mean((x - q(x; mbits[m], grid[i, c]))^2)   (fp8q_mse_grid_f32)   (:337-347)
 *                vote of the mantissa width, per-row argmin -> state.mbits / vote / maxval / xmin (fp8q_mse_select_f32) (:350-369)
 *   y != NULL    y = quantize(x; maxval, voted width)  (fp8q_quantize_f32 when n_m == 1, else fp8q_quantize_dm_f32).  For a
 *                per-tensor quantizer (C == 1, x and y 16-byte co-aligned) vote + argmin ride in the prologue of this K1 launch
 *                instead of a launch of their own: same outputs in state.mbits / vote / maxval / xmin, same y.
 * Everything is enqueued on `stream`; nothing comes back to the host.  The state is caller-owned device memory
 * (any layout; the torch host allocates one block per estimator) and persists between the batches of a calibration.
 * Workspaces: ws_minmax as fp8q_minmax_f32 (zeroed, left zero; only read when first != 0), ws_select as fp8q_mse_select_f32
 * (zero header), ws_mse as fp8q_mse_grid_f32; fp8q_mse_calibrate_workspace_bytes returns the last size and writes the other two.
 * Results are those of the four entry points called one after the other (bit for bit).
 * pre != NULL (per-tensor quantizers, C == 1, inner == N * C * HW of pre): the quantizer sits behind a batch norm + activation
 * (+ residual); `x` is then a caller-owned scratch of `inner` floats that RECEIVES t = act(bn(pre->x) + residual) -- written by
 * fp8q_affine_act_minmax_linspace_f32 (first batch: abs-max and grid from the same launch) or fp8q_affine_act_f32 -- and the
 * search / quantization run on it: quantized_folded_bn.py:39-55 + quantization_manager.py:114-122 in one call.
 * ws_minmax then needs max(fp8q_minmax_workspace_bytes, fp8q_affine_act_minmax_workspace_bytes) bytes.
 */
typedef struct fp8q_affine_pre {
    const float *x;          /* [N, C, HW] the producer's output (convolution / linear) */
    const float *residual;   /* [N, C, HW] or NULL */
    const float *alpha_beta; /* folded [C, 2] batch-norm vector (fp8q_bn_fold_f32) or NULL */
    int64_t N, C, HW;
    int act;                 /* 0 none, 1 ReLU, 2 ReLU6 */
} fp8q_affine_pre;
typedef struct fp8q_mse_state {
    float *cur_min, *cur_max; /* [C] row minimum / maximum of the first batch */
    float *absmax;            /* [C] max|x| of the first batch: defines the grid */
    float *grid;              /* [n_cand, C] */
    float *mses;              /* [n_m, n_cand, C], accumulated over the batches */
    float *maxval;            /* [C] the winner's clipping value (what set_quant_range(xmin, maxval) stores) */
    float *xmin;              /* [C] -sign_bits * maxval; may be NULL */
    float *mbits;             /* [1] voted mantissa width */
    int *vote;                /* [1] its index in mbits_host; may be NULL */
} fp8q_mse_state;
size_t fp8q_mse_calibrate_workspace_bytes(int64_t C, int64_t inner, int64_t n_cand, int n_m, size_t *minmax_bytes,
                                          size_t *select_bytes);
int fp8q_mse_calibrate_f32(float *x, float *y, int64_t C, int64_t inner, const fp8q_mse_state *state, int first,
                           int n_cand, const float *mbits_host, int n_m, int n_bits, int sign_bits, const fp8q_affine_pre *pre,
                           void *ws_minmax, size_t ws_minmax_bytes, void *ws_select, size_t ws_select_bytes, void *ws_mse,
                           size_t ws_mse_bytes, fp8q_stream_t stream);

/*
 * The float64 lane -- BASELINE config 1.  compute_quant_error.py:19-20 draws float64 samples; LineSearchEstimator
 * (quantization/range_estimators.py:161-169, 205-222, 236-256) takes their min / max and scores 1000 clipping candidates
 * on them, and quant_error_estimator.py:67-73 runs the quantizer's forward on a float64 sample.  Under ATen's type
 * promotion quantize_to_fp8_ste_MM (fp8_quantizer.py:105-133) then keeps M, E and bias in float32 (maxval and the mantissa
 * bits are float32 tensors) and evaluates everything downstream of x in float64.  Arithmetic contract: bit-identical to
 * oracle/fp8q_oracle.c:orc_quant1_f64 (double log2 / 2^x = the table-driven 1-ulp evaluations defined there; against the
 * reference's own 1-ulp Sleef routines: <= 2 ulp(double) per element, the same grid point -- tests/golden/g1c_*.npz).
 *   fp8q_quantize_f64   K1: x, y [C, inner] float64 (y may alias x), 8-byte aligned; maxval fp32 [1] or [C].  16 B / element.
 *   fp8q_minmax_f64     row_min / row_max [C] float64 of x [C, inner] (NaN anywhere in a row -> NaN); two launches.
 *   fp8q_mse_grid_f64   K4: out[n_m, n_cand, C] (float64) += sum over the row of (x - q(x; mbits[m], grid[i, c]))^2
 *                       (reduce_sum != 0: LineSearchEstimator.loss_fx, torch.sum) or its mean (reduce_sum == 0:
 *                       FP_MSE_Estimator.forward :337-347).  Per element the K1-f64 arithmetic, bit for bit; the sums
 *                       are formed in another order than ATen's (or the oracle's compensated one): ~1e-15 relative.
 *                       ws: at least fp8q_mse_f64_workspace_bytes() bytes, 8-byte aligned, need not be initialised.
 */
int fp8q_quantize_f64(const double *x, double *y, int64_t C, int64_t inner, const float *maxval, int64_t n_maxval,
                      float mbits, int n_bits, int sign_bits, fp8q_stream_t stream);
size_t fp8q_minmax_f64_workspace_bytes(int64_t C, int64_t inner);
int fp8q_minmax_f64(const double *x, int64_t C, int64_t inner, double *row_min, double *row_max, void *ws,
                    size_t ws_bytes, fp8q_stream_t stream);
size_t fp8q_mse_f64_workspace_bytes(int64_t C, int64_t inner, int64_t n_cand, int n_m);
int fp8q_mse_grid_f64(const double *x, int64_t C, int64_t inner, const float *grid, int64_t n_cand,
                      const float *mbits_host, int n_m, int n_bits, int sign_bits, double *out, int reduce_sum, void *ws,
                      size_t ws_bytes, fp8q_stream_t stream);

/*
 * N2 -- producer epilogue fused with the activation quantizer: eval-mode batch norm
 * (NCHW, per-channel mean / invstd = 1/sqrt(var+eps) / gamma / beta, all four NULL to skip),
 * + optional residual add, + optional activation (act: 0 none, 1 ReLU, 2 ReLU6), then the
 * per-tensor FP8 quantizer -- BNFusedHijacker.forward, quantization/quantized_folded_bn.py:39-55,
 * and the residual tail of models/resnet_quantized.py:43-46, in one pass (8 B / element, 12 with a residual).
 * The _minmax twin produces the range of the same pre-quantization tensor for calibration
 * (same fold semantics as fp8q_minmax_f32; ws of at least fp8q_affine_act_minmax_workspace_bytes(N, C, HW)
 * bytes).  x, residual, y: [N, C, HW] fp32, 16-byte aligned;
 * C*HW must be a multiple of 4 (FP8Q_EUNSUPPORTED otherwise: use the unfused calls).
 */
int fp8q_affine_act_quantize_f32(const float *x, const float *residual, float *y, int64_t N, int64_t C,
                                 int64_t HW, const float *mean, const float *invstd, const float *gamma,
                                 const float *beta, int act, const float *maxval, float mbits, int n_bits,
                                 int sign_bits, fp8q_stream_t stream);
/*
 * The same epilogue with the batch norm's per-channel constants folded once: fp8q_bn_fold_f32 writes
 * alpha_beta[c] = {alpha_c, beta'_c} = {invstd_c * gamma_c, fma(-mean_c, alpha_c, beta_c)} ([C, 2] fp32, 8-byte aligned) --
 * the two numbers the kernels above form per plane -- and fp8q_affine_act_quantize_ab_f32 reads them (one 8-byte load
 * per plane instead of four 4-byte ones).  Bit-identical results; in eval mode the vector changes only when the BN
 * parameters do, so a caller folds once and reuses it for every forward.
 */
int fp8q_bn_fold_f32(const float *mean, const float *invstd, const float *gamma, const float *beta, int64_t C,
                     float *alpha_beta, fp8q_stream_t stream);
/*
 * A per-tensor quantizer whose range is FIXED can also be prepared once: fp8q_quantizer_prepare_f32 writes the channel
 * constants and the {s, 1/s} table every launch otherwise rebuilds from maxval (FP8Q_PREP_BYTES bytes, 16-byte aligned,
 * device memory).  Passed as `prep` (NULL = rebuild), it takes 0.5-1 us off the critical path of launches on cache-sized
 * activations (4-10 us each: most of a ResNet-18 / MobileNetV2 forward's quantizer launches).  `prep` must have been
 * prepared from the same maxval VALUE, mbits, n_bits and sign_bits as the call it is passed to; results are bit-identical.
 * alpha_beta may be NULL in the _ab entry point (no batch norm: the residual tails).
 */
#define FP8Q_PREP_BYTES 1056 /* float4 {maxval, lo, bias, threshold} + 130 x float2 {s_p, 1 / s_p} */
int fp8q_quantizer_prepare_f32(const float *maxval, float mbits, int n_bits, int sign_bits, float *prep, fp8q_stream_t stream);
int fp8q_affine_act_quantize_ab_f32(const float *x, const float *residual, float *y, int64_t N, int64_t C, int64_t HW,
                                    const float *alpha_beta, int act, const float *maxval, const float *prep, float mbits,
                                    int n_bits, int sign_bits, fp8q_stream_t stream);

/* The epilogue WITHOUT the quantizer: y = act(bn(x) + residual) (same kernels, same BN arithmetic, the quantizer switched off).
 * What an MSE range estimator behind a BN + activation searches on during calibration (quantized_folded_bn.py:39-55 followed by
 * range_estimators.py:318-369): one 8 B / element pass instead of torch's batch_norm + activation passes.  alpha_beta: the folded
 * [C, 2] vector of fp8q_bn_fold_f32, or NULL (no batch norm). */
int fp8q_affine_act_f32(const float *x, const float *residual, float *y, int64_t N, int64_t C, int64_t HW, const float *alpha_beta,
                        int act, fp8q_stream_t stream);
/* The same pass for the FIRST calibration batch of an MSE estimator: t = act(bn(x) + residual) is written AND its minimum /
 * maximum / abs-max and the search grid linspace(lo_frac * max|t|, hi_frac * max|t|, n_cand) ([n_cand, 1]) come out of the same
 * launch (what fp8q_affine_act_f32 + fp8q_minmax_linspace_f32 on t give, bit for bit; 8 B / element instead of 12).
 * ws: as fp8q_affine_act_minmax_f32 (zeroed, left zero; fp8q_affine_act_minmax_workspace_bytes). */
int fp8q_affine_act_minmax_linspace_f32(const float *x, const float *residual, float *t, int64_t N, int64_t C, int64_t HW,
                                        const float *alpha_beta, int act, float *cur_min, float *cur_max, float *maxval_out,
                                        float *grid, int n_cand, double lo_frac, double hi_frac, void *ws, size_t ws_bytes,
                                        fp8q_stream_t stream);
size_t fp8q_affine_act_minmax_workspace_bytes(int64_t N, int64_t C, int64_t HW);
int fp8q_affine_act_minmax_f32(const float *x, const float *residual, int64_t N, int64_t C, int64_t HW,
                               const float *mean, const float *invstd, const float *gamma, const float *beta,
                               int act, float *cur_min, float *cur_max, float *maxval_out, int fold_mode,
                               double momentum, int first, void *ws, size_t ws_bytes, fp8q_stream_t stream);

int fp8q_affine_act_minmax_packed_f32(const float *x, const float *residual, int64_t N, int64_t C, int64_t HW,
                                      const float *mean, const float *invstd, const float *gamma, const float *beta,
                                      int act, float *cur_min, float *cur_max, float *maxval_out, float *packed,
                                      int fold_mode, double momentum, int first, void *ws, size_t ws_bytes,
                                      fp8q_stream_t stream);

/*
 * N3 -- real FP8 storage codes.  The reference only simulates the format; its
 * enumerator generate_all_values_fp (fp8_quantizer.py:13-41) defines the byte layout
 * [sign | E exponent bits | M fraction bits] (exponent code 0 subnormal, no inf/NaN codes).
 *   fp8q_encode_u8: codes[i] = the byte whose value is quantize_to_fp8_ste_MM(x)[i]
 *   fp8q_decode_u8: y[i] = value of codes[i].  decode(encode(x)) == fp8q_quantize_f32(x) bit for bit whenever
 *                   the channel's scales are exactly geometric in fp32 (s_(p+1) == 2 s_p: true for every range
 *                   with |k - bias| below bias's binade, e.g. all weight-sized ranges); otherwise an element
 *                   that rounds UP into the next binade is 2^(M+1) s_p in K1 and 2^M s_(p+1) after decoding,
 *                   two fp32 renderings of the same grid point that can differ by ulp(k - bias) ln 2 relative (a few ULP, <= 5e-6 for |bias| < 100; the reference has the
 *                   same ambiguity: it emits either, depending on which side x came from)
 * n_bits <= 8 and at least one exponent bit (FP8Q_EUNSUPPORTED otherwise).  The format has no NaN: NaN inputs and degenerate channels (maxval 0/inf/NaN, whose
 * K1 output is NaN) encode as 0.  HBM traffic: 5 B / element each.
 */
int fp8q_encode_u8(const float *x, uint8_t *codes, int64_t C, int64_t inner, const float *maxval,
                   int64_t n_maxval, float mbits, int n_bits, int sign_bits, fp8q_stream_t stream);
int fp8q_decode_u8(const uint8_t *codes, float *y, int64_t C, int64_t inner, const float *maxval,
                   int64_t n_maxval, float mbits, int n_bits, int sign_bits, fp8q_stream_t stream);

/*
 * Multi-tensor K1 -- all weight tensors of a model in one launch: the reference quantizes
 * each layer's weight in that layer's forward (hijacker.py:70-108 -> quantize_weights), 21 tiny launches
 * for ResNet-18; with fixed ranges they are independent and can be enqueued together.
 *   descs  HOST array of n descriptors (read during the call only); every field as in fp8q_quantize_f32
 * Tensors are batched 32 per launch; a tensor that cannot be batched (pointers not 16-byte aligned, rows
 * shorter than 4 or too short for per-row tables, >= 64 MiB) gets its own fp8q_quantize_f32 launch, in order.
 * Results are bit-identical to n separate fp8q_quantize_f32 calls.  Nothing is enqueued if a descriptor is bad.
 */
typedef struct fp8q_tensor_desc {
    const float *x;
    float *y;
    const float *maxval;
    int64_t C, inner, n_maxval;
    float mbits;
    int n_bits, sign_bits;
} fp8q_tensor_desc;
int fp8q_multi_quantize_f32(const fp8q_tensor_desc *descs, int n, fp8q_stream_t stream);

/*
 * Multi-tensor K2+K5+K1 -- all weight tensors of a model in ESTIMATE state (per-channel current_minmax with
 * set_maxval: QuantizationManager.forward, quantization_manager.py:114-122, once per layer in the reference): one launch
 * finds every row's range and writes maxval_out[i][c] = |max(|min_c|, max_c)| ([C] floats per tensor, a HOST array of n
 * device pointers), a second one is fp8q_multi_quantize_f32 reading those ranges.  descs[i].n_maxval must equal C;
 * descs[i].maxval is an INPUT field and is not written: it must be NULL or name the same buffer as maxval_out[i]
 * (FP8Q_EINVAL otherwise -- a range buffer elsewhere would be ignored).  Two launches for a whole model instead of one
 * per layer; results bit-identical to fp8q_minmax_quantize_f32 per tensor.  This is what a rank of the channel-sharded
 * multi-GPU weight path runs on its shards before the all-gather (fp8q.dist.quantize_weights_sharded_bucketed).
 */
int fp8q_multi_minmax_quantize_f32(const fp8q_tensor_desc *descs, float *const *maxval_out, int n, fp8q_stream_t stream);
/*
 * Storage codes (fp8q_encode_u8 / fp8q_decode_u8) for many tensors at once: what the bucketed all-gather of channel-sharded
 * weights puts on the wire -- 1 byte per element instead of 4.  Same descriptor table; the CODE side is typed through the
 * float pointers of fp8q_tensor_desc:  _encode_: x = fp32 values (16-byte aligned), y = (float *) of the uint8 codes
 * (4-byte aligned);  _decode_: x = (const float *) of the codes, y = fp32 values.  Other alignments take one
 * fp8q_encode_u8 / fp8q_decode_u8 call per tensor.  _minmax_encode_: per-channel current_minmax ranges first
 * (maxval_out, as fp8q_multi_minmax_quantize_f32), then the codes: two launches for a whole bucket.
 * Bit-identical to the single-tensor entry points; n_bits <= 8 and at least one exponent bit (FP8Q_EUNSUPPORTED otherwise).
 */
int fp8q_multi_encode_u8(const fp8q_tensor_desc *descs, int n, fp8q_stream_t stream);
int fp8q_multi_minmax_encode_u8(const fp8q_tensor_desc *descs, float *const *maxval_out, int n, fp8q_stream_t stream);
int fp8q_multi_decode_u8(const fp8q_tensor_desc *descs, int n, fp8q_stream_t stream);

/*
 * Prepared multi-tensor launch.  fp8q_multi_quantize_f32 validates, classifies and packs its descriptors on every
 * call; for a fixed set of tensors (a model's weights after fix_ranges(), re-quantized whenever the weights change)
 * that work is done ONCE here and a call costs one kernel launch per 32 tensors.  The plan is a small HOST object
 * (the only allocation this library ever makes; no device memory); it records pointers, so the tensors must stay
 * where they are and keep their shapes -- their CONTENTS (weights, ranges) may change between launches.
 *   create    0 and *plan on success; a negative FP8Q_E* / positive hipError_t (out of host memory) otherwise
 *   launch    bit-identical to fp8q_multi_quantize_f32 on the same descriptors; enqueue-only; thread-safe
 *   launches  kernel launches one fp8q_multi_plan_launch enqueues
 */
typedef struct fp8q_multi_plan fp8q_multi_plan;
int fp8q_multi_plan_create(const fp8q_tensor_desc *descs, int n, fp8q_multi_plan **plan);
int fp8q_multi_plan_launch(const fp8q_multi_plan *plan, fp8q_stream_t stream);
int fp8q_multi_plan_launches(const fp8q_multi_plan *plan);
void fp8q_multi_plan_destroy(fp8q_multi_plan *plan);

/* Plain float4 copy kernel with the same launch shape as K1: the measured HBM ceiling that
 * bench.py reports next to the 8 TB/s spec figure. */
int fp8q_copy_f32(const float *x, float *y, int64_t n, fp8q_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FP8Q_H */
