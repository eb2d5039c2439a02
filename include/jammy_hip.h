/* jammy_hip.h -- C ABI of libjammy_hip.so: the MI355X (gfx950) kernels behind the per-layer forward / inverse +
 * log-det hot path of thoglu/jammy_flows.
 *
 * Conventions:
 *   - plain pointers to DEVICE memory + sizes, no torch types; every matrix argument carries its row stride (elements)
 *   - `params` is the reference's `extra_inputs` row block: row-major (param_batch, n_params) with param_batch 1
 *     (permanent / broadcast parameters) or B (one row per sample, as the amortisation MLP emits them,
 *     jammy_flows/main/default.py:956, 1002-1012, 1488); row layout per layer letter: see each struct below
 *   - inputs are never written (reference contract: tests/test_general.py:519, 533-550); outputs are fresh buffers
 *   - `status` (nullable) points to JF_STATUS_WORDS int32 counters that kernels bump atomically; the host turns them
 *     into the reference's warnings / exceptions (layers/bisection_n_newton.py:84-133, layers/spline_fns.py:57-59)
 *   - functions return JF_OK or a negative error code; they never throw, allocate, or synchronise;
 *     `stream` is a hipStream_t (the caller passes torch.cuda.current_stream().cuda_stream)
 *   - every entry point exists as _f32 and _f64
 */
#ifndef JAMMY_HIP_H
#define JAMMY_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JF_OK 0
#define JF_ERR_BADARG (-1)
#define JF_ERR_UNSUPPORTED (-2)
#define JF_ERR_LAUNCH (-3)

#define JF_STATUS_NONCONVERGED 0 /* rows whose Newton iteration ended above the reference's print threshold */
#define JF_STATUS_NONFINITE 1    /* non-finite iterates / outputs */
#define JF_STATUS_OUT_OF_RANGE 2 /* spline inputs outside [left,right] */
#define JF_STATUS_NEWTON_STEPS 3 /* g layers, sampling direction: row-steps of the Newton stage (sum over iterations of the rows still
                                    iterating) -- what the reference's masked iteration spends (bisection_n_newton.py:74-120) */
#define JF_STATUS_WORDS 4

#define JF_MAX_CHAIN 8 /* max layers fused into one launch */

int jf_abi_version(void);

/* Audit build of the iterative solvers (sampling direction of 'g', the solves of 'm' and 'v'; ABI 8).  The product library's rules: a safeguarded
 * Newton approach phase, float64 rows stop at an update of 1e-9, float32 rows at their rounding floor, the sphere Newton of 'v' at a
 * Gauss-Newton step of 1e-7.  csrc/Makefile also builds libjammy_hip_audit.so (-DJF_NEWTON_RULE_REFERENCE): the same entry points with the
 * reference's own iteration, bisection_n_newton.py:11-135 (25 bisections on [-1e5, 1e5], then Newton until the row's update sum is below 1e-14
 * or 20 steps are done) and :330-465 ('v': 1e-12).  jf_get_newton_rule() returns 1 for that library, 0 for the product library.  The two agree to
 * rounding (tests/test_gpu_parity.py::test_reference_newton_rule_*); the audit library exists so that the reference's iteration counts can be
 * reproduced.  The Python package loads it when JF_NEWTON_RULE=reference is set in the environment. */
int jf_get_newton_rule(void);

/* one column range of an MLP input row cat[conditional_input, embed(x_0), embed(x_1), ...] (main/default.py:946-962): `src` row-major with row
 * stride `stride` (elements); kind 0 = n_in columns copied, 1 = an S1 angle -> (cos, sin), 2 = S2 (theta, phi) -> (x, y, z)
 * (sphere_base.py:305-332, 786-794).  Used by jf_conditioning_rows (the rows materialised) and by the *_seg / split3 entry points (read in place). */
#define JF_MAX_SEGMENTS 16
typedef struct jf_cond_segment { const void* src; int64_t stride; int32_t kind; int32_t n_in; } jf_cond_segment;

/* ------------------------------------------------------------------------------------------------------------
 * 'g' Gaussianization flow (replaces gf_block._inv_flow_mapping / _flow_mapping + euclidean_base offset handling:
 * jammy_flows/layers/euclidean/gaussianization_flow.py:911-989, 995-1114; euclidean_base.py:34-76;
 * layers/bisection_n_newton.py:11-135).
 * Row layout: [offset D if model_offset][householder v: hh_iter*D][means K*D][log_widths K*D][log_weights K*D if fit_normalization]
 *             ((K,D) K-major)   -- gaussianization_flow.py:739-742, 820-834
 * ------------------------------------------------------------------------------------------------------------ */
enum { JF_GF_ISIGMOID = 0, JF_GF_INORMAL_PARTLY_PRECISE = 1, JF_GF_INORMAL_PARTLY_CRUDE = 2, JF_GF_INORMAL_FULL_PADE = 3 };
enum { JF_GF_WIDTH_SMOOTH_SATURATION = 0, JF_GF_WIDTH_EXP = 1, JF_GF_WIDTH_SOFTPLUS = 2 };
enum { JF_GF_STRETCH_CLASSIC = 0, JF_GF_STRETCH_RQ_SPLINES = 1 };
/* rotation_mode of gf_block (gaussianization_flow.py:95-100, 711-799): HOUSEHOLDER with hh_iter == 0 is "none" */
enum { JF_GF_ROT_HOUSEHOLDER = 0, JF_GF_ROT_ANGLES = 1, JF_GF_ROT_CAYLEY = 2, JF_GF_ROT_TRIANGULAR = 3 };

typedef struct jf_gf_layer {
    int32_t num_kde;                /* K */
    int32_t hh_iter;                /* number of Householder reflections, 0 = no rotation */
    int32_t model_offset;           /* row starts with D offsets (last layer of an e-block) */
    int32_t fit_normalization;      /* log_weights present */
    int32_t regulate_normalization; /* soft-clamp log_weights to [ln norm_min, ln norm_max] */
    int32_t inverse_function_type;  /* JF_GF_* */
    int32_t width_mode;             /* JF_GF_WIDTH_* */
    int32_t clamp_widths;
    int32_t nonlinear_stretch_type; /* JF_GF_STRETCH_*: CLASSIC = logistic mixture + inverse-CDF stage; RQ_SPLINES = per-dimension
                                       rational-quadratic spline with learnable box and linear tails (spline_fns.py:188-358), row layout
                                       [offset][rot][log_w D*K][log_h D*K][log_d D*(K+1)][box D*4] (gaussianization_flow.py:873-909) */
    int32_t rotation_mode;          /* JF_GF_ROT_*: ANGLES = D(D-1)/2 Givens angles in itertools.combinations order (:747-780), CAYLEY = one
                                       parameter (D == 2, :782-798), TRIANGULAR = [lower D(D-1)/2][log-diagonal D-1][upper D(D-1)/2] of
                                       L diag(e^d) U with unit triangles and sum(d) = 0 (:711-729, 942-964) */
    int32_t center_mean;            /* the means section holds K-1 rows; the last mean makes the weighted mean vanish (:846-852) */
    int32_t add_skewness;           /* a trailing K*D section of log-exponents of the skewed-logistic components (:352-368, 411-442);
                                       float64 only, as in the reference (extra_functions.py:28) */
    double width_min, width_max;    /* width_max <= 0: no upper bound */
    double norm_min, norm_max;
} jf_gf_layer;
/* Layers with rotation_mode != HOUSEHOLDER, center_mean or add_skewness run in the general-option kernel of jf_gf_chain_inv / _fwd
 * (one lane per row, D <= 8); the fused block entry points and jf_gf_chain_inv_bwd return JF_ERR_UNSUPPORTED for them.  D <= 64 for layers at the
 * other options (groups of up to 64 lanes -- a whole wave -- per row; 32 until ABI 8's second build). */

/* log-prob direction of a chain of `n_layers` g layers applied in REVERSE order (layer n-1 first), all in one launch.
 * params row = the layers' rows concatenated in layer order 0..n-1.  log_det_in / base_logp_in may be NULL (= 0);
 * base_logp_out (nullable) receives base_logp_in + sum_d N(0,1).log_prob(x_out_d)  (main/default.py:1110-1115).
 * bins (nullable, row stride bins_stride): raw searchsorted result of every rq_splines layer in execution order, D columns per such
 * layer (spline_fns.py:13-19, 252-258) -- the integer output the bit-exact parity tests compare. */
/* LDS bytes one launch of this chain needs in the log-prob / sampling direction (jf_gf_chain_lds_bytes_*) and in the backward direction
 * (jf_gf_chain_inv_bwd_lds_bytes_*), or a negative JF_ERR_* (JF_ERR_UNSUPPORTED: more than the 160 KB of a CU).  param_batch_is_one: the
 * broadcast regime.  The host cuts chains of wide layers (D up to 32: groups of 16 / 32 lanes per row) into launches that fit. */
int64_t jf_gf_chain_lds_bytes_f32(int32_t D, int32_t n_layers, const jf_gf_layer* layers, int32_t param_batch_is_one);
int64_t jf_gf_chain_lds_bytes_f64(int32_t D, int32_t n_layers, const jf_gf_layer* layers, int32_t param_batch_is_one);
int64_t jf_gf_chain_inv_bwd_lds_bytes_f32(int32_t D, int32_t n_layers, const jf_gf_layer* layers, int32_t param_batch_is_one);
int64_t jf_gf_chain_inv_bwd_lds_bytes_f64(int32_t D, int32_t n_layers, const jf_gf_layer* layers, int32_t param_batch_is_one);
int jf_gf_chain_inv_f32(const float* x, int64_t x_stride, const float* log_det_in, const float* params, int64_t param_stride,
                        int32_t param_batch, int64_t B, int32_t D, int32_t n_layers, const jf_gf_layer* layers, float* x_out,
                        int64_t x_out_stride, float* log_det_out, const float* base_logp_in, float* base_logp_out,
                        int64_t* bins, int64_t bins_stride, int32_t* status, void* stream);
int jf_gf_chain_inv_f64(const double* x, int64_t x_stride, const double* log_det_in, const double* params, int64_t param_stride,
                        int32_t param_batch, int64_t B, int32_t D, int32_t n_layers, const jf_gf_layer* layers, double* x_out,
                        int64_t x_out_stride, double* log_det_out, const double* base_logp_in, double* base_logp_out,
                        int64_t* bins, int64_t bins_stride, int32_t* status, void* stream);
/* jf_gf_chain_inv that also writes total_out[b] = base_logp_out[b] + log_det_out[b] (base_logp_out required): for a pdf that ends with this
 * chain, log_prob = log_prob_base + log_det (main/default.py:1110-1117) without a launch of its own -- a quarter of the step at 4096 rows */
int jf_gf_chain_inv_total_f32(const float* x, int64_t x_stride, const float* log_det_in, const float* params, int64_t param_stride,
                              int32_t param_batch, int64_t B, int32_t D, int32_t n_layers, const jf_gf_layer* layers, float* x_out,
                              int64_t x_out_stride, float* log_det_out, const float* base_logp_in, float* base_logp_out, float* total_out,
                              int64_t* bins, int64_t bins_stride, int32_t* status, void* stream);
int jf_gf_chain_inv_total_f64(const double* x, int64_t x_stride, const double* log_det_in, const double* params, int64_t param_stride,
                              int32_t param_batch, int64_t B, int32_t D, int32_t n_layers, const jf_gf_layer* layers, double* x_out,
                              int64_t x_out_stride, double* log_det_out, const double* base_logp_in, double* base_logp_out, double* total_out,
                              int64_t* bins, int64_t bins_stride, int32_t* status, void* stream);

/* sampling direction: layers applied in order 0..n-1; each solves its mixture-CDF map by 25 bisection steps on [-1e5,1e5]
 * + <= 20 Newton steps (row stops when sum_d |update| < 1e-14).  log_det_out = log_det_in - sum log-derivatives. */
/* Co-vector transport through the log-prob direction (ABI v6): cot_out = J^{-T} cot_in with J = d x_out / d x the chain's Jacobian at x -- per
 * layer the same reflections as x and a division by the stage's derivative.  What the implicit-function adjoint of SAMPLING needs: the
 * gradient of a loss on samples x*(theta) is -(dF/dtheta)^T lambda with J^T lambda = g (bisection_n_newton.py:74-93 differentiates through the
 * Newton iterations instead).  x_out / log_det_out receive the chain's values.  Default-option layers and rq_splines stretch; layers with the
 * general options (jf_gf_ext.h) return JF_ERR_UNSUPPORTED. */
int jf_gf_chain_inv_cot_f32(const float* x, int64_t x_stride, const float* params, int64_t param_stride, int32_t param_batch, int64_t B, int32_t D,
                            int32_t n_layers, const jf_gf_layer* layers, const float* cot_in, int64_t cot_in_stride, float* cot_out,
                            int64_t cot_out_stride, float* x_out, int64_t x_out_stride, float* log_det_out, void* stream);
int jf_gf_chain_inv_cot_f64(const double* x, int64_t x_stride, const double* params, int64_t param_stride, int32_t param_batch, int64_t B, int32_t D,
                            int32_t n_layers, const jf_gf_layer* layers, const double* cot_in, int64_t cot_in_stride, double* cot_out,
                            int64_t cot_out_stride, double* x_out, int64_t x_out_stride, double* log_det_out, void* stream);
int jf_gf_chain_fwd_f32(const float* z, int64_t z_stride, const float* log_det_in, const float* params, int64_t param_stride,
                        int32_t param_batch, int64_t B, int32_t D, int32_t n_layers, const jf_gf_layer* layers, float* x_out,
                        int64_t x_out_stride, float* log_det_out, int64_t* bins, int64_t bins_stride, int32_t* status, void* stream);
int jf_gf_chain_fwd_f64(const double* z, int64_t z_stride, const double* log_det_in, const double* params, int64_t param_stride,
                        int32_t param_batch, int64_t B, int32_t D, int32_t n_layers, const jf_gf_layer* layers, double* x_out,
                        int64_t x_out_stride, double* log_det_out, int64_t* bins, int64_t bins_stride, int32_t* status, void* stream);
/* jf_gf_chain_fwd with a START TABLE for broadcast parameters (param_batch == 1): every (layer, coordinate) then inverts one fixed monotone
 * function x(z) for all rows (gaussianization_flow.py:921 calls the bisection / Newton solver per row regardless).  A first launch solves it on
 * 512 intervals of z (+-20 for isigmoid stages, +-8 for the normal-type stages) and keeps each interval's cubic Hermite polynomial in `table`
 * (jf_gf_chain_fwd_table_elems(D, n_layers) elements of the function's precision, scratch of this call); the chain launch starts the
 * reference's Newton stage from the polynomial's value wherever it reproduces the solved interval midpoint to 2e-5, and solves every other
 * lane as jf_gf_chain_fwd does.  Same results to the solver's tolerance, fewer Newton row-steps in `status`.  Ten-component layers at default
 * options use the table; other layers, per-sample parameters and the general-option kernel ignore it.  Worth it from ~10^4 rows. */
int64_t jf_gf_chain_fwd_table_elems(int32_t D, int32_t n_layers);
int jf_gf_chain_fwd_tab_f32(const float* z, int64_t z_stride, const float* log_det_in, const float* params, int64_t param_stride,
                            int32_t param_batch, int64_t B, int32_t D, int32_t n_layers, const jf_gf_layer* layers, float* x_out,
                            int64_t x_out_stride, float* log_det_out, int64_t* bins, int64_t bins_stride, int32_t* status, float* table, void* stream);
int jf_gf_chain_fwd_tab_f64(const double* z, int64_t z_stride, const double* log_det_in, const double* params, int64_t param_stride,
                            int32_t param_batch, int64_t B, int32_t D, int32_t n_layers, const jf_gf_layer* layers, double* x_out,
                            int64_t x_out_stride, double* log_det_out, int64_t* bins, int64_t bins_stride, int32_t* status, double* table, void* stream);

/* Backward of jf_gf_chain_inv_* (classic stretch): vector-Jacobian product for upstream gradients g_x_out (B, D; nullable = 0),
 * g_log_det (B; nullable) and g_base_logp (B; nullable) of the three outputs -- what torch.autograd computes by replaying
 * gf_block._inv_flow_mapping (gaussianization_flow.py:995-1114) + the offset (euclidean_base.py:34-51) layer by layer.  The kernel re-runs
 * the chain itself (nothing has to be saved by the forward launch) and writes g_x (B, D) and g_params:
 *   param_batch == B: (B, P) rows in the layout of `params` (what the amortisation MLP's backward consumes);
 *   param_batch == 1: jf_gf_chain_inv_bwd_partials(B, D) rows of partial sums (one per workgroup; rows of workgroups that took no tile are zero), to be added up by the caller.  Inside a workgroup the
 *   sums are accumulated in float64 by LDS atomics (both precisions): the float32 result is reproducible in practice, the float64 one to the last
 *   bits only up to the order of those additions.
 * The gradients of log_det_in and base_logp_in are g_log_det and g_base_logp themselves. */
int64_t jf_gf_chain_inv_bwd_partials(int64_t B, int32_t D);
int jf_gf_chain_inv_bwd_f32(const float* x, int64_t x_stride, const float* params, int64_t param_stride, int32_t param_batch, int64_t B,
                            int32_t D, int32_t n_layers, const jf_gf_layer* layers, const float* g_x_out, int64_t g_x_out_stride,
                            const float* g_log_det, const float* g_base_logp, float* g_x, int64_t g_x_stride, float* g_params,
                            int64_t g_params_stride, int32_t* status, void* stream);
int jf_gf_chain_inv_bwd_f64(const double* x, int64_t x_stride, const double* params, int64_t param_stride, int32_t param_batch, int64_t B,
                            int32_t D, int32_t n_layers, const jf_gf_layer* layers, const double* g_x_out, int64_t g_x_out_stride,
                            const double* g_log_det, const double* g_base_logp, double* g_x, int64_t g_x_stride, double* g_params,
                            int64_t g_params_stride, int32_t* status, void* stream);

/* A whole conditional / autoregressive Euclidean block in ONE launch, log-prob direction: the default amortisation MLP
 * params = tanh(in @ W1^T + b1) @ W2^T + b2  (nn.Sequential(Linear, Tanh, Linear), main/default.py:656-670, input = cat of conditional
 * input and the embeddings of the previous sub-manifolds, :946-962) followed by jf_gf_chain_inv on those per-sample parameters
 * (main/default.py:998-1031).  The (B, sum of the layers' row lengths) parameter block never reaches HBM: it is produced on the matrix
 * cores into LDS tiles and consumed there ("fused" accounting).  W2 (N, H) with N = sum of the layers' row lengths.
 * Limits: D in {3, 4}, K1 <= 28, H <= 128, classic stretch; otherwise JF_ERR_UNSUPPORTED (use jf_mlp2 + jf_gf_chain_inv). */
int jf_cond_gf_chain_inv_f32(const float* in, int64_t in_stride, const float* W1, int64_t w1_stride, const float* b1, const float* W2,
                             int64_t w2_stride, const float* b2, int32_t K1, int32_t H, const float* x, int64_t x_stride,
                             const float* log_det_in, int64_t B, int32_t D, int32_t n_layers, const jf_gf_layer* layers, float* x_out,
                             int64_t x_out_stride, float* log_det_out, const float* base_logp_in, float* base_logp_out, int32_t* status,
                             void* stream);
int jf_cond_gf_chain_inv_f64(const double* in, int64_t in_stride, const double* W1, int64_t w1_stride, const double* b1, const double* W2,
                             int64_t w2_stride, const double* b2, int32_t K1, int32_t H, const double* x, int64_t x_stride,
                             const double* log_det_in, int64_t B, int32_t D, int32_t n_layers, const jf_gf_layer* layers, double* x_out,
                             int64_t x_out_stride, double* log_det_out, const double* base_logp_in, double* base_logp_out, int32_t* status,
                             void* stream);

/* The same block with the 128 -> N product on split-bf16 MFMA (every f32 operand = three bf16 pieces, the six products whose piece
 * indices sum to <= 2, f32 accumulation: products exact, dropped terms <= 3 * 2^-24 |w||h|) and the parameter block kept in the register
 * file (W2's rows are permuted so that the MFMA result registers of lane (row, coordinate) are that lane's parameters).  float32 only;
 * D in {3, 4}; layers at the reference's default options (num_kde 10, smooth-saturation widths without clamping, fitted + regulated
 * normalisation, hh_iter <= 4, classic stretch); K1 <= 28, H <= 128.  jf_cond_gf_packed_bytes returns the size of the packed image or
 * JF_ERR_UNSUPPORTED; jf_cond_gf_pack_f32 builds it from W2 (N, H) / b2 (N) -- redo whenever the weights change (one small launch);
 * jf_cond_gf_chain_inv_split_f32 = jf_cond_gf_chain_inv_f32 with (W2, b2) replaced by the packed image. */
int64_t jf_cond_gf_packed_bytes(int32_t D, int32_t n_layers, const jf_gf_layer* layers);
/* The SAMPLING direction of the same block in one launch (amortisation MLP + the g layers' bisection / Newton solves, first layer first;
 * main/default.py:1420-1506 with gaussianization_flow.py:911-989 and bisection_n_newton.py:11-135): same packed image, same limits.  Each
 * layer's parameters are regulated once in the MFMA result registers and the 25 + <= 20 mixture evaluations of its solve read registers only.
 * z: base points (B, D); status counts Newton row-steps / non-converged / non-finite rows like jf_gf_chain_fwd. */
int jf_cond_gf_chain_fwd_split_f32(const float* in, int64_t in_stride, const float* W1, int64_t w1_stride, const float* b1, const void* packed,
                                   int32_t K1, int32_t H, const float* z, int64_t z_stride, const float* log_det_in, int64_t B, int32_t D,
                                   int32_t n_layers, const jf_gf_layer* layers, float* x_out, int64_t x_out_stride, float* log_det_out,
                                   int32_t* status, void* stream);
/* The same block with a selectable split arithmetic for the 128 -> N product (pack and launch must name the same one):
 *   JF_SPLIT_BF16X3  three bf16 pieces per operand, six products (exact f32 products, dropped terms <= 3 * 2^-24) -- what the entry points
 *                    without the suffix 2 use;
 *   JF_SPLIT_F16X2   two f16 pieces per operand (11 + 11 bits; W2 and h scaled by powers of two so that the pieces are normal numbers, the
 *                    low pieces by another 2^11), three products, half the MFMA work and two thirds of the LDS traffic.  Representation
 *                    error <= 2^-22 per operand: below the rounding of the f32 accumulation (measured rms error of a parameter 2.4e-8 vs
 *                    6.8e-8 for a plain f32 matrix product of the same operands).
 * jf_cond_gf_chain_split2_f32: direction JF_DIR_INV (x -> z; base_logp_* and aux as in jf_cond_gf_chain_inv_split_save_f32, aux may be NULL)
 * or JF_DIR_FWD (sampling; base_logp_in / base_logp_out / aux must be NULL). */
#define JF_SPLIT_BF16X3 0
#define JF_SPLIT_F16X2 1
#define JF_DIR_INV 0
#define JF_DIR_FWD 1
int64_t jf_cond_gf_packed_bytes2(int32_t D, int32_t n_layers, const jf_gf_layer* layers, int32_t arithmetic);
int jf_cond_gf_pack2_f32(const float* W2, int64_t w2_stride, const float* b2, int32_t H, int32_t D, int32_t n_layers, const jf_gf_layer* layers,
                         int32_t arithmetic, void* packed, void* stream);
int jf_cond_gf_chain_split2_f32(int32_t direction, int32_t arithmetic, const float* in, int64_t in_stride, const float* W1, int64_t w1_stride,
                                const float* b1, const void* packed, int32_t K1, int32_t H, const float* x, int64_t x_stride,
                                const float* log_det_in, int64_t B, int32_t D, int32_t n_layers, const jf_gf_layer* layers, float* x_out,
                                int64_t x_out_stride, float* log_det_out, const float* base_logp_in, float* base_logp_out, float* aux,
                                int32_t* status, void* stream);
/* the same launch with the MLP's input rows READ WHERE THEY ARE: `segments` (<= 4, widths adding up to K1) describe
 * cat[conditional_input, embed(x_0), ...] (main/default.py:946-962) as column ranges of the caller's tensors -- see jf_conditioning_rows --, so
 * neither a conditioning launch nor the (B, K1) matrix exists.
 * ld_pre / blp_pre / total (all nullable; log-prob direction with base_logp_out only): this block is the LAST of a pdf whose blocks were
 * evaluated independently (see jf_combine_rows) -- the lists (<= 4 entries each) hold the log-dets / base log-probs of the blocks before it,
 * log_det_out / base_logp_out then receive the pdf's totals, summed in list order with this block last, and total = base_logp_out +
 * log_det_out: bit for bit what jf_combine_rows returns, without its launch (main/default.py:1110-1117: log_prob = log_prob_base + log_det). */
#define JF_MAX_ROW_LISTS 16
typedef struct jf_row_list {
    const void* p[JF_MAX_ROW_LISTS];
    int32_t n;
} jf_row_list;
int jf_cond_gf_chain_split3_f32(int32_t direction, int32_t arithmetic, const jf_cond_segment* segments, int32_t n_segments, const float* W1,
                                int64_t w1_stride, const float* b1, const void* packed, int32_t K1, int32_t H, const float* x, int64_t x_stride,
                                const float* log_det_in, int64_t B, int32_t D, int32_t n_layers, const jf_gf_layer* layers, float* x_out,
                                int64_t x_out_stride, float* log_det_out, const float* base_logp_in, float* base_logp_out, float* aux,
                                const jf_row_list* ld_pre, const jf_row_list* blp_pre, float* total, int32_t* status, void* stream);
/* Training step of the same block (float32; replaces torch.autograd's replay of main/default.py:656-670, 946-962, 998-1031 +
 * gaussianization_flow.py:995-1114 for the loss of examples/jammy_flows.py:381-412).
 * jf_cond_gf_chain_inv_split_save_f32 = jf_cond_gf_chain_inv_split_f32 that also leaves, per (layer, row, coordinate lane), the layer's input
 * coordinate and its normalised mixture sums (cdf, sf, pdf, 1 / N) in aux (jf_cond_gf_aux_floats(B, n_layers) floats, 16-byte aligned).
 * jf_cond_gf_chain_inv_split_bwd_f32 recomputes the hidden activations and each layer's parameters like the forward launch, overwrites the
 * parameter registers with their gradients and returns, for upstream gradients g_x_out (B, D) / g_log_det (B) / g_base_logp (B) (each may be
 * NULL = zero):  g_x (B, D);  h (B, H), the hidden activations;  g_h (B, H) = g_params W2, the gradient arriving at them;  g_pp
 * (B, n_layers * 144), the gradient of the parameter rows in PACKED column order [layer][coordinate lane 0..3][slot 0..35] (slot order: 10
 * means, 10 log-widths, 10 log-weights, 4 Householder vectors, offset, pad; absent columns are zero).  g_W2 / g_b2 are then one
 * jf_linear_wgrad_split_f32(g_pp, h) whose rows the caller gathers back into W2's row order; g_W1 / g_b1 one jf_mlp_hidden_bwd_f32(g_h).
 * packedT: W2^T as MFMA fragments (jf_cond_gf_bwd_packed_bytes / jf_cond_gf_bwd_pack_f32, redo whenever W2 changes).  z: x_out of the
 * forward launch.  Same limits as jf_cond_gf_chain_inv_split_f32, H % 4 == 0. */
int jf_cond_gf_chain_inv_split_save_f32(const float* in, int64_t in_stride, const float* W1, int64_t w1_stride, const float* b1, const void* packed,
                                        int32_t K1, int32_t H, const float* x, int64_t x_stride, const float* log_det_in, int64_t B, int32_t D,
                                        int32_t n_layers, const jf_gf_layer* layers, float* x_out, int64_t x_out_stride, float* log_det_out,
                                        const float* base_logp_in, float* base_logp_out, float* aux, int32_t* status, void* stream);
int64_t jf_cond_gf_aux_floats(int64_t B, int32_t n_layers);
int64_t jf_cond_gf_bwd_packed_bytes(int32_t D, int32_t n_layers, const jf_gf_layer* layers);
int jf_cond_gf_bwd_pack_f32(const float* W2, int64_t w2_stride, int32_t H, int32_t D, int32_t n_layers, const jf_gf_layer* layers, void* packedT,
                            void* stream);
int jf_cond_gf_chain_inv_split_bwd_f32(const float* in, int64_t in_stride, const float* W1, int64_t w1_stride, const float* b1, const void* packed,
                                       const void* packedT, int32_t K1, int32_t H, const float* z, int64_t z_stride, const float* aux, int64_t B,
                                       int32_t D, int32_t n_layers, const jf_gf_layer* layers, const float* g_x_out, int64_t g_x_out_stride,
                                       const float* g_log_det, const float* g_base_logp, float* g_x, int64_t g_x_stride, float* g_pp,
                                       int64_t g_pp_stride, float* h_out, int64_t h_stride, float* g_h, int64_t g_h_stride, float* g_absmax,
                                       void* stream);
/* g_absmax (optional, one float the caller zeroed): the launch leaves max |g_pp| there (atomic max), the power-of-two scale for
 * jf_linear_wgrad_split16_f32 = jf_linear_wgrad_split_f32 on f16 pairs (three products instead of six): g scaled so that g_absmax lands in
 * [2^14, 2^15), `in` by 2^in_exp (14 for activations in (-1, 1)); absolute error of an entry <= max(2^-22 |g|, 2^-39 g_absmax). */
int jf_linear_wgrad_split16_f32(const float* g, int64_t g_stride, const float* in, int64_t in_stride, int64_t B, int32_t K, int32_t N,
                                const float* g_absmax, int32_t in_exp, float* partial_w, float* partial_b, void* stream);
/* 16-row groups per wave of the split kernel: 0 = chosen by batch size (default), 1 or 2 force a variant (process-wide; for A/B timing and the
 * determinism stress tests, which must cover both).  Returns the previous setting; other values only query. */
int jf_cond_gf_split_row_groups(int32_t row_groups);
int jf_cond_gf_pack_f32(const float* W2, int64_t w2_stride, const float* b2, int32_t H, int32_t D, int32_t n_layers, const jf_gf_layer* layers,
                        void* packed, void* stream);
int jf_cond_gf_chain_inv_split_f32(const float* in, int64_t in_stride, const float* W1, int64_t w1_stride, const float* b1, const void* packed,
                                   int32_t K1, int32_t H, const float* x, int64_t x_stride, const float* log_det_in, int64_t B, int32_t D,
                                   int32_t n_layers, const jf_gf_layer* layers, float* x_out, int64_t x_out_stride, float* log_det_out,
                                   const float* base_logp_in, float* base_logp_out, int32_t* status, void* stream);

/* A conditional Euclidean block whose parameters come from a two-stage LOW-RANK AmortizableMLP (highway_mode 0), in ONE launch:
 * params = U2 (V2 tanh(W1 c + b1)) + b2 with W1 = U1 V1 (rank r1) or a full matrix (V1 == NULL, U1 = W1 (H, K1)) and a rank-r2 last stage
 * (amortizable_mlp.py:508-578), followed by jf_gf_chain_inv on those parameters.  Every parameter of a row is b2[j] + <U2[j, :], t2> for ONE
 * r2-vector t2 per row, so the (B, N) block is never formed: lanes regenerate their parameters from U2 / b2 held in LDS.
 * Limits: K1 <= 32, H <= 128, r1, r2 <= 16, D <= 8, layers at the reference's default options (as jf_cond_gf_chain_inv_split_f32, with up to 8
 * Householder reflections); otherwise JF_ERR_UNSUPPORTED.  U1 (H, r1), V1 (r1, K1), V2 (r2, H), U2 (N, r2): dense row-major. */
int jf_amlp_gf_chain_inv_f32(const float* in, int64_t in_stride, const float* V1, const float* U1, const float* b1, const float* V2, const float* U2,
                             const float* b2, int32_t K1, int32_t H, int32_t r1, int32_t r2, const float* x, int64_t x_stride,
                             const float* log_det_in, int64_t B, int32_t D, int32_t n_layers, const jf_gf_layer* layers, float* x_out,
                             int64_t x_out_stride, float* log_det_out, const float* base_logp_in, float* base_logp_out, int32_t* status,
                             void* stream);
int jf_amlp_gf_chain_inv_f64(const double* in, int64_t in_stride, const double* V1, const double* U1, const double* b1, const double* V2,
                             const double* U2, const double* b2, int32_t K1, int32_t H, int32_t r1, int32_t r2, const double* x, int64_t x_stride,
                             const double* log_det_in, int64_t B, int32_t D, int32_t n_layers, const jf_gf_layer* layers, double* x_out,
                             int64_t x_out_stride, double* log_det_out, const double* base_logp_in, double* base_logp_out, int32_t* status,
                             void* stream);
/* The SAMPLING direction of the same block in one launch (float64, both ranks <= 8, H % 16 == 0: the matrix-core variant; otherwise
 * JF_ERR_UNSUPPORTED and the caller runs jf_amlp2 + jf_gf_chain_fwd): the (B, N) parameter block is never materialised, every layer's
 * bisection / Newton solves read register-resident derived rows.  z: base points (B, D); status as jf_gf_chain_fwd. */
int jf_amlp_gf_chain_fwd_f64(const double* in, int64_t in_stride, const double* V1, const double* U1, const double* b1, const double* V2,
                             const double* U2, const double* b2, int32_t K1, int32_t H, int32_t r1, int32_t r2, const double* z, int64_t z_stride,
                             const double* log_det_in, int64_t B, int32_t D, int32_t n_layers, const jf_gf_layer* layers, double* x_out,
                             int64_t x_out_stride, double* log_det_out, int32_t* status, void* stream);
/* Training: the head of a two-stage low-rank AmortizableMLP -- t1 = V1 c, h = tanh(U1 t1 + b1), t2 = V2 h (amortizable_mlp.py:508-578: everything
 * in front of the last stage's U product; V1 (r1, K1), U1 (H, r1), b1 (H), V2 (r2, H) dense row-major; K1 <= 32, H <= 128 and a multiple of 16,
 * ranks <= 8, float64 -- else JF_ERR_UNSUPPORTED).  jf_lowrank_head_f64 writes t1 (B, 8), h (B, H) and t2 (B, 8) (columns >= the rank are zero)
 * in one launch; jf_lowrank_head_bwd_f64 takes g_t2 (B rows, stride g_t2_stride) and returns g_V1, g_U1, g_b1, g_V2 (the shapes of the weights) and,
 * when g_in is not NULL, g_in (B, K1) -- one launch + one reduction, nothing of size (B, H) written.  `workspace`: jf_lowrank_head_workspace_doubles. */
int jf_lowrank_head_f64(const double* in, int64_t in_stride, const double* V1, const double* U1, const double* b1, const double* V2, int64_t B,
                        int32_t K1, int32_t H, int32_t r1, int32_t r2, double* t1, double* h, double* t2, void* stream);
int64_t jf_lowrank_head_workspace_doubles(int64_t B, int32_t K1, int32_t H);
int jf_lowrank_head_bwd_f64(const double* in, int64_t in_stride, const double* V1, const double* U1, const double* V2, int64_t B, int32_t K1,
                            int32_t H, int32_t r1, int32_t r2, const double* t1, const double* h, const double* g_t2, int64_t g_t2_stride,
                            double* g_in, int64_t g_in_stride, double* g_V1, double* g_U1, double* g_b1, double* g_V2, double* workspace,
                            void* stream);
/* Training on a low-rank last MLP stage (float64, r2 <= 8, D <= 8, default layer options -- else JF_ERR_UNSUPPORTED): the chain of g layers on
 * the parameter rows U2 t2[row] + b2 (t2 (B, r2), U2 (N, r2), b2 (N): amortizable_mlp.py:508-578), the (B, N) block never materialised.
 * jf_lowrank_gf_chain_inv_f64: jf_gf_chain_inv's outputs; aux (nullable) receives (n_layers, 5, 2, B, 4) doubles -- every layer's input
 * coordinates and normalised mixture sums -- which jf_lowrank_gf_chain_inv_bwd_f64 reads instead of re-running the chain.
 * jf_lowrank_gf_chain_inv_bwd_f64: what torch.autograd returns for (x, t2, U2, b2) given the upstream gradients of (x_out, log_det_out,
 * base_logp_out) (each nullable): g_x (B, D), g_t2 (B, 8: columns >= r2 are zero), g_U2 (N, r2), g_b2 (N).  One launch per layer (the
 * parameter gradients are contracted with U2 and [t2 | 1] in the matrix-core registers) + one reduction; `workspace` holds
 * jf_lowrank_gf_workspace_doubles(B, n_layers) doubles.  x_out: the forward's output (needed when g_base_logp is given). */
int jf_lowrank_gf_chain_inv_f64(const double* t2, int64_t t2_stride, const double* U2, const double* b2, int32_t r2, const double* x, int64_t x_stride,
                                const double* log_det_in, int64_t B, int32_t D, int32_t n_layers, const jf_gf_layer* layers, double* x_out,
                                int64_t x_out_stride, double* log_det_out, const double* base_logp_in, double* base_logp_out, double* aux,
                                int32_t* status, void* stream);
int64_t jf_lowrank_gf_workspace_doubles(int64_t B, int32_t n_layers);
int jf_lowrank_gf_chain_inv_bwd_f64(const double* t2, int64_t t2_stride, const double* U2, const double* b2, int32_t r2, const double* aux,
                                    const double* x_out, int64_t x_out_stride, int64_t B, int32_t D, int32_t n_layers, const jf_gf_layer* layers,
                                    const double* g_x_out, int64_t g_x_out_stride, const double* g_log_det, const double* g_base_logp, double* g_x,
                                    int64_t g_x_stride, double* g_t2, double* g_U2, double* g_b2, double* workspace, int32_t* status, void* stream);

/* The same two-stage low-rank AmortizableMLP alone, ONE launch instead of four dense launches:
 * out (B, N) = U2 (V2 tanh(W1 in + b1)) + b2 (same operand conventions and limits as jf_amlp_gf_chain_inv). */
int jf_amlp2_f32(const float* in, int64_t in_stride, const float* V1, const float* U1, const float* b1, const float* V2, const float* U2,
                 const float* b2, int64_t B, int32_t K1, int32_t H, int32_t r1, int32_t r2, int32_t N, float* out, int64_t out_stride, void* stream);
int jf_amlp2_f64(const double* in, int64_t in_stride, const double* V1, const double* U1, const double* b1, const double* V2, const double* U2,
                 const double* b2, int64_t B, int32_t K1, int32_t H, int32_t r1, int32_t r2, int32_t N, double* out, int64_t out_stride,
                 void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Dense layer of the parameter-emitting MLPs: out = act(in @ W^T + bias)   (MFMA)
 * replaces torch.nn.Linear + tanh of the default nn.Sequential (main/default.py:656-670) and the U / V^T products of
 * AmortizableMLP._apply_amortized_mlp with permanent parameters (amortizable_mlp.py:508-578).
 * W row-major (N, K) with row stride w_stride; bias nullable; act: 0 identity, 1 tanh.
 * ------------------------------------------------------------------------------------------------------------ */
int jf_linear_f32(const float* in, int64_t in_stride, const float* W, int64_t w_stride, const float* bias, int64_t B, int32_t K,
                  int32_t N, int32_t act, float* out, int64_t out_stride, void* stream);
int jf_linear_f64(const double* in, int64_t in_stride, const double* W, int64_t w_stride, const double* bias, int64_t B,
                  int32_t K, int32_t N, int32_t act, double* out, int64_t out_stride, void* stream);

/* Backward of jf_linear with respect to the weights: the product that reduces over the batch,
 *   g_W (N, K) = sum_b g[b, :]^T in[b, :],  g_bias (N) = sum_b g[b, :]        (what autograd's `g.t() @ inp`, `g.sum(0)` compute for nn.Linear)
 * split over the grid: partials_w (S, N, K) and partials_b (S, N; nullable) receive one slab per row chunk, S = jf_linear_wgrad_splits_<dt>(B, K, N)
 * (ABI 3: the shape decides between the tiled MFMA kernel and the streaming kernel for min(N, K) <= 16); the caller adds the slabs
 * (deterministic, no atomics).  K <= 128 or min(N, K) <= 16, otherwise JF_ERR_UNSUPPORTED (use a library GEMM). */
int64_t jf_linear_wgrad_splits_f32(int64_t B, int32_t K, int32_t N);
int64_t jf_linear_wgrad_splits_f64(int64_t B, int32_t K, int32_t N);
int jf_linear_wgrad_f32(const float* g, int64_t g_stride, const float* in, int64_t in_stride, int64_t B, int32_t K, int32_t N, float* partials_w,
                        float* partials_b, void* stream);
int jf_linear_wgrad_f64(const double* g, int64_t g_stride, const double* in, int64_t in_stride, int64_t B, int32_t K, int32_t N, double* partials_w,
                        double* partials_b, void* stream);

/* Whole backward of a narrow Linear - tanh - Linear head (K1 <= 32 inputs, H <= 128 hidden units, N <= 16 outputs: e.g. the 4 -> 128 -> 10 MLP
 * of mlp_predictors (main/default.py:656-670) that parametrises an 'f' layer) in one launch: the hidden activations are recomputed from x, and
 * the gradients of W1, b1, W2, b2 -- what autograd computes with six launches -- leave as partial slabs, S = jf_mlp2_small_bwd_slabs(B):
 *   slab (S, H, K1 + 1 + N): per hidden unit j  [g_W1[j][0..K1) | g_b1[j] | g_W2[0..N)[j]],   slab_b2 (S, N);   the caller adds over S.
 * No gradient with respect to x (JF_ERR_UNSUPPORTED shapes / a needed input gradient: use the per-layer entry points). */
int64_t jf_mlp2_small_bwd_slabs(int64_t B);
int jf_mlp2_small_bwd_f32(const float* x, int64_t x_stride, const float* W1, int64_t w1_stride, const float* b1, const float* W2, int64_t w2_stride,
                          const float* g_out, int64_t g_stride, int64_t B, int32_t K1, int32_t H, int32_t N, float* slab, float* slab_b2, void* stream);
int jf_mlp2_small_bwd_f64(const double* x, int64_t x_stride, const double* W1, int64_t w1_stride, const double* b1, const double* W2, int64_t w2_stride,
                          const double* g_out, int64_t g_stride, int64_t B, int32_t K1, int32_t H, int32_t N, double* slab, double* slab_b2, void* stream);

/* The first layer of such an MLP alone (K1 <= 32, H <= 128; any second layer): g_hidden (B, H) = the gradient with respect to the tanh OUTPUT
 * (the second layer's grad_output @ weight) -> tanh derivative + weight / bias gradient of the first layer in one launch, the activations
 * recomputed from x.  slab (S, H, K1 + 1): [g_W1[j][0..K1) | g_b1[j]], S = jf_mlp2_small_bwd_slabs(B), added up by the caller. */
int jf_mlp_hidden_bwd_f32(const float* x, int64_t x_stride, const float* W1, int64_t w1_stride, const float* b1, const float* g_hidden,
                          int64_t g_stride, int64_t B, int32_t K1, int32_t H, float* slab, void* stream);
int jf_mlp_hidden_bwd_f64(const double* x, int64_t x_stride, const double* W1, int64_t w1_stride, const double* b1, const double* g_hidden,
                          int64_t g_stride, int64_t B, int32_t K1, int32_t H, double* slab, void* stream);

/* One AmortizableMLP stage with PER-SAMPLE weights (amortize_everything / fully_amortized_pdf: _apply_amortized_mlp with extra_inputs,
 * amortizable_mlp.py:508-578): out[b] = act(W_b in[b] + bias_b) (+ residual[b]); `segment` points at this stage's [U | V | bias] slice of
 * row 0 of the per-sample parameter block (row stride segment_stride): rank == 0: U = W (n_out x n_in); else U (n_out x rank), V (rank x n_in).
 * n_in, n_out <= 1024, rank <= 64.  jf_amlp_stage_bwd: g_out -> g_in (nullable) and g_segment (B, n_u + n_v + n_b); `y` = the stage's
 * activated output (needed for act == 1). */
#define JF_DECLARE_AMLP(T, suffix)                                                                                                         \
    int jf_amlp_stage_##suffix(const T* in, int64_t in_stride, const T* segment, int64_t segment_stride, int64_t B, int32_t n_in,            \
                               int32_t n_out, int32_t rank, int32_t has_bias, int32_t act, const T* residual, int64_t residual_stride,      \
                               T* out, int64_t out_stride, void* stream);                                                                   \
    int jf_amlp_stage_bwd_##suffix(const T* in, int64_t in_stride, const T* segment, int64_t segment_stride, int64_t B, int32_t n_in,        \
                                   int32_t n_out, int32_t rank, int32_t has_bias, int32_t act, const T* y, int64_t y_stride,                \
                                   const T* g_out, int64_t g_out_stride, T* g_in, int64_t g_in_stride, T* g_segment,                        \
                                   int64_t g_segment_stride, void* stream);
JF_DECLARE_AMLP(float, f32)
JF_DECLARE_AMLP(double, f64)

/* The default amortisation MLP with ONE hidden layer in a single launch: out = tanh(in @ W1^T + b1) @ W2^T + b2
 * (nn.Sequential(Linear, Tanh, Linear), main/default.py:656-670); hidden activations stay in registers (never in HBM).  K1 <= 32, H <= 128,
 * otherwise JF_ERR_UNSUPPORTED (use jf_linear per layer).  W1 (H, K1), W2 (N, H) row-major. */
int jf_mlp2_f32(const float* in, int64_t in_stride, const float* W1, int64_t w1_stride, const float* b1, const float* W2, int64_t w2_stride,
                const float* b2, int64_t B, int32_t K1, int32_t H, int32_t N, float* out, int64_t out_stride, void* stream);
int jf_mlp2_f64(const double* in, int64_t in_stride, const double* W1, int64_t w1_stride, const double* b1, const double* W2, int64_t w2_stride,
                const double* b2, int64_t B, int32_t K1, int32_t H, int32_t N, double* out, int64_t out_stride, void* stream);
/* jf_mlp2_f64 with the second product on the INT8 matrix cores (float64 matrix cores run at the float64 vector rate on MI355X: 2.94 ms per 2^20
 * rows for the 128 -> 548 block of the benchmark configuration; int8 MFMA: 50x that rate, exact int32 accumulation).  h = tanh(..) in [-1, 1]
 * and every row of W2 (scaled by its own power of two) are cut into `slices` balanced base-128 digits, the digit pairs (i, j) with
 * i + j < slices are multiplied with v_mfma_i32_16x16x64_i8 and the levels i + j recombined in float64: an error-free product of the
 * truncated operands.  slices = 6: operands kept to 2^-41, ~2e-12 |w|_max per output; slices = 5: 2^-34, ~3e-10.
 * jf_mlp2_i8_packed_bytes / jf_mlp2_i8_pack_f64: size and contents of the digit image of W2 (N, H) / b2 (N; may be NULL) -- device memory,
 * 16-byte aligned, rebuilt whenever the weights change.  jf_mlp2_i8_f64: arguments as jf_mlp2_f64 with the image in place of W2 / b2.
 * K1 <= 28, H <= 128, otherwise JF_ERR_UNSUPPORTED.  Weights must be finite (the exact path propagates non-finite weights, this one cannot). */
int64_t jf_mlp2_i8_packed_bytes(int32_t N, int32_t slices);
int jf_mlp2_i8_pack_f64(const double* W2, int64_t w2_stride, const double* b2, int32_t H, int32_t N, int32_t slices, void* packed, void* stream);
int jf_mlp2_i8_f64(const double* in, int64_t in_stride, const double* W1, int64_t w1_stride, const double* b1, const void* packed, int64_t B,
                   int32_t K1, int32_t H, int32_t N, int32_t slices, double* out, int64_t out_stride, void* stream);
/* the same with the input rows read in place from `segments` (see jf_cond_gf_chain_split3_f32) */
int jf_mlp2_i8_seg_f64(const jf_cond_segment* segments, int32_t n_segments, const double* W1, int64_t w1_stride, const double* b1, const void* packed,
                       int64_t B, int32_t K1, int32_t H, int32_t N, int32_t slices, double* out, int64_t out_stride, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * 't' affine flow / multivariate normal (replaces mvn_block._inv_flow_mapping / _flow_mapping + the euclidean_base offset:
 * jammy_flows/layers/euclidean/multivariate_normal.py:226-263, layers/matrix_fns.py:4-146, euclidean_base.py:34-76).
 * Row: [offset D if model_offset][raw log-diagonal: 1 (diagonal_symmetric) | D (diagonal, full)][strictly-lower entries D(D-1)/2 (full),
 * sub-diagonal by sub-diagonal from the bottom-left corner, matrix_fns.py:36-50].  log L_ii = the same width regulators as 'g'.  D <= 32.
 * base_logp_out (nullable) as for jf_gf_chain_inv.  jf_t_layer_inv_bwd: backward of the log-prob direction (same conventions as
 * jf_<fam>_chain_inv_bwd below: per-sample g_params (B, P), broadcast row sums ADDED into (1, P)); reverse mode in closed form: one forward
 * and one backward substitution per row.
 * ------------------------------------------------------------------------------------------------------------ */
enum { JF_T_IDENTITY = 0, JF_T_DIAGONAL_SYMMETRIC = 1, JF_T_DIAGONAL = 2, JF_T_FULL = 3 };
typedef struct jf_t_layer { int32_t cov_type, model_offset, width_mode, clamp_widths; double width_min, width_max; } jf_t_layer;
#define JF_DECLARE_T(T, suffix)                                                                                                          \
    int jf_t_layer_inv_##suffix(const T* x, int64_t x_stride, const T* log_det_in, const T* params, int64_t param_stride, int32_t param_batch, \
                                int64_t B, int32_t D, const jf_t_layer* layer, T* x_out, int64_t x_out_stride, T* log_det_out,              \
                                const T* base_logp_in, T* base_logp_out, int32_t* status, void* stream);                                  \
    int jf_t_layer_fwd_##suffix(const T* z, int64_t z_stride, const T* log_det_in, const T* params, int64_t param_stride, int32_t param_batch, \
                                int64_t B, int32_t D, const jf_t_layer* layer, T* x_out, int64_t x_out_stride, T* log_det_out,              \
                                const T* base_logp_in, T* base_logp_out, int32_t* status, void* stream);                                  \
    int jf_t_layer_inv_bwd_##suffix(const T* x, int64_t x_stride, const T* params, int64_t param_stride, int32_t param_batch, int64_t B,      \
                                    int32_t D, const jf_t_layer* layer, const T* g_x_out, int64_t g_x_out_stride, const T* g_log_det,       \
                                    const T* g_base_logp, T* g_x, int64_t g_x_stride, T* g_params, int64_t g_params_stride,                \
                                    int32_t* status, void* stream);
JF_DECLARE_T(float, f32)
JF_DECLARE_T(double, f64)

/* ------------------------------------------------------------------------------------------------------------
 * Manifold layers (interval / S1 / S2).  All take / return INTRINSIC coordinates (interval value, angle, (theta, phi));
 * `first` = the layer also applies the chart to / from the Euclidean base space (euclidean_to_{interval,sphere}_as_first).
 * A chain = the layers of one sub-manifold block, applied n-1..0 in the log-prob direction and 0..n-1 when sampling;
 * params row = the layers' rows concatenated in layer order.  `bins` (nullable) receives the raw spline bin index of every
 * spline call in call order, one int64 column per call (what the reference's spline_fns.searchsorted returns; -2 = not called
 * for this row), row stride bins_stride.
 * ------------------------------------------------------------------------------------------------------------ */
#define JF_MAX_MCHAIN 4
#define JF_MAX_NESTED 4

/* width / height / derivative bookkeeping shared by 'r' and 'o'
 * (layers/intervals/rational_quadratic_spline.py:99-178, layers/spheres/splines_1d.py:39-109); row: [w n_w][h n_h][d n_d] */
typedef struct jf_spline_opts {
    int32_t num_bins, smooth, fix_first, fix_second, independent, fix_bd, n_w, n_h, n_d, reserved;
    double fix_bd_value; /* un-softplus'ed value of a fixed boundary derivative */
    double min_w, min_h, min_d, ratio; /* ratio <= 0: restrict_max_min_width_height_ratio off */
} jf_spline_opts;

/* 'r' rational-quadratic spline on [lo, hi] (rational_quadratic_spline.py:180-400 + interval_base.py:33-79) */
typedef struct jf_r_layer { jf_spline_opts sp; double lo, hi; int32_t first, reserved; } jf_r_layer;
/* 'o' circular spline (splines_1d.py:111-306 + sphere_base.py:601-695); row: [householder hh_iter*2][spline row] */
typedef struct jf_o_layer { jf_spline_opts sp; int32_t natural_direction, hh_iter, first, reserved; } jf_o_layer;
/* 'm' Moebius mixture (moebius_1d.py:57-259, bisection_n_newton.py:137-256); row: [householder hh_iter*2][(wx,wy,logit-len,log-w) x nc] */
typedef struct jf_m_layer { int32_t num_components, natural_direction, hh_iter, first, omega_pars; } jf_m_layer;
/* omega_pars: 4 = (omega_x, omega_y, logit length, log weight) per component (use_moebius_xyz_parametrization, the default); 3 = (omega angle,
 * logit length, log weight) (moebius_1d.py:39-46, 175-178); 0 is read as 4. */
/* hh_iter of every sphere layer also encodes the reference's rotation_mode (sphere_base.py:112-240): >= 0 = that many Householder reflections
 * (hh_iter * E raw vectors, E = embedding dimension); -1 = "angles" (Givens rotations, E (E - 1) / 2 angles); -2 = "xyz" (3 parameters, S2);
 * -3 = "quaternion" (4 parameters, S2). */
/* 'f' von-Mises-Fisher z-scaling + optional vertical 'r' / circular 'o' flows (fvm_2d.py:273-726);
 * row: [householder hh_iter*3][log kappa][vertical rows][circular rows]
 * correlated != 0 (add_correlated_rq_spline_flow, fvm_2d.py:244-262, 406-409, 575-578): the circular rows are not in the row but are
 * emitted PER SAMPLE by a tanh MLP  z -> corr_hidden -> sum(circular row lengths)  whose own weights are in the row
 * (amortize_everything layout of AmortizableMLP, amortizable_mlp.py:284-375: stage 1 [W1 H][b1 H] (input dim 1 => full matrix),
 * stage 2 [U n_out x H][b2] if corr_full2 else [U n_out x rank][V rank x H][b2]); row: [householder][log kappa][vertical rows][MLP];
 * the circular layers then carry their own Householder rotation (hh_iter of jf_o_layer) and no azimuthal scaling is applied. */
enum { JF_F_KAPPA_DIRECT_LOG = 0, JF_F_KAPPA_SOFTPLUS = 1, JF_F_KAPPA_LOG_BOUNDED = 2, JF_F_KAPPA_MU = 3, JF_F_KAPPA_MU_SQUARED = 4,
       JF_F_KAPPA_QUATVEC = 5, JF_F_KAPPA_QUATVEC_SQUARED = 6 };
typedef struct jf_f_layer {
    int32_t hh_iter, first, n_vertical, n_circular;
    int32_t correlated, corr_hidden, corr_rank, corr_full2;
    int32_t kappa_mode, kappa_clamping; /* JF_F_KAPPA_* (fvm_2d.py:105-139); modes >= JF_F_KAPPA_MU read kappa off the rotation parameters and
                                           the row has no kappa entry */
    int32_t extra_rotation, reserved;   /* add_extra_rotation_inbetween (fvm_2d.py:381-402, 664-688): a fixed quarter turn that moves the pole
                                           onto the equator between the kappa step and the nested spline flows */
    double z_sign, min_kappa, identity_region;
    jf_r_layer vertical[JF_MAX_NESTED];
    jf_o_layer circular[JF_MAX_NESTED];
} jf_f_layer;
/* 'v' exponential map on S2, float64 only (exponential_map_s2.py:248-528, bisection_n_newton.py:330-465);
 * row: [householder hh_iter*3][(mu_x,mu_y,mu_z,log-w[,log-beta]) as (n_pot, nc)] */
enum { JF_V_LINEAR = 0, JF_V_QUADRATIC = 1, JF_V_EXPONENTIAL = 2, JF_V_SPLINES = 3 /* rows: mu 3, log-w, 10 widths, 10 heights, 11 derivatives */ };
typedef struct jf_v_layer { int32_t num_components, exp_map_type, natural_direction, hh_iter, max_newton_iter, first; } jf_v_layer;

/* base-class steps only: optional Householder rotation in embedding space + optional first-layer chart; kind 0 interval, 1 S1, 2 S2
 * (identity layers 'y' / 'z'; sphere_base.py:601-695 and interval_base.py:61-79 around third-party subclasses); row: [householder] */
typedef struct jf_c_layer { int32_t kind, hh_iter, first, reserved; double lo, hi; } jf_c_layer;

#define JF_DECLARE_MCHAIN(fam, T, suffix)                                                                                            \
    int jf_##fam##_chain_inv_##suffix(const T* x, int64_t x_stride, const T* log_det_in, const T* params, int64_t param_stride,        \
                                      int32_t param_batch, int64_t B, int32_t n_layers, const jf_##fam##_layer* layers, T* x_out,      \
                                      int64_t x_out_stride, T* log_det_out, const T* base_logp_in, T* base_logp_out, int64_t* bins,    \
                                      int64_t bins_stride, int32_t* status, void* stream);                                             \
    int jf_##fam##_chain_fwd_##suffix(const T* x, int64_t x_stride, const T* log_det_in, const T* params, int64_t param_stride,        \
                                      int32_t param_batch, int64_t B, int32_t n_layers, const jf_##fam##_layer* layers, T* x_out,      \
                                      int64_t x_out_stride, T* log_det_out, const T* base_logp_in, T* base_logp_out, int64_t* bins,    \
                                      int64_t bins_stride, int32_t* status, void* stream);                                             \
    /* _inv as the LAST block of a pdf whose blocks were evaluated independently: ld_pre / blp_pre (<= 4 entries each, nullable) hold the   \
     * earlier blocks' per-row sums, log_det_out / base_logp_out receive the pdf's totals (list order, this block last) and total_out =    \
     * base_logp_out + log_det_out -- see jf_cond_gf_chain_split3_f32 / jf_combine_rows */                                                 \
    int jf_##fam##_chain_inv_sum_##suffix(const T* x, int64_t x_stride, const T* log_det_in, const T* params, int64_t param_stride,    \
                                          int32_t param_batch, int64_t B, int32_t n_layers, const jf_##fam##_layer* layers, T* x_out,  \
                                          int64_t x_out_stride, T* log_det_out, const T* base_logp_in, T* base_logp_out,               \
                                          const jf_row_list* ld_pre, const jf_row_list* blp_pre, T* total_out, int64_t* bins,          \
                                          int64_t bins_stride, int32_t* status, void* stream);
JF_DECLARE_MCHAIN(r, float, f32)
JF_DECLARE_MCHAIN(r, double, f64)
JF_DECLARE_MCHAIN(o, float, f32)
JF_DECLARE_MCHAIN(o, double, f64)
JF_DECLARE_MCHAIN(m, float, f32)
JF_DECLARE_MCHAIN(m, double, f64)
JF_DECLARE_MCHAIN(f, float, f32)
JF_DECLARE_MCHAIN(f, double, f64)
JF_DECLARE_MCHAIN(v, float, f32) /* returns JF_ERR_UNSUPPORTED: the reference asserts float64 (exponential_map_s2.py:450) */
JF_DECLARE_MCHAIN(v, double, f64)
JF_DECLARE_MCHAIN(c, float, f32) /* x has 2 columns for kind 2, else 1 */
JF_DECLARE_MCHAIN(c, double, f64)

/* A conditional manifold block in ONE launch: the default amortisation MLP params = tanh(in @ W1^T + b1) @ W2^T + b2
 * (main/default.py:656-670) followed by jf_<fam>_chain_inv (log-prob direction) or jf_<fam>_chain_fwd (sampling direction,
 * main/default.py:1482-1506: z = base points, no base log-prob) on those per-sample parameters.  The (B, N) block stays in LDS.
 * Limits: K1 <= 28, H <= 128, N = sum of the layers' row lengths <= 64, no correlated 'f'; otherwise JF_ERR_UNSUPPORTED. */
#define JF_DECLARE_COND_MCHAIN(fam, T, suffix)                                                                                            \
    int jf_cond_##fam##_chain_inv_##suffix(const T* in, int64_t in_stride, const T* W1, int64_t w1_stride, const T* b1, const T* W2,        \
                                           int64_t w2_stride, const T* b2, int32_t K1, int32_t H, const T* x, int64_t x_stride,            \
                                           const T* log_det_in, int64_t B, int32_t n_layers, const jf_##fam##_layer* layers, T* x_out,      \
                                           int64_t x_out_stride, T* log_det_out, const T* base_logp_in, T* base_logp_out, int32_t* status, \
                                           void* stream);                                                                                   \
    int jf_cond_##fam##_chain_fwd_##suffix(const T* in, int64_t in_stride, const T* W1, int64_t w1_stride, const T* b1, const T* W2,        \
                                           int64_t w2_stride, const T* b2, int32_t K1, int32_t H, const T* z, int64_t z_stride,            \
                                           const T* log_det_in, int64_t B, int32_t n_layers, const jf_##fam##_layer* layers, T* x_out,      \
                                           int64_t x_out_stride, T* log_det_out, int32_t* status, void* stream);
JF_DECLARE_COND_MCHAIN(r, float, f32)
JF_DECLARE_COND_MCHAIN(r, double, f64)
JF_DECLARE_COND_MCHAIN(o, float, f32)
JF_DECLARE_COND_MCHAIN(o, double, f64)
JF_DECLARE_COND_MCHAIN(m, float, f32)
JF_DECLARE_COND_MCHAIN(m, double, f64)
JF_DECLARE_COND_MCHAIN(f, float, f32)
JF_DECLARE_COND_MCHAIN(f, double, f64)

/* Backward of jf_<fam>_chain_inv_*: vector-Jacobian product for upstream gradients of (x_out, log_det_out, base_logp_out) (each nullable = 0),
 * i.e. torch.autograd over the layer loop of all_layer_inverse (main/default.py:998-1031) for a manifold block.  Evaluated in forward mode
 * (one pass of the chain per group of input directions on dual numbers through the same device code as the forward kernels) except for
 * 'v', whose exponential map and closed-form potentials have hand-written reverse-mode adjoints.  g_x (B, dim);
 * g_params: (B, P) for per-sample parameters; for param_batch == 1 the row sums are ADDED into g_params (1, P) (zero it first).
 * g_log_det_in = g_log_det, g_base_logp_in = g_base_logp (pass through). */
#define JF_DECLARE_MCHAIN_BWD(fam, T, suffix)                                                                                          \
    int jf_##fam##_chain_inv_bwd_##suffix(const T* x, int64_t x_stride, const T* params, int64_t param_stride, int32_t param_batch,       \
                                          int64_t B, int32_t n_layers, const jf_##fam##_layer* layers, const T* g_x_out,                  \
                                          int64_t g_x_out_stride, const T* g_log_det, const T* g_base_logp, T* g_x, int64_t g_x_stride,  \
                                          T* g_params, int64_t g_params_stride, int32_t* status, void* stream);
JF_DECLARE_MCHAIN_BWD(r, float, f32)
JF_DECLARE_MCHAIN_BWD(r, double, f64)
JF_DECLARE_MCHAIN_BWD(o, float, f32)
JF_DECLARE_MCHAIN_BWD(o, double, f64)
JF_DECLARE_MCHAIN_BWD(m, float, f32)
JF_DECLARE_MCHAIN_BWD(m, double, f64)
JF_DECLARE_MCHAIN_BWD(f, float, f32)
JF_DECLARE_MCHAIN_BWD(f, double, f64)
JF_DECLARE_MCHAIN_BWD(v, float, f32) /* JF_ERR_UNSUPPORTED ('v' is float64 only) */
JF_DECLARE_MCHAIN_BWD(v, double, f64)
JF_DECLARE_MCHAIN_BWD(c, float, f32)
JF_DECLARE_MCHAIN_BWD(c, double, f64)

/* intrinsic <-> embedding coordinates of S1 (angle <-> (cos, sin)) and S2 ((theta, phi) <-> (x, y, z)) with the log-det of
 * sphere_base.spherical_to_eucl_embedding / eucl_to_spherical_embedding (sphere_base.py:242-335); dim = 1 or 2 */
int jf_sphere_to_embedding_f32(const float* x, int64_t x_stride, const float* log_det_in, int64_t B, int32_t dim, float* x_out,
                               int64_t x_out_stride, float* log_det_out, void* stream);
int jf_sphere_to_embedding_f64(const double* x, int64_t x_stride, const double* log_det_in, int64_t B, int32_t dim, double* x_out,
                               int64_t x_out_stride, double* log_det_out, void* stream);
int jf_sphere_from_embedding_f32(const float* x, int64_t x_stride, const float* log_det_in, int64_t B, int32_t dim, float* x_out,
                                 int64_t x_out_stride, float* log_det_out, void* stream);
int jf_sphere_from_embedding_f64(const double* x, int64_t x_stride, const double* log_det_in, int64_t B, int32_t dim, double* x_out,
                                 int64_t x_out_stride, double* log_det_out, void* stream);

/* input rows of the amortisation MLPs in one launch: out row = the segments' contributions side by side, i.e.
 * cat[conditional_input, embed(x_0), embed(x_1), ...] of main/default.py:946-962 with embed = identity for Euclidean / interval targets
 * (kind 0, n_in columns copied), (cos, sin) for an S1 angle (kind 1) and (x, y, z) for S2 (theta, phi) (kind 2)
 * (sphere_base.py:305-332, 786-794).  Block i's MLP reads a prefix of the row (autoregressive conditioning). */
/* (jf_cond_segment / JF_MAX_SEGMENTS: declared at the top of this header) */
int jf_conditioning_rows_f32(const jf_cond_segment* segments, int32_t n_segments, int64_t B, float* out, int64_t out_stride, void* stream);
int jf_conditioning_rows_f64(const jf_cond_segment* segments, int32_t n_segments, int64_t B, double* out, int64_t out_stride, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * base-distribution log-prob: out[b] = (in ? in[b] : 0) + sum_d N(0,1).log_prob(z[b,d])
 * (torch.distributions.Normal(0,1).log_prob(base_pos).sum(-1), jammy_flows/main/default.py:1110-1115, 1657, 1670)
 * ------------------------------------------------------------------------------------------------------------ */
int jf_normal_logp_f32(const float* z, int64_t z_stride, int64_t B, int32_t D, const float* in, float* out, void* stream);
int jf_normal_logp_f64(const double* z, int64_t z_stride, int64_t B, int32_t D, const double* in, double* out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Dense layer on split-bf16 matrix arithmetic (float32 in / out, float32-equivalent accuracy: every operand as three bf16 pieces, six
 * v_mfma_f32_16x16x32_bf16 per product, f32 accumulation):  out (B, N) = X (B, K) W^T + bias.  Serves the large products of the conditional
 * block's backward -- the nn.Linear of mlp_predictors (main/default.py:656-670) re-evaluated, and grad_output @ weight of its autograd --
 * at ~2.5x the rate of the exact-f32 MFMA kernel (jf_linear).
 *   jf_linear_split_packed_bytes(N, K): size of the packed weight image;  jf_linear_split_pack_f32: W (N, K), element (n, k) at
 *   W[n * w_row_stride + k * w_col_stride] (so a transposed view packs without a copy) -> packed (16-byte aligned), once per weight version;
 *   jf_linear_split_f32: the product; needs K % 4 == 0, N % 4 == 0, 16-byte aligned rows (else JF_ERR_UNSUPPORTED: use jf_linear).
 * ------------------------------------------------------------------------------------------------------------ */
int64_t jf_linear_split_packed_bytes(int32_t N, int32_t K);
int jf_linear_split_pack_f32(const float* W, int64_t w_row_stride, int64_t w_col_stride, int32_t N, int32_t K, void* packed, void* stream);
int jf_linear_split_f32(const float* X, int64_t x_stride, const void* packed, const float* bias, int64_t B, int32_t K, int32_t N, float* out,
                        int64_t out_stride, void* stream);

/* jf_linear_wgrad on the same split-bf16 arithmetic (float32, K <= 128, K % 4 == 0, N % 4 == 0, 16-byte aligned rows; else
 * JF_ERR_UNSUPPORTED: use jf_linear_wgrad): partial_w (S, N, K) and partial_b (S, N) or NULL with S = jf_linear_wgrad_split_splits(B, N),
 * added up by the caller. */
int64_t jf_linear_wgrad_split_splits(int64_t B, int32_t N);
int jf_linear_wgrad_split_f32(const float* g, int64_t g_stride, const float* in, int64_t in_stride, int64_t B, int32_t K, int32_t N, float* partial_w,
                              float* partial_b, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * backward of a tanh activation: out[i] = g[i] * (1 - y[i]^2) for n contiguous elements (y = the saved activation).  What torch.autograd
 * runs as three elementwise launches (y*y, 1 - ., g * .) behind nn.Tanh in the amortisation MLPs (main/default.py:656-670,
 * amortizable_mlp.py:508-578); out may alias g.
 * ------------------------------------------------------------------------------------------------------------ */
int jf_tanh_bwd_f32(const float* g, const float* y, int64_t n, float* out, void* stream);
int jf_tanh_bwd_f64(const double* g, const double* y, int64_t n, double* out, void* stream);
/* Sum of the S partial slabs a batch-reducing backward launch left (jf_linear_wgrad*, jf_mlp2_small_bwd, jf_mlp_hidden_bwd, the broadcast
 * regime of jf_gf_chain_inv_bwd), in chunks of `chunk` slabs: out_a[c * na + i] = sum over s in [c * chunk, min(S, (c + 1) * chunk)) of
 * a[s * na + i] and, when nb > 0, the same for b in the same launch (weights and bias of one layer); out_a / out_b hold ceil(S / chunk) slabs.
 * chunk >= S: the total.  Thousands of slabs: one launch with chunk ~ sqrt(S), a second with chunk = all.  Fixed summation order: results are
 * bit-identical from run to run (what torch.sum(0) on the slabs did in five launches per training step). */
int jf_slab_sum_f32(const float* a, int64_t na, float* out_a, const float* b, int64_t nb, float* out_b, int32_t S, int32_t chunk, void* stream);
int jf_slab_sum_f64(const double* a, int64_t na, double* out_a, const double* b, int64_t nb, double* out_b, int32_t S, int32_t chunk, void* stream);
/* The totals of ALL S slabs, laid out for their consumer (ABI v7): out[map[i]] = sum_s a[s * na + i] for i < na, out[map[na + i]] = sum_s
 * b[s * nb + i]; map: na + nb device int32 entries, negative = dropped (padding rows of a packed gradient).  One launch where the plain sum was
 * followed by copy / gather launches (a transposed weight gradient, bias columns split off a slab, packed parameter rows back in natural
 * order: main/default.py's torch.autograd does the same with views + .contiguous()).  Same fixed summation order as jf_slab_sum. */
int jf_slab_sum_map_f32(const float* a, int64_t na, const float* b, int64_t nb, const int32_t* map, float* out, int32_t S, void* stream);
int jf_slab_sum_map_f64(const double* a, int64_t na, const double* b, int64_t nb, const int32_t* map, double* out, int32_t S, void* stream);
/* AmortizableMLP nonlinearities other than tanh (extra_functions.py:81-89; tanh is fused into the dense kernels): out = act(z) on the layer's
 * pre-activation, and the backward out = g * act'(z).  n elements, contiguous. */
#define JF_ACT_RELU 2
#define JF_ACT_SOFTPLUS 3
#define JF_ACT_ELU 4
#define JF_ACT_SWISH 5
#define JF_ACT_SQUARE 6
#define JF_ACT_IDENTITY 7
int jf_activation_f32(const float* z, int64_t n, int32_t code, float* out, void* stream);
int jf_activation_f64(const double* z, int64_t n, int32_t code, double* out, void* stream);
int jf_activation_bwd_f32(const float* g, const float* z, int64_t n, int32_t code, float* out, void* stream);
int jf_activation_bwd_f64(const double* g, const double* z, int64_t n, int32_t code, double* out, void* stream);
/* The device math functions the flow kernels are built on (csrc/jf_math.h), elementwise -- so that their accuracy can be measured from the
 * host: exp_fast (float32: v_exp_f32), log_fast, tanh_fast (hidden layers: absolute accuracy), rcp. */
#define JF_MATH_EXP_FAST 0
#define JF_MATH_LOG_FAST 1
#define JF_MATH_TANH_FAST 2
#define JF_MATH_RCP 3
int jf_device_math_f32(const float* x, int64_t n, int32_t fn, float* out, void* stream);
int jf_device_math_f64(const double* x, int64_t n, int32_t fn, double* out, void* stream);
/* The last operations of pdf.forward / all_layer_inverse (main/default.py:1110-1117): the sub-manifold blocks of the log-prob direction are
 * independent given the targets, so every block returns its OWN log-det and base log-prob (ld_in = blp_in = NULL) and one launch adds them up,
 * in list order: ld_out[b] = sum_i ld.p[i][b], blp_out[b] = sum_i blp.p[i][b], total_out[b] = blp_out[b] + ld_out[b] (each output nullable). */
/* Broadcast-parameter g chains of 2 .. 4 dimensions, log-prob direction: batches of fewer than `rows` rows run with one lane per (row,
 * coordinate) instead of one lane per row -- more waves for small batches, bit-identical results (csrc/jf_gfb.h).  Sets the threshold (0: always
 * one lane per row; negative: back to the default 2^16 / JF_GFB_LANE_ROWS) and returns the previous one. */
int64_t jf_gf_bcast_lane_rows(int64_t rows);
/* Two side blocks of a log-prob step in ONE launch (csrc/merged_kernels.hip).  The blocks of the log-prob direction are independent given the
 * targets (main/default.py:946-962); between jf_merge_begin and jf_merge_end the launches this thread's entry points would issue are captured
 * instead, and jf_merge_end issues ONE grid that runs the captured blocks side by side (the blocks' own device code: bit-identical per-row
 * results).  Accepted: exactly one broadcast g chain (jf_gf_chain_inv_f32, param_batch 1, D <= 4, classic layers) and one conditional `f` block
 * (jf_cond_f_chain_inv_f32, one layer) over the same B rows.  Otherwise the captured launches are issued one by one in their order and
 * JF_MERGE_DECLINED is returned; the results are in place either way.  Inside a step-plan recording the (merged or replayed) launches go to
 * the plan.  jf_merge_abort drops the captured launches (error paths); jf_merge_captured = launches captured so far. */
#define JF_MERGE_DECLINED 1
int jf_merge_begin(void);
int jf_merge_abort(void);
int jf_merge_captured(void);
int jf_merge_end(void* stream);
/* (jf_row_list / JF_MAX_ROW_LISTS: defined above, before jf_cond_gf_chain_split3_f32) */
int jf_combine_rows_f32(const jf_row_list* ld, const jf_row_list* blp, int64_t B, float* ld_out, float* blp_out, float* total_out, void* stream);
int jf_combine_rows_f64(const jf_row_list* ld, const jf_row_list* blp, int64_t B, double* ld_out, double* blp_out, double* total_out, void* stream);
/* torch.optim.Adam (amsgrad off, no weight decay: what examples/jammy_flows.py:381-412 trains with) over up to JF_ADAM_MAX_TENSORS parameter
 * tensors in ONE launch: m <- m + (g - m)(1 - b1); v <- v b2 + (1 - b2) g^2; p <- p - lr / (1 - b1^step) m / (sqrt(v) / sqrt(1 - b2^step) + eps).
 * `step` counts from 1.  All four arrays of a tensor are contiguous, n elements, of the function's precision. */
#define JF_ADAM_MAX_TENSORS 64
typedef struct jf_adam_tensor { void* param; const void* grad; void* exp_avg; void* exp_avg_sq; int64_t n; } jf_adam_tensor;
int jf_adam_step_f32(const jf_adam_tensor* tensors, int32_t n_tensors, double lr, double beta1, double beta2, double eps, int64_t step, void* stream);
int jf_adam_step_f64(const jf_adam_tensor* tensors, int32_t n_tensors, double lr, double beta1, double beta2, double eps, int64_t step, void* stream);
/* the same update with the step count read from DEVICE memory at execution time (*step_dev >= 1, already incremented for this step): a launch
 * recorded in a HIP graph then replays with the count of the replay (torch.optim.Adam(capturable=True) keeps its step on the device for the
 * same reason).  The caller increments *step_dev on the same stream before the launch. */
int jf_adam_step_dev_f32(const jf_adam_tensor* tensors, int32_t n_tensors, double lr, double beta1, double beta2, double eps, const int64_t* step_dev, void* stream);
int jf_adam_step_dev_f64(const jf_adam_tensor* tensors, int32_t n_tensors, double lr, double beta1, double beta2, double eps, const int64_t* step_dev, void* stream);
/* out = a + b (n elements) */
int jf_add_rows_f32(const float* a, const float* b, int64_t n, float* out, void* stream);
int jf_add_rows_f64(const double* a, const double* b, int64_t n, double* out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Step plans (ABI v5): a whole evaluation step -- every launch pdf.forward / pdf.sample makes for one input shape, i.e. the loop over the
 * sub-manifolds and their layers that the reference runs in Python on every call (jammy_flows/main/default.py:879-1057 all_layer_inverse,
 * :1059-1117 forward, :1373-1531 all_layer_forward) -- recorded once and re-issued from C by ONE call.
 *
 *   p = jf_plan_create();
 *   jf_plan_add_slot(p, x, bytes_of_x); ... one slot per buffer whose ADDRESS changes between replays (inputs, outputs)
 *   jf_plan_record_begin(p);
 *       jf_plan_add_memset(p, status, 0, 16);                 optional explicit ops
 *       jf_gf_chain_inv_f32(...); jf_cond_f_chain_inv_f32(...); ...   ANY entry points of this header, on this thread: recorded, not launched
 *       jf_plan_add_copy_to_host(p, pinned_host_status, status, 16);
 *   n_ops = jf_plan_record_end(p);
 *   jf_plan_launch(p, slot_bases, n_slots, stream);            as often as wanted: re-issues the ops on `stream`, every device pointer that
 *                                                              pointed into slot i at record time now points to slot_bases[i] + same offset
 * Everything an entry point decides on the host (kernel variant, grid, occupancy query, attribute setting) is decided at record time; a plan
 * is therefore valid for the shapes, strides, options and non-slot addresses (weights, packed images, intermediate buffers) it was recorded
 * with.  jf_plan_launch never allocates or synchronises.  One host thread replays a plan at a time.
 * Timing: with jf_plan_set_timing(p, n), n >= 1, every n-th replay records HIP events around every op on the launch stream (0 = off);
 * jf_plan_read_timing waits for them and returns the summed milliseconds per op and the number of replays they cover.
 * ------------------------------------------------------------------------------------------------------------ */
/* plans are named by int64 HANDLES (never by addresses): a stale or random value is answered with JF_ERR_BADARG */
int64_t jf_plan_create(void);                                               /* -> handle (> 0) or a negative error */
int32_t jf_plan_destroy(int64_t plan);
int32_t jf_plan_add_slot(int64_t plan, const void* base, int64_t bytes);    /* -> slot index, or a negative error (overlapping slots are refused) */
int32_t jf_plan_record_begin(int64_t plan);
int32_t jf_plan_record_end(int64_t plan);                                   /* -> number of recorded ops */
int32_t jf_plan_add_memset(int64_t plan, void* dst, int32_t value, int64_t bytes);
int32_t jf_plan_add_copy_to_host(int64_t plan, void* host_dst, const void* src, int64_t bytes);
/* lanes: ops recorded after jf_plan_set_lane(plan, l), 0 < l < 4, are issued on side stream l of the plan (0 = the caller's stream).  Between
 * jf_plan_add_fork and the next jf_plan_add_join the ops of different lanes must be independent of each other: the fork makes the side streams
 * wait for everything given to the caller's stream so far, the join makes the caller's stream wait for them.  (Small batches: the blocks of a
 * log-prob step are independent given the targets, and their launch / drain tails overlap.) */
/* any_order: launches recorded while it is on are issued without the queue's barrier bit (hipExtAnyOrderLaunch) and may run beside the launches
 * recorded before them; the first launch recorded after it is switched off waits for all of them.  Same independence contract as for lanes. */
int32_t jf_plan_set_any_order(int64_t plan, int32_t on);
int32_t jf_plan_set_lane(int64_t plan, int32_t lane);
int32_t jf_plan_add_fork(int64_t plan);
int32_t jf_plan_add_join(int64_t plan);
int32_t jf_plan_num_ops(int64_t plan);
int32_t jf_plan_num_relocations(int64_t plan);
int32_t jf_plan_launch(int64_t plan, const void* const* slot_bases, int32_t n_slots, void* stream);
int32_t jf_plan_set_timing(int64_t plan, int32_t on);
int32_t jf_plan_debug_words(int64_t plan, int32_t op, uint64_t* out, int32_t cap);   /* argument storage of one op (debugging aid) */
int32_t jf_plan_read_timing(int64_t plan, double* ms_sum_per_op, int32_t n_ops, int64_t* replays, int32_t reset);

/* ------------------------------------------------------------------------------------------------------------
 * Reductions of the analysis utilities, so that 1e5 .. 1e6 evaluated rows never travel to the host:
 * jf_coverage_histogram: the counting loop of calculate_approximate_coverage (jammy_flows/helper_fns/coverage.py:45-65) behind
 *   pdf.approximate_coverage (main/default.py:1954-2022).  twice[b] = 2 (log_at_zero - log_prob_base[b]); thresholds = chi2.ppf of the expected
 *   coverage probabilities (ascending, n <= 1024); hist[i] (i <= n, int64, ADDED to) = rows whose first threshold with twice < thr is i
 *   (i = n: none), so #(twice < thr[i]) = hist[0] + ... + hist[i].  twice_out nullable.
 * jf_segment_reduce: the sample means of pdf.entropy (main/default.py:2263-2454): out[g] = -mean_s in[g, s] (mode 0) or
 *   logsumexp_s in[g, s] - log S (mode 1), in row-major (n_seg, seg_len).
 * ------------------------------------------------------------------------------------------------------------ */
int jf_coverage_histogram_f32(const float* log_prob_base, int64_t B, double log_at_zero, const float* thresholds, int32_t n, int64_t* hist,
                              float* twice_out, void* stream);
int jf_coverage_histogram_f64(const double* log_prob_base, int64_t B, double log_at_zero, const double* thresholds, int32_t n, int64_t* hist,
                              double* twice_out, void* stream);
int jf_segment_reduce_f32(const float* in, int64_t n_seg, int64_t seg_len, int32_t mode, float* out, void* stream);
int jf_segment_reduce_f64(const double* in, int64_t n_seg, int64_t seg_len, int32_t mode, double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* JAMMY_HIP_H */
