/* xgpr_hip.h -- C ABI of libxgpr_hip.so, the MI355X (gfx950) implementation of the
 * xGPR random-feature / CG hot path.
 *
 * This is the drop-in boundary: every entry point below replaces one operator of the
 * reference's GPU extension module
 *     src/xGPR/random_feature_generation/gpu_rf_gen/xgpr_cuda_rfgen_cpp_ext.cpp:20-93
 * (or the CPU twin cpu_rf_gen/xgpr_cpu_rfgen_cpp_ext.cpp:23-147 where the CUDA module
 * has no counterpart), or one per-chunk step of the reference's CG / preconditioner
 * loops that this library fuses with feature generation.  Citations are relative to
 * /root/reference/.
 *
 * Conventions
 *   - plain pointers and sizes, no framework types; `stream` is a hipStream_t passed as
 *     void* (NULL = the null stream).  Every call is asynchronous and stream-ordered.
 *   - all array arguments are DEVICE pointers to C-contiguous data unless the name ends
 *     in `_host`.  Nothing is retained after return; nothing is allocated (scratch comes
 *     in through `workspace`, sized by the matching *_workspace_bytes() function).
 *   - return value: 0 on success, a negative code otherwise; xgpr_last_error() returns
 *     the message the reference would have thrown as std::runtime_error (thread-local).
 *   - "_f32"/"_f64" is the element type T of (input, chi) -- the reference overloads each
 *     operator for float and double (e.g. xgpr_cuda_rfgen_cpp_ext.cpp:32-40).
 */
#ifndef XGPR_HIP_H
#define XGPR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes (messages mirror the reference's throw sites) */
#define XGPR_OK                 0
#define XGPR_ERR_NO_DATAPOINTS (-1)  /* rbf_ops.cpp:49-50   "no datapoints" */
#define XGPR_ERR_ODD_OUTPUT    (-2)  /* rbf_ops.cpp:51-52   "last dim of output must be even number" */
#define XGPR_ERR_RFFS_FREQS    (-3)  /* rbf_ops.cpp:53-54,61-62 "incorrect number of rffs and or freqs." */
#define XGPR_ERR_ARRAY_SIZES   (-4)  /* rbf_ops.cpp:168-170 "Wrong array sizes." */
#define XGPR_ERR_SEQLEN_SIZE   (-5)  /* rbf_convolution.cpp:55-56 "wrong array sizes" */
#define XGPR_ERR_CONV_WIDTH    (-6)  /* rbf_convolution.cpp:57-58 "invalid conv_width" */
#define XGPR_ERR_SEQLEN_RANGE  (-7)  /* rbf_convolution.cpp:78-82 "All sequence lengths must be >= conv width and < array size." */
#define XGPR_ERR_ARRAY_DIMS    (-8)  /* transform_functions.cpp:109-110 "incorrect array dims passed" */
#define XGPR_ERR_NOT_POW2      (-9)  /* transform_functions.cpp:111-114 "last dim not power of 2 > 1" */
#define XGPR_ERR_UNSUPPORTED   (-20) /* shape outside what this build supports (message says which) */
#define XGPR_ERR_WORKSPACE     (-21) /* workspace missing / too small / misaligned pointer */
#define XGPR_ERR_HIP           (-100)/* a HIP runtime call or kernel launch failed */

const char *xgpr_last_error(void);
/* "gfx950" etc: the offload architecture the device code was built for. */
const char *xgpr_build_arch(void);
/* sha256 (hex) of the sources (xgpr_amd/csrc/ *, this header) and compiler flags the binary was built from, as
 * xgpr_amd/build.py source_id() computes it; "unidentified" for a hand-compiled library */
const char *xgpr_build_id(void);

/* ---- bare FHT: cudaFastHadamardTransform2D (xgpr_cuda_rfgen_cpp_ext.cpp:21-24) and the
 * CPU-only 3-D form cpuFastHadamardTransform (xgpr_cpu_rfgen_cpp_ext.cpp:24-30).
 * In place, un-normalised, over the last axis of x[n, dim1, dim2] (dim1 = 1 for 2-D). */
int xgpr_fht_f32(float *x, long n, long dim1, long dim2, void *stream);
int xgpr_fht_f64(double *x, long n, long dim1, long dim2, void *stream);

/* ---- cudaSRHT (xgpr_cuda_rfgen_cpp_ext.cpp:25-30): x[n, dim] <- FHT(x * radem * 2^(-log2(dim)/2)) */
int xgpr_srht_f32(float *x, const int8_t *radem, long n, long dim, long radem_len, void *stream);
int xgpr_srht_f64(double *x, const int8_t *radem, long n, long dim, long radem_len, void *stream);

/* ---- cudaRBFFeatureGen (xgpr_cuda_rfgen_cpp_ext.cpp:32-40)
 * x[n, d] (T), out[out_rows, num_rffs] (f64), radem[3, 1, radem_shape2] (int8),
 * chi[num_freqs] (T).  Like the reference's CUDA kernel (rbf_ops.cu:121-127) the output
 * is OVERWRITTEN: out[i, 2f] = s*cos(chi[f]*sorf(x_i)[f]), out[i, 2f+1] = s*sin(...).
 * float32 input, padded width P <= 4096: wave-tile kernels (a transform of P = 2048 / 4096 elements spans two / four
 * wave tiles with one cross-wave exchange per round); float64 input, P <= 8192: float64 wave tiles (16 doubles per
 * lane), which also serve float32 input at P = 8192; beyond that: the any-width path (one workgroup per transform, butterflies in
 * LDS).  The gradient operator (xgpr_rbf_grad_*) runs on wave tiles for the same widths.
 * Numerics: the float32 argument of every cos / sin is bit-identical to the reference's (same butterfly order, no
 * contraction).  For even log2(P) the three normalisers 2^(-log2(P)/2) are exact powers of two and are applied ONCE,
 * folded into chi: intermediate values of the transform are P^(1/2) .. P^(3/2) times the reference's (round by round), so
 * inputs whose transform the reference still holds finite overflow here once |x| exceeds roughly FLT_MAX / P^(3/2) (1e34 at
 * P = 1024; 3e29 = FLT_MAX / P^3 for the worst-aligned input) -- untested territory, far outside sigma-scaled data. */
size_t xgpr_rbf_workspace_bytes(long radem_shape2);
/* Workspace for any SORF operator below (feature-gen, grad, conv, max-pool): covers the
 * packed sign masks of the float wave-tile path (padded width <= 1024; for the float feature
 * operator and the feature cache: <= 4096) and, for padded widths beyond the LDS capacity
 * (> 32768 float / > 16384 double), the global scratch of the any-width path.  `width` is the un-padded transform width (d, or conv_width * C);
 * elem_size is sizeof(T). */
size_t xgpr_sorf_workspace_bytes(long radem_shape2, long width, int elem_size);
/* Workspace of the convolution operators (xgpr_conv1d_fgen_*, xgpr_conv_grad_*, xgpr_conv1d_maxpool_*) with room
 * for the processing order of the nseq sequences: given at least this much, a launch runs its longest sequences
 * first (a wave runs for as long as its sequence has k-mers; in the caller's order the long sequences at the end of
 * a chunk leave most of the GPU idle).  With only xgpr_sorf_workspace_bytes the operators run in the caller's order;
 * results are the same either way.  No reference counterpart: the reference allocates its scratch per call
 * (gpu_rf_gen/convolution_ops/rbf_convolution.cu:368-379). */
size_t xgpr_conv_workspace_bytes(long radem_shape2, long width, int elem_size, long nseq);
int xgpr_rbf_feature_gen_f32(const float *x, double *out, const int8_t *radem, const float *chi,
                             long n, long d, long out_rows, long num_rffs, long num_freqs,
                             long radem_shape2, int fit_intercept,
                             void *workspace, size_t workspace_bytes, void *stream);
int xgpr_rbf_feature_gen_f64(const double *x, double *out, const int8_t *radem, const double *chi,
                             long n, long d, long out_rows, long num_rffs, long num_freqs,
                             long radem_shape2, int fit_intercept,
                             void *workspace, size_t workspace_bytes, void *stream);

/* ---- cudaRBFGrad (xgpr_cuda_rfgen_cpp_ext.cpp:41-49): features + d/dsigma.
 * grad[grad_rows, grad_cols, 1] (f64).  x is NOT pre-multiplied by sigma.  sigma is a
 * double as in the CPU module (cpu_rf_gen/rbf_ops/rbf_ops.h:57). */
int xgpr_rbf_grad_f32(const float *x, double *out, double *grad, const int8_t *radem,
                      const float *chi, long n, long d, long out_rows, long num_rffs,
                      long grad_rows, long grad_cols, long num_freqs, long radem_shape2,
                      double sigma, int fit_intercept,
                      void *workspace, size_t workspace_bytes, void *stream);
int xgpr_rbf_grad_f64(const double *x, double *out, double *grad, const int8_t *radem,
                      const double *chi, long n, long d, long out_rows, long num_rffs,
                      long grad_rows, long grad_cols, long num_freqs, long radem_shape2,
                      double sigma, int fit_intercept,
                      void *workspace, size_t workspace_bytes, void *stream);

/* ---- cudaConv1dFGen (xgpr_cuda_rfgen_cpp_ext.cpp:70-80): x[n, L, C]; k-mer windows of
 * conv_width*C contiguous elements; results are ADDED into out (which callers zero), as in
 * the reference (rbf_convolution.cu:140-146).  scaling_type 0 none / 1 sqrt / 2 full.
 * seqlen_host[nseq]: int32 on the HOST, as in the reference's CUDA module
 * (gpu_rf_gen/convolution_ops/rbf_convolution.h:19) -- validated there; seqlen_dev is the
 * same array on the device (the reference H2D-copies it on every call,
 * rbf_convolution.cu:368-379; here the caller owns that copy).
 * Windows (conv_width * C, padded to P) up to 4096 elements run on wave-tile kernels for both element types (float32 up to
 * 1024: wave_conv_kernel; float32 2048 / 4096 and float64: wave_tile_conv_kernel); wider windows on the any-width path. */
int xgpr_conv1d_fgen_f32(const float *x, double *out, const int8_t *radem, const float *chi,
                         const int32_t *seqlen_host, const int32_t *seqlen_dev,
                         long n, long L, long C, long out_rows, long num_rffs, long num_freqs,
                         long radem_shape2, long nseq, int conv_width, int scaling_type,
                         void *workspace, size_t workspace_bytes, void *stream);
int xgpr_conv1d_fgen_f64(const double *x, double *out, const int8_t *radem, const double *chi,
                         const int32_t *seqlen_host, const int32_t *seqlen_dev,
                         long n, long L, long C, long out_rows, long num_rffs, long num_freqs,
                         long radem_shape2, long nseq, int conv_width, int scaling_type,
                         void *workspace, size_t workspace_bytes, void *stream);

/* ---- cudaConvGrad (xgpr_cuda_rfgen_cpp_ext.cpp:81-92) */
int xgpr_conv_grad_f32(const float *x, double *out, double *grad, const int8_t *radem,
                       const float *chi, const int32_t *seqlen_host, const int32_t *seqlen_dev,
                       long n, long L, long C, long out_rows, long num_rffs, long grad_rows,
                       long grad_cols, long num_freqs, long radem_shape2, long nseq,
                       double sigma, int conv_width, int scaling_type,
                       void *workspace, size_t workspace_bytes, void *stream);
int xgpr_conv_grad_f64(const double *x, double *out, double *grad, const int8_t *radem,
                       const double *chi, const int32_t *seqlen_host, const int32_t *seqlen_dev,
                       long n, long L, long C, long out_rows, long num_rffs, long grad_rows,
                       long grad_cols, long num_freqs, long radem_shape2, long nseq,
                       double sigma, int conv_width, int scaling_type,
                       void *workspace, size_t workspace_bytes, void *stream);

/* ---- cudaConv1dMaxpool (xgpr_cuda_rfgen_cpp_ext.cpp:61-69): out[n, num_rffs] float32,
 * out = max(out, chi * sorf(window)) over k-mers; num_freqs == num_rffs;
 * radem_shape2 == reps * P exactly (conv1d_operations.cpp:65-68). */
int xgpr_conv1d_maxpool_f32(const float *x, float *out, const int8_t *radem, const float *chi,
                            const int32_t *seqlen_host, const int32_t *seqlen_dev,
                            long n, long L, long C, long out_rows, long num_rffs, long num_freqs,
                            long radem_shape2, long nseq, int conv_width,
                            void *workspace, size_t workspace_bytes, void *stream);
int xgpr_conv1d_maxpool_f64(const double *x, float *out, const int8_t *radem, const double *chi,
                            const int32_t *seqlen_host, const int32_t *seqlen_dev,
                            long n, long L, long C, long out_rows, long num_rffs, long num_freqs,
                            long radem_shape2, long nseq, int conv_width,
                            void *workspace, size_t workspace_bytes, void *stream);

/* ---- fused CG matvec: the per-chunk body of CPU/GPU_ConjugateGrad._matvec
 * (src/xGPR/fitting_toolkit/cg_tools.py:173-200 / :26-53) for one shard of rows,
 *     w_out[m] = sum_i Z[i, m] * (Z[i, :] . v),   Z = transform_x(x)  (never written to HBM),
 * i.e. `for chunk: Z = kernel.transform_x(x); matvec += Z.T @ (Z @ vec)` with feature
 * generation (kernel_baseclass.py:269-299, incl. the intercept column Z[:,0] = 1) fused in.
 * x[n, d] float32 ALREADY multiplied by sigma (sorf_kernel_baseclass.py:117); v, w_out
 * [num_rffs] f64.  lambda^2 * v is NOT added (the caller adds it after the all-reduce).
 * Deterministic: per-workgroup partial sums are combined in a fixed order.
 * Supported: padded width P = 2^ceil(log2(max(d,2))) <= 4096, num_freqs <= 65536.  One pass over the
 * datapoints while a workgroup can hold every tile (1024 frequencies) of a datapoint: up to num_freqs = 7168
 * at P <= 1024, up to 4096 at P = 2048 / 4096 (xgpr_ztz_matvec_plan says which); beyond that -- and for EIGHT
 * tiles, 7168 < num_freqs <= 8192, which do not divide the kernel's twelve waves -- a dot pass and an update pass
 * per window of 131072 datapoints, i.e. the features are generated twice.
 * The call packs radem into sign masks at the head of the workspace first; radem == NULL says the
 * workspace still holds the masks of an earlier call with the same radem (a CG solve calls this once per
 * iteration with one workspace), and the packing launch is skipped. */
size_t xgpr_ztz_matvec_workspace_bytes(long num_rffs, long radem_shape2);
/* Which plan xgpr_ztz_matvec_f32 runs for rows of d floats (16-byte aligned) and num_freqs frequencies -- what a caller
 * that can also keep the features resident (xgpr_rbf_feature_cache_f32 + xgpr_zcache_matvec_f32) decides by:
 * 1 = one pass on the three-wave kernel (at padded width <= 1024 regenerating is as fast as streaming the cache; the wide
 * transforms of padded width 2048 / 4096 cost 1.2x / 1.5x that per tile); 2 = one pass, two-wave kernel (seven tiles per
 * datapoint, or one tile at padded width >= 128: slower than the cache stream); 3 = two feature passes (num_freqs > 8192;
 * eight tiles per datapoint: 7168 < num_freqs <= 8192; more than 4096 frequencies at padded width > 1024);
 * 0 = unsupported shape (padded width > 4096, num_freqs > 65536). */
int xgpr_ztz_matvec_plan(long d, long num_freqs);
int xgpr_ztz_matvec_f32(const float *x, const int8_t *radem, const float *chi, const double *v,
                        double *w_out, long n, long d, long num_rffs, long num_freqs,
                        long radem_shape2, int fit_intercept,
                        void *workspace, size_t workspace_bytes, void *stream);

/* ---- z^T y for one shard (scoring_toolkit/exact_nmll_calcs.py:13-39 calc_zty, and the
 * `zty += Z.T @ y` line of rand_nys_constructors.py:96-123), fused with feature
 * generation: zty_out[m] = sum_i Z[i, m] * y[i].  Same support envelope as the matvec. */
int xgpr_zty_f32(const float *x, const int8_t *radem, const float *chi, const double *y,
                 double *zty_out, long n, long d, long num_rffs, long num_freqs,
                 long radem_shape2, int fit_intercept,
                 void *workspace, size_t workspace_bytes, void *stream);

/* ---- fused CG vector updates for one right-hand side: the per-iteration arithmetic of
 * CPU/GPU_ConjugateGrad.fit (src/xGPR/fitting_toolkit/cg_tools.py:255-274 / :108-127) between
 * the matvec and the next search direction, as two single-workgroup kernels (the preconditioner apply
 * between them is xgpr_precond_apply_f64).  All vectors are
 * float64 [M] on the device; scal is float64 [4] = { r.z, alpha, err, beta }.
 *   step1: w += lam2 * p (w arrives holding the all-reduced Z^T Z p); alpha = (r.z)/(p.w);
 *          x += alpha p; r_next = r - alpha w; err = |r| / init_norm   (cg_tools.py:256-265)
 *   step2: beta = (r_next.z_next)/(r.z); p_next = z_next + beta p     (cg_tools.py:271-274)
 * stop_tol = 0: plain steps (the host decides when to stop, cg_tools.py:266-269).  stop_tol > 0: for
 * iterations queued ahead of the host's check (replayed from a HIP graph) the convergence test of
 * cg_tools.py:266-269 is applied on the device as well: once the error of the previous iteration is
 * below stop_tol, this and all later steps leave every vector untouched, so x holds exactly the iterate
 * the host-checked loop returns.  scal is then float64 [8 + max_iterations], zero-initialised except
 * scal[2] = +inf: scal[4] = stopped flag, scal[5] = iterations applied, scal[8 + i] = err of iteration i.
 * err_out (may be NULL): a second place for err -- e.g. pinned host memory, which the device writes directly
 * (followed by a system-scope fence), so that the host's lagging read of the error needs no copy command. */
int xgpr_cg_step1_f64(double *w, const double *p, double *x, const double *r, double *r_next,
                      const double *z, double *scal, double lam2, double init_norm, long M,
                      double stop_tol, double *err_out, void *stream);
int xgpr_cg_step2_f64(const double *r_next, const double *z_next, const double *p, double *p_next,
                      double *scal, long M, double stop_tol, void *stream);

/* The same two steps for a block of k right-hand sides (GPU_ConjugateGrad.fit with k > 1,
 * src/xGPR/fitting_toolkit/cg_tools.py:121-141 / :255-274; the approximate NMLL's 26-column solve,
 * src/xGPR/xgp_regression.py:338-367): every vector is float64 [M, k] row-major, as the block matvec takes and returns
 * them.  Per column j: rz[j] carries r.z from step 1 to step 2; init_norm[j] = |r_0[:, j]|; alpha_out[j], beta_out[j],
 * err_out[j] are this iteration's rows of the caller's [iterations, k] tables (err_out may be pinned host memory, written
 * by the device and followed by a system-scope fence).  k <= 32.  Each step is two launches over row blocks (per-block
 * partial dot products in the workspace, added in block order by every workgroup of the second launch). */
size_t xgpr_cg_block_workspace_bytes(long M, long k);
int xgpr_cg_step1_block_f64(double *w, const double *p, double *x, const double *r, double *r_next, const double *z,
                            double *rz, double *alpha_out, double *err_out, const double *init_norm, double lam2,
                            long M, long k, void *workspace, size_t workspace_bytes, void *stream);
int xgpr_cg_step2_block_f64(const double *r_next, const double *z_next, const double *p, double *p_next, const double *rz,
                            double *beta_out, long M, long k, void *workspace, size_t workspace_bytes, void *stream);

/* The classifier's cost function between its projection and back-projection
 * (src/xGPR/fitting_toolkit/nonlinear_cg_toolkit.py:243-262): pred [n, ncls] float64 row-major is replaced, row by row,
 * by softmax_2.71828(pred) - onehot(label); loss_partials [ceil(n / 256)] receives per-workgroup sums of
 * -log(max(p[label], 1e-16)) (the caller adds them); labels int64 [n] in [0, ncls) -- a label outside that range makes its
 * workgroup's partial NaN (the reference's gather / scatter fails on such an index). */
int xgpr_softmax_residual_f64(double *pred, const long *labels, long n, long ncls, double *loss_partials, void *stream);

/* The first product of RandNysPreconditioner.batch_matvec for a block of right-hand sides
 * (src/xGPR/preconditioners/rand_nys_preconditioners.py:68, `self.u_mat.T @ xvec`): t_out [rank, k] = U^T R with
 * U [M, rank], R [M, k] float64 row-major, k <= 32; per-row-block partial sums in the workspace, added in block order. */
size_t xgpr_precond_utr_block_workspace_bytes(long M, long rank, long k);
int xgpr_precond_utr_block_f64(const double *u, const double *r, double *t_out, long M, long rank, long k,
                               void *workspace, size_t workspace_bytes, void *stream);

/* RandNysPreconditioner.batch_matvec for a block of k <= 32 right-hand sides, both products on the float64 matrix cores:
 * z [M, k] = r + U ((inv_eig * prefactor - 1) .* (U^T r)) -- the reference's xprod2 + xprod1
 * (src/xGPR/preconditioners/rand_nys_preconditioners.py:66-72) as two products; r, z [M, k] row-major, z != r. */
size_t xgpr_precond_apply_block_workspace_bytes(long M, long rank, long k);
int xgpr_precond_apply_block_f64(const double *u, const double *inv_eig, double prefactor, const double *r, double *z,
                                 long M, long rank, long k, void *workspace, size_t workspace_bytes, void *stream);

/* ---- RandNysPreconditioner.batch_matvec for one right-hand side
 * (src/xGPR/preconditioners/rand_nys_preconditioners.py:66-72):
 *   z = U (inv_eig * prefactor .* U^T r) + (r - U U^T r),   U [M, rank] float64 row-major. */
size_t xgpr_precond_apply_workspace_bytes(long rank);
int xgpr_precond_apply_f64(const double *u, const double *inv_eig, double prefactor, const double *r,
                           double *z, long M, long rank, void *workspace, size_t workspace_bytes,
                           void *stream);

/* ---- resident feature cache (an MI355X-side option, no reference counterpart: the reference
 * regenerates Z chunk by chunk on every CG iteration because it cannot hold it).
 * xgpr_rbf_feature_cache_f32 is cudaRBFFeatureGen (xgpr_cuda_rfgen_cpp_ext.cpp:32-40) writing the
 * float32 (cos, sin) pairs *before* scaling into zc[n, num_rffs] -- exactly the values whose
 * widening times the scale is the float64 output -- so a shard's Z stays in HBM (32 KB per
 * datapoint at 8192 features).  xgpr_zcache_matvec_f32 is then the chunk body of the CG matvec
 * (fitting_toolkit/cg_tools.py:189-191) streamed from that cache at HBM speed:
 * w_out = sum_i z_i (z_i . v), z_i = scale * zc[i] with Z[:,0] = 1 under fit_intercept; float64
 * accumulation, deterministic.  num_freqs <= 16384 (one tile of 1024 frequencies per wave up to 8192,
 * two per wave beyond); workspace as for xgpr_ztz_matvec_f32. */
int xgpr_rbf_feature_cache_f32(const float *x, float *zc, const int8_t *radem, const float *chi,
                               long n, long d, long num_rffs, long num_freqs, long radem_shape2,
                               void *workspace, size_t workspace_bytes, void *stream);
int xgpr_zcache_matvec_f32(const float *zc, const double *v, double *w_out, long n, long num_rffs,
                           int fit_intercept, void *workspace, size_t workspace_bytes, void *stream);
/* the same over a cache that holds any kernel's feature rows z_i / scale as float32 (e.g. the
 * Conv1d / graph kernels' transform_x output rounded to float32, scale = 1, intercept column
 * already set): w_out = scale^2 * sum_i c_i (c_i . v). */
int xgpr_zcache_matvec_scaled_f32(const float *zc, const double *v, double *w_out, long n, long num_rffs,
                                  double scale, void *workspace, size_t workspace_bytes, void *stream);

/* ---- SRHTCompressor.transform_x in one pass (srht_compressor.py:87-97: zero-pad the chunk to the padded
 * width, cudaSRHT in place, gather the sampled columns): out[i, c] = SRHT(z_i)[sampler[c]] for c < ncols,
 * z [n, m] and out [n, ldo] of the same type, radem int8 [padded_width], sampler int64 [>= ncols], all on the
 * device.  z is not modified.  With y != NULL (float64 [n]) the same read of z also produces the chunk's
 * z^T y (`z_trans_y += xdata.T @ ydata`, rand_nys_constructors.py:115) in zty_out [m] (overwritten;
 * deterministic), using xgpr_srht_sample_workspace_bytes(m) of workspace.  The padded row must fit in LDS
 * (float64: padded_width <= 16384). */
size_t xgpr_srht_sample_workspace_bytes(long m);
int xgpr_srht_sample_f32(const float *z, const int8_t *radem, const long *sampler, float *out, const double *y,
                         double *zty_out, long n, long m, long padded_width, long ncols, long ldo,
                         void *workspace, size_t workspace_bytes, void *stream);
int xgpr_srht_sample_f64(const double *z, const int8_t *radem, const long *sampler, double *out, const double *y,
                         double *zty_out, long n, long m, long padded_width, long ncols, long ldo,
                         void *workspace, size_t workspace_bytes, void *stream);

/* ---- the same compressor step from FLOAT32 feature rows (the resident cache / a regenerated window:
 * Z = scale * zc, Z[:, 0] = 1 when fit_intercept; scale <= 0 selects the RBF-family constant of
 * xgpr_rbf_feature_cache_f32, a positive scale is for caches that hold complete feature rows): the float64 Z of
 * rand_nys_constructors.py:113 is never written.  out [n, ldo] float64: columns < ncols as xgpr_srht_sample_f64,
 * columns ncols .. ldo - 1 zeroed (xgpr_sketch_gemm_f64 reads whole 64-column groups: pass ldo = ncols rounded up
 * to a multiple of 64).  Any padded width up to 32768 with (padded_width / 8192) * ncols <= 8192: rows beyond the
 * LDS capacity (cfg5: 32768 float64 = 256 KiB) are transformed block by block, the last stages on the sampled
 * columns only (bit-identical to pad + SRHT + gather, transform_functions.cpp:95-121). */
int xgpr_srht_sample_rows_f32(const float *zc, const int8_t *radem, const long *sampler, double *out, const double *y,
                              double *zty_out, long n, long m, long padded_width, long ncols, long ldo, double scale,
                              int fit_intercept, void *workspace, size_t workspace_bytes, void *stream);

/* ---- the dense contractions of the preconditioner passes on the float64 matrix cores, B operand = float32
 * feature rows (rand_nys_constructors.py:34, :54, :119; the reference calls a library GEMM on float64 Z):
 *     C[I, J] (+)= sum_k A[k, i] * B(k, j),   B = Z (bt = 0: K = datapoints, J = num_rffs)
 *                                              B = Z^T (bt = 1: K = num_rffs, J = datapoints)
 * with Z = scale * zc [n, num_rffs], Z[:, 0] = 1 when fit_intercept (scale as above).  A float64 [K, lda], lda a
 * multiple of 64 >= I with the columns I .. lda - 1 zero; C float64 [I, ldc] or, trans_out, [J, ldc];
 * accumulate != 0 adds to C.  Deterministic (contraction ranges are combined in a fixed order).  bt = 1 needs
 * num_rffs % 4 == 0.  Workspace: xgpr_sketch_gemm_workspace_bytes(I, J, K, ldc, trans_out). */
size_t xgpr_sketch_gemm_workspace_bytes(long I, long J, long K, long ldc, int trans_out);
int xgpr_sketch_gemm_f64(const double *A, long lda, const float *zc, long n, long num_rffs, double *C, long ldc,
                         long I, int bt, int trans_out, double scale, int fit_intercept, int accumulate,
                         void *workspace, size_t workspace_bytes, void *stream);

/* ---- the dense Z^T Z accumulation of exact mode, the variance matrix and crude tuning
 * (scoring_toolkit/exact_nmll_calcs.py:42-78 calc_design_mat `z_trans_z += xfeatures.T @ xfeatures`, :116-139
 * calc_var_design_mat, scoring_toolkit/lb_optimizer.py:68-117 get_eigvals) on the float64 matrix cores with BOTH
 * operands the float32 feature rows (Z = scale * zc, Z[:, 0] = 1 when fit_intercept; scale as above):
 *     C[msub, msub] (+)= Z[:, :msub]^T Z[:, :msub]
 * float64 Z is never written.  Only tiles on or above the diagonal are computed; both triangles are stored.  msub a
 * multiple of 128 <= num_rffs (the reference's variance step uses the leading variance_rffs features), num_rffs a
 * multiple of 4, ldc even >= msub.  Deterministic.  Workspace: xgpr_ztz_gram_workspace_bytes(msub, n). */
size_t xgpr_ztz_gram_workspace_bytes(long msub, long n);
int xgpr_ztz_gram_f64(const float *zc, long n, long num_rffs, double *C, long ldc, long msub, double scale,
                      int fit_intercept, int accumulate, void *workspace, size_t workspace_bytes, void *stream);

/* ---- cudaMiniARDGrad(inputArr, outputArr, precompWeights, sigmaMap, sigmaVals, gradArr, fitIntercept)
 * (gpu_rf_gen/xgpr_cuda_rfgen_cpp_ext.cpp:50-60; cpu_rf_gen/rbf_ops/ard_ops.cpp:39-124): MiniARD random
 * features out[n, num_rffs] and their gradient grad[n, num_rffs, num_lengthscales] w.r.t. the per-group
 * inverse lengthscales, from the dense precomputed weights[num_freqs, d]; sigma_map[d] int32 (group of each
 * input feature), sigma_vals[d] float64, all on the device.  fit_intercept only selects the constant; the
 * caller sets column 0 (kernel_baseclass.py:356-359).  Up to 8 groups; n <= 262140 per call. */
int xgpr_mini_ard_grad_f32(const float *x, double *out, const float *weights, const int32_t *sigma_map,
                           const double *sigma_vals, double *grad, long n, long d, long out_rows,
                           long num_rffs, long num_freqs, long w_cols, long map_len, long sig_len,
                           long grad_rows, long grad_cols, long num_lengthscales, int fit_intercept,
                           void *stream);
int xgpr_mini_ard_grad_f64(const double *x, double *out, const double *weights, const int32_t *sigma_map,
                           const double *sigma_vals, double *grad, long n, long d, long out_rows,
                           long num_rffs, long num_freqs, long w_cols, long map_len, long sig_len,
                           long grad_rows, long grad_cols, long num_lengthscales, int fit_intercept,
                           void *stream);

/* ---- Block matvec over the resident cache for k right-hand sides, the matrix-core part of the
 * CG path: replaces `matvec += Z.T @ (Z @ vec)` of GPU_ConjugateGrad._matvec for vec of shape
 * [M, k] (fitting_toolkit/cg_tools.py:41-44; k = nsamples + 1 = 26 in approximate_nmll,
 * xgp_regression.py:338-367).  v, w_out: float64 [num_rffs, k] C-contiguous; zc as above.
 * w_out (+)= scale^2 * Zc^T (Zc v) with Z[:,0] = 1 under fit_intercept, float64 MFMA
 * (v_mfma_f64_16x16x4_f64), deterministic.  scale <= 0 selects the RBF-family scale
 * sqrt(1/F) or sqrt(1/(F - 0.5)) (rbf_ops.cpp:68-72); accumulate != 0 adds into w_out.
 * 1 <= k <= 32 per call, num_rffs a multiple of 4. */
size_t xgpr_zcache_block_workspace_bytes(long n, long num_rffs, long k);
int xgpr_zcache_block_matvec_f32(const float *zc, const double *v, double *w_out, long n, long num_rffs,
                                 long k, int fit_intercept, double scale, int accumulate,
                                 void *workspace, size_t workspace_bytes, void *stream);

/* The two contractions of the block matvec on their own -- what the classifier's cost function
 * does per chunk (fitting_toolkit/nonlinear_cg_toolkit.py:251-269: `pred = xd @ wvec`, then
 * `grad[:,k] += ((pred[:,k] - targets)[:,None] * xd).sum(axis=0)`), and `xfeatures @ weights` in
 * predict (xgp_classification.py:96-102):
 *   project:      t_out[n, k]        =  Z v        v     [num_rffs, k]
 *   backproject:  g_out[num_rffs, k] (+)= Z^T r    r     [n, k]
 * with Z = scale * zc and Z[:,0] = 1 under fit_intercept; all float64, C-contiguous.  The back-projection takes
 * xgpr_zcache_block_workspace_bytes(n, num_rffs, k).  The projection runs with workspace == NULL; given
 * xgpr_zcache_block_project_workspace_bytes(n, num_rffs, k) (0 for long launches; a workspace sized for the
 * back-projection is large enough), a short launch (fewer than 65536 rows: a chunk of ~2000 rows is what the reference
 * feeds it, cg_tools.py:41-44) splits the contraction over the features across workgroups and adds the partial sums in
 * a fixed order -- deterministic for a given (n, num_rffs, k). */
size_t xgpr_zcache_block_project_workspace_bytes(long n, long num_rffs, long k);
int xgpr_zcache_block_project_f32(const float *zc, const double *v, double *t_out, long n, long num_rffs,
                                  long k, int fit_intercept, double scale,
                                  void *workspace, size_t workspace_bytes, void *stream);
int xgpr_zcache_block_backproject_f32(const float *zc, const double *r, double *g_out, long n, long num_rffs,
                                      long k, int fit_intercept, double scale, int accumulate,
                                      void *workspace, size_t workspace_bytes, void *stream);

/* ---- self test of the cross-lane butterfly stages the wave-level FHT is built on: for each
 * of the 6 lane strides h = 1, 2, 4, 8, 16, 32 runs one stage on v[r] = lane + 64 r
 * (r = 0..15) and writes the result to out[6][16][64] (int32, device).  Expected:
 * bit h of lane clear -> 2 lane + h + 128 r, set -> -h. */
int xgpr_selftest_lane_xor(int32_t *out, void *stream);

/* ---- the exchange step of the sharded path issued on the caller's stream.  The reference is single-device
 * (docs/FAQ.rst:12-15); its loops `for chunk: w += Z.T @ (Z @ v)` (fitting_toolkit/cg_tools.py:189-191) and
 * `acc += ...` (preconditioners/rand_nys_constructors.py:115-119) become per-rank partial sums, and these entry points
 * add them up with RCCL's all-reduce ENQUEUED ON `stream` -- directly behind the kernel that produced the partial sum,
 * with no hand-off to another stream.  xgpr_rccl_load(path) resolves the RCCL library the process already uses
 * (dlopen; NULL / "" = "librccl.so"); xgpr_rccl_unique_id fills 128 bytes (ncclUniqueId) on one rank, which the host
 * side distributes by any means (here: the torch.distributed store); every rank then calls xgpr_rccl_comm_init.
 * buf: float64 [n] on the device, summed in place over all ranks. */
int xgpr_rccl_load(const char *path);
int xgpr_rccl_unique_id(char *out128);
int xgpr_rccl_comm_init(void **comm, int nranks, const char *id128, int rank);
int xgpr_allreduce_sum_f64(void *comm, double *buf, long n, void *stream);
int xgpr_rccl_comm_destroy(void *comm);

#ifdef __cplusplus
}
#endif
#endif /* XGPR_HIP_H */
