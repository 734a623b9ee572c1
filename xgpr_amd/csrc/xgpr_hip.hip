// xgpr_hip.hip -- MI355X (gfx950 / CDNA4) kernels + C ABI for the xGPR hot path:
// SORF / fast-Hadamard random-feature generation and the fused Z^T(Zv) CG matvec.
//
// Written for wave64 CDNA4 only (no CUDA/portability layer).  Build:
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared xgpr_hip.hip -o libxgpr_hip.so
// -ffp-contract=off is REQUIRED: the butterflies and the Rademacher/normaliser multiplies
// must round exactly like the reference's scalar C++ (which is built without FMA), so that
// the f32 argument of every cos/sin is bit-identical to the reference's.  Places where a
// fused multiply-add is wanted call __builtin_fma[f] explicitly.
//
// Reference behaviour restated here (paths relative to /root/reference/src/xGPR/):
//   FHT butterflies      random_feature_generation/cpu_rf_gen/shared_fht_functions/hadamard_transforms.cpp:83-127
//   SORF (3 x D,H)       .../shared_fht_functions/shared_rfgen_ops.cpp:51-78
//   RBF post-process     .../shared_fht_functions/shared_rfgen_ops.cpp:92-114 (grad :125-156)
//   row drivers          .../rbf_ops/rbf_ops.cpp:27-106, convolution_ops/rbf_convolution.cpp:23-140,
//                        convolution_ops/conv1d_operations.cpp:23-168, basic_ops/transform_functions.cpp:22-121
//   CG matvec            fitting_toolkit/cg_tools.py:173-200 ; z^T y  scoring_toolkit/exact_nmll_calcs.py:13-39
//
// Layout of one wave's data ("wave tile"): 16 VGPRs x 64 lanes = 1024 consecutive random
// frequencies of one datapoint; register r, lane l holds frequency  f = 1024*b + 64*r + l.
// For padded width P <= 1024 that is 1024/P complete SORF transforms; within a transform the
// element index is (64*r + l) mod P, so butterfly strides < 64 are cross-lane (DPP /
// v_permlane{16,32}_swap, no LDS) and strides >= 64 are register-local.  Loads of x and the
// f64 (cos,sin) stores are fully coalesced (256 B / 1 KiB per wave instruction).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <dlfcn.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>          // types only: the library is resolved at run time (xgpr_rccl_load), never linked
#else
// No RCCL development headers on this install: the handful of ABI-stable NCCL types the dlsym'ed entry points use
// (rccl.h: ncclComm_t opaque, 128-byte id, ncclSuccess = 0, ncclSum = 0, ncclFloat64 = 8).
typedef struct ncclComm *ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef enum { ncclFloat64 = 8 } ncclDataType_t;
#endif

#include "../../include/xgpr_hip.h"

namespace {

#include "common.inc"
#include "generic_fht.inc"
#include "wave_sorf.inc"
#include "wave_kernels.inc"
#include "wave_tile.inc"
#include "fused_ztz.inc"
#include "zcache.inc"
#include "zblock.inc"
#include "sketch_gemm.inc"
#include "gram.inc"
#include "mini_ard.inc"
#include "cg_kernels.inc"
#include "launchers.inc"

}  // namespace

// ------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------
extern "C" {

const char *xgpr_last_error(void) { return g_err.c_str(); }
const char *xgpr_build_arch(void) { return "gfx950"; }

// sha256 of the sources and flags this binary was compiled from (xgpr_amd/build.py source_id(), passed as
// -DXGPR_BUILD_ID); the marker in front lets build.py read it from the file without loading the library
#ifndef XGPR_BUILD_ID
#define XGPR_BUILD_ID "unidentified"
#endif
const char *xgpr_build_id(void) {
    static const char id[] = "xgpr-build-id:" XGPR_BUILD_ID;
    return id + 14;
}

int xgpr_fht_f32(float *x, long n, long dim1, long dim2, void *stream) { return fht_impl<float>(x, n, dim1, dim2, stream); }
int xgpr_fht_f64(double *x, long n, long dim1, long dim2, void *stream) { return fht_impl<double>(x, n, dim1, dim2, stream); }

int xgpr_srht_f32(float *x, const int8_t *radem, long n, long dim, long radem_len, void *stream) {
    return srht_impl<float>(x, radem, n, dim, radem_len, stream);
}
int xgpr_srht_f64(double *x, const int8_t *radem, long n, long dim, long radem_len, void *stream) {
    return srht_impl<double>(x, radem, n, dim, radem_len, stream);
}

size_t xgpr_rbf_workspace_bytes(long radem_shape2) { return masks_bytes(radem_shape2); }
size_t xgpr_conv_workspace_bytes(long radem_shape2, long width, int elem_size, long nseq) {
    return xgpr_sorf_workspace_bytes(radem_shape2, width, elem_size) + align_up((size_t)(nseq > 0 ? nseq : 0) * sizeof(int32_t), 256);
}
size_t xgpr_sorf_workspace_bytes(long radem_shape2, long width, int elem_size) {
    const size_t a = masks_bytes(radem_shape2);
    const size_t b = generic_scratch_bytes(padded_width(width), (size_t)elem_size);
    return a > b ? a : b;
}

int xgpr_rbf_feature_gen_f32(const float *x, double *out, const int8_t *radem, const float *chi, long n, long d,
                             long out_rows, long num_rffs, long num_freqs, long radem_shape2, int fit_intercept,
                             void *workspace, size_t workspace_bytes, void *stream) {
    return rbf_impl<float>(x, out, nullptr, radem, chi, n, d, out_rows, num_rffs, 0, 0, num_freqs, radem_shape2, 0.0,
                           fit_intercept, false, workspace, workspace_bytes, stream);
}
int xgpr_rbf_feature_gen_f64(const double *x, double *out, const int8_t *radem, const double *chi, long n, long d,
                             long out_rows, long num_rffs, long num_freqs, long radem_shape2, int fit_intercept,
                             void *workspace, size_t workspace_bytes, void *stream) {
    return rbf_impl<double>(x, out, nullptr, radem, chi, n, d, out_rows, num_rffs, 0, 0, num_freqs, radem_shape2, 0.0,
                            fit_intercept, false, workspace, workspace_bytes, stream);
}
int xgpr_rbf_grad_f32(const float *x, double *out, double *grad, const int8_t *radem, const float *chi, long n,
                      long d, long out_rows, long num_rffs, long grad_rows, long grad_cols, long num_freqs,
                      long radem_shape2, double sigma, int fit_intercept, void *workspace, size_t workspace_bytes,
                      void *stream) {
    return rbf_impl<float>(x, out, grad, radem, chi, n, d, out_rows, num_rffs, grad_rows, grad_cols, num_freqs,
                           radem_shape2, sigma, fit_intercept, true, workspace, workspace_bytes, stream);
}
int xgpr_rbf_grad_f64(const double *x, double *out, double *grad, const int8_t *radem, const double *chi, long n,
                      long d, long out_rows, long num_rffs, long grad_rows, long grad_cols, long num_freqs,
                      long radem_shape2, double sigma, int fit_intercept, void *workspace, size_t workspace_bytes,
                      void *stream) {
    return rbf_impl<double>(x, out, grad, radem, chi, n, d, out_rows, num_rffs, grad_rows, grad_cols, num_freqs,
                            radem_shape2, sigma, fit_intercept, true, workspace, workspace_bytes, stream);
}

int xgpr_mini_ard_grad_f32(const float *x, double *out, const float *weights, const int32_t *sigma_map,
                           const double *sigma_vals, double *grad, long n, long d, long out_rows, long num_rffs,
                           long num_freqs, long w_cols, long map_len, long sig_len, long grad_rows, long grad_cols,
                           long num_lengthscales, int fit_intercept, void *stream) {
    return mini_ard_impl<float>(x, out, weights, sigma_map, sigma_vals, grad, n, d, out_rows, num_rffs, num_freqs, w_cols,
                                map_len, sig_len, grad_rows, grad_cols, num_lengthscales, fit_intercept, stream);
}
int xgpr_mini_ard_grad_f64(const double *x, double *out, const double *weights, const int32_t *sigma_map,
                           const double *sigma_vals, double *grad, long n, long d, long out_rows, long num_rffs,
                           long num_freqs, long w_cols, long map_len, long sig_len, long grad_rows, long grad_cols,
                           long num_lengthscales, int fit_intercept, void *stream) {
    return mini_ard_impl<double>(x, out, weights, sigma_map, sigma_vals, grad, n, d, out_rows, num_rffs, num_freqs, w_cols,
                                 map_len, sig_len, grad_rows, grad_cols, num_lengthscales, fit_intercept, stream);
}

int xgpr_conv1d_fgen_f32(const float *x, double *out, const int8_t *radem, const float *chi,
                         const int32_t *seqlen_host, const int32_t *seqlen_dev, long n, long L, long C,
                         long out_rows, long num_rffs, long num_freqs, long radem_shape2, long nseq, int conv_width,
                         int scaling_type, void *workspace, size_t workspace_bytes, void *stream) {
    return conv_impl<float>(x, out, nullptr, nullptr, radem, chi, seqlen_host, seqlen_dev, n, L, C, out_rows, num_rffs,
                            0, 0, num_freqs, radem_shape2, nseq, 0.0, conv_width, scaling_type, MODE_CONV, workspace,
                            workspace_bytes, stream);
}
int xgpr_conv1d_fgen_f64(const double *x, double *out, const int8_t *radem, const double *chi,
                         const int32_t *seqlen_host, const int32_t *seqlen_dev, long n, long L, long C,
                         long out_rows, long num_rffs, long num_freqs, long radem_shape2, long nseq, int conv_width,
                         int scaling_type, void *workspace, size_t workspace_bytes, void *stream) {
    return conv_impl<double>(x, out, nullptr, nullptr, radem, chi, seqlen_host, seqlen_dev, n, L, C, out_rows,
                             num_rffs, 0, 0, num_freqs, radem_shape2, nseq, 0.0, conv_width, scaling_type, MODE_CONV,
                             workspace, workspace_bytes, stream);
}
int xgpr_conv_grad_f32(const float *x, double *out, double *grad, const int8_t *radem, const float *chi,
                       const int32_t *seqlen_host, const int32_t *seqlen_dev, long n, long L, long C, long out_rows,
                       long num_rffs, long grad_rows, long grad_cols, long num_freqs, long radem_shape2, long nseq,
                       double sigma, int conv_width, int scaling_type, void *workspace, size_t workspace_bytes,
                       void *stream) {
    return conv_impl<float>(x, out, grad, nullptr, radem, chi, seqlen_host, seqlen_dev, n, L, C, out_rows, num_rffs,
                            grad_rows, grad_cols, num_freqs, radem_shape2, nseq, sigma, conv_width, scaling_type,
                            MODE_CONV_GRAD, workspace, workspace_bytes, stream);
}
int xgpr_conv_grad_f64(const double *x, double *out, double *grad, const int8_t *radem, const double *chi,
                       const int32_t *seqlen_host, const int32_t *seqlen_dev, long n, long L, long C, long out_rows,
                       long num_rffs, long grad_rows, long grad_cols, long num_freqs, long radem_shape2, long nseq,
                       double sigma, int conv_width, int scaling_type, void *workspace, size_t workspace_bytes,
                       void *stream) {
    return conv_impl<double>(x, out, grad, nullptr, radem, chi, seqlen_host, seqlen_dev, n, L, C, out_rows, num_rffs,
                             grad_rows, grad_cols, num_freqs, radem_shape2, nseq, sigma, conv_width, scaling_type,
                             MODE_CONV_GRAD, workspace, workspace_bytes, stream);
}
int xgpr_conv1d_maxpool_f32(const float *x, float *out, const int8_t *radem, const float *chi,
                            const int32_t *seqlen_host, const int32_t *seqlen_dev, long n, long L, long C,
                            long out_rows, long num_rffs, long num_freqs, long radem_shape2, long nseq,
                            int conv_width, void *workspace, size_t workspace_bytes, void *stream) {
    return conv_impl<float>(x, nullptr, nullptr, out, radem, chi, seqlen_host, seqlen_dev, n, L, C, out_rows, num_rffs,
                            0, 0, num_freqs, radem_shape2, nseq, 0.0, conv_width, 0, MODE_MAXPOOL, workspace,
                            workspace_bytes, stream);
}
int xgpr_conv1d_maxpool_f64(const double *x, float *out, const int8_t *radem, const double *chi,
                            const int32_t *seqlen_host, const int32_t *seqlen_dev, long n, long L, long C,
                            long out_rows, long num_rffs, long num_freqs, long radem_shape2, long nseq,
                            int conv_width, void *workspace, size_t workspace_bytes, void *stream) {
    return conv_impl<double>(x, nullptr, nullptr, out, radem, chi, seqlen_host, seqlen_dev, n, L, C, out_rows,
                             num_rffs, 0, 0, num_freqs, radem_shape2, nseq, 0.0, conv_width, 0, MODE_MAXPOOL,
                             workspace, workspace_bytes, stream);
}

size_t xgpr_ztz_matvec_workspace_bytes(long num_rffs, long radem_shape2) {
    return ztz_workspace_bytes(num_rffs, radem_shape2);
}
int xgpr_ztz_matvec_plan(long d, long num_freqs) {
    const long P = padded_width(d);
    if (P > 4096 || num_freqs < 1 || num_freqs > 65536) return 0;
    if (ztz_takes_two_passes(d, P, num_freqs, true)) return 3;
    if (P > 1024) return ztz3_shape_ok(d, P, ztz3_compute_tiles(P, num_freqs)) ? 1 : 0;
    return ztz3_shape_ok(d, P, (int)((num_freqs + 1023) / 1024)) ? 1 : 2;
}
int xgpr_ztz_matvec_f32(const float *x, const int8_t *radem, const float *chi, const double *v, double *w_out,
                        long n, long d, long num_rffs, long num_freqs, long radem_shape2, int fit_intercept,
                        void *workspace, size_t workspace_bytes, void *stream) {
    return ztz_impl<true>(x, radem, chi, v, w_out, n, d, num_rffs, num_freqs, radem_shape2, fit_intercept, workspace,
                          workspace_bytes, stream);
}
int xgpr_zty_f32(const float *x, const int8_t *radem, const float *chi, const double *y, double *zty_out, long n,
                 long d, long num_rffs, long num_freqs, long radem_shape2, int fit_intercept, void *workspace,
                 size_t workspace_bytes, void *stream) {
    return ztz_impl<false>(x, radem, chi, y, zty_out, n, d, num_rffs, num_freqs, radem_shape2, fit_intercept, workspace,
                           workspace_bytes, stream);
}

int xgpr_srht_sample_rows_f32(const float *zc, const int8_t *radem, const long *sampler, double *out, const double *y,
                              double *zty_out, long n, long m, long padded_width, long ncols, long ldo, double scale,
                              int fit_intercept, void *workspace, size_t workspace_bytes, void *stream) {
    return srht_sample_rows_impl(zc, radem, sampler, out, y, zty_out, n, m, padded_width, ncols, ldo, scale, fit_intercept,
                                 workspace, workspace_bytes, stream);
}
size_t xgpr_sketch_gemm_workspace_bytes(long I, long J, long K, long ldc, int trans_out) {
    return sk_geometry(I, J, K, ldc, trans_out).ws_bytes;
}
int xgpr_sketch_gemm_f64(const double *A, long lda, const float *zc, long n, long num_rffs, double *C, long ldc,
                         long I, int bt, int trans_out, double scale, int fit_intercept, int accumulate,
                         void *workspace, size_t workspace_bytes, void *stream) {
    return sketch_gemm_impl(A, lda, zc, n, num_rffs, C, ldc, I, bt, trans_out, scale, fit_intercept, accumulate, workspace,
                            workspace_bytes, stream);
}

size_t xgpr_ztz_gram_workspace_bytes(long msub, long n) { return gram_workspace_bytes(msub, n); }
int xgpr_ztz_gram_f64(const float *zc, long n, long num_rffs, double *C, long ldc, long msub, double scale,
                      int fit_intercept, int accumulate, void *workspace, size_t workspace_bytes, void *stream) {
    return gram_impl(zc, n, num_rffs, C, ldc, msub, scale, fit_intercept, accumulate, workspace, workspace_bytes, stream);
}

int xgpr_cg_step1_f64(double *w, const double *p, double *x, const double *r, double *r_next, const double *z,
                      double *scal, double lam2, double init_norm, long M, double stop_tol, double *err_out,
                      void *stream) {
    if (M <= 0) return fail(XGPR_ERR_ARRAY_DIMS, "incorrect array dims passed");
    hipLaunchKernelGGL(cg_step1_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, w, p, x, r, r_next, z, scal, lam2,
                       init_norm, M, stop_tol, err_out);
    HIP_TRY(hipGetLastError(), "cg_step1_kernel launch");
    return 0;
}
int xgpr_cg_step2_f64(const double *r_next, const double *z_next, const double *p, double *p_next, double *scal, long M,
                      double stop_tol, void *stream) {
    if (M <= 0) return fail(XGPR_ERR_ARRAY_DIMS, "incorrect array dims passed");
    hipLaunchKernelGGL(cg_step2_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, r_next, z_next, p, p_next, scal, M,
                       stop_tol);
    HIP_TRY(hipGetLastError(), "cg_step2_kernel launch");
    return 0;
}

namespace {
constexpr long UTRB_MAX_BLOCKS = 64;
struct UtrbGeom { long nrb; long rows_per; };
UtrbGeom utrb_geometry(long M) {
    long nrb = (M + 63) / 64;
    if (nrb > UTRB_MAX_BLOCKS) nrb = UTRB_MAX_BLOCKS;
    if (nrb < 1) nrb = 1;
    const long rows_per = ((M + nrb - 1) / nrb + 3) / 4 * 4;
    return {(M + rows_per - 1) / rows_per, rows_per};
}
int utr_block_launch(const double *u, const double *r, double *t_out, long M, long rank, long k, double *part, hipStream_t st) {
    const UtrbGeom gm = utrb_geometry(M);
    const dim3 grid((unsigned)((rank + 127) / 128), (unsigned)gm.nrb);
    if (k <= 16) hipLaunchKernelGGL((precond_utr_mfma_kernel<1>), grid, dim3(512), 0, st, u, r, part, M, rank, (int)k, gm.rows_per);
    else hipLaunchKernelGGL((precond_utr_mfma_kernel<2>), grid, dim3(512), 0, st, u, r, part, M, rank, (int)k, gm.rows_per);
    HIP_TRY(hipGetLastError(), "precond_utr_mfma_kernel launch");
    const long total = rank * k;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, part, t_out, total, gm.nrb);
    HIP_TRY(hipGetLastError(), "reduce_slabs_kernel launch");
    return 0;
}
}  // namespace
size_t xgpr_precond_utr_block_workspace_bytes(long M, long rank, long k) {
    return (size_t)utrb_geometry(M).nrb * (size_t)rank * (size_t)k * sizeof(double);
}
int xgpr_precond_utr_block_f64(const double *u, const double *r, double *t_out, long M, long rank, long k,
                               void *workspace, size_t workspace_bytes, void *stream) {
    if (M <= 0 || rank <= 0 || k <= 0) return fail(XGPR_ERR_ARRAY_DIMS, "incorrect array dims passed");
    if (k > 32) return fail(XGPR_ERR_UNSUPPORTED, "U^T R block kernel takes at most 32 right-hand sides");
    if (!workspace || workspace_bytes < xgpr_precond_utr_block_workspace_bytes(M, rank, k) || !aligned16(workspace))
        return fail(XGPR_ERR_WORKSPACE, "workspace too small (see xgpr_precond_utr_block_workspace_bytes)");
    return utr_block_launch(u, r, t_out, M, rank, k, reinterpret_cast<double *>(workspace), (hipStream_t)stream);
}
size_t xgpr_precond_apply_block_workspace_bytes(long M, long rank, long k) {
    return xgpr_precond_utr_block_workspace_bytes(M, rank, k) + (size_t)rank * (size_t)k * sizeof(double);
}
int xgpr_precond_apply_block_f64(const double *u, const double *inv_eig, double prefactor, const double *r, double *z,
                                 long M, long rank, long k, void *workspace, size_t workspace_bytes, void *stream) {
    if (M <= 0 || rank <= 0 || k <= 0) return fail(XGPR_ERR_ARRAY_DIMS, "incorrect array dims passed");
    if (k > 32) return fail(XGPR_ERR_UNSUPPORTED, "block preconditioner apply takes at most 32 right-hand sides");
    if (!workspace || workspace_bytes < xgpr_precond_apply_block_workspace_bytes(M, rank, k) || !aligned16(workspace))
        return fail(XGPR_ERR_WORKSPACE, "workspace too small (see xgpr_precond_apply_block_workspace_bytes)");
    hipStream_t st = (hipStream_t)stream;
    double *part = reinterpret_cast<double *>(workspace);
    double *t = part + (size_t)utrb_geometry(M).nrb * rank * k;
    int rc = utr_block_launch(u, r, t, M, rank, k, part, st);
    if (rc) return rc;
    const dim3 grid((unsigned)((M + 15) / 16));
    if (k <= 16) hipLaunchKernelGGL((precond_uz_mfma_kernel<1>), grid, dim3(256), 0, st, u, t, inv_eig, prefactor, r, z, M, rank, (int)k);
    else hipLaunchKernelGGL((precond_uz_mfma_kernel<2>), grid, dim3(256), 0, st, u, t, inv_eig, prefactor, r, z, M, rank, (int)k);
    HIP_TRY(hipGetLastError(), "precond_uz_mfma_kernel launch");
    return 0;
}

size_t xgpr_cg_block_workspace_bytes(long M, long k) {
    const long nblk = (M + CGBLK_ROWS - 1) / CGBLK_ROWS;
    return (size_t)(nblk > 0 ? nblk : 1) * 3 * (size_t)(k > 0 ? k : 1) * sizeof(double);
}
int xgpr_cg_step1_block_f64(double *w, const double *p, double *x, const double *r, double *r_next, const double *z,
                            double *rz, double *alpha_out, double *err_out, const double *init_norm, double lam2,
                            long M, long k, void *workspace, size_t workspace_bytes, void *stream) {
    if (M <= 0 || k <= 0) return fail(XGPR_ERR_ARRAY_DIMS, "incorrect array dims passed");
    if (k > CGBLK_KMAX) return fail(XGPR_ERR_UNSUPPORTED, "block CG steps take at most 32 right-hand sides");
    if (!workspace || workspace_bytes < xgpr_cg_block_workspace_bytes(M, k))
        return fail(XGPR_ERR_WORKSPACE, "workspace too small (see xgpr_cg_block_workspace_bytes)");
    const unsigned nblk = (unsigned)((M + CGBLK_ROWS - 1) / CGBLK_ROWS);
    double *part = reinterpret_cast<double *>(workspace);
    hipLaunchKernelGGL(cg_step1a_block_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, w, p, r, z, part, lam2, M, (int)k);
    HIP_TRY(hipGetLastError(), "cg_step1a_block_kernel launch");
    hipLaunchKernelGGL(cg_step1b_block_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, w, p, x, r, r_next, part, rz,
                       alpha_out, err_out, init_norm, M, (int)k);
    HIP_TRY(hipGetLastError(), "cg_step1b_block_kernel launch");
    return 0;
}
int xgpr_cg_step2_block_f64(const double *r_next, const double *z_next, const double *p, double *p_next, const double *rz,
                            double *beta_out, long M, long k, void *workspace, size_t workspace_bytes, void *stream) {
    if (M <= 0 || k <= 0) return fail(XGPR_ERR_ARRAY_DIMS, "incorrect array dims passed");
    if (k > CGBLK_KMAX) return fail(XGPR_ERR_UNSUPPORTED, "block CG steps take at most 32 right-hand sides");
    if (!workspace || workspace_bytes < xgpr_cg_block_workspace_bytes(M, k))
        return fail(XGPR_ERR_WORKSPACE, "workspace too small (see xgpr_cg_block_workspace_bytes)");
    const unsigned nblk = (unsigned)((M + CGBLK_ROWS - 1) / CGBLK_ROWS);
    double *part = reinterpret_cast<double *>(workspace);
    hipLaunchKernelGGL(cg_step2a_block_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, r_next, z_next, part, M, (int)k);
    HIP_TRY(hipGetLastError(), "cg_step2a_block_kernel launch");
    hipLaunchKernelGGL(cg_step2b_block_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, z_next, p, p_next, part, rz,
                       beta_out, M, (int)k);
    HIP_TRY(hipGetLastError(), "cg_step2b_block_kernel launch");
    return 0;
}

int xgpr_softmax_residual_f64(double *pred, const long *labels, long n, long ncls, double *loss_partials, void *stream) {
    if (n <= 0 || ncls <= 0 || ncls > 2147483647L) return fail(XGPR_ERR_ARRAY_DIMS, "incorrect array dims passed");
    const long nblk = (n + 255) / 256;
    if (nblk > 2147483647L) return fail(XGPR_ERR_UNSUPPORTED, "too many datapoints for one launch");
    hipLaunchKernelGGL(softmax_residual_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, pred, labels, n, (int)ncls,
                       loss_partials);
    HIP_TRY(hipGetLastError(), "softmax_residual_kernel launch");
    return 0;
}

size_t xgpr_precond_apply_workspace_bytes(long rank) { return (size_t)(PRE_BLOCKS + 1) * rank * sizeof(double); }
int xgpr_precond_apply_f64(const double *u, const double *inv_eig, double prefactor, const double *r, double *z,
                           long M, long rank, void *workspace, size_t workspace_bytes, void *stream) {
    if (M <= 0 || rank <= 0) return fail(XGPR_ERR_ARRAY_DIMS, "incorrect array dims passed");
    if (!workspace || workspace_bytes < xgpr_precond_apply_workspace_bytes(rank) || !aligned16(workspace))
        return fail(XGPR_ERR_WORKSPACE, "workspace too small (see xgpr_precond_apply_workspace_bytes)");
    hipStream_t st = (hipStream_t)stream;
    double *part = reinterpret_cast<double *>(workspace);
    double *t = part + (size_t)PRE_BLOCKS * rank;
    const int nb = (int)(M < PRE_BLOCKS ? M : PRE_BLOCKS);
    hipLaunchKernelGGL(precond_utr_kernel, dim3(nb), dim3(256), 0, st, u, r, part, M, rank);
    HIP_TRY(hipGetLastError(), "precond_utr_kernel launch");
    hipLaunchKernelGGL(reduce_slabs_short_kernel, dim3((unsigned)((rank + 15) / 16)), dim3(256), 0, st, part, t, rank, (long)nb);
    HIP_TRY(hipGetLastError(), "reduce_slabs_short_kernel launch");
    hipLaunchKernelGGL(precond_uz_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, u, t, inv_eig, prefactor, r, z,
                       M, rank);
    HIP_TRY(hipGetLastError(), "precond_uz_kernel launch");
    return 0;
}

int xgpr_rbf_feature_cache_f32(const float *x, float *zc, const int8_t *radem, const float *chi, long n, long d,
                               long num_rffs, long num_freqs, long radem_shape2, void *workspace,
                               size_t workspace_bytes, void *stream) {
    return zcache_build_impl(x, zc, radem, chi, n, d, num_rffs, num_freqs, radem_shape2, workspace, workspace_bytes, stream);
}
int xgpr_zcache_matvec_f32(const float *zc, const double *v, double *w_out, long n, long num_rffs, int fit_intercept,
                           void *workspace, size_t workspace_bytes, void *stream) {
    return zcache_matvec_impl(zc, v, w_out, n, num_rffs, fit_intercept, 0.0, workspace, workspace_bytes, stream);
}
int xgpr_zcache_matvec_scaled_f32(const float *zc, const double *v, double *w_out, long n, long num_rffs, double scale,
                                  void *workspace, size_t workspace_bytes, void *stream) {
    if (!(scale > 0.0)) return fail(XGPR_ERR_ARRAY_DIMS, "scale must be positive");
    return zcache_matvec_impl(zc, v, w_out, n, num_rffs, 0, scale, workspace, workspace_bytes, stream);
}

size_t xgpr_zcache_block_workspace_bytes(long n, long num_rffs, long k) {
    if (n <= 0 || num_rffs <= 0 || k < 1) return 0;
    const ZbGeom gm = zb_geometry(n, num_rffs, k > 32 ? 32 : k);
    return gm.t_bytes + gm.slab_bytes + gm.tpart_bytes;
}
size_t xgpr_zcache_block_project_workspace_bytes(long n, long num_rffs, long k) {
    if (n <= 0 || num_rffs <= 0 || k < 1) return 0;
    return zb_geometry(n, num_rffs, k > 32 ? 32 : k).tpart_bytes;
}
int xgpr_zcache_block_matvec_f32(const float *zc, const double *v, double *w_out, long n, long num_rffs, long k,
                                 int fit_intercept, double scale, int accumulate, void *workspace,
                                 size_t workspace_bytes, void *stream) {
    return zcache_block_impl(ZB_MATVEC, zc, v, w_out, n, num_rffs, k, fit_intercept, scale, accumulate, workspace,
                             workspace_bytes, stream);
}
int xgpr_zcache_block_project_f32(const float *zc, const double *v, double *t_out, long n, long num_rffs, long k,
                                  int fit_intercept, double scale, void *workspace, size_t workspace_bytes, void *stream) {
    return zcache_block_impl(ZB_PROJECT, zc, v, t_out, n, num_rffs, k, fit_intercept, scale, 0, workspace, workspace_bytes,
                             stream);
}
int xgpr_zcache_block_backproject_f32(const float *zc, const double *r, double *g_out, long n, long num_rffs, long k,
                                      int fit_intercept, double scale, int accumulate, void *workspace,
                                      size_t workspace_bytes, void *stream) {
    return zcache_block_impl(ZB_BACKPROJECT, zc, r, g_out, n, num_rffs, k, fit_intercept, scale, accumulate, workspace,
                             workspace_bytes, stream);
}

size_t xgpr_srht_sample_workspace_bytes(long m) { return (size_t)SRHT_ZTY_MAX_BLOCKS * (m > 0 ? m : 0) * sizeof(double); }
int xgpr_srht_sample_f64(const double *z, const int8_t *radem, const long *sampler, double *out, const double *y,
                         double *zty_out, long n, long m, long padded_width_, long ncols, long ldo, void *workspace,
                         size_t workspace_bytes, void *stream) {
    return srht_sample_impl<double>(z, radem, sampler, out, y, zty_out, n, m, padded_width_, ncols, ldo, workspace,
                                    workspace_bytes, stream);
}
int xgpr_srht_sample_f32(const float *z, const int8_t *radem, const long *sampler, float *out, const double *y,
                         double *zty_out, long n, long m, long padded_width_, long ncols, long ldo, void *workspace,
                         size_t workspace_bytes, void *stream) {
    return srht_sample_impl<float>(z, radem, sampler, out, y, zty_out, n, m, padded_width_, ncols, ldo, workspace,
                                   workspace_bytes, stream);
}

#ifdef XGPR_ZB_STAMPS      /* development build only (tools/zblock_stamps.py): copies the stamp buffer of zblock.inc to the host */
int xgpr_debug_zb_stamps(void *dst, size_t bytes) {
    if (bytes > sizeof(g_zb_stamps)) bytes = sizeof(g_zb_stamps);
    HIP_TRY(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_zb_stamps), bytes), "hipMemcpyFromSymbol(g_zb_stamps)");
    return (int)(sizeof(g_zb_stamps) / 8);
}
#endif

int xgpr_selftest_lane_xor(int32_t *out, void *stream) {
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
    HIP_TRY(hipGetLastError(), "selftest_kernel launch");
    return 0;
}

// ---- RCCL on the caller's stream.  torch.distributed's all-reduce runs RCCL on ProcessGroupNCCL's own stream and
// chains it to the compute stream with an event on each side; the 64 KiB exchange of a CG iteration is latency-bound, so
// these entry points enqueue ncclAllReduce directly on the stream the matvec's slab reduction was launched on.  RCCL is
// the copy the process already uses (the path handed to xgpr_rccl_load: torch ships its own librccl.so), resolved with
// dlopen / dlsym so that libxgpr_hip.so carries no link-time dependency on it.
namespace {
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*get_unique_id)(ncclUniqueId *) = nullptr;
    ncclResult_t (*comm_init_rank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*all_reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    const char *(*get_error_string)(ncclResult_t) = nullptr;
} g_rccl;
int rccl_fail(ncclResult_t r, const char *what) {
    g_err = std::string(what) + ": " + (g_rccl.get_error_string ? g_rccl.get_error_string(r) : "RCCL error");
    return XGPR_ERR_HIP;
}
}  // namespace

int xgpr_rccl_load(const char *path) {
    if (g_rccl.lib) return 0;
    void *lib = dlopen(path && path[0] ? path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return fail(XGPR_ERR_UNSUPPORTED, "xgpr_rccl_load: could not open the RCCL library");
    g_rccl.get_unique_id = reinterpret_cast<decltype(g_rccl.get_unique_id)>(dlsym(lib, "ncclGetUniqueId"));
    g_rccl.comm_init_rank = reinterpret_cast<decltype(g_rccl.comm_init_rank)>(dlsym(lib, "ncclCommInitRank"));
    g_rccl.all_reduce = reinterpret_cast<decltype(g_rccl.all_reduce)>(dlsym(lib, "ncclAllReduce"));
    g_rccl.comm_destroy = reinterpret_cast<decltype(g_rccl.comm_destroy)>(dlsym(lib, "ncclCommDestroy"));
    g_rccl.get_error_string = reinterpret_cast<decltype(g_rccl.get_error_string)>(dlsym(lib, "ncclGetErrorString"));
    if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.all_reduce || !g_rccl.comm_destroy)
        return fail(XGPR_ERR_UNSUPPORTED, "xgpr_rccl_load: the library does not export the RCCL entry points");
    g_rccl.lib = lib;
    return 0;
}
int xgpr_rccl_unique_id(char *out128) {
    if (!g_rccl.lib) return fail(XGPR_ERR_UNSUPPORTED, "RCCL not loaded (xgpr_rccl_load)");
    ncclUniqueId id;
    ncclResult_t r = g_rccl.get_unique_id(&id);
    if (r != ncclSuccess) return rccl_fail(r, "ncclGetUniqueId");
    memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}
int xgpr_rccl_comm_init(void **comm, int nranks, const char *id128, int rank) {
    if (!g_rccl.lib) return fail(XGPR_ERR_UNSUPPORTED, "RCCL not loaded (xgpr_rccl_load)");
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return fail(XGPR_ERR_ARRAY_DIMS, "incorrect communicator arguments");
    ncclUniqueId id;
    memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t c = nullptr;
    ncclResult_t r = g_rccl.comm_init_rank(&c, nranks, id, rank);
    if (r != ncclSuccess) return rccl_fail(r, "ncclCommInitRank");
    *comm = c;
    return 0;
}
int xgpr_allreduce_sum_f64(void *comm, double *buf, long n, void *stream) {
    if (!g_rccl.lib || !comm) return fail(XGPR_ERR_UNSUPPORTED, "no RCCL communicator (xgpr_rccl_comm_init)");
    if (n <= 0) return 0;
    ncclResult_t r = g_rccl.all_reduce(buf, buf, (size_t)n, ncclFloat64, ncclSum, (ncclComm_t)comm, (hipStream_t)stream);
    if (r != ncclSuccess) return rccl_fail(r, "ncclAllReduce");
    return 0;
}
int xgpr_rccl_comm_destroy(void *comm) {
    if (!g_rccl.lib || !comm) return 0;
    ncclResult_t r = g_rccl.comm_destroy((ncclComm_t)comm);
    if (r != ncclSuccess) return rccl_fail(r, "ncclCommDestroy");
    return 0;
}

}  // extern "C"
