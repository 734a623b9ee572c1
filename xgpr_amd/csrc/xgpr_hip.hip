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
#include <string>

#include "../../include/xgpr_hip.h"

namespace {

// ------------------------------------------------------------------------------------
// host-side helpers
// ------------------------------------------------------------------------------------
thread_local std::string g_err;

int fail(int code, const char *msg) { g_err = msg; return code; }

int hip_fail(hipError_t e, const char *what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return XGPR_ERR_HIP;
}

#define HIP_TRY(expr, what) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return hip_fail(e_, what); } while (0)

// padded SORF width: rbf_ops.cpp:56-59 / rbf_convolution.cpp:60-64
long padded_width(long w) {
    double e = w > 2 ? (double)w : 2.0;
    return (long)pow(2.0, ceil(log2(e)));
}

// the SORF / SRHT normaliser, computed in T exactly as shared_rfgen_ops.cpp:54-55 does
template <typename T> T norm_constant(long dim) {
    T nc = (T)(log2((double)dim) / 2);
    nc = (T)(1 / pow(2.0, (double)nc));
    return nc;
}

int ilog2(long v) { int l = 0; while ((1L << l) < v) l++; return l; }

int device_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0; hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess)
            cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ------------------------------------------------------------------------------------
// device math
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float as_f(int x) { return __builtin_bit_cast(float, x); }
__device__ __forceinline__ int as_i(float x) { return __builtin_bit_cast(int, x); }

// Cephes single-precision sin / cos kernels on the reduced argument r in [-pi/4, pi/4],
// quadrant q (v = q * pi/2 + r).
__device__ __forceinline__ void sincos_poly(float r, int q, float &s, float &c) {
    float r2 = r * r;
    float ps = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2,
                              -1.6666654611e-1f), r2 * r, r);
    float pc = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2,
                              4.166664568298827e-2f), r2 * r2, __builtin_fmaf(-0.5f, r2, 1.0f));
    float ss = (q & 1) ? pc : ps;
    float cc = (q & 1) ? ps : pc;
    s = as_f(as_i(ss) ^ ((q & 2) << 30));
    c = as_f(as_i(cc) ^ (((q + 1) & 2) << 30));
}

// sin and cos of a float argument, <= 1.6 ulp each for |v| < 2^18 (max abs error 9.3e-8,
// measured against double-precision libm over 4e7 arguments): Cody-Waite reduction by pi/2
// in three fmas + the Cephes kernels.  The reference evaluates glibc cosf/sinf (<1 ulp), so
// features agree to ~2 ulp(f32) * scale.
__device__ __forceinline__ void sincos_f32_core(float v, float &s, float &c) {
    float kf = __builtin_rintf(v * 0.6366197466850281f);
    float r = __builtin_fmaf(kf, -1.5707963705062866f, v);
    r = __builtin_fmaf(kf, 4.371138828673793e-08f, r);
    r = __builtin_fmaf(kf, 1.7151245100058819e-15f, r);
    sincos_poly(r, (int)kf, s, c);
}

// Rare path (2^18 <= |v|): the reduction is done in double precision (exact quadrant and a
// reduced argument good to 1e-16 |v| for |v| < 2^31), call-free so that it costs the hot
// kernels no registers.  Beyond 2^31 (un-normalised inputs: |chi * x| > 2e9), inf and nan
// give NaN -- loudly -- where glibc would run Payne-Hanek.
__device__ __forceinline__ void sincos_f32_big(float v, float &s, float &c) {
    double vd = (double)v;
    double kd = __builtin_rint(vd * 0.63661977236758134308);
    double r = __builtin_fma(kd, -1.57079632679489655800e+00, vd);
    r = __builtin_fma(kd, -6.12323399573676603587e-17, r);
    sincos_poly((float)r, (int)((long)kd & 3), s, c);
    if (!(fabsf(v) < 2147483648.0f)) { s = __builtin_nanf(""); c = s; }
}

constexpr float SINCOS_FAST_LIMIT = 262144.0f;

__device__ __forceinline__ void sincos_f32(float v, float &s, float &c) {
    sincos_f32_core(v, s, c);
    if (__builtin_expect(!(fabsf(v) < SINCOS_FAST_LIMIT), 0)) sincos_f32_big(v, s, c);
}

template <typename T> struct Math;
template <> struct Math<float> {
    static __device__ __forceinline__ void sincos(float v, float &s, float &c) { sincos_f32(v, s, c); }
};
template <> struct Math<double> {
    static __device__ __forceinline__ void sincos(double v, double &s, double &c) { s = sin(v); c = cos(v); }
};

// ------------------------------------------------------------------------------------
// generic (any width) path: one workgroup per transform, butterflies in LDS.
// Used for P > 1024, for double precision, for the gradient ops and as the bare FHT / SRHT.
// ------------------------------------------------------------------------------------

// In-place FHT of every aligned P-block of buf[0:len) (len a multiple of P), stages in the
// reference's order h = 1, 2, 4, ...; two stages per barrier (radix-4 pass = stage h then
// stage 2h with identical operand pairing, so every rounding matches the radix-2 chain).
template <typename T>
__device__ __forceinline__ void lds_fht(T *buf, int len, int P, int tid, int nt) {
    int h = 1;
    for (; (h << 1) < P; h <<= 2) {
        for (int idx = tid; idx < (len >> 2); idx += nt) {
            int lo = idx & (h - 1);
            int j = ((idx - lo) << 2) | lo;
            T a = buf[j], b = buf[j + h], c = buf[j + 2 * h], d = buf[j + 3 * h];
            T ab = a + b, amb = a - b, cd = c + d, cmd = c - d;
            buf[j] = ab + cd;
            buf[j + h] = amb + cmd;
            buf[j + 2 * h] = ab - cd;
            buf[j + 3 * h] = amb - cmd;
        }
        __syncthreads();
    }
    if (h < P) {
        for (int idx = tid; idx < (len >> 1); idx += nt) {
            int lo = idx & (h - 1);
            int j = ((idx - lo) << 1) | lo;
            T a = buf[j], b = buf[j + h];
            buf[j] = a + b;
            buf[j + h] = a - b;
        }
        __syncthreads();
    }
}

// bare FHT / SRHT over a flat array of `total` elements made of vectors of length P.
// Each workgroup owns CH consecutive elements (CH a multiple of P, or CH < P when P exceeds
// the LDS capacity, in which case the stages h >= CH are finished by global_stage_kernel).
template <typename T, bool SRHT>
__global__ void generic_fht_kernel(T *x, const int8_t *__restrict__ radem, long total, int P, int CH, T nc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *buf = reinterpret_cast<T *>(smem);
    const int tid = threadIdx.x, nt = blockDim.x;
    const long base = (long)blockIdx.x * CH;
    for (int e = tid; e < CH; e += nt) {
        long idx = base + e;
        T val = idx < total ? x[idx] : (T)0;
        if (SRHT && idx < total) val *= radem[idx & (long)(P - 1)] * nc;
        buf[e] = val;
    }
    __syncthreads();
    lds_fht<T>(buf, CH, P < CH ? P : CH, tid, nt);
    for (int e = tid; e < CH; e += nt) {
        long idx = base + e;
        if (idx < total) x[idx] = buf[e];
    }
}

// SRHT of each row of z[n, m] (zero padded to P) followed by the column sample, out of place:
// out[i, c] = FHT(z_i * radem * nc)[sampler[c]], c < ncols (srht_compressor.py:87-97, where the
// reference pads, transforms the whole chunk in place and then gathers).  A workgroup walks rows
// blockIdx.x, blockIdx.x + gridDim.x, ...; each row is read once and only ncols of the P transformed
// values are written.  With y != nullptr the same read also accumulates the chunk's z^T y
// (rand_nys_constructors.py:115) -- thread t keeps the columns t, t + nt, ... in registers and the
// workgroup's partial sums go to zty_part[blockIdx.x, m] for an ordered reduction.
constexpr int SRHT_ZTY_COLS = 16;      // columns per thread: P <= 16 * blockDim

template <typename T>
__global__ void srht_sample_kernel(const T *__restrict__ z, const int8_t *__restrict__ radem,
                                   const long *__restrict__ sampler, T *out, const double *__restrict__ y,
                                   double *zty_part, long n, long m, int P, long ncols, long ldo, T nc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *buf = reinterpret_cast<T *>(smem);
    const int tid = threadIdx.x, nt = blockDim.x;
    double acc[SRHT_ZTY_COLS];
    #pragma unroll
    for (int q = 0; q < SRHT_ZTY_COLS; q++) acc[q] = 0.0;
    for (long row = blockIdx.x; row < n; row += gridDim.x) {
        const T *zr = z + row * m;
        const double yi = y ? y[row] : 0.0;
        #pragma unroll
        for (int q = 0; q < SRHT_ZTY_COLS; q++) {
            const int e = tid + q * nt;
            if (e < P) {
                const T v = e < m ? zr[e] : (T)0;
                acc[q] = __builtin_fma(yi, (double)v, acc[q]);
                buf[e] = v * (radem[e] * nc);
            }
        }
        __syncthreads();
        lds_fht<T>(buf, P, P, tid, nt);
        T *orow = out + row * ldo;
        for (long c = tid; c < ncols; c += nt) orow[c] = buf[sampler[c]];
        __syncthreads();                 // buf is rewritten for the next row
    }
    if (y) {
        double *slab = zty_part + (long)blockIdx.x * m;
        #pragma unroll
        for (int q = 0; q < SRHT_ZTY_COLS; q++) {
            const int e = tid + q * nt;
            if (e < m) slab[e] = acc[q];
        }
    }
}

// one butterfly stage of stride h straight in global memory (only for P > LDS capacity)
template <typename T>
__global__ void global_stage_kernel(T *x, long npairs, long h) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npairs) return;
    long lo = idx & (h - 1);
    long j = ((idx - lo) << 1) | lo;
    T a = x[j], b = x[j + h];
    x[j] = a + b;
    x[j + h] = a - b;
}

enum { MODE_RBF = 0, MODE_RBF_GRAD = 1, MODE_CONV = 2, MODE_CONV_GRAD = 3, MODE_MAXPOOL = 4 };

template <typename T> struct SorfArgs {
    const T *x; double *out; double *grad; float *outf;
    const int8_t *radem; const T *chi; const int32_t *seqlen;
    long n; long row_stride; long F; long R;
    int d;            // elements copied per transform (input width, or conv_width * C)
    int kmer_stride;  // C for the conv ops
    int conv_width; int P; int reps; int scaling_type;
    T nc; double scale; double sigma;
    T *scratch;       // GLOBALBUF only: one P-element buffer per workgroup, in global memory
};

// one workgroup per (datapoint i = blockIdx.x, repeat k = blockIdx.y); for the conv ops the
// k-mer loop runs inside the workgroup in the reference's order (j ascending), each thread
// owning its output elements, so the f64 accumulation order equals the reference's.
template <typename T, int MODE, bool GLOBALBUF>
__global__ void generic_sorf_kernel(SorfArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // widths beyond the LDS capacity (P > 32768 float / 16384 double) run the same code on a
    // per-workgroup buffer in global memory (__syncthreads orders it at workgroup scope)
    T *buf = GLOBALBUF ? a.scratch + (long)blockIdx.x * a.P : reinterpret_cast<T *>(smem);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int P = a.P;
    constexpr bool CONVLIKE = (MODE == MODE_CONV || MODE == MODE_CONV_GRAD || MODE == MODE_MAXPOOL);
  for (long item = blockIdx.x; item < a.n * a.reps; item += gridDim.x) {
    const long i = item / a.reps;
    const int k = (int)(item % a.reps);
    int nk = 1;
    double rs = a.scale;
    if (CONVLIKE) {
        nk = a.seqlen[i] - a.conv_width + 1;
        if (MODE != MODE_MAXPOOL) {
            if (a.scaling_type == 1) rs = a.scale / sqrt((double)nk);
            else if (a.scaling_type == 2) rs = a.scale / (double)nk;
        }
    }
    const int out0 = k * P;
    long endp = a.F < (long)(k + 1) * P ? a.F : (long)(k + 1) * P;
    const int cnt = (int)(endp - out0);
    const int8_t *re = a.radem + out0;

    for (int j = 0; j < nk; j++) {
        const T *xe = a.x + i * a.row_stride + (long)j * a.kmer_stride;
        for (int e = tid; e < P; e += nt) buf[e] = e < a.d ? xe[e] : (T)0;
        for (int s = 0; s < 3; s++) {
            // same thread touches the same elements as in the load above / the pass below
            for (int e = tid; e < P; e += nt) buf[e] *= re[(long)s * a.R + e] * a.nc;
            __syncthreads();
            lds_fht<T>(buf, P, P, tid, nt);
        }
        for (int e = tid; e < cnt; e += nt) {
            const T chv = a.chi[out0 + e];
            if (MODE == MODE_RBF || MODE == MODE_CONV) {
                T prod = buf[e] * chv;
                T sn, cs;
                Math<T>::sincos(prod, sn, cs);
                double *o = a.out + i * 2 * a.F + 2 * (long)(out0 + e);
                if (MODE == MODE_RBF) {
                    double2 val = make_double2(cs * rs, sn * rs);
                    *reinterpret_cast<double2 *>(o) = val;
                } else {
                    o[0] += cs * rs;
                    o[1] += sn * rs;
                }
            } else if (MODE == MODE_RBF_GRAD || MODE == MODE_CONV_GRAD) {
                // shared_rfgen_ops.cpp:140-155, including the roundings back to T
                T grad_val = buf[e] * chv;
                T prod_val = (T)(grad_val * a.sigma);
                T sn, cs;
                Math<T>::sincos(prod_val, sn, cs);
                T cos_val = (T)(cs * rs);
                T sin_val = (T)(sn * rs);
                double *o = a.out + i * 2 * a.F + 2 * (long)(out0 + e);
                double *g = a.grad + i * 2 * a.F + 2 * (long)(out0 + e);
                T gs = sin_val * grad_val, gc = cos_val * grad_val;
                if (MODE == MODE_RBF_GRAD) {
                    o[0] = cos_val; o[1] = sin_val;
                    g[0] = -(double)gs; g[1] = gc;
                } else {
                    o[0] += cos_val; o[1] += sin_val;
                    g[0] -= gs; g[1] += gc;
                }
            } else {  // MODE_MAXPOOL: conv1d_operations.cpp:160-166
                float prod = (float)(buf[e] * chv);
                float *o = a.outf + i * a.F + out0 + e;
                float old = *o;
                *o = old > prod ? old : prod;
            }
        }
        __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------
// Rademacher sign masks: masks[s * MW + g] bit l = (radem[s, 0, 64 g + l] < 0); one 64-bit
// lane mask per (diagonal s, group of 64 frequencies), zero beyond R.  MW is R rounded up to
// a multiple of 1024, over 64 -- so a wave tile's 16 masks per diagonal are always in range.
// The masks are wave-uniform, so the kernels read them with scalar loads and apply them with
// one v_cndmask per element.
// ------------------------------------------------------------------------------------
__global__ void pack_radem_kernel(const int8_t *__restrict__ radem, uint64_t *masks, long R, int MW) {
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (item >= 3L * MW) return;
    const int s = (int)(item / MW);
    const long g = item % MW;
    const long f = g * 64 + lane;
    int8_t val = f < R ? radem[(long)s * R + f] : (int8_t)1;
    uint64_t m = __ballot(val < 0);
    if (lane == 0) masks[item] = m;
}

// ------------------------------------------------------------------------------------
// wave-level FHT (P <= 1024, float)
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void bfly(float &a, float &b) {
    float s = a + b, d = a - b;
    a = s; b = d;
}

// cross-lane butterfly of stride H (< 64) on all 16 registers: lanes with bit H clear get
// x + partner, lanes with bit H set get partner - x, partner = lane ^ H.
template <int H> __device__ __forceinline__ void xstage(float (&v)[16], int lane) {
    if constexpr (H == 1 || H == 2 || H == 8) {
        // partner through a DPP operand (quad_perm / row_ror:8), own term with the sign folded in
        const int sm = (lane & H) ? (int)0x80000000 : 0;
        #pragma unroll
        for (int r = 0; r < 16; r++) {
            int xi = as_i(v[r]);
            int p;
            if constexpr (H == 1) p = __builtin_amdgcn_mov_dpp(xi, 0xB1, 0xf, 0xf, true);       // quad_perm:[1,0,3,2]
            else if constexpr (H == 2) p = __builtin_amdgcn_mov_dpp(xi, 0x4E, 0xf, 0xf, true);  // quad_perm:[2,3,0,1]
            else p = __builtin_amdgcn_mov_dpp(xi, 0x128, 0xf, 0xf, true);                        // row_ror:8
            v[r] = as_f(xi ^ sm) + as_f(p);
        }
    } else if constexpr (H == 4) {
        // lane^4 is row_ror:12 (lane+4) for DPP banks 0,2 and row_ror:4 (lane-4) for banks 1,3:
        // two bank-masked DPP ops write the butterfly directly.  s_nop 1 covers the
        // VALU-write -> DPP-read hazard on both sides (the compiler cannot see into the asm).
        #pragma unroll
        for (int r0 = 0; r0 < 16; r0 += 8) {
            float o0, o1, o2, o3, o4, o5, o6, o7;
            asm volatile(
                "s_nop 1\n\t"
                "v_add_f32_dpp %0, %8, %8 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
                "v_add_f32_dpp %1, %9, %9 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
                "v_add_f32_dpp %2, %10, %10 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
                "v_add_f32_dpp %3, %11, %11 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
                "v_add_f32_dpp %4, %12, %12 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
                "v_add_f32_dpp %5, %13, %13 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
                "v_add_f32_dpp %6, %14, %14 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
                "v_add_f32_dpp %7, %15, %15 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
                "v_sub_f32_dpp %0, %8, %8 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
                "v_sub_f32_dpp %1, %9, %9 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
                "v_sub_f32_dpp %2, %10, %10 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
                "v_sub_f32_dpp %3, %11, %11 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
                "v_sub_f32_dpp %4, %12, %12 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
                "v_sub_f32_dpp %5, %13, %13 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
                "v_sub_f32_dpp %6, %14, %14 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
                "v_sub_f32_dpp %7, %15, %15 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
                "s_nop 1"
                : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "=&v"(o4), "=&v"(o5), "=&v"(o6), "=&v"(o7)
                : "v"(v[r0]), "v"(v[r0 + 1]), "v"(v[r0 + 2]), "v"(v[r0 + 3]), "v"(v[r0 + 4]), "v"(v[r0 + 5]),
                  "v"(v[r0 + 6]), "v"(v[r0 + 7]));
            v[r0] = o0; v[r0 + 1] = o1; v[r0 + 2] = o2; v[r0 + 3] = o3;
            v[r0 + 4] = o4; v[r0 + 5] = o5; v[r0 + 6] = o6; v[r0 + 7] = o7;
        }
    } else {
        // H = 16 / 32: v_permlane16_swap / v_permlane32_swap on a register pair (A, B) puts the
        // two butterfly operands of both registers into the same lanes: swap, add/sub, swap back.
        #pragma unroll
        for (int r = 0; r < 16; r += 2) {
            int a = as_i(v[r]), b = as_i(v[r + 1]);
            if constexpr (H == 16) {
                auto t = __builtin_amdgcn_permlane16_swap(a, b, false, false);
                float s = as_f(t[0]) + as_f(t[1]), d = as_f(t[0]) - as_f(t[1]);
                auto u = __builtin_amdgcn_permlane16_swap(as_i(s), as_i(d), false, false);
                v[r] = as_f(u[0]); v[r + 1] = as_f(u[1]);
            } else {
                auto t = __builtin_amdgcn_permlane32_swap(a, b, false, false);
                float s = as_f(t[0]) + as_f(t[1]), d = as_f(t[0]) - as_f(t[1]);
                auto u = __builtin_amdgcn_permlane32_swap(as_i(s), as_i(d), false, false);
                v[r] = as_f(u[0]); v[r + 1] = as_f(u[1]);
            }
        }
    }
}

// FHT of every length-P transform in the wave tile, stages h = 1, 2, ..., P/2 in order.
template <int LOG2P> __device__ __forceinline__ void wave_fht(float (&v)[16], int lane) {
    constexpr int P = 1 << LOG2P;
    if constexpr (P > 1) xstage<1>(v, lane);
    if constexpr (P > 2) xstage<2>(v, lane);
    if constexpr (P > 4) xstage<4>(v, lane);
    if constexpr (P > 8) xstage<8>(v, lane);
    if constexpr (P > 16) xstage<16>(v, lane);
    if constexpr (P > 32) xstage<32>(v, lane);
    #pragma unroll
    for (int q = 1; q < P / 64; q <<= 1) {
        #pragma unroll
        for (int r = 0; r < 16; r++)
            if (!(r & q)) bfly(v[r], v[r + q]);
    }
}

// Scalar view of the packed sign masks: constant address space, so the (wave-uniform) loads
// are s_load_dwordx16 into SGPRs and each mask is applied with one v_cndmask.
typedef const __attribute__((address_space(4))) uint64_t *cmask_t;

__device__ __forceinline__ cmask_t as_cmask(const uint64_t *p) { return (cmask_t)p; }

// Inside a per-datapoint loop: stops the compiler from hoisting the 48 mask loads out of the
// loop (96 live SGPRs would be spilled to VGPR lanes); re-reading 384 B from the scalar cache
// per datapoint costs no vector-ALU issue slots.
__device__ __forceinline__ cmask_t launder(cmask_t p) {
    asm volatile("" : "+s"(p));
    return p;
}

// three rounds of { x *= radem * norm ; FHT }.  mk points at this tile's first mask of
// diagonal 0; diagonal s is MW masks further on.  For even log2(P) the normaliser is an
// exact power of two and is folded into chi by the caller (exact), so only the sign flip
// remains; for odd log2(P) the rounded constant is applied per round as the reference does.
template <int LOG2P>
__device__ __forceinline__ void wave_sorf(float (&v)[16], cmask_t mk, int MW, float nc, int lane) {
    uint64_t m[16];
    #pragma unroll
    for (int r = 0; r < 16; r++) m[r] = mk[r];
    #pragma unroll
    for (int s = 0; s < 3; s++) {
        #pragma unroll
        for (int r = 0; r < 16; r++) {
            float t = (LOG2P & 1) ? v[r] * nc : v[r];
            v[r] = __builtin_amdgcn_inverse_ballot_w64(m[r]) ? -t : t;
        }
        if (s < 2) {
            // next round's 16 masks: issued here, after this round's were consumed (the empty asm
            // ties the loads to the data), so they arrive under the FHT and at most two rounds of
            // masks are ever live in SGPRs
            asm volatile("" : "+s"(mk) : "v"(v[0]));
            #pragma unroll
            for (int r = 0; r < 16; r++) m[r] = mk[(s + 1) * MW + r];
        }
        wave_fht<LOG2P>(v, lane);
    }
}

// ---- two-layout SORF for 128 <= P <= 1024 ("T path").  The vector pipe is what bounds these
// kernels, and a cross-lane butterfly costs it 3-5x a register-local one (DPP ops issue at half
// rate, v_permlane*_swap at quarter rate).  So the tile alternates between two register layouts
// through a wave-private LDS buffer (the LDS pipe is otherwise idle):
//   layout C ("columns", the tile layout everything else uses): lane = t[5:0], register = t[9:6]
//   layout R ("rows"):                                          lane = t[9:4], register = t[3:0]
// In R strides 1..8 are register-local and strides 16, 32 are quad_perm DPP; in C strides >= 64
// are register-local: 8 of the 10 stages of an FHT-1024 need no cross-lane instruction.  Stage
// order is still h = 1, 2, 4, ..., so the result is bit-identical to wave_sorf (tools/sorf_variants.hip).
// The buffer holds the tile as 64 rows of 16 floats (4 KiB); inside row p the four 16-byte groups
// are XOR-swizzled with (p >> 2) & 3, which makes both the row accesses (ds_read/write_b128, one
// row per lane) and the column accesses (ds_read/write_b32, lane = t[5:0]) bank-conflict-free.
constexpr int TBUF_FLOATS = 64 * 16;

__device__ __forceinline__ int tswz(int e) {       // element t of the tile -> dword index in the buffer
    const int row = e >> 4, col = e & 15;
    return row * 16 + ((col & 12) ^ (((row >> 2) & 3) << 2)) + (col & 3);
}

__device__ __forceinline__ void wave_lds_sync() {
    // the buffer is private to the wave and a wave's LDS operations execute in order: only the
    // compiler has to be kept from reordering across the exchange
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void cols_to_rows(float (&v)[16], float *tb, int lane) {
    #pragma unroll
    for (int r = 0; r < 16; r++) tb[tswz(r * 64 + lane)] = v[r];
    wave_lds_sync();
    const int sz = (lane >> 2) & 3;
    #pragma unroll
    for (int g = 0; g < 4; g++) {
        float4 t = *reinterpret_cast<const float4 *>(tb + lane * 16 + ((g ^ sz) << 2));
        v[4 * g] = t.x; v[4 * g + 1] = t.y; v[4 * g + 2] = t.z; v[4 * g + 3] = t.w;
    }
    wave_lds_sync();
}

__device__ __forceinline__ void rows_to_cols(float (&v)[16], float *tb, int lane) {
    const int sz = (lane >> 2) & 3;
    #pragma unroll
    for (int g = 0; g < 4; g++)
        *reinterpret_cast<float4 *>(tb + lane * 16 + ((g ^ sz) << 2)) =
            make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
    wave_lds_sync();
    #pragma unroll
    for (int r = 0; r < 16; r++) v[r] = tb[tswz(r * 64 + lane)];
    wave_lds_sync();
}

// sign words of a tile in layout R: bit j of sw[s] = sign of element 16 * lane + j of diagonal s,
// i.e. 16-bit slice `lane` of the tile's sixteen 64-bit masks (little endian)
__device__ __forceinline__ void load_sign_words(uint32_t (&sw)[3], const uint64_t *masks, int MW, int b, int lane) {
    const uint16_t *m16 = reinterpret_cast<const uint16_t *>(masks);
    #pragma unroll
    for (int s = 0; s < 3; s++) sw[s] = m16[((long)s * MW + (long)b * 16) * 4 + lane];
}

template <int LOG2P>
__device__ __forceinline__ void wave_sorf_t(float (&v)[16], const uint32_t (&sw)[3], float *tb, float nc, int lane) {
    static_assert(LOG2P >= 7 && LOG2P <= 10, "two-layout SORF is for 128 <= P <= 1024");
    constexpr int P = 1 << LOG2P;
    #pragma unroll
    for (int s = 0; s < 3; s++) {
        cols_to_rows(v, tb, lane);
        #pragma unroll
        for (int j = 0; j < 16; j++) {
            float t = (LOG2P & 1) ? v[j] * nc : v[j];
            v[j] = as_f(as_i(t) ^ (int)((sw[s] << (31 - j)) & 0x80000000u));
        }
        #pragma unroll
        for (int q = 1; q < 16; q <<= 1) {             // strides 1, 2, 4, 8
            #pragma unroll
            for (int j = 0; j < 16; j++)
                if (!(j & q)) bfly(v[j], v[j + q]);
        }
        xstage<1>(v, lane);                             // stride 16 = lane bit 0 in layout R
        xstage<2>(v, lane);                             // stride 32 = lane bit 1
        rows_to_cols(v, tb, lane);
        #pragma unroll
        for (int q = 1; q < P / 64; q <<= 1) {          // strides 64 ... P/2
            #pragma unroll
            for (int r = 0; r < 16; r++)
                if (!(r & q)) bfly(v[r], v[r + q]);
        }
    }
}

// one entry point for the kernels: T path for P >= 128 when a buffer is supplied, register path otherwise
template <int LOG2P, bool TP>
__device__ __forceinline__ void tile_sorf(float (&v)[16], cmask_t mk, const uint32_t (&sw)[3], float *tb, int MW,
                                          float nc, int lane) {
    if constexpr (TP && LOG2P >= 7) wave_sorf_t<LOG2P>(v, sw, tb, nc, lane);
    else wave_sorf<LOG2P>(v, mk, MW, nc, lane);
}

// load one datapoint (or k-mer window) into the wave tile: element (64 r + l) mod P, zero
// padded from d up to P, replicated over the tile's 1024 / P transforms.  The padded lanes
// read element 0 and are zeroed by a select, so there is no branch around the loads.
template <int LOG2P>
__device__ __forceinline__ void wave_load(float (&v)[16], const float *__restrict__ xe, int d, int lane) {
    constexpr int P = 1 << LOG2P;
    if constexpr (P >= 64) {
        constexpr int RP = P / 64;
        #pragma unroll
        for (int r = 0; r < RP; r++) {
            const int e = r * 64 + lane;
            const bool ok = e < d;
            float t = xe[ok ? e : 0];
            v[r] = ok ? t : 0.0f;
        }
        #pragma unroll
        for (int r = RP; r < 16; r++) v[r] = v[r & (RP - 1)];
    } else {
        const int e = lane & (P - 1);
        const bool ok = e < d;
        float t = xe[ok ? e : 0];
        t = ok ? t : 0.0f;
        #pragma unroll
        for (int r = 0; r < 16; r++) v[r] = t;
    }
}

// cos/sin of the 16 arguments of a tile.  The common path is branch-free; the rare large
// arguments (|v| >= 2^18, inf, nan) are fixed up behind ONE wave-level test.
__device__ __forceinline__ void tile_sincos(const float (&arg)[16], float (&sn)[16], float (&cs)[16]) {
    bool big = false;
    #pragma unroll
    for (int r = 0; r < 16; r++) {
        sincos_f32_core(arg[r], sn[r], cs[r]);
        big |= !(fabsf(arg[r]) < SINCOS_FAST_LIMIT);
    }
    if (__builtin_expect(__any(big), 0)) {
        #pragma unroll
        for (int r = 0; r < 16; r++) {
            float s2, c2;
            sincos_f32_big(arg[r], s2, c2);
            const bool b = !(fabsf(arg[r]) < SINCOS_FAST_LIMIT);
            sn[r] = b ? s2 : sn[r];
            cs[r] = b ? c2 : cs[r];
            __builtin_amdgcn_sched_barrier(0);   // cold path: one element at a time, no extra registers
        }
    }
}

struct WaveArgs {
    const float *x; double *out; float *outf;
    const uint64_t *masks; const float *chi; const int32_t *seqlen;
    const double *vec; double *wpart;
    long n; long row_stride; long F;
    int d; int kmer_stride; int conv_width; int scaling_type;
    int MW; int nb;           // masks per diagonal; wave tiles (1024 frequencies) per datapoint
    int G;                    // matvec: datapoints in flight per workgroup
    int fit_intercept;
    float nc; float chi_scale; // per-round normaliser (odd log2 P) / folded normaliser^3 (even)
    double scale;
    double *tpart;            // two-pass matvec: per-(datapoint, tile) partial dots [rows, nb]
    int add_to_slab;          // two-pass matvec: slabs accumulate over row windows
    double *grad; double sigma;   // gradient operators
};

// ---- cudaRBFFeatureGen: one wave per (datapoint, tile); 4 waves per workgroup.  With CACHE the
// kernel writes the float32 (cos, sin) pairs before scaling -- the exact values the float64
// output is the widening of -- into a.outf [n, 2F] (the resident feature cache).
enum { OUT_F64 = 0, OUT_CACHE = 1, OUT_GRAD = 2 };

template <int LOG2P, int OUT>
__global__ __launch_bounds__(256) void wave_rbf_kernel(WaveArgs a) {
    constexpr bool CACHE = OUT == OUT_CACHE;
    const int lane = threadIdx.x & 63;
    const long item = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) + (long)blockIdx.x * 4;
    if (item >= a.n * a.nb) return;
    const long i = item / a.nb;
    const int b = __builtin_amdgcn_readfirstlane((int)(item % a.nb));
    float v[16];
    wave_load<LOG2P>(v, a.x + i * a.row_stride, a.d, lane);
    constexpr bool TP = LOG2P >= 7;
    __shared__ __attribute__((aligned(16))) float tbuf[TP ? 4 * TBUF_FLOATS : 4];
    uint32_t sw[3] = {0, 0, 0};
    if constexpr (TP) load_sign_words(sw, a.masks, a.MW, b, lane);
    tile_sorf<LOG2P, TP>(v, as_cmask(a.masks + (long)b * 16), sw, tbuf + (threadIdx.x >> 6) * (TP ? TBUF_FLOATS : 1),
                         a.MW, a.nc, lane);
    const long f0 = (long)b * 1024 + lane;
    const bool full = f0 - lane + 1024 <= a.F;      // wave-uniform: the whole tile is inside F
    float arg[16], sn[16], cs[16];
    #pragma unroll
    for (int r = 0; r < 16; r++) {
        const long f = f0 + r * 64;
        const float ch = a.chi[(full || f < a.F) ? f : 0];
        arg[r] = v[r] * (ch * a.chi_scale);
    }
    if constexpr (OUT == OUT_GRAD) {
        // cudaRBFGrad: shared_rfgen_ops.cpp:140-155 with its roundings back to float; the input is
        // not pre-multiplied by sigma, and the scale is a double here (rbf_ops.cpp:180-185)
        float gv[16];
        #pragma unroll
        for (int r = 0; r < 16; r++) { gv[r] = arg[r]; arg[r] = (float)(gv[r] * a.sigma); }
        tile_sincos(arg, sn, cs);
        double *orow = a.out + i * 2 * a.F, *grow = a.grad + i * 2 * a.F;
        #pragma unroll
        for (int r = 0; r < 16; r++) {
            const long f = f0 + r * 64;
            if (full || f < a.F) {
                const float cos_val = (float)(cs[r] * a.scale), sin_val = (float)(sn[r] * a.scale);
                const float gs = sin_val * gv[r], gc = cos_val * gv[r];
                *reinterpret_cast<double2 *>(orow + 2 * f) = make_double2(cos_val, sin_val);
                *reinterpret_cast<double2 *>(grow + 2 * f) = make_double2(-(double)gs, gc);
            }
        }
        return;
    }
    tile_sincos(arg, sn, cs);
    if constexpr (CACHE) {
        float *crow = a.outf + i * 2 * a.F;
        #pragma unroll
        for (int r = 0; r < 16; r++) {
            const long f = f0 + r * 64;
            if (full || f < a.F) *reinterpret_cast<float2 *>(crow + 2 * f) = make_float2(cs[r], sn[r]);
        }
        return;
    }
    double *orow = a.out + i * 2 * a.F;
    if (full) {
        #pragma unroll
        for (int r = 0; r < 16; r++)
            *reinterpret_cast<double2 *>(orow + 2 * (f0 + r * 64)) = make_double2(cs[r] * a.scale, sn[r] * a.scale);
    } else {
        #pragma unroll
        for (int r = 0; r < 16; r++) {
            const long f = f0 + r * 64;
            if (f < a.F) *reinterpret_cast<double2 *>(orow + 2 * f) = make_double2(cs[r] * a.scale, sn[r] * a.scale);
        }
    }
}

// ---- cudaConv1dFGen / cudaConv1dMaxpool: one wave per (sequence, tile), k-mers looped inside
// the wave in the reference's order, sums kept in registers, one read-modify-write at the end.
template <int LOG2P, bool MAXPOOL>
__global__ __launch_bounds__(256) void wave_conv_kernel(WaveArgs a) {
    const int lane = threadIdx.x & 63;
    const long item = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) + (long)blockIdx.x * 4;
    if (item >= a.n * a.nb) return;
    const long i = item / a.nb;
    const int b = __builtin_amdgcn_readfirstlane((int)(item % a.nb));
    const int nk = a.seqlen[i] - a.conv_width + 1;
    const long f0 = (long)b * 1024 + lane;
    float ch[16];
    #pragma unroll
    for (int r = 0; r < 16; r++) {
        const long f = f0 + r * 64;
        ch[r] = f < a.F ? a.chi[f] * a.chi_scale : 0.0f;
    }
    cmask_t mk = as_cmask(a.masks + (long)b * 16);
    constexpr bool TP = LOG2P >= 7;
    __shared__ __attribute__((aligned(16))) float tbuf[TP ? 4 * TBUF_FLOATS : 4];
    float *tb = tbuf + (threadIdx.x >> 6) * (TP ? TBUF_FLOATS : 1);
    uint32_t sw[3] = {0, 0, 0};
    if constexpr (TP) load_sign_words(sw, a.masks, a.MW, b, lane);
    const float *xrow = a.x + i * a.row_stride;
    if constexpr (MAXPOOL) {
        float acc[16];
        float *orow = a.outf + i * a.F;
        #pragma unroll
        for (int r = 0; r < 16; r++) {
            const long f = f0 + r * 64;
            acc[r] = f < a.F ? orow[f] : 0.0f;
        }
        for (int j = 0; j < nk; j++) {
            mk = launder(mk);
            float v[16];
            wave_load<LOG2P>(v, xrow + (long)j * a.kmer_stride, a.d, lane);
            tile_sorf<LOG2P, TP>(v, mk, sw, tb, a.MW, a.nc, lane);
            #pragma unroll
            for (int r = 0; r < 16; r++) {
                float prod = v[r] * ch[r];
                acc[r] = acc[r] > prod ? acc[r] : prod;
            }
        }
        #pragma unroll
        for (int r = 0; r < 16; r++) {
            const long f = f0 + r * 64;
            if (f < a.F) orow[f] = acc[r];
        }
    } else if (a.grad) {
        // cudaConvGrad: per k-mer the roundings of shared_rfgen_ops.cpp:140-155 (values rounded back
        // to float before they are accumulated), sums over k-mers in float64 in the reference's order
        double rs = a.scale;
        if (a.scaling_type == 1) rs = a.scale / sqrt((double)nk);
        else if (a.scaling_type == 2) rs = a.scale / (double)nk;
        double oc[16], os[16], gc[16], gs[16];
        #pragma unroll
        for (int r = 0; r < 16; r++) { oc[r] = 0.0; os[r] = 0.0; gc[r] = 0.0; gs[r] = 0.0; }
        for (int j = 0; j < nk; j++) {
            mk = launder(mk);
            float v[16], gv[16], sn[16], cs[16];
            wave_load<LOG2P>(v, xrow + (long)j * a.kmer_stride, a.d, lane);
            tile_sorf<LOG2P, TP>(v, mk, sw, tb, a.MW, a.nc, lane);
            #pragma unroll
            for (int r = 0; r < 16; r++) { gv[r] = v[r] * ch[r]; v[r] = (float)(gv[r] * a.sigma); }
            tile_sincos(v, sn, cs);
            #pragma unroll
            for (int r = 0; r < 16; r++) {
                const float cos_val = (float)(cs[r] * rs), sin_val = (float)(sn[r] * rs);
                oc[r] += cos_val;
                os[r] += sin_val;
                gc[r] -= sin_val * gv[r];
                gs[r] += cos_val * gv[r];
            }
        }
        double *orow = a.out + i * 2 * a.F, *grow = a.grad + i * 2 * a.F;
        #pragma unroll
        for (int r = 0; r < 16; r++) {
            const long f = f0 + r * 64;
            if (f < a.F) {
                double2 *o = reinterpret_cast<double2 *>(orow + 2 * f), *g = reinterpret_cast<double2 *>(grow + 2 * f);
                double2 ov = *o, gvv = *g;
                ov.x += oc[r]; ov.y += os[r];
                gvv.x += gc[r]; gvv.y += gs[r];
                *o = ov; *g = gvv;
            }
        }
    } else {
        double ac[16], as[16];
        #pragma unroll
        for (int r = 0; r < 16; r++) { ac[r] = 0.0; as[r] = 0.0; }
        for (int j = 0; j < nk; j++) {
            mk = launder(mk);
            float v[16], sn[16], cs[16];
            wave_load<LOG2P>(v, xrow + (long)j * a.kmer_stride, a.d, lane);
            tile_sorf<LOG2P, TP>(v, mk, sw, tb, a.MW, a.nc, lane);
            #pragma unroll
            for (int r = 0; r < 16; r++) v[r] *= ch[r];
            tile_sincos(v, sn, cs);
            #pragma unroll
            for (int r = 0; r < 16; r++) {
                ac[r] += (double)cs[r];
                as[r] += (double)sn[r];
            }
        }
        double rs = a.scale;
        if (a.scaling_type == 1) rs = a.scale / sqrt((double)nk);
        else if (a.scaling_type == 2) rs = a.scale / (double)nk;
        double *orow = a.out + i * 2 * a.F;
        #pragma unroll
        for (int r = 0; r < 16; r++) {
            const long f = f0 + r * 64;
            if (f < a.F) {
                double2 *o = reinterpret_cast<double2 *>(orow + 2 * f);
                double2 old = *o;
                old.x += ac[r] * rs;
                old.y += as[r] * rs;
                *o = old;
            }
        }
    }
}

// sum of a double over the 64 lanes, returned wave-uniform.  Rows of 16 lanes are reduced with
// DPP moves (row_shr 8/4/2/1 on the two dwords + v_add_f64: short VALU chains, no LDS round
// trips); the four row totals are read out of lanes 15, 31, 47, 63 and added in a fixed order.
__device__ __forceinline__ double dpp_shr_f64(double x, const int ctrl_sel) {
    int lo = __builtin_bit_cast(int2, x).x, hi = __builtin_bit_cast(int2, x).y;
    int lo2, hi2;
    if (ctrl_sel == 8) { lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x118, 0xf, 0xf, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x118, 0xf, 0xf, true); }
    else if (ctrl_sel == 4) { lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x114, 0xf, 0xf, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x114, 0xf, 0xf, true); }
    else if (ctrl_sel == 2) { lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x112, 0xf, 0xf, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x112, 0xf, 0xf, true); }
    else { lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, true); }
    int2 r; r.x = lo2; r.y = hi2;
    return __builtin_bit_cast(double, r);
}

__device__ __forceinline__ double readlane_f64(double x, int l) {
    int2 t = __builtin_bit_cast(int2, x);
    int2 r;
    r.x = __builtin_amdgcn_readlane(t.x, l);
    r.y = __builtin_amdgcn_readlane(t.y, l);
    return __builtin_bit_cast(double, r);
}

__device__ __forceinline__ double wave_sum(double u) {
    // row_shr:n gives lane i the value of lane i-n (0.0 shifted in): after the four steps lane 15
    // of every row holds the row total
    u += dpp_shr_f64(u, 8);
    u += dpp_shr_f64(u, 4);
    u += dpp_shr_f64(u, 2);
    u += dpp_shr_f64(u, 1);
    return (readlane_f64(u, 15) + readlane_f64(u, 31)) + (readlane_f64(u, 47) + readlane_f64(u, 63));
}

// ---- fused CG matvec / z^T y.  A workgroup holds G datapoints in flight; datapoint slot g is
// served by nb waves, wave (g, b) owning tile b = frequencies [1024 b, 1024 b + 1024): its f64
// accumulators stay in registers for the whole launch while the workgroup strides over its
// datapoints; the vector v (MATVEC) sits in LDS as (cos, sin) pairs, shared by the G slots.
// Per datapoint: SORF -> cos/sin (registers) -> partial dot with v -> nb partials meet in LDS
// (one barrier, double-buffered) -> rank-1 update of the accumulators.  Z is never written.
// At the end every slot writes its accumulators as one slab wpart[slot, :];
// reduce_slabs_kernel adds the slabs in slot order (deterministic).
template <int LOG2P, bool MATVEC, bool TPREQ>
__global__ __launch_bounds__(512, 2) void wave_ztz_kernel(WaveArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool TP = TPREQ && LOG2P >= 7;
    double2 *pv = reinterpret_cast<double2 *>(smem);                       // [nb * 1024] (cos, sin) of v
    const size_t pv_bytes = MATVEC ? (size_t)a.nb * 1024 * 16 : 0;
    double *part = reinterpret_cast<double *>(smem + pv_bytes);            // [2][16]
    float *tb = reinterpret_cast<float *>(smem + pv_bytes + 256) + (threadIdx.x >> 6) * (TP ? TBUF_FLOATS : 0);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int b, g;
    long slot, nslots;
    if (MATVEC) {                      // workgroup = G slots x nb tiles, coupled through LDS
        b = w % a.nb; g = w / a.nb;
        slot = (long)blockIdx.x * a.G + g;
        nslots = (long)gridDim.x * a.G;
    } else {                           // independent waves: flat (slot, tile) numbering over the grid
        const long gw = (long)blockIdx.x * (blockDim.x >> 6) + w;
        b = __builtin_amdgcn_readfirstlane((int)(gw % a.nb)); g = 0;
        slot = gw / a.nb;
        nslots = ((long)gridDim.x * (blockDim.x >> 6)) / a.nb;
    }
    const long iters = (a.n + nslots - 1) / nslots;
    const long f0 = (long)b * 1024 + lane;
    cmask_t mk = as_cmask(a.masks + (long)b * 16);
    uint32_t sw[3] = {0, 0, 0};
    if constexpr (TP) load_sign_words(sw, a.masks, a.MW, b, lane);

    if (MATVEC) {
        for (long f = threadIdx.x; f < (long)a.nb * 1024; f += blockDim.x)
            pv[f] = f < a.F ? *reinterpret_cast<const double2 *>(a.vec + 2 * f) : make_double2(0.0, 0.0);
        __syncthreads();
    }
    float ch[16];
    double ac[16], as[16];
    #pragma unroll
    for (int r = 0; r < 16; r++) {
        const long f = f0 + r * 64;
        ch[r] = f < a.F ? a.chi[f] * a.chi_scale : 0.0f;
        ac[r] = 0.0; as[r] = 0.0;
    }
    // Z[:, 0] = 1 (kernel_baseclass.py:296-297): feature (f = 0, cos) is 1 / scale before scaling
    const bool icpt = a.fit_intercept && b == 0 && lane == 0;
    const double inv_scale = 1.0 / a.scale;
    const double s2 = a.scale * a.scale;
    const double2 *pw = pv + (long)b * 1024 + lane;

    float v[16];
    if (slot < a.n) wave_load<LOG2P>(v, a.x + slot * a.row_stride, a.d, lane);
    for (long it = 0; it < iters; it++) {
        const long row = it * nslots + slot;
        const bool active = row < a.n;
        float cs[16], sn[16];
        if constexpr (!TP) mk = launder(mk);
        if (active) {
            tile_sorf<LOG2P, TP>(v, mk, sw, tb, a.MW, a.nc, lane);
            #pragma unroll
            for (int r = 0; r < 16; r++) v[r] *= ch[r];
            tile_sincos(v, sn, cs);
        } else {
            #pragma unroll
            for (int r = 0; r < 16; r++) { cs[r] = 0.0f; sn[r] = 0.0f; }
        }
        // next datapoint's x: issued before the f64 work so its latency is covered
        const long nrow = row + nslots;
        if (nrow < a.n) wave_load<LOG2P>(v, a.x + nrow * a.row_stride, a.d, lane);
        const double c0 = (icpt && active) ? inv_scale : (double)cs[0];
        double u;
        if (MATVEC) {
            double u0 = 0.0, u1 = 0.0;
            {
                const double2 p = pw[0];
                u0 = __builtin_fma(c0, p.x, u0);
                u1 = __builtin_fma((double)sn[0], p.y, u1);
            }
            #pragma unroll
            for (int r = 1; r < 16; r++) {
                const double2 p = pw[r * 64];
                u0 = __builtin_fma((double)cs[r], p.x, u0);
                u1 = __builtin_fma((double)sn[r], p.y, u1);
            }
            u = wave_sum(u0 + u1);
            if (lane == 0) part[(it & 1) * 16 + w] = u;
            __syncthreads();
            double t = 0.0;
            for (int bb = 0; bb < a.nb; bb++) t += part[(it & 1) * 16 + g * a.nb + bb];
            u = t * s2;
        } else if (a.tpart) {          // second pass of the two-pass matvec: t = sum of the tile partials
            double t = 0.0;
            if (active)
                for (int bb = 0; bb < a.nb; bb++) t += a.tpart[row * a.nb + bb];
            u = t * s2;
        } else {
            u = active ? a.vec[row] * a.scale : 0.0;   // y[row] * scale
        }
        if (active) {
            ac[0] = __builtin_fma(c0, u, ac[0]);
            as[0] = __builtin_fma((double)sn[0], u, as[0]);
            #pragma unroll
            for (int r = 1; r < 16; r++) {
                ac[r] = __builtin_fma((double)cs[r], u, ac[r]);
                as[r] = __builtin_fma((double)sn[r], u, as[r]);
            }
        }
    }
    double *slab = a.wpart + slot * 2 * a.F;
    #pragma unroll
    for (int r = 0; r < 16; r++) {
        const long f = f0 + r * 64;
        if (f < a.F) {
            double2 *o = reinterpret_cast<double2 *>(slab + 2 * f);
            double2 val = make_double2(ac[r], as[r]);
            if (a.add_to_slab) { const double2 old = *o; val.x += old.x; val.y += old.y; }
            *o = val;
        }
    }
}

// ---- first pass of the two-pass matvec (num_freqs > 8192, where one workgroup cannot hold a
// datapoint's tiles): independent waves, tile b fixed per wave with its slice of v in registers;
// tpart[row, b] = partial dot of tile b of datapoint `row` with v (unscaled cos/sin, as in the
// fused kernel).  The second pass is wave_ztz_kernel<.., false, ..> with a.tpart set.
template <int LOG2P, bool TPREQ>
__global__ __launch_bounds__(256) void wave_dot_kernel(WaveArgs a) {
    constexpr bool TP = TPREQ && LOG2P >= 7;
    __shared__ __attribute__((aligned(16))) float tbuf[TP ? 4 * TBUF_FLOATS : 4];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long gw = (long)blockIdx.x * 4 + w;
    const int b = __builtin_amdgcn_readfirstlane((int)(gw % a.nb));
    const long slot = gw / a.nb;
    const long nslots = ((long)gridDim.x * 4) / a.nb;
    const long f0 = (long)b * 1024 + lane;
    cmask_t mk = as_cmask(a.masks + (long)b * 16);
    float *tb = tbuf + w * (TP ? TBUF_FLOATS : 1);
    uint32_t sw[3] = {0, 0, 0};
    if constexpr (TP) load_sign_words(sw, a.masks, a.MW, b, lane);
    float ch[16];
    double pc[16], ps[16];
    #pragma unroll
    for (int r = 0; r < 16; r++) {
        const long f = f0 + r * 64;
        const bool ok = f < a.F;
        ch[r] = ok ? a.chi[f] * a.chi_scale : 0.0f;
        double2 p = ok ? *reinterpret_cast<const double2 *>(a.vec + 2 * f) : make_double2(0.0, 0.0);
        pc[r] = p.x; ps[r] = p.y;
    }
    const bool icpt = a.fit_intercept && b == 0 && lane == 0;
    const double inv_scale = 1.0 / a.scale;
    for (long row = slot; row < a.n; row += nslots) {
        float v[16], cs[16], sn[16];
        if constexpr (!TP) mk = launder(mk);
        wave_load<LOG2P>(v, a.x + row * a.row_stride, a.d, lane);
        tile_sorf<LOG2P, TP>(v, mk, sw, tb, a.MW, a.nc, lane);
        #pragma unroll
        for (int r = 0; r < 16; r++) v[r] *= ch[r];
        tile_sincos(v, sn, cs);
        const double c0 = icpt ? inv_scale : (double)cs[0];
        double u0 = __builtin_fma(c0, pc[0], 0.0), u1 = __builtin_fma((double)sn[0], ps[0], 0.0);
        #pragma unroll
        for (int r = 1; r < 16; r++) {
            u0 = __builtin_fma((double)cs[r], pc[r], u0);
            u1 = __builtin_fma((double)sn[r], ps[r], u1);
        }
        const double u = wave_sum(u0 + u1);
        if (lane == 0) a.tpart[row * a.nb + b] = u;
    }
}

// ---- CG matvec over a resident float32 feature cache: w = sum_i z_i (z_i . v) with z_i read from
// HBM instead of regenerated.  288 GB of HBM hold the cache of a whole shard (32 KB per datapoint at
// M = 8192), and streaming it is faster than regenerating the features on the vector pipe.  Same
// ownership as wave_ztz_kernel (wave b of a datapoint slot owns tile b, float64 accumulators in
// registers, v in LDS, one barrier per datapoint, slabs reduced in order).  The kernel is bound by
// bytes in flight (HBM latency under load is several microseconds), so every wave keeps a ring of
// RING datapoints in registers: RING - 1 of them loading while one is consumed.  The lane <->
// frequency map is chosen for 16-byte loads: lane l, load q holds frequencies
// f = 1024 b + 128 q + 2 l and f + 1 as (cos, sin, cos, sin).
struct ZcArgs {
    const float *zc; const double *vec; double *wpart;
    long n; long F; int nb; int G; int fit_intercept; double scale;
};

template <bool VEC4, int RING>
__global__ __launch_bounds__(512, 2) void zcache_ztz_kernel(ZcArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double2 *pv = reinterpret_cast<double2 *>(smem);                              // [nb * 1024] (cos, sin) of v
    double *part = reinterpret_cast<double *>(smem + (size_t)a.nb * 1024 * 16);   // [2][G][8], zero padded
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = w % a.nb, g = w / a.nb;
    const long slot = (long)blockIdx.x * a.G + g;
    const long nslots = (long)gridDim.x * a.G;
    const long iters = (a.n + nslots - 1) / nslots;
    for (long f = threadIdx.x; f < (long)a.nb * 1024; f += blockDim.x)
        pv[f] = f < a.F ? *reinterpret_cast<const double2 *>(a.vec + 2 * f) : make_double2(0.0, 0.0);
    if (threadIdx.x < 2 * 8 * 8) part[threadIdx.x] = 0.0;
    __syncthreads();
    const long fb = (long)b * 1024 + 2 * lane;           // first frequency of load q is fb + 128 q
    double ac[32];
    #pragma unroll
    for (int j = 0; j < 32; j++) ac[j] = 0.0;
    const bool icpt = a.fit_intercept && b == 0 && lane == 0;
    const double inv_scale = 1.0 / a.scale;
    const double s2 = a.scale * a.scale;

    auto load_row = [&](long row, float4 (&dst)[8]) {
        const float *zr = a.zc + row * 2 * a.F;
        #pragma unroll
        for (int q = 0; q < 8; q++) {
            const long f = fb + 128 * q;
            if (VEC4) {
                dst[q] = f + 1 < a.F ? *reinterpret_cast<const float4 *>(zr + 2 * f) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                float2 lo = f < a.F ? *reinterpret_cast<const float2 *>(zr + 2 * f) : make_float2(0.f, 0.f);
                float2 hi = f + 1 < a.F ? *reinterpret_cast<const float2 *>(zr + 2 * f + 2) : make_float2(0.f, 0.f);
                dst[q] = make_float4(lo.x, lo.y, hi.x, hi.y);
            }
        }
    };
    float4 buf[RING][8];
    #pragma unroll
    for (int k = 0; k < RING; k++) {
        #pragma unroll
        for (int q = 0; q < 8; q++) buf[k][q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k * nslots + slot < a.n) load_row(k * nslots + slot, buf[k]);
    }
    for (long it0 = 0; it0 < iters; it0 += RING) {
        #pragma unroll
        for (int k = 0; k < RING; k++) {
            const long it = it0 + k;
            if (it >= iters) break;                      // uniform over the workgroup
            const long row = it * nslots + slot;
            const bool active = row < a.n;
            double u0 = 0.0, u1 = 0.0;
            #pragma unroll
            for (int q = 0; q < 8; q++) {
                const double2 p0 = pv[fb + 128 * q], p1 = pv[fb + 128 * q + 1];
                const double c0 = (icpt && q == 0) ? inv_scale : (double)buf[k][q].x;
                u0 = __builtin_fma(c0, p0.x, u0);
                u1 = __builtin_fma((double)buf[k][q].y, p0.y, u1);
                u0 = __builtin_fma((double)buf[k][q].z, p1.x, u0);
                u1 = __builtin_fma((double)buf[k][q].w, p1.y, u1);
            }
            const double u = wave_sum(u0 + u1);
            double *pp = part + ((it & 1) * 8 + g) * 8;
            if (lane == 0) pp[b] = active ? u : 0.0;
            __syncthreads();
            const double2 t01 = *reinterpret_cast<const double2 *>(pp), t23 = *reinterpret_cast<const double2 *>(pp + 2);
            const double2 t45 = *reinterpret_cast<const double2 *>(pp + 4), t67 = *reinterpret_cast<const double2 *>(pp + 6);
            const double us = (((t01.x + t01.y) + (t23.x + t23.y)) + ((t45.x + t45.y) + (t67.x + t67.y))) * s2;
            if (active) {
                #pragma unroll
                for (int q = 0; q < 8; q++) {
                    const double c0 = (icpt && q == 0) ? inv_scale : (double)buf[k][q].x;
                    ac[4 * q] = __builtin_fma(c0, us, ac[4 * q]);
                    ac[4 * q + 1] = __builtin_fma((double)buf[k][q].y, us, ac[4 * q + 1]);
                    ac[4 * q + 2] = __builtin_fma((double)buf[k][q].z, us, ac[4 * q + 2]);
                    ac[4 * q + 3] = __builtin_fma((double)buf[k][q].w, us, ac[4 * q + 3]);
                }
            }
            const long nrow = (it + RING) * nslots + slot;
            if (nrow < a.n) load_row(nrow, buf[k]);      // refill this ring entry
        }
    }
    double *slab = a.wpart + slot * 2 * a.F;
    #pragma unroll
    for (int q = 0; q < 8; q++) {
        const long f = fb + 128 * q;
        if (f < a.F) *reinterpret_cast<double2 *>(slab + 2 * f) = make_double2(ac[4 * q], ac[4 * q + 1]);
        if (f + 1 < a.F) *reinterpret_cast<double2 *>(slab + 2 * f + 2) = make_double2(ac[4 * q + 2], ac[4 * q + 3]);
    }
}

// w_out[m] = sum over slabs (fixed order) of wpart[slab, m]; 64 columns x 4 slab phases per
// workgroup, each thread 8 independent partial sums to keep loads in flight.
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const double *__restrict__ wpart, double *w_out, long M,
                                                           long nslabs) {
    __shared__ double red[4][64];
    const int col = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const long m = (long)blockIdx.x * 64 + col;
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (m < M) {
        long k = ph;
        for (; k + 28 < nslabs; k += 32) {
            #pragma unroll
            for (int q = 0; q < 8; q++) s[q] += wpart[(k + 4 * q) * M + m];
        }
        double tail = 0.0;
        for (; k < nslabs; k += 4) tail += wpart[k * M + m];
        s[0] += tail;
    }
    red[ph][col] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (ph == 0 && m < M) w_out[m] = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
}

// ------------------------------------------------------------------------------------
// Block matvec over the resident feature cache for k right-hand sides (the approximate-NMLL
// probes: k = nsamples + 1 = 26, xgp_regression.py:338-367; any batched solve):
//     W[M, k] = s^2 * Zc^T (Zc V),  Zc = the float32 cache (exact in float64), V, W float64.
// This is the reference's `Z.T @ (Z @ vec)` (cg_tools.py:41-44) as two dense contractions, and it
// is the matrix-core part of the CG path: v_mfma_f64_16x16x4_f64 with float64 accumulation.
//   zblock_t_kernel:  T[n, KP]  = Zc V         (contract features; a wave owns 64 datapoints)
//   zblock_w_kernel:  slab[M, KP] = Zc^T T     (contract datapoints; a wave owns 64 features)
// Operand maps (guide §3): A lane l -> A[i = l & 15][k = l >> 4], B lane l -> B[k = l >> 4][j = l & 15],
// D register r -> D[i = (l >> 4) + 4 r][j = l & 15].  The contraction index of one instruction is
// only a label, so a lane loads 16 bytes (4 consecutive features) and spends one element per
// instruction: instruction e of a group contracts features {16 q + 4 (l >> 4) + e} (T kernel) or
// owns output features {f0 + 4 (l & 15) + e} (W kernel); both are bijections undone at the store.
// ------------------------------------------------------------------------------------
typedef double double4_t __attribute__((ext_vector_type(4)));

struct ZbArgs {
    const float *zc; const double *V; double *T; double *wpart;
    long n; long M; int k; int fit_intercept; double scale;
    long rows_per_range;
    long ldt;            // row pitch of T in doubles: 16 CT inside the fused matvec, k for a caller's [n, k] array
    double tscale;       // factor applied to T at the store (1 inside the fused matvec, scale for the projection)
};

constexpr int ZB_FEATS = 32;      // features per LDS chunk of V (T kernel): a lane's two loads per row tile cover whole 128-B lines
                                  // (with 16 the other half of each line was fetched from HBM a second time: PMC traffic 2.0x)
constexpr int ZB_ROWS = 32;       // datapoints per LDS chunk of T (W kernel)
constexpr int ZB_RING = 4;        // chunks of the streamed operand per wave: ZB_RING - 1 in flight, one consumed
constexpr int ZB_WFEATS = 512;    // features per workgroup of the W kernel (8 waves x 64)

// Both kernels: 8 waves per workgroup (the CU places a 4-wave workgroup's waves on two SIMDs, which
// halves the matrix-pipe rate: tools/mfma_probe.hip), loads are branch-free (clamped addresses; what a
// clamped load returns is multiplied by a zero from the other operand or never stored) so the
// compiler can count vmcnt, and every wave keeps a ring of ZB_RING chunks of its streamed operand in
// registers (all but one in flight) because HBM latency under load is several microseconds.
__device__ __forceinline__ double elem(const float4 &v, int e) {
    return (double)(e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w);
}

template <int CT, int ZB_RT>
__global__ __launch_bounds__(512, ZB_RT == 1 ? 2 : 1) void zblock_t_kernel(ZbArgs a) {
    constexpr int KP = 16 * CT;
    constexpr int VS = KP + 4;                     // +32 B per row: lane groups g and g+1 land 128 B apart
    constexpr int QG = ZB_FEATS / 16;              // groups of 16 features per chunk
    constexpr int VPT = (ZB_FEATS * KP + 511) / 512;   // V elements staged per thread per chunk
    __shared__ __attribute__((aligned(16))) double vs[2][ZB_FEATS * VS];
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long rbase = (long)blockIdx.x * (128 * ZB_RT) + 16 * ZB_RT * w;
    const float *rp[ZB_RT];
    #pragma unroll
    for (int rt = 0; rt < ZB_RT; rt++) {
        long row = rbase + 16 * rt + c;
        if (row >= a.n) row = a.n - 1;             // clamped rows are computed and never stored
        rp[rt] = a.zc + row * a.M + 4 * g;
    }
    double4_t acc[ZB_RT][CT];
    #pragma unroll
    for (int rt = 0; rt < ZB_RT; rt++)
        #pragma unroll
        for (int ct = 0; ct < CT; ct++) acc[rt][ct] = (double4_t){0.0, 0.0, 0.0, 0.0};
    const long nchunks = (a.M + ZB_FEATS - 1) / ZB_FEATS;
    const double inv_scale = 1.0 / a.scale;
    const long fmax = a.M - 4 - 4 * g;             // last float4 of a row, relative to rp

    auto load_a = [&](long ch, float4 (&dst)[QG][ZB_RT]) {
        #pragma unroll
        for (int q = 0; q < QG; q++) {
            long f = ch * ZB_FEATS + 16 * q;
            if (f > fmax) f = fmax;                // past the last feature: V supplies the zeros
            #pragma unroll
            for (int rt = 0; rt < ZB_RT; rt++) dst[q][rt] = *reinterpret_cast<const float4 *>(rp[rt] + f);
        }
    };
    auto load_v = [&](long ch, double (&dst)[VPT]) {
        #pragma unroll
        for (int e = 0; e < VPT; e++) {
            const int idx = threadIdx.x + 512 * e;
            const long f = ch * ZB_FEATS + idx / KP;
            const int col = idx % KP;
            const bool ok = idx < ZB_FEATS * KP && f < a.M && col < a.k;
            const double v = a.V[ok ? f * a.k + col : 0];
            dst[e] = ok ? v : 0.0;
        }
    };
    auto store_v = [&](int buf, const double (&src)[VPT]) {
        #pragma unroll
        for (int e = 0; e < VPT; e++) {
            const int idx = threadIdx.x + 512 * e;
            if (idx < ZB_FEATS * KP) vs[buf][(idx / KP) * VS + idx % KP] = src[e];
        }
    };
    auto compute = [&](long ch, const float4 (&cur)[QG][ZB_RT]) {
        const bool first = a.fit_intercept && ch == 0 && g == 0;
        const double *vb = vs[ch & 1];
        #pragma unroll
        for (int q = 0; q < QG; q++) {
            #pragma unroll
            for (int e = 0; e < 4; e++) {
                double b[CT];
                #pragma unroll
                for (int ct = 0; ct < CT; ct++) b[ct] = vb[(16 * q + 4 * g + e) * VS + 16 * ct + c];
                #pragma unroll
                for (int rt = 0; rt < ZB_RT; rt++) {
                    double av = elem(cur[q][rt], e);
                    if (q == 0 && e == 0 && first) av = inv_scale;     // Z[:, 0] = 1 (kernel_baseclass.py:296-297)
                    #pragma unroll
                    for (int ct = 0; ct < CT; ct++)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b[ct], acc[rt][ct], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);     // keep the conversions of later groups out of this one's registers
            }
        }
    };

    static_assert(ZB_RING == 4, "the stage loop below is unrolled for a ring of 4");
    float4 ring[ZB_RING][QG][ZB_RT];
    double vreg[ZB_RING][VPT];
    #pragma unroll
    for (int i = 0; i < ZB_RING - 1; i++) { load_v(i, vreg[i]); load_a(i, ring[i]); }
    store_v(0, vreg[0]);
    __syncthreads();
#define ZB_STAGE(S, CH)                                                                                       \
    {                                                                                                         \
        load_v((CH) + 3, vreg[((S) + 3) % 4]); load_a((CH) + 3, ring[((S) + 3) % 4]);                         \
        compute((CH), ring[(S)]);                                                                             \
        store_v((int)(((CH) + 1) & 1), vreg[((S) + 1) % 4]);                                                  \
        __syncthreads();                                                                                      \
    }
    for (long ch = 0; ch < nchunks; ch += 4) {
        ZB_STAGE(0, ch)                // chunks past the last one multiply clamped loads by zeros
        ZB_STAGE(1, ch + 1)
        ZB_STAGE(2, ch + 2)
        ZB_STAGE(3, ch + 3)
    }
#undef ZB_STAGE
    #pragma unroll
    for (int rt = 0; rt < ZB_RT; rt++)
        #pragma unroll
        for (int r = 0; r < 4; r++) {
            const long row = rbase + 16 * rt + g + 4 * r;
            if (row < a.n) {
                #pragma unroll
                for (int ct = 0; ct < CT; ct++)
                    if (16 * ct + c < a.ldt) a.T[row * a.ldt + 16 * ct + c] = acc[rt][ct][r] * a.tscale;
            }
        }
}

template <int CT>
__global__ __launch_bounds__(512, 1) void zblock_w_kernel(ZbArgs a) {
    constexpr int KP = 16 * CT;
    constexpr int TS = KP + 4;
    constexpr int STEPS = ZB_ROWS / 4;
    constexpr int TPT = (ZB_ROWS * KP + 511) / 512;  // T elements staged per thread per chunk
    __shared__ __attribute__((aligned(16))) double ts[2][ZB_ROWS * TS];
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long fw = (long)blockIdx.x * ZB_WFEATS + 64 * w;
    long f0 = fw + 4 * c;                            // this lane's 4 features (the A row index)
    const bool icpt = a.fit_intercept && f0 == 0;
    if (f0 > a.M - 4) f0 = a.M - 4;                  // clamped features are computed and never stored
    const long rbeg = (long)blockIdx.y * a.rows_per_range;
    long rend = rbeg + a.rows_per_range;
    if (rend > a.n) rend = a.n;
    const long nchunks = (rend - rbeg + ZB_ROWS - 1) / ZB_ROWS;      // >= 1 by construction of the grid
    const double inv_scale = 1.0 / a.scale;
    double4_t acc[4][CT];
    #pragma unroll
    for (int e = 0; e < 4; e++)
        #pragma unroll
        for (int ct = 0; ct < CT; ct++) acc[e][ct] = (double4_t){0.0, 0.0, 0.0, 0.0};
    const float *zf = a.zc + f0;

    auto load_a = [&](long ch, float4 (&dst)[STEPS]) {
        #pragma unroll
        for (int s = 0; s < STEPS; s++) {
            long row = rbeg + ch * ZB_ROWS + 4 * s + g;
            if (row > rend - 1) row = rend - 1;      // past the range: T supplies the zeros
            dst[s] = *reinterpret_cast<const float4 *>(zf + row * a.M);
        }
    };
    auto load_t = [&](long ch, double (&dst)[TPT]) {
        #pragma unroll
        for (int e = 0; e < TPT; e++) {
            const int idx = threadIdx.x + 512 * e;
            const long row = rbeg + ch * ZB_ROWS + idx / KP;
            const bool ok = idx < ZB_ROWS * KP && row < rend && idx % KP < a.ldt;
            const double v = a.T[ok ? row * a.ldt + idx % KP : 0];
            dst[e] = ok ? v : 0.0;
        }
    };
    auto store_t = [&](int buf, const double (&src)[TPT]) {
        #pragma unroll
        for (int e = 0; e < TPT; e++) {
            const int idx = threadIdx.x + 512 * e;
            if (idx < ZB_ROWS * KP) ts[buf][(idx / KP) * TS + idx % KP] = src[e];
        }
    };
    auto compute = [&](long ch, const float4 (&cur)[STEPS]) {
        const double *tb = ts[ch & 1];
        #pragma unroll
        for (int s = 0; s < STEPS; s++) {
            double b[CT];
            #pragma unroll
            for (int ct = 0; ct < CT; ct++) b[ct] = tb[(4 * s + g) * TS + 16 * ct + c];
            #pragma unroll
            for (int e = 0; e < 4; e++) {
                double av = elem(cur[s], e);
                if (e == 0 && icpt) av = inv_scale;
                #pragma unroll
                for (int ct = 0; ct < CT; ct++)
                    acc[e][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b[ct], acc[e][ct], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);         // keep the conversions of later steps out of this one's registers
        }
    };

    static_assert(ZB_RING == 4, "the stage loop below is unrolled for a ring of 4");
    float4 ring[ZB_RING][STEPS];
    double treg[ZB_RING][TPT];
    #pragma unroll
    for (int i = 0; i < ZB_RING - 1; i++) { load_t(i, treg[i]); load_a(i, ring[i]); }
    store_t(0, treg[0]);
    __syncthreads();
#define ZB_STAGE(S, CH)                                                                                       \
    {                                                                                                         \
        load_t((CH) + 3, treg[((S) + 3) % 4]); load_a((CH) + 3, ring[((S) + 3) % 4]);                         \
        compute((CH), ring[(S)]);                                                                             \
        store_t((int)(((CH) + 1) & 1), treg[((S) + 1) % 4]);                                                  \
        __syncthreads();                                                                                      \
    }
    for (long ch = 0; ch < nchunks; ch += 4) {
        ZB_STAGE(0, ch)                // chunks past the last one multiply clamped loads by zeros
        ZB_STAGE(1, ch + 1)
        ZB_STAGE(2, ch + 2)
        ZB_STAGE(3, ch + 3)
    }
#undef ZB_STAGE
    // D register r of instruction e holds output feature fw + 4 (g + 4 r) + e, column 16 ct + c.
    double *slab = a.wpart + (long)blockIdx.y * a.M * KP;
    #pragma unroll
    for (int e = 0; e < 4; e++)
        #pragma unroll
        for (int r = 0; r < 4; r++) {
            const long f = fw + 4 * (g + 4 * r) + e;
            if (f < a.M) {
                #pragma unroll
                for (int ct = 0; ct < CT; ct++) slab[f * KP + 16 * ct + c] = acc[e][ct][r];
            }
        }
}

// W[f, col] (+)= s2 * sum over row ranges (fixed order) of slab[range, f, col]; KP -> k compaction.
__global__ __launch_bounds__(256) void reduce_block_slabs_kernel(const double *__restrict__ wpart, double *w_out, long M,
                                                                 int KP, int k, long nslabs, double s2, int accumulate) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * KP) return;
    const long f = idx / KP;
    const int col = (int)(idx % KP);
    if (col >= k) return;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    long r = 0;
    for (; r + 3 < nslabs; r += 4) {
        #pragma unroll
        for (int q = 0; q < 4; q++) s[q] += wpart[(r + q) * M * KP + idx];
    }
    double tail = 0.0;
    for (; r < nslabs; r++) tail += wpart[r * M * KP + idx];
    const double v = (((s[0] + s[1]) + (s[2] + s[3])) + tail) * s2;
    w_out[f * k + col] = accumulate ? w_out[f * k + col] + v : v;
}

// ------------------------------------------------------------------------------------
// MiniARD features + gradient (cudaMiniARDGrad; rbf_ops/ard_ops.cpp:39-124, ard_ops.cu): a dense
// projection with a precomputed [num_freqs, d] weight matrix, the product x[k] w[j,k] grouped by
// lengthscale.  Not part of the SORF path -- it completes the operator surface.  Workgroup =
// 64 frequencies x 4 datapoints; weights and inputs go through LDS in 64-wide slices of d so that
// the global reads are coalesced along d; the per-(datapoint, frequency) sums run in k order, as in
// the reference (x * w in T, everything after in float64).
// ------------------------------------------------------------------------------------
constexpr int ARD_MAX_GROUPS = 8;

template <typename T>
__global__ __launch_bounds__(256) void mini_ard_grad_kernel(const T *__restrict__ x, double *out,
                                                            const T *__restrict__ weights,
                                                            const int32_t *__restrict__ sigma_map,
                                                            const double *__restrict__ sigma_vals, double *grad,
                                                            long n, long d, long F, int nl, double norm) {
    __shared__ T ws[64][65];
    __shared__ T xs[4][64];
    __shared__ int32_t ms[64];
    __shared__ double ss[64];
    const int fj = threadIdx.x & 63, ri = threadIdx.x >> 6;
    const long j = (long)blockIdx.x * 64 + fj;
    const long i = (long)blockIdx.y * 4 + ri;
    double g[ARD_MAX_GROUPS];
    #pragma unroll
    for (int l = 0; l < ARD_MAX_GROUPS; l++) g[l] = 0.0;
    double rf = 0.0;
    for (long k0 = 0; k0 < d; k0 += 64) {
        // stage: lane fj walks d (coalesced), 16 weight rows per pass of the 4 thread rows
        #pragma unroll
        for (int p = 0; p < 16; p++) {
            const int jr = 4 * p + ri;
            const long jj = (long)blockIdx.x * 64 + jr;
            ws[jr][fj] = (jj < F && k0 + fj < d) ? weights[jj * d + k0 + fj] : (T)0;
        }
        xs[ri][fj] = (i < n && k0 + fj < d) ? x[i * d + k0 + fj] : (T)0;
        if (ri == 0) {
            ms[fj] = k0 + fj < d ? sigma_map[k0 + fj] : 0;
            ss[fj] = k0 + fj < d ? sigma_vals[k0 + fj] : 0.0;
        }
        __syncthreads();
        const int kmax = (int)(d - k0 < 64 ? d - k0 : 64);
        for (int k = 0; k < kmax; k++) {
            const double dot = (double)(xs[ri][k] * ws[fj][k]);
            const int grp = ms[k];
            #pragma unroll
            for (int l = 0; l < ARD_MAX_GROUPS; l++) g[l] += (grp == l) ? dot : 0.0;
            rf += ss[k] * dot;
        }
        __syncthreads();
    }
    if (i >= n || j >= F) return;
    double sn, cs;
    sincos(rf, &sn, &cs);
    cs *= norm;
    sn *= norm;
    out[i * 2 * F + 2 * j] = cs;
    out[i * 2 * F + 2 * j + 1] = sn;
    double *gp = grad + (i * 2 * F + 2 * j) * nl;
    #pragma unroll
    for (int l = 0; l < ARD_MAX_GROUPS; l++) {
        if (l < nl) {
            gp[l] = -g[l] * sn;
            gp[l + nl] = g[l] * cs;
        }
    }
}

// ------------------------------------------------------------------------------------
// CG vector updates for one right-hand side (fitting_toolkit/cg_tools.py:255-274), fused into
// two single-workgroup kernels: M is only 10^3..10^5, so one workgroup reduces and updates the
// whole vector in a few microseconds, deterministically, instead of ~20 library launches.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum(double v, double *red /* [16] */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v = wave_sum(v);
    __syncthreads();               // red may still be read from a previous call
    if (lane == 0) red[w] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < nw; i++) t += red[i];
    return t;
}

// step 1 (cg_tools.py:256-265): w += lambda^2 p ; alpha = (r.z)/(p.w) ; x += alpha p ;
// r_next = r - alpha w ; err = |r| / |r_0|  (the lagging error, from the CURRENT residual).
// scal[0] = r.z, scal[1] = alpha, scal[2] = err.
__global__ __launch_bounds__(1024) void cg_step1_kernel(double *w, const double *__restrict__ p, double *x,
                                                        const double *__restrict__ r, double *r_next,
                                                        const double *__restrict__ z, double *scal, double lam2,
                                                        double init_norm, long M) {
    __shared__ double red[16];
    double rz = 0.0, pw = 0.0, rr = 0.0;
    for (long i = threadIdx.x; i < M; i += blockDim.x) {
        const double pi = p[i], ri = r[i];
        const double wi = w[i] + lam2 * pi;
        w[i] = wi;
        rz += ri * z[i];
        pw += pi * wi;
        rr += ri * ri;
    }
    rz = block_sum(rz, red);
    pw = block_sum(pw, red);
    rr = block_sum(rr, red);
    const double alpha = rz / pw;
    for (long i = threadIdx.x; i < M; i += blockDim.x) {
        x[i] += alpha * p[i];
        r_next[i] = r[i] - alpha * w[i];
    }
    if (threadIdx.x == 0) { scal[0] = rz; scal[1] = alpha; scal[2] = sqrt(rr) / init_norm; }
}

// step 2 (cg_tools.py:271-274): beta = (r_next.z_next)/(r.z) ; p_next = z_next + beta p.  scal[3] = beta.
__global__ __launch_bounds__(1024) void cg_step2_kernel(const double *__restrict__ r_next, const double *__restrict__ z_next,
                                                        const double *__restrict__ p, double *p_next, double *scal, long M) {
    __shared__ double red[16];
    double rz = 0.0;
    for (long i = threadIdx.x; i < M; i += blockDim.x) rz += r_next[i] * z_next[i];
    rz = block_sum(rz, red);
    const double beta = rz / scal[0];
    for (long i = threadIdx.x; i < M; i += blockDim.x) p_next[i] = z_next[i] + beta * p[i];
    if (threadIdx.x == 0) scal[3] = beta;
}

// ------------------------------------------------------------------------------------
// Preconditioner apply for one right-hand side (rand_nys_preconditioners.py:66-72):
//   z = U (inv_eig * prefactor .* U^T r) + (r - U U^T r) = r + U ((inv_eig * prefactor - 1) .* (U^T r)),
// U [M, rank] float64 row-major.  Two HBM/MALL-bound passes over U (33.5 MB at M = 8192, rank = 512):
// (1) per-row-block partial column sums, (2) slab reduce (shared with the matvec), (3) one wave
// per row: z_i = r_i + U_i . s.  The library GEMV these replace ran at 0.24 ms per product.
// ------------------------------------------------------------------------------------
constexpr int PRE_BLOCKS = 256;

__global__ __launch_bounds__(256) void precond_utr_kernel(const double *__restrict__ u, const double *__restrict__ r,
                                                          double *part, long M, long rank) {
    const long rows_per = (M + gridDim.x - 1) / gridDim.x;
    const long i0 = (long)blockIdx.x * rows_per;
    const long i1 = i0 + rows_per < M ? i0 + rows_per : M;
    for (long j = threadIdx.x; j < rank; j += blockDim.x) {
        double acc = 0.0;
        for (long i = i0; i < i1; i++) acc = __builtin_fma(u[i * rank + j], r[i], acc);
        part[(long)blockIdx.x * rank + j] = acc;
    }
}

__global__ __launch_bounds__(256) void precond_uz_kernel(const double *__restrict__ u, const double *__restrict__ t,
                                                         const double *__restrict__ inv_eig, double prefactor,
                                                         const double *__restrict__ r, double *z, long M, long rank) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const double *ur = u + row * rank;
    double acc = 0.0;
    for (long j = lane; j < rank; j += 64) acc = __builtin_fma(ur[j], (inv_eig[j] * prefactor - 1.0) * t[j], acc);
    acc = wave_sum(acc);
    if (lane == 0) z[row] = r[row] + acc;
}

// ---- self test of the cross-lane stages: v[r] = lane + 64 r, one stage of stride H,
// registers 0 and 1 written out.  Expected: bit H of lane clear -> 2 lane + H + 128 r, set -> -H.
__global__ void selftest_kernel(int32_t *out) {
    const int lane = threadIdx.x & 63;
    float v[16];
    #pragma unroll
    for (int q = 0; q < 6; q++) {
        #pragma unroll
        for (int r = 0; r < 16; r++) v[r] = (float)(lane + 64 * r);
        if (q == 0) xstage<1>(v, lane);
        else if (q == 1) xstage<2>(v, lane);
        else if (q == 2) xstage<4>(v, lane);
        else if (q == 3) xstage<8>(v, lane);
        else if (q == 4) xstage<16>(v, lane);
        else xstage<32>(v, lane);
        #pragma unroll
        for (int r = 0; r < 16; r++) out[(q * 16 + r) * 64 + lane] = (int32_t)v[r];
    }
}

// ------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------
constexpr long LDS_CAP_BYTES = 128 * 1024;   // per-workgroup LDS the generic path will ask for

template <typename T> long lds_cap_elems() { return LDS_CAP_BYTES / (long)sizeof(T); }

template <typename K> int allow_big_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    }
    return 0;
}

int threads_for(long P) { return P >= 4096 ? 1024 : (P >= 512 ? 256 : 64); }

template <typename T, bool SRHT>
int launch_fht(T *x, const int8_t *radem, long nvec, long P, hipStream_t st) {
    const long total = nvec * P;
    const long cap = lds_cap_elems<T>();
    long CH;
    if (P <= cap) {
        CH = P >= 1024 ? P : 1024;   // several short vectors per workgroup
    } else {
        CH = cap;
    }
    const long nblocks = (total + CH - 1) / CH;
    const size_t lds = (size_t)CH * sizeof(T);
    auto kern = generic_fht_kernel<T, SRHT>;
    int rc = allow_big_lds(kern, lds);
    if (rc) return rc;
    const T nc = SRHT ? norm_constant<T>(P) : (T)1;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(threads_for(CH)), lds, st, x, radem, total, (int)P, (int)CH, nc);
    HIP_TRY(hipGetLastError(), "generic_fht_kernel launch");
    for (long h = CH; h < P; h <<= 1) {
        const long npairs = total / 2;
        hipLaunchKernelGGL(global_stage_kernel<T>, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, st, x, npairs, h);
        HIP_TRY(hipGetLastError(), "global_stage_kernel launch");
    }
    return 0;
}

constexpr long GENERIC_SCRATCH_SLOTS = 256;

// bytes of global scratch the generic path needs for padded width P (0 when it fits in LDS)
size_t generic_scratch_bytes(long P, size_t elem) {
    return (size_t)P * elem > (size_t)LDS_CAP_BYTES ? (size_t)GENERIC_SCRATCH_SLOTS * P * elem : 0;
}

template <typename T, int MODE>
int launch_generic_sorf(SorfArgs<T> a, void *workspace, size_t wbytes, hipStream_t st) {
    long items = a.n * a.reps;
    if ((long)a.P > lds_cap_elems<T>()) {
        const size_t need = generic_scratch_bytes(a.P, sizeof(T));
        if (!workspace || wbytes < need || !aligned16(workspace))
            return fail(XGPR_ERR_WORKSPACE, "workspace too small (see xgpr_sorf_workspace_bytes)");
        a.scratch = reinterpret_cast<T *>(workspace);
        const long nblocks = items < GENERIC_SCRATCH_SLOTS ? items : GENERIC_SCRATCH_SLOTS;
        hipLaunchKernelGGL((generic_sorf_kernel<T, MODE, true>), dim3((unsigned)nblocks), dim3(1024), 0, st, a);
        HIP_TRY(hipGetLastError(), "generic_sorf_kernel launch");
        return 0;
    }
    const size_t lds = (size_t)a.P * sizeof(T);
    auto kern = generic_sorf_kernel<T, MODE, false>;
    int rc = allow_big_lds(kern, lds);
    if (rc) return rc;
    const long nblocks = items < (1L << 20) ? items : (1L << 20);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(threads_for(a.P)), lds, st, a);
    HIP_TRY(hipGetLastError(), "generic_sorf_kernel launch");
    return 0;
}

int pack_masks(const int8_t *radem, uint64_t *masks, long R, int MW, hipStream_t st) {
    const long items = 3L * MW;
    hipLaunchKernelGGL(pack_radem_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, radem, masks, R, MW);
    HIP_TRY(hipGetLastError(), "pack_radem_kernel launch");
    return 0;
}

int masks_per_diag(long R) { return (int)(align_up((size_t)R, 1024) / 64); }
size_t masks_bytes(long R) { return align_up((size_t)3 * masks_per_diag(R) * sizeof(uint64_t), 256); }

#define DISPATCH_LOG2P(lg, CALL)                                                 \
    switch (lg) {                                                                \
        case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break;  \
        case 4: CALL(4); break; case 5: CALL(5); break; case 6: CALL(6); break;  \
        case 7: CALL(7); break; case 8: CALL(8); break; case 9: CALL(9); break;  \
        case 10: CALL(10); break;                                                \
        default: return fail(XGPR_ERR_UNSUPPORTED, "padded width > 1024 on the wave path"); \
    }

void fill_norms(WaveArgs &a, int lg) {
    const float nc = norm_constant<float>(1L << lg);
    if (lg & 1) { a.nc = nc; a.chi_scale = 1.0f; }
    else { a.nc = 1.0f; a.chi_scale = nc * nc * nc; }   // exact power of two
}

// ------------------------------------------------------------------------------------
// validation shared by the entry points (mirrors the reference's throw sites)
// ------------------------------------------------------------------------------------
int check_seqlens(const int32_t *seqlen_host, long nseq, long n, long L, int conv_width) {
    if (nseq != n) return fail(XGPR_ERR_SEQLEN_SIZE, "wrong array sizes");
    if (L < conv_width || conv_width <= 0) return fail(XGPR_ERR_CONV_WIDTH, "invalid conv_width");
    if (!seqlen_host) return fail(XGPR_ERR_SEQLEN_RANGE, "seqlen_host is required (sequence lengths are validated on the host)");
    int32_t mn = 2147483647, mx = 0;
    for (long i = 0; i < nseq; i++) {
        if (seqlen_host[i] > mx) mx = seqlen_host[i];
        if (seqlen_host[i] < mn) mn = seqlen_host[i];
    }
    if (mx > L || mn < conv_width)
        return fail(XGPR_ERR_SEQLEN_RANGE, "All sequence lengths must be >= conv width and < array size.");
    return 0;
}

template <typename T> double rbf_scale(long num_freqs, int fit_intercept) {
    // rbf_ops.cpp:64-69: the constant is rounded to T there
    T s = fit_intercept ? (T)sqrt(1.0 / ((double)num_freqs - 0.5)) : (T)sqrt(1.0 / (double)num_freqs);
    return (double)s;
}

// ------------------------------------------------------------------------------------
// typed implementations behind the C entry points
// ------------------------------------------------------------------------------------
template <typename T>
int fht_impl(T *x, long n, long dim1, long dim2, void *stream) {
    if (n == 0) return fail(XGPR_ERR_NO_DATAPOINTS, "no datapoints");
    if (dim2 < 2) return fail(XGPR_ERR_NOT_POW2, "last dim not power of 2 > 1");
    if ((dim2 & (dim2 - 1)) != 0) return fail(XGPR_ERR_NOT_POW2, "last dim not power of 2");
    if (dim1 < 1) return fail(XGPR_ERR_ARRAY_DIMS, "incorrect array dims passed");
    return launch_fht<T, false>(x, nullptr, n * dim1, dim2, (hipStream_t)stream);
}

template <typename T>
int srht_impl(T *x, const int8_t *radem, long n, long dim, long radem_len, void *stream) {
    if (n == 0) return fail(XGPR_ERR_NO_DATAPOINTS, "no datapoints");
    if (dim != radem_len) return fail(XGPR_ERR_ARRAY_DIMS, "incorrect array dims passed");
    if (dim < 2) return fail(XGPR_ERR_NOT_POW2, "last dim not power of 2 > 1");
    if ((dim & (dim - 1)) != 0) return fail(XGPR_ERR_NOT_POW2, "last dim not power of 2");
    return launch_fht<T, true>(x, radem, n, dim, (hipStream_t)stream);
}

template <typename T>
int rbf_impl(const T *x, double *out, double *grad, const int8_t *radem, const T *chi, long n, long d,
             long out_rows, long num_rffs, long grad_rows, long grad_cols, long num_freqs, long R,
             double sigma, int fit_intercept, bool want_grad, void *workspace, size_t wbytes, void *stream) {
    const long P = padded_width(d);
    if (n == 0 || out_rows != n) return fail(XGPR_ERR_NO_DATAPOINTS, "no datapoints");
    if (num_rffs < 2 || (num_rffs & 1) != 0) return fail(XGPR_ERR_ODD_OUTPUT, "last dim of output must be even number");
    if (2 * num_freqs != num_rffs || num_freqs > R) return fail(XGPR_ERR_RFFS_FREQS, "incorrect number of rffs and or freqs.");
    if (want_grad && (grad_rows != out_rows || grad_cols != num_rffs)) return fail(XGPR_ERR_ARRAY_SIZES, "Wrong array sizes.");
    if (R % P != 0) return fail(XGPR_ERR_RFFS_FREQS, "incorrect number of rffs and or freqs.");
    if (!aligned16(out)) return fail(XGPR_ERR_WORKSPACE, "output pointer must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int reps = (int)((num_freqs + P - 1) / P);

    if constexpr (sizeof(T) == 4) {
        if (P <= 1024) {
            if (!workspace || wbytes < masks_bytes(R)) return fail(XGPR_ERR_WORKSPACE, "workspace too small (see xgpr_rbf_workspace_bytes)");
            WaveArgs a = {};
            a.x = x; a.out = out; a.masks = (const uint64_t *)workspace; a.chi = chi;
            a.n = n; a.row_stride = d; a.F = num_freqs; a.d = (int)d;
            a.MW = masks_per_diag(R); a.nb = (int)((num_freqs + 1023) / 1024);
            a.scale = rbf_scale<float>(num_freqs, fit_intercept);
            if (want_grad) {
                if (!aligned16(grad)) return fail(XGPR_ERR_WORKSPACE, "gradient pointer must be 16-byte aligned");
                a.grad = grad; a.sigma = sigma;
                a.scale = fit_intercept ? sqrt(1.0 / ((double)num_freqs - 0.5)) : sqrt(1.0 / (double)num_freqs);
            }
            const int lg = ilog2(P);
            fill_norms(a, lg);
            int rc = pack_masks(radem, (uint64_t *)workspace, R, a.MW, st);
            if (rc) return rc;
            const long items = n * a.nb;
            const long nblocks = (items + 3) / 4;
            if (nblocks > 2147483647L) return fail(XGPR_ERR_UNSUPPORTED, "too many datapoints for one launch");
            if (want_grad) {
#define CALL_RBFG(LG) hipLaunchKernelGGL((wave_rbf_kernel<LG, OUT_GRAD>), dim3((unsigned)nblocks), dim3(256), 0, st, a)
                DISPATCH_LOG2P(lg, CALL_RBFG)
#undef CALL_RBFG
            } else {
#define CALL_RBF(LG) hipLaunchKernelGGL((wave_rbf_kernel<LG, OUT_F64>), dim3((unsigned)nblocks), dim3(256), 0, st, a)
                DISPATCH_LOG2P(lg, CALL_RBF)
#undef CALL_RBF
            }
            HIP_TRY(hipGetLastError(), "wave_rbf_kernel launch");
            return 0;
        }
    }
    SorfArgs<T> a = {};
    a.x = x; a.out = out; a.grad = grad; a.radem = radem; a.chi = chi;
    a.n = n; a.row_stride = d; a.F = num_freqs; a.R = R; a.d = (int)d;
    a.P = (int)P; a.reps = reps; a.nc = norm_constant<T>(P); a.sigma = sigma;
    if (want_grad) {
        // rbf_ops.cpp:180-185: a double constant in the gradient op
        a.scale = fit_intercept ? sqrt(1.0 / ((double)num_freqs - 0.5)) : sqrt(1.0 / (double)num_freqs);
        return launch_generic_sorf<T, MODE_RBF_GRAD>(a, workspace, wbytes, st);
    }
    a.scale = rbf_scale<T>(num_freqs, fit_intercept);
    return launch_generic_sorf<T, MODE_RBF>(a, workspace, wbytes, st);
}

template <typename T>
int conv_impl(const T *x, double *out, double *grad, float *outf, const int8_t *radem, const T *chi,
              const int32_t *seqlen_host, const int32_t *seqlen_dev, long n, long L, long C, long out_rows,
              long num_rffs, long grad_rows, long grad_cols, long num_freqs, long R, long nseq, double sigma,
              int conv_width, int scaling_type, int mode, void *workspace, size_t wbytes, void *stream) {
    if (n == 0 || out_rows != n) return fail(XGPR_ERR_NO_DATAPOINTS, "no datapoints");
    if (num_rffs < 2 || (num_rffs & 1) != 0) return fail(XGPR_ERR_ODD_OUTPUT, "last dim of output must be even number");
    if (mode == MODE_MAXPOOL) {
        if (num_freqs != num_rffs || num_freqs > R) return fail(XGPR_ERR_RFFS_FREQS, "incorrect number of rffs and or freqs.");
    } else {
        if (2 * num_freqs != num_rffs || num_freqs > R) return fail(XGPR_ERR_RFFS_FREQS, "incorrect number of rffs and or freqs.");
    }
    if (mode == MODE_CONV_GRAD && (grad_rows != out_rows || grad_cols != num_rffs))
        return fail(XGPR_ERR_ARRAY_SIZES, "Wrong array sizes.");
    if (nseq != n) return fail(XGPR_ERR_SEQLEN_SIZE, "wrong array sizes");
    if (L < conv_width || conv_width <= 0) return fail(XGPR_ERR_CONV_WIDTH, "invalid conv_width");
    const long win = (long)conv_width * C;
    const long P = padded_width(win);
    const int reps = (int)((num_freqs + P - 1) / P);
    if (R % P != 0) return fail(XGPR_ERR_RFFS_FREQS, "incorrect number of rffs and or freqs.");
    if (mode == MODE_MAXPOOL && R != (long)reps * P) return fail(XGPR_ERR_RFFS_FREQS, "incorrect number of rffs and or freqs.");
    int rc = check_seqlens(seqlen_host, nseq, n, L, conv_width);
    if (rc) return rc;
    if (!seqlen_dev) return fail(XGPR_ERR_WORKSPACE, "seqlen_dev (device copy of the sequence lengths) is required");
    hipStream_t st = (hipStream_t)stream;

    if constexpr (sizeof(T) == 4) {
        if (P <= 1024) {
            if (mode != MODE_MAXPOOL && !aligned16(out)) return fail(XGPR_ERR_WORKSPACE, "output pointer must be 16-byte aligned");
            if (mode == MODE_CONV_GRAD && !aligned16(grad)) return fail(XGPR_ERR_WORKSPACE, "gradient pointer must be 16-byte aligned");
            if (!workspace || wbytes < masks_bytes(R)) return fail(XGPR_ERR_WORKSPACE, "workspace too small (see xgpr_rbf_workspace_bytes)");
            WaveArgs a = {};
            a.x = x; a.out = out; a.outf = outf; a.masks = (const uint64_t *)workspace; a.chi = chi; a.seqlen = seqlen_dev;
            a.n = n; a.row_stride = L * C; a.F = num_freqs; a.d = (int)win; a.kmer_stride = (int)C;
            a.conv_width = conv_width; a.scaling_type = scaling_type;
            a.MW = masks_per_diag(R); a.nb = (int)((num_freqs + 1023) / 1024);
            a.scale = sqrt(1.0 / (double)num_freqs);
            if (mode == MODE_CONV_GRAD) { a.grad = grad; a.sigma = sigma; }
            const int lg = ilog2(P);
            fill_norms(a, lg);
            rc = pack_masks(radem, (uint64_t *)workspace, R, a.MW, st);
            if (rc) return rc;
            const long nblocks = (n * a.nb + 3) / 4;
            if (nblocks > 2147483647L) return fail(XGPR_ERR_UNSUPPORTED, "too many datapoints for one launch");
            if (mode != MODE_MAXPOOL) {
#define CALL_CONV(LG) hipLaunchKernelGGL((wave_conv_kernel<LG, false>), dim3((unsigned)nblocks), dim3(256), 0, st, a)
                DISPATCH_LOG2P(lg, CALL_CONV)
#undef CALL_CONV
            } else {
#define CALL_MAXP(LG) hipLaunchKernelGGL((wave_conv_kernel<LG, true>), dim3((unsigned)nblocks), dim3(256), 0, st, a)
                DISPATCH_LOG2P(lg, CALL_MAXP)
#undef CALL_MAXP
            }
            HIP_TRY(hipGetLastError(), "wave_conv_kernel launch");
            return 0;
        }
    }
    SorfArgs<T> a = {};
    a.x = x; a.out = out; a.grad = grad; a.outf = outf; a.radem = radem; a.chi = chi; a.seqlen = seqlen_dev;
    a.n = n; a.row_stride = L * C; a.F = num_freqs; a.R = R; a.d = (int)win; a.kmer_stride = (int)C;
    a.conv_width = conv_width; a.P = (int)P; a.reps = reps; a.scaling_type = scaling_type;
    a.nc = norm_constant<T>(P); a.scale = sqrt(1.0 / (double)num_freqs); a.sigma = sigma;
    if (mode == MODE_CONV) return launch_generic_sorf<T, MODE_CONV>(a, workspace, wbytes, st);
    if (mode == MODE_CONV_GRAD) return launch_generic_sorf<T, MODE_CONV_GRAD>(a, workspace, wbytes, st);
    return launch_generic_sorf<T, MODE_MAXPOOL>(a, workspace, wbytes, st);
}

constexpr long ZTZ_MAX_SLABS = 2048;
constexpr long ZTZ_ROW_WINDOW = 65536;     // two-pass matvec: rows per (dot, update) launch pair

size_t ztz_workspace_bytes(long num_rffs, long R) {
    size_t b = masks_bytes(R) + (size_t)ZTZ_MAX_SLABS * num_rffs * sizeof(double);
    if (num_rffs / 2 > 8192) b += (size_t)ZTZ_ROW_WINDOW * ((num_rffs / 2 + 1023) / 1024) * sizeof(double);
    return b;
}

// independent-wave kernels: 4 waves per workgroup, two waves per SIMD over the chip, a whole number
// of datapoint slots (nb waves each), at most ZTZ_MAX_SLABS slots and no more slots than datapoints
void flat_geometry(int nb, long n, int &waves_per_wg, long &nblocks, long &nslots) {
    waves_per_wg = 4;
    long waves = (long)device_cus() * 8;
    nslots = waves / nb > 0 ? waves / nb : 1;
    if (nslots > ZTZ_MAX_SLABS) nslots = ZTZ_MAX_SLABS;
    if (nslots > n) nslots = n;
    // nslots * nb must be a multiple of 4
    while ((nslots * nb) % 4 != 0) nslots++;
    nblocks = nslots * nb / 4;
}

int ztz_two_pass(const float *x, const int8_t *radem, const float *chi, const double *vec, double *w_out, long n,
                 long d, long num_rffs, long num_freqs, long R, int fit_intercept, void *workspace, hipStream_t st);

template <bool MATVEC>
int ztz_impl(const float *x, const int8_t *radem, const float *chi, const double *vec, double *w_out, long n,
             long d, long num_rffs, long num_freqs, long R, int fit_intercept, void *workspace, size_t wbytes,
             void *stream) {
    const long P = padded_width(d);
    if (n == 0) return fail(XGPR_ERR_NO_DATAPOINTS, "no datapoints");
    if (num_rffs < 2 || (num_rffs & 1) != 0) return fail(XGPR_ERR_ODD_OUTPUT, "last dim of output must be even number");
    if (2 * num_freqs != num_rffs || num_freqs > R) return fail(XGPR_ERR_RFFS_FREQS, "incorrect number of rffs and or freqs.");
    if (R % P != 0) return fail(XGPR_ERR_RFFS_FREQS, "incorrect number of rffs and or freqs.");
    if (P > 1024) return fail(XGPR_ERR_UNSUPPORTED, "fused matvec supports padded width <= 1024");
    if (num_freqs > 65536) return fail(XGPR_ERR_UNSUPPORTED, "fused matvec supports num_freqs <= 65536");
    if ((MATVEC && !aligned16(vec)) || !aligned16(w_out)) return fail(XGPR_ERR_WORKSPACE, "vector pointers must be 16-byte aligned");
    const size_t mb = masks_bytes(R);
    const size_t need = ztz_workspace_bytes(num_rffs, R);
    if (!workspace || wbytes < need || !aligned16(workspace))
        return fail(XGPR_ERR_WORKSPACE, "workspace too small (see xgpr_ztz_matvec_workspace_bytes)");
    hipStream_t st = (hipStream_t)stream;
    if (MATVEC && num_freqs > 8192)
        return ztz_two_pass(x, radem, chi, vec, w_out, n, d, num_rffs, num_freqs, R, fit_intercept, workspace, st);

    WaveArgs a = {};
    a.x = x; a.masks = (const uint64_t *)workspace; a.chi = chi; a.vec = vec;
    a.wpart = reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(workspace) + mb);
    a.n = n; a.row_stride = d; a.F = num_freqs; a.d = (int)d;
    a.MW = masks_per_diag(R); a.nb = (int)((num_freqs + 1023) / 1024);
    // 8 waves per workgroup where possible (2 per SIMD, the register budget of this kernel): the
    // G datapoint slots of a workgroup share one copy of v in LDS
    a.G = a.nb >= 8 ? 1 : 8 / a.nb;
    if ((long)a.G > n) a.G = (int)n;
    a.fit_intercept = fit_intercept;
    a.scale = rbf_scale<float>(num_freqs, fit_intercept);
    const int lg = ilog2(P);
    fill_norms(a, lg);
    int rc = pack_masks(radem, (uint64_t *)workspace, R, a.MW, st);
    if (rc) return rc;

    int waves_per_wg = a.nb * a.G;
    int wg_per_cu = 8 / waves_per_wg > 0 ? 8 / waves_per_wg : 1;     // 2 waves per SIMD
    long nblocks = (long)device_cus() * wg_per_cu;
    long nslabs;
    if (MATVEC) {
        const long max_by_rows = (n + a.G - 1) / a.G;
        if (nblocks > max_by_rows) nblocks = max_by_rows;
        if (nblocks * a.G > ZTZ_MAX_SLABS) nblocks = ZTZ_MAX_SLABS / a.G;
        nslabs = nblocks * a.G;
    } else {
        flat_geometry(a.nb, n, waves_per_wg, nblocks, nslabs);
        wg_per_cu = 2;
    }
    const size_t lds_base = (MATVEC ? (size_t)a.nb * 1024 * 16 : 0) + 256;
    const size_t lds_t = lds_base + (size_t)waves_per_wg * TBUF_FLOATS * sizeof(float);
    // the two-layout FHT needs 5 KiB of LDS per wave; without room for it (M = 16384) the register-only FHT runs
    const bool tp = lg >= 7 && lds_t * wg_per_cu <= 160 * 1024;
    const size_t lds = tp ? lds_t : lds_base;
#define CALL_ZTZ(LG)                                                                                        \
    if (tp) {                                                                                               \
        auto kern = wave_ztz_kernel<LG, MATVEC, true>;                                                      \
        int rc2 = allow_big_lds(kern, lds);                                                                 \
        if (rc2) return rc2;                                                                                \
        hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(waves_per_wg * 64), lds, st, a);            \
    } else {                                                                                                \
        auto kern = wave_ztz_kernel<LG, MATVEC, false>;                                                     \
        int rc2 = allow_big_lds(kern, lds);                                                                 \
        if (rc2) return rc2;                                                                                \
        hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(waves_per_wg * 64), lds, st, a);            \
    }
    DISPATCH_LOG2P(lg, CALL_ZTZ)
#undef CALL_ZTZ
    HIP_TRY(hipGetLastError(), "wave_ztz_kernel launch");
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((num_rffs + 63) / 64)), dim3(256), 0, st, a.wpart, w_out,
                       num_rffs, nslabs);
    HIP_TRY(hipGetLastError(), "reduce_slabs_kernel launch");
    return 0;
}

// Z^T(Z v) for num_freqs > 8192: per window of rows, (1) wave_dot_kernel writes the per-tile partial
// dots, (2) the update kernel recomputes the features and accumulates z_i * (z_i . v) into its slabs;
// then the slabs are reduced as in the single-pass path.  Costs one extra SORF + sincos per feature
// but has no coupling between the tiles of a datapoint, so it works for any number of tiles.
int ztz_two_pass(const float *x, const int8_t *radem, const float *chi, const double *vec, double *w_out, long n,
                 long d, long num_rffs, long num_freqs, long R, int fit_intercept, void *workspace, hipStream_t st) {
    const long P = padded_width(d);
    const size_t mb = masks_bytes(R);
    WaveArgs a = {};
    a.masks = (const uint64_t *)workspace; a.chi = chi; a.vec = vec;
    a.wpart = reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(workspace) + mb);
    a.tpart = a.wpart + (size_t)ZTZ_MAX_SLABS * num_rffs;
    a.row_stride = d; a.F = num_freqs; a.d = (int)d;
    a.MW = masks_per_diag(R); a.nb = (int)((num_freqs + 1023) / 1024);
    a.G = 1; a.fit_intercept = fit_intercept;
    a.scale = rbf_scale<float>(num_freqs, fit_intercept);
    const int lg = ilog2(P);
    fill_norms(a, lg);
    int rc = pack_masks(radem, (uint64_t *)workspace, R, a.MW, st);
    if (rc) return rc;
    long nslabs_max = 0;
    for (long row0 = 0; row0 < n; row0 += ZTZ_ROW_WINDOW) {
        const long rows = n - row0 < ZTZ_ROW_WINDOW ? n - row0 : ZTZ_ROW_WINDOW;
        a.x = x + row0 * d;
        a.n = rows;
        a.add_to_slab = row0 > 0;
        int wpw; long nblocks, nslots;
        flat_geometry(a.nb, row0 == 0 ? rows : ZTZ_ROW_WINDOW, wpw, nblocks, nslots);
        if (row0 == 0) nslabs_max = nslots;
        // later (shorter) windows keep the first window's slot count so that every slab is revisited
        if (row0 > 0) { nslots = nslabs_max; nblocks = nslots * a.nb / 4; }
        const bool tp = lg >= 7;
#define CALL_DOT(LG)                                                                                          \
        if (tp) hipLaunchKernelGGL((wave_dot_kernel<LG, true>), dim3((unsigned)nblocks), dim3(256), 0, st, a);   \
        else hipLaunchKernelGGL((wave_dot_kernel<LG, false>), dim3((unsigned)nblocks), dim3(256), 0, st, a);
        DISPATCH_LOG2P(lg, CALL_DOT)
#undef CALL_DOT
        HIP_TRY(hipGetLastError(), "wave_dot_kernel launch");
        const size_t lds = 256 + (tp ? (size_t)4 * TBUF_FLOATS * sizeof(float) : 0);
#define CALL_UPD(LG)                                                                                          \
        if (tp) hipLaunchKernelGGL((wave_ztz_kernel<LG, false, true>), dim3((unsigned)nblocks), dim3(256), lds, st, a);  \
        else hipLaunchKernelGGL((wave_ztz_kernel<LG, false, false>), dim3((unsigned)nblocks), dim3(256), lds, st, a);
        DISPATCH_LOG2P(lg, CALL_UPD)
#undef CALL_UPD
        HIP_TRY(hipGetLastError(), "wave_ztz_kernel (update pass) launch");
    }
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((num_rffs + 63) / 64)), dim3(256), 0, st, a.wpart, w_out,
                       num_rffs, nslabs_max);
    HIP_TRY(hipGetLastError(), "reduce_slabs_kernel launch");
    return 0;
}

int zcache_build_impl(const float *x, float *zc, const int8_t *radem, const float *chi, long n, long d,
                      long num_rffs, long num_freqs, long R, void *workspace, size_t wbytes, void *stream) {
    const long P = padded_width(d);
    if (n == 0) return fail(XGPR_ERR_NO_DATAPOINTS, "no datapoints");
    if (num_rffs < 2 || (num_rffs & 1) != 0) return fail(XGPR_ERR_ODD_OUTPUT, "last dim of output must be even number");
    if (2 * num_freqs != num_rffs || num_freqs > R) return fail(XGPR_ERR_RFFS_FREQS, "incorrect number of rffs and or freqs.");
    if (R % P != 0) return fail(XGPR_ERR_RFFS_FREQS, "incorrect number of rffs and or freqs.");
    if (P > 1024) return fail(XGPR_ERR_UNSUPPORTED, "the feature cache supports padded width <= 1024");
    if (!workspace || wbytes < masks_bytes(R)) return fail(XGPR_ERR_WORKSPACE, "workspace too small (see xgpr_rbf_workspace_bytes)");
    if ((reinterpret_cast<uintptr_t>(zc) & 7) != 0) return fail(XGPR_ERR_WORKSPACE, "cache pointer must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    WaveArgs a = {};
    a.x = x; a.outf = zc; a.masks = (const uint64_t *)workspace; a.chi = chi;
    a.n = n; a.row_stride = d; a.F = num_freqs; a.d = (int)d;
    a.MW = masks_per_diag(R); a.nb = (int)((num_freqs + 1023) / 1024);
    const int lg = ilog2(P);
    fill_norms(a, lg);
    int rc = pack_masks(radem, (uint64_t *)workspace, R, a.MW, st);
    if (rc) return rc;
    const long nblocks = (n * a.nb + 3) / 4;
    if (nblocks > 2147483647L) return fail(XGPR_ERR_UNSUPPORTED, "too many datapoints for one launch");
#define CALL_RBFC(LG) hipLaunchKernelGGL((wave_rbf_kernel<LG, OUT_CACHE>), dim3((unsigned)nblocks), dim3(256), 0, st, a)
    DISPATCH_LOG2P(lg, CALL_RBFC)
#undef CALL_RBFC
    HIP_TRY(hipGetLastError(), "wave_rbf_kernel (cache) launch");
    return 0;
}

int zcache_matvec_impl(const float *zc, const double *vec, double *w_out, long n, long num_rffs, int fit_intercept,
                       double scale_override, void *workspace, size_t wbytes, void *stream) {
    if (n == 0) return fail(XGPR_ERR_NO_DATAPOINTS, "no datapoints");
    if (num_rffs < 2 || (num_rffs & 1) != 0) return fail(XGPR_ERR_ODD_OUTPUT, "last dim of output must be even number");
    const long F = num_rffs / 2;
    if (F > 8192) return fail(XGPR_ERR_UNSUPPORTED, "cached matvec supports num_freqs <= 8192");
    if (!aligned16(vec) || !aligned16(w_out) || !aligned16(zc)) return fail(XGPR_ERR_WORKSPACE, "pointers must be 16-byte aligned");
    const size_t need = (size_t)ZTZ_MAX_SLABS * num_rffs * sizeof(double);
    if (!workspace || wbytes < need || !aligned16(workspace))
        return fail(XGPR_ERR_WORKSPACE, "workspace too small (see xgpr_ztz_matvec_workspace_bytes)");
    hipStream_t st = (hipStream_t)stream;
    ZcArgs a = {};
    a.zc = zc; a.vec = vec; a.wpart = reinterpret_cast<double *>(workspace);
    a.n = n; a.F = F; a.nb = (int)((F + 1023) / 1024);
    a.G = 8 / a.nb;                                // 8 waves per workgroup = 2 per SIMD, deep register rings
    if ((long)a.G > n) a.G = (int)n;
    a.fit_intercept = fit_intercept;
    a.scale = scale_override > 0.0 ? scale_override : rbf_scale<float>(F, fit_intercept);
    const int waves = a.nb * a.G;
    long nblocks = device_cus();
    const long max_by_rows = (n + a.G - 1) / a.G;
    if (nblocks > max_by_rows) nblocks = max_by_rows;
    if (nblocks * a.G > ZTZ_MAX_SLABS) nblocks = ZTZ_MAX_SLABS / a.G;
    constexpr int RING = 2;     // measured: 6.3 TB/s with 2 (222 VGPRs), 6.0 with 3, 3.8 with 4 (spills)
    const size_t lds = (size_t)a.nb * 1024 * 16 + 2 * 8 * 8 * sizeof(double);
    if (F % 2 == 0) {
        auto kern = zcache_ztz_kernel<true, RING>;
        int rc = allow_big_lds(kern, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(waves * 64), lds, st, a);
    } else {
        auto kern = zcache_ztz_kernel<false, RING>;
        int rc = allow_big_lds(kern, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(waves * 64), lds, st, a);
    }
    HIP_TRY(hipGetLastError(), "zcache_ztz_kernel launch");
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((num_rffs + 63) / 64)), dim3(256), 0, st, a.wpart, w_out,
                       num_rffs, nblocks * a.G);
    HIP_TRY(hipGetLastError(), "reduce_slabs_kernel launch");
    return 0;
}

struct ZbGeom { int kp; long mblk; long nrb; long rows_per_range; size_t t_bytes; size_t slab_bytes; };

ZbGeom zb_geometry(long n, long num_rffs, long k) {
    ZbGeom gm;
    gm.kp = k <= 16 ? 16 : 32;
    gm.mblk = (num_rffs + ZB_WFEATS - 1) / ZB_WFEATS;
    long target = (2L * device_cus() + gm.mblk - 1) / gm.mblk;      // row ranges wanted: ~2 workgroups per CU
    const long max_ranges = (n + ZB_ROWS - 1) / ZB_ROWS;
    if (target > max_ranges) target = max_ranges;
    if (target < 1) target = 1;
    gm.rows_per_range = ((n + target - 1) / target + ZB_ROWS - 1) / ZB_ROWS * ZB_ROWS;
    gm.nrb = (n + gm.rows_per_range - 1) / gm.rows_per_range;
    gm.t_bytes = ((size_t)n * gm.kp * sizeof(double) + 255) / 256 * 256;
    gm.slab_bytes = (size_t)gm.nrb * num_rffs * gm.kp * sizeof(double);
    return gm;
}

enum { ZB_MATVEC = 0, ZB_PROJECT = 1, ZB_BACKPROJECT = 2 };

// mode ZB_MATVEC:      out[M, k] (+)= s^2 Zc^T (Zc in),  in = V [M, k]
// mode ZB_PROJECT:     out[n, k]  =  s Zc in,            in = V [M, k]      (T kernel only)
// mode ZB_BACKPROJECT: out[M, k] (+)= s Zc^T in,         in = R [n, k]      (W kernel + reduce)
int zcache_block_impl(int mode, const float *zc, const double *in, double *out, long n, long num_rffs, long k,
                      int fit_intercept, double scale_override, int accumulate, void *workspace, size_t wbytes,
                      void *stream) {
    if (n <= 0) return fail(XGPR_ERR_NO_DATAPOINTS, "no datapoints");
    if (num_rffs < 4 || (num_rffs & 3) != 0) return fail(XGPR_ERR_UNSUPPORTED, "block matvec needs num_rffs to be a multiple of 4");
    if (k < 1 || k > 32) return fail(XGPR_ERR_UNSUPPORTED, "block matvec takes 1..32 right-hand sides per call");
    if (!aligned16(zc)) return fail(XGPR_ERR_WORKSPACE, "cache pointer must be 16-byte aligned");
    const ZbGeom gm = zb_geometry(n, num_rffs, k);
    if (mode != ZB_PROJECT && (!workspace || wbytes < gm.t_bytes + gm.slab_bytes || !aligned16(workspace)))
        return fail(XGPR_ERR_WORKSPACE, "workspace too small (see xgpr_zcache_block_workspace_bytes)");
    hipStream_t st = (hipStream_t)stream;
    ZbArgs a = {};
    a.zc = zc; a.V = in;
    a.n = n; a.M = num_rffs; a.k = (int)k; a.fit_intercept = fit_intercept;
    a.scale = scale_override > 0.0 ? scale_override : rbf_scale<float>(num_rffs / 2, fit_intercept);
    a.rows_per_range = gm.rows_per_range;
    a.ldt = gm.kp; a.tscale = 1.0;
    if (mode == ZB_MATVEC) {
        a.T = reinterpret_cast<double *>(workspace);
    } else if (mode == ZB_PROJECT) {
        a.T = out; a.ldt = k; a.tscale = a.scale;
    } else {
        a.T = const_cast<double *>(in); a.ldt = k;
    }
    if (mode != ZB_PROJECT)
        a.wpart = reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(workspace) + gm.t_bytes);
    // T kernel: 16 RT datapoints per wave, 8 waves; RT = 4 reuses each LDS operand most, smaller RT
    // keeps every CU busy when the shard (or window) is short
    const long cus = device_cus();
    const int rt = n >= 2 * cus * 512 ? 4 : n >= 2 * cus * 256 ? 2 : 1;
    const long tblocks = (n + 128 * rt - 1) / (128 * rt);
    if (tblocks > 2147483647L || gm.nrb > 65535) return fail(XGPR_ERR_UNSUPPORTED, "too many datapoints for one launch");
    const dim3 tg((unsigned)tblocks), wg((unsigned)gm.mblk, (unsigned)gm.nrb), blk(512);
#define ZB_LAUNCH_T(CT, RT) hipLaunchKernelGGL((zblock_t_kernel<CT, RT>), tg, blk, 0, st, a)
    if (mode != ZB_BACKPROJECT) {
        if (gm.kp == 16) {
            if (rt == 4) ZB_LAUNCH_T(1, 4); else if (rt == 2) ZB_LAUNCH_T(1, 2); else ZB_LAUNCH_T(1, 1);
        } else {
            if (rt == 4) ZB_LAUNCH_T(2, 4); else if (rt == 2) ZB_LAUNCH_T(2, 2); else ZB_LAUNCH_T(2, 1);
        }
        HIP_TRY(hipGetLastError(), "zblock_t_kernel launch");
    }
#undef ZB_LAUNCH_T
    if (mode == ZB_PROJECT) return 0;
    if (gm.kp == 16) hipLaunchKernelGGL(zblock_w_kernel<1>, wg, blk, 0, st, a);
    else hipLaunchKernelGGL(zblock_w_kernel<2>, wg, blk, 0, st, a);
    HIP_TRY(hipGetLastError(), "zblock_w_kernel launch");
    const long total = num_rffs * gm.kp;
    hipLaunchKernelGGL(reduce_block_slabs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a.wpart, out,
                       num_rffs, gm.kp, (int)k, gm.nrb, mode == ZB_MATVEC ? a.scale * a.scale : a.scale, accumulate);
    HIP_TRY(hipGetLastError(), "reduce_block_slabs_kernel launch");
    return 0;
}

template <typename T>
int mini_ard_impl(const T *x, double *out, const T *weights, const int32_t *sigma_map, const double *sigma_vals,
                  double *grad, long n, long d, long out_rows, long num_rffs, long num_freqs, long w_cols, long map_len,
                  long sig_len, long grad_rows, long grad_cols, long num_lengthscales, int fit_intercept, void *stream) {
    if (n == 0 || out_rows != n) return fail(XGPR_ERR_NO_DATAPOINTS, "no datapoints");
    if (grad_rows != out_rows || grad_cols != num_rffs) return fail(XGPR_ERR_ARRAY_SIZES, "Wrong array sizes.");
    if (w_cols != d) return fail(XGPR_ERR_ARRAY_SIZES, "Wrong array sizes.");
    if (num_rffs != 2 * num_freqs || map_len != w_cols) return fail(XGPR_ERR_ARRAY_SIZES, "Wrong array sizes.");
    if (sig_len != map_len) return fail(XGPR_ERR_ARRAY_SIZES, "Wrong array sizes.");
    if (num_lengthscales < 1 || num_lengthscales > ARD_MAX_GROUPS)
        return fail(XGPR_ERR_UNSUPPORTED, "MiniARD gradient supports up to 8 lengthscale groups");
    const long yblocks = (n + 3) / 4;
    if (yblocks > 65535) return fail(XGPR_ERR_UNSUPPORTED, "too many datapoints for one launch (chunk the input)");
    // the constant is typed T in the reference (ard_ops.cpp:86-91)
    const double norm = (double)(T)std::sqrt(1.0 / (fit_intercept ? (double)num_freqs - 0.5 : (double)num_freqs));
    hipLaunchKernelGGL(mini_ard_grad_kernel<T>, dim3((unsigned)((num_freqs + 63) / 64), (unsigned)yblocks), dim3(256), 0,
                       (hipStream_t)stream, x, out, weights, sigma_map, sigma_vals, grad, n, d, num_freqs,
                       (int)num_lengthscales, norm);
    HIP_TRY(hipGetLastError(), "mini_ard_grad_kernel launch");
    return 0;
}

constexpr long SRHT_ZTY_MAX_BLOCKS = 512;

template <typename T>
int srht_sample_impl(const T *z, const int8_t *radem, const long *sampler, T *out, const double *y, double *zty_out,
                     long n, long m, long P, long ncols, long ldo, void *workspace, size_t wbytes, void *stream) {
    if (n <= 0 || m <= 0) return fail(XGPR_ERR_ARRAY_DIMS, "incorrect array dims passed");
    if (P < 2 || (P & (P - 1)) != 0 || m > P) return fail(XGPR_ERR_NOT_POW2, "last dim not power of 2 > 1");
    if (ncols < 1 || ncols > P || ldo < ncols) return fail(XGPR_ERR_ARRAY_DIMS, "incorrect array dims passed");
    if (P > lds_cap_elems<T>()) return fail(XGPR_ERR_UNSUPPORTED, "fused SRHT + sample needs the padded row to fit in LDS");
    const int nt = threads_for(P);
    if (P > (long)SRHT_ZTY_COLS * nt) return fail(XGPR_ERR_UNSUPPORTED, "fused SRHT + sample: padded width too large");
    long nblocks = 2L * device_cus();
    if (nblocks > n) nblocks = n;
    if (nblocks > SRHT_ZTY_MAX_BLOCKS) nblocks = SRHT_ZTY_MAX_BLOCKS;
    double *part = nullptr;
    if (y) {
        if (!zty_out) return fail(XGPR_ERR_ARRAY_DIMS, "incorrect array dims passed");
        if (!workspace || wbytes < (size_t)nblocks * m * sizeof(double) || !aligned16(workspace))
            return fail(XGPR_ERR_WORKSPACE, "workspace too small (see xgpr_srht_sample_workspace_bytes)");
        part = reinterpret_cast<double *>(workspace);
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)P * sizeof(T);
    auto kern = srht_sample_kernel<T>;
    int rc = allow_big_lds(kern, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(nt), lds, st, z, radem, sampler, out, y, part, n, m, (int)P, ncols,
                       ldo, norm_constant<T>(P));
    HIP_TRY(hipGetLastError(), "srht_sample_kernel launch");
    if (y) {
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((m + 63) / 64)), dim3(256), 0, st, part, zty_out, m, nblocks);
        HIP_TRY(hipGetLastError(), "reduce_slabs_kernel launch");
    }
    return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------
extern "C" {

const char *xgpr_last_error(void) { return g_err.c_str(); }
const char *xgpr_build_arch(void) { return "gfx950"; }

int xgpr_fht_f32(float *x, long n, long dim1, long dim2, void *stream) { return fht_impl<float>(x, n, dim1, dim2, stream); }
int xgpr_fht_f64(double *x, long n, long dim1, long dim2, void *stream) { return fht_impl<double>(x, n, dim1, dim2, stream); }

int xgpr_srht_f32(float *x, const int8_t *radem, long n, long dim, long radem_len, void *stream) {
    return srht_impl<float>(x, radem, n, dim, radem_len, stream);
}
int xgpr_srht_f64(double *x, const int8_t *radem, long n, long dim, long radem_len, void *stream) {
    return srht_impl<double>(x, radem, n, dim, radem_len, stream);
}

size_t xgpr_rbf_workspace_bytes(long radem_shape2) { return masks_bytes(radem_shape2); }
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

int xgpr_cg_step1_f64(double *w, const double *p, double *x, const double *r, double *r_next, const double *z,
                      double *scal, double lam2, double init_norm, long M, void *stream) {
    if (M <= 0) return fail(XGPR_ERR_ARRAY_DIMS, "incorrect array dims passed");
    hipLaunchKernelGGL(cg_step1_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, w, p, x, r, r_next, z, scal, lam2,
                       init_norm, M);
    HIP_TRY(hipGetLastError(), "cg_step1_kernel launch");
    return 0;
}
int xgpr_cg_step2_f64(const double *r_next, const double *z_next, const double *p, double *p_next, double *scal, long M,
                      void *stream) {
    if (M <= 0) return fail(XGPR_ERR_ARRAY_DIMS, "incorrect array dims passed");
    hipLaunchKernelGGL(cg_step2_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, r_next, z_next, p, p_next, scal, M);
    HIP_TRY(hipGetLastError(), "cg_step2_kernel launch");
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
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((rank + 63) / 64)), dim3(256), 0, st, part, t, rank, (long)nb);
    HIP_TRY(hipGetLastError(), "reduce_slabs_kernel launch");
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
    return gm.t_bytes + gm.slab_bytes;
}
int xgpr_zcache_block_matvec_f32(const float *zc, const double *v, double *w_out, long n, long num_rffs, long k,
                                 int fit_intercept, double scale, int accumulate, void *workspace,
                                 size_t workspace_bytes, void *stream) {
    return zcache_block_impl(ZB_MATVEC, zc, v, w_out, n, num_rffs, k, fit_intercept, scale, accumulate, workspace,
                             workspace_bytes, stream);
}
int xgpr_zcache_block_project_f32(const float *zc, const double *v, double *t_out, long n, long num_rffs, long k,
                                  int fit_intercept, double scale, void *stream) {
    return zcache_block_impl(ZB_PROJECT, zc, v, t_out, n, num_rffs, k, fit_intercept, scale, 0, nullptr, 0, stream);
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

int xgpr_selftest_lane_xor(int32_t *out, void *stream) {
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
    HIP_TRY(hipGetLastError(), "selftest_kernel launch");
    return 0;
}

}  // extern "C"
