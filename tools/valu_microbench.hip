// valu_microbench.hip -- issue cost (cycles per wave-instruction per SIMD) of the VALU / cross-lane
// instructions the wave-level FHT is built from, on gfx950.  One workgroup per CU, W waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_microbench.hip -o tools/valu_microbench && tools/valu_microbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define ITERS 2000

#define REP8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)

template <int T>
__global__ void bench(float *out, unsigned long long *cyc, unsigned long long sm) {
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    float q0 = 1.5f, q1 = 2.5f, q2 = 3.5f, q3 = 4.5f, q4 = 5.5f, q5 = 6.5f, q6 = 7.5f, q7 = 8.5f;
    double d0 = r0, d1 = r1, d2 = r2, d3 = r3;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) {
        if (T == 0) {   // v_add_f32 x32
#define I(k) "v_add_f32 %" #k ", %" #k ", %8\n\t"
            asm volatile(REP8(I) REP8(I) REP8(I) REP8(I)
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(q0));
#undef I
        } else if (T == 1) {  // v_pk_add_f32 on register pairs x32 (each = 2 adds)
            asm volatile(
                "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4\n\t"
                "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4\n\t"
                "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4\n\t"
                "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4\n\t"
                "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4\n\t"
                "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4\n\t"
                "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4\n\t"
                "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4\n\t"
                : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d0 * 0 + 1.0));
        } else if (T == 2) {  // v_add_f32_dpp quad_perm x32
#define I(k) "v_add_f32_dpp %" #k ", %" #k ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            asm volatile(REP8(I) REP8(I) REP8(I) REP8(I)
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(q0));
#undef I
        } else if (T == 3) {  // v_add_f32_dpp row_ror:8 x32
#define I(k) "v_add_f32_dpp %" #k ", %" #k ", %8 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            asm volatile(REP8(I) REP8(I) REP8(I) REP8(I)
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(q0));
#undef I
        } else if (T == 4) {  // v_permlane32_swap x32 (pairs)
            asm volatile(
                "v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                "v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                "v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                "v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                "v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                "v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                "v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                "v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
        } else if (T == 5) {  // v_permlane16_swap x32
            asm volatile(
                "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t"
                "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t"
                "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t"
                "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t"
                "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t"
                "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t"
                "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t"
                "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
        } else if (T == 6) {  // v_cndmask_b32_e64 with an SGPR-pair mask and neg modifier x32
#define I(k) "v_cndmask_b32_e64 %" #k ", %" #k ", -%" #k ", %8\n\t"
            asm volatile(REP8(I) REP8(I) REP8(I) REP8(I)
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "s"(sm));
#undef I
        } else if (T == 7) {  // v_fma_f32 x32
#define I(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n\t"
            asm volatile(REP8(I) REP8(I) REP8(I) REP8(I)
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(q0), "v"(q1));
#undef I
        } else if (T == 8) {  // v_pk_fma_f32 x32 on pairs
            asm volatile(
                "v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5\n\t"
                "v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5\n\t"
                "v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5\n\t"
                "v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5\n\t"
                "v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5\n\t"
                "v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5\n\t"
                "v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5\n\t"
                "v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5\n\t"
                : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d0 * 0 + 1.0), "v"(d1 * 0 + 0.5));
        } else if (T == 9) {  // v_fma_f64 x32
            asm volatile(
                "v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5\n\t"
                "v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5\n\t"
                "v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5\n\t"
                "v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5\n\t"
                "v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5\n\t"
                "v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5\n\t"
                "v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5\n\t"
                "v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5\n\t"
                : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d0 * 0 + 1.0), "v"(d1 * 0 + 0.5));
        } else if (T == 10) {  // v_cvt_f64_f32 x32
            asm volatile(
                "v_cvt_f64_f32 %0, %4\n\tv_cvt_f64_f32 %1, %5\n\tv_cvt_f64_f32 %2, %6\n\tv_cvt_f64_f32 %3, %7\n\t"
                "v_cvt_f64_f32 %0, %4\n\tv_cvt_f64_f32 %1, %5\n\tv_cvt_f64_f32 %2, %6\n\tv_cvt_f64_f32 %3, %7\n\t"
                "v_cvt_f64_f32 %0, %4\n\tv_cvt_f64_f32 %1, %5\n\tv_cvt_f64_f32 %2, %6\n\tv_cvt_f64_f32 %3, %7\n\t"
                "v_cvt_f64_f32 %0, %4\n\tv_cvt_f64_f32 %1, %5\n\tv_cvt_f64_f32 %2, %6\n\tv_cvt_f64_f32 %3, %7\n\t"
                "v_cvt_f64_f32 %0, %4\n\tv_cvt_f64_f32 %1, %5\n\tv_cvt_f64_f32 %2, %6\n\tv_cvt_f64_f32 %3, %7\n\t"
                "v_cvt_f64_f32 %0, %4\n\tv_cvt_f64_f32 %1, %5\n\tv_cvt_f64_f32 %2, %6\n\tv_cvt_f64_f32 %3, %7\n\t"
                "v_cvt_f64_f32 %0, %4\n\tv_cvt_f64_f32 %1, %5\n\tv_cvt_f64_f32 %2, %6\n\tv_cvt_f64_f32 %3, %7\n\t"
                "v_cvt_f64_f32 %0, %4\n\tv_cvt_f64_f32 %1, %5\n\tv_cvt_f64_f32 %2, %6\n\tv_cvt_f64_f32 %3, %7\n\t"
                : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(r0), "v"(r1), "v"(r2), "v"(r3));
        } else if (T == 11) {  // ds_swizzle_b32 xor 16 x32 (LDS crossbar)
#define I(k) "ds_swizzle_b32 %" #k ", %" #k " offset:swizzle(BITMASK_PERM,\"0000p\")\n\t"
            asm volatile(REP8(I) REP8(I) REP8(I) REP8(I) "s_waitcnt lgkmcnt(0)\n\t"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
#undef I
        } else if (T == 12) {  // v_mul_f64 x32
            asm volatile(
                "v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4\n\t"
                "v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4\n\t"
                "v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4\n\t"
                "v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4\n\t"
                "v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4\n\t"
                "v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4\n\t"
                "v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4\n\t"
                "v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4\n\t"
                : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d0 * 0 + 1.0000001));
        } else if (T == 13) {  // v_rndne_f32 / transcendental-class check: v_sin_f32 x32
#define I(k) "v_sin_f32 %" #k ", %" #k "\n\t"
            asm volatile(REP8(I) REP8(I) REP8(I) REP8(I)
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
#undef I
        } else if (T == 14) {  // v_pk_mul_f32 x32
            asm volatile(
                "v_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4\n\t"
                "v_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4\n\t"
                "v_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4\n\t"
                "v_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4\n\t"
                "v_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4\n\t"
                "v_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4\n\t"
                "v_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4\n\t"
                "v_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4\n\t"
                : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d0 * 0 + 1.0));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + (float)(d0 + d1 + d2 + d3) + q7 + q2 + q3 + q4 + q5 + q6;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int T> void run(const char *name, int waves_per_simd) {
    int cus = 256;
    int threads = 64 * 4 * waves_per_simd;
    float *out; unsigned long long *cyc;
    hipMalloc(&out, sizeof(float) * cus * threads);
    hipMalloc(&cyc, sizeof(unsigned long long) * cus * threads / 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    bench<T><<<cus, threads>>>(out, cyc, 0xAAAAAAAAAAAAAAAAull);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    bench<T><<<cus, threads>>>(out, cyc, 0xAAAAAAAAAAAAAAAAull);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(cus * threads / 64);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    double med = (double)h[h.size() / 2];
    double ninstr = (double)ITERS * 32;
    // s_memtime ticks at a constant 100 MHz on this family; convert through the wall time too
    printf("%-28s W=%d  wall %.3f ms  -> %.2f ns per wave-instr per SIMD (x%d waves)  memtime ticks/instr %.3f\n", name,
           waves_per_simd, ms, ms * 1e6 / (ninstr * waves_per_simd), waves_per_simd, med / ninstr);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_add_f32", w);
        run<1>("v_pk_add_f32", w);
        run<2>("v_add_f32_dpp quad_perm", w);
        run<3>("v_add_f32_dpp row_ror:8", w);
        run<4>("v_permlane32_swap", w);
        run<5>("v_permlane16_swap", w);
        run<6>("v_cndmask_e64 sgpr neg", w);
        run<7>("v_fma_f32", w);
        run<8>("v_pk_fma_f32", w);
        run<9>("v_fma_f64", w);
        run<10>("v_cvt_f64_f32", w);
        run<11>("ds_swizzle_b32", w);
        run<12>("v_mul_f64", w);
        run<13>("v_sin_f32", w);
        run<14>("v_pk_mul_f32", w);
    }
    return 0;
}
