// mfma_probe2.hip -- price of ONE filler instruction beside v_mfma_f64_16x16x4_f64 on gfx950: a loop of 8 independent
// MFMAs per wave plus N fillers of one kind (inline assembly, so the compiler neither removes nor moves them), at two
// and four waves per SIMD.  Prints matrix-pipe cycles per MFMA and the increment per filler.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_probe2.hip -o tools/mfma_probe2 && tools/mfma_probe2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));
enum { F_NONE, F_INT, F_F32, F_CVT, F_F64MUL, F_DSB64, F_DSB128, F_DSB32, F_CNDMASK, F_SNOP };

template <int KIND, int N, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void probe(double *out, long iters) {
    __shared__ __attribute__((aligned(16))) double lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 1.0 + i * 1e-3;
    __syncthreads();
    double4_t acc[8];
    for (int i = 0; i < 8; i++) acc[i] = (double4_t){0, 0, 0, 0};
    double a0 = 1.0 + lane, b0 = 2.0 + lane;
    unsigned xi = lane; float xf = lane; double xd = lane; float4_t q4 = {0, 0, 0, 0}; double q2 = 0; float q1 = 0;
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) double *)lds + lane * 16;
    for (long it = 0; it < iters; it++) {
        #pragma unroll
        for (int f = 0; f < N; f++) {
            if (KIND == F_INT) asm volatile("v_add_u32 %0, %0, 1" : "+v"(xi));
            if (KIND == F_F32) asm volatile("v_add_f32 %0, %0, 1.0" : "+v"(xf));
            if (KIND == F_CVT) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(xd) : "v"(xf));
            if (KIND == F_F64MUL) asm volatile("v_mul_f64 %0, %0, 1.0" : "+v"(xd));
            if (KIND == F_DSB64) asm volatile("ds_read_b64 %0, %1" : "=v"(q2) : "v"(addr));
            if (KIND == F_DSB128) asm volatile("ds_read_b128 %0, %1" : "=v"(q4) : "v"(addr));
            if (KIND == F_DSB32) asm volatile("ds_read_b32 %0, %1" : "=v"(q1) : "v"(addr));
            if (KIND == F_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(xi));
            if (KIND == F_SNOP) asm volatile("s_nop 1");
        }
        if (KIND == F_DSB64 || KIND == F_DSB128 || KIND == F_DSB32) asm volatile("s_waitcnt lgkmcnt(0)");
        #pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[i], 0, 0, 0);
    }
    double s = xi + xf + xd + q4.x + q2 + q1;
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(long)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double base_cyc[3];
template <int KIND, int N, int WAVES>
void run(double *out, const char *name) {
    const long iters = 4096;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<KIND, N, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, out, iters);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<KIND, N, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = 256.0 * WAVES * iters * 8;
    const double cyc = ms * 1e-3 * 2.4e9 * 1024 / mfmas;
    const int wi = WAVES == 8 ? 0 : WAVES == 16 ? 1 : 2;
    if (KIND == F_NONE) base_cyc[wi] = cyc;
    printf("%-10s x%d per 8 MFMAs, %2d waves/CU: %6.1f cycles/MFMA/SIMD (at 2.4 GHz)", name, N, WAVES, cyc);
    if (KIND != F_NONE && N > 0) printf("   +%.1f matrix-pipe cycles per filler (all waves of the SIMD counted: x%d issued)", (cyc - base_cyc[wi]) * 8 / N, WAVES / 4);
    printf("\n");
}

#define BOTH(KIND, N, name) run<KIND, N, 8>(out, name); run<KIND, N, 16>(out, name);
int main() {
    double *out;
    (void)hipMalloc(&out, 256 * 16 * 64 * sizeof(double));
    BOTH(F_NONE, 0, "none")
    BOTH(F_INT, 4, "v_add_u32") BOTH(F_INT, 8, "v_add_u32")
    BOTH(F_F32, 4, "v_add_f32") BOTH(F_CNDMASK, 4, "v_cndmask")
    BOTH(F_CVT, 1, "cvt_f64_f32") BOTH(F_CVT, 4, "cvt_f64_f32")
    BOTH(F_F64MUL, 2, "v_mul_f64") BOTH(F_F64MUL, 4, "v_mul_f64")
    BOTH(F_DSB32, 2, "ds_read_b32") BOTH(F_DSB64, 4, "ds_read_b64") BOTH(F_DSB128, 4, "ds_read_b128") BOTH(F_DSB128, 8, "ds_read_b128")
    BOTH(F_SNOP, 4, "s_nop 1")
    return 0;
}
