// Can the float64 VECTOR pipe add to the float64 MATRIX pipe on gfx950?  12-wave workgroups, one per CU: waves 0..7
// (two per SIMD) issue independent v_mfma_f64_16x16x4_f64 chains, waves 8..11 (one per SIMD) issue independent
// v_fma_f64 chains; each role is also run alone.  Prints flop rates per role (matrix peak = vector peak = 78.6 TFLOP/s).
//   hipcc -O3 --offload-arch=gfx950 tools/dualpipe_probe.hip -o tools/dualpipe_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: both roles, 1: MFMA waves only (others exit), 2: VALU waves only
__global__ __launch_bounds__(768) void k(double *out, int iters, long long *ticks) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool mf = w < 8;
    if ((MODE == 1 && !mf) || (MODE == 2 && mf)) return;
    double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    double s = 0.0;
    if (mf) {
        double4_t acc[8];
        for (int i = 0; i < 8; i++) acc[i] = (double4_t){0, 0, 0, 0};
        for (int it = 0; it < iters; it++) {
            #pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][3];
    } else {
        double acc[32];
        for (int i = 0; i < 32; i++) acc[i] = i;
        for (int it = 0; it < iters; it++) {
            #pragma unroll
            for (int i = 0; i < 32; i++) acc[i] = __builtin_fma(a, b, acc[i]);
            asm volatile("" : "+v"(a));
        }
        for (int i = 0; i < 32; i++) s += acc[i];
    }
    const long long t1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 768 + threadIdx.x] = s;
    if (lane == 0) ticks[blockIdx.x * 12 + w] = t1 - t0;
}

int main() {
    double *d; long long *t;
    hipMalloc(&d, 8 * 768 * 256); hipMalloc(&t, 8 * 12 * 256);
    const int iters = 40000;
    for (int mode = 0; mode < 3; mode++) {
        hipMemset(t, 0, 8 * 12 * 256);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(768), 0, 0, d, iters, t);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(768), 0, 0, d, iters, t);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(768), 0, 0, d, iters, t);
        long long h[12 * 256];
        hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
        double tm = 0, tv = 0; int nm = 0, nv = 0;
        for (int b = 0; b < 256; b++) for (int w = 0; w < 12; w++) { if (h[b * 12 + w] == 0) continue; if (w < 8) { tm += h[b * 12 + w]; nm++; } else { tv += h[b * 12 + w]; nv++; } }
        // s_memrealtime ticks at 100 MHz
        const double mfma_flop = 8.0 * iters * 2048, valu_flop = 32.0 * iters * 128;
        printf("mode %d:", mode);
        if (nm) printf("  matrix waves: %.1f us each -> %.1f TFLOP/s chip-wide", tm / nm / 100.0, mfma_flop * 8 * 256 / (tm / nm / 1e8) / 1e12);
        if (nv) printf("  vector waves: %.1f us each -> %.1f TFLOP/s chip-wide", tv / nv / 100.0, valu_flop * 4 * 256 / (tv / nv / 1e8) / 1e12);
        printf("\n");
    }
    return 0;
}
