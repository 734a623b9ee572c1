// Probes v_mfma_f64_4x4x4_4b_f64 on gfx950: (1) which lanes hold which (block, row, k) / (block, k, col) / (block, row, col)
// elements, by multiplying one-hot operands; (2) its issue rate beside v_mfma_f64_16x16x4_f64 (cycles per instruction per
// SIMD with 8 waves per CU issuing independent accumulations).   hipcc -O3 --offload-arch=gfx950 tools/mfma4x4_probe.hip -o tools/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(double *out) {            // grid 64 x 64 waves: (la, lb) one-hot lanes
    const int lane = threadIdx.x, la = blockIdx.x, lb = blockIdx.y;
    const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[((long)la * 64 + lb) * 64 + lane] = d;
}

template <int KIND>
__global__ __launch_bounds__(512) void rate_kernel(double *out, int iters, long long *cycles) {
    const int lane = threadIdx.x & 63;
    double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
    double acc1[8];
    double4_t acc4[8];
    for (int i = 0; i < 8; i++) { acc1[i] = 0.0; acc4[i] = (double4_t){0, 0, 0, 0}; }
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        #pragma unroll
        for (int i = 0; i < 8; i++) {
            if (KIND == 0) acc1[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1[i], 0, 0, 0);
            else acc4[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc4[i], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
    for (int i = 0; i < 8; i++) s += KIND == 0 ? acc1[i] : acc4[i][0] + acc4[i][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
    double *d; long long *cyc;
    hipMalloc(&d, sizeof(double) * 64 * 64 * 64);
    hipMalloc(&cyc, sizeof(long long) * 256);
    hipLaunchKernelGGL(layout_kernel, dim3(64, 64), dim3(64), 0, 0, d);
    std::vector<double> h(64 * 64 * 64);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    // for every output lane: which (la, lb) pairs feed it
    printf("output lane <- list of (a_lane, b_lane) pairs contributing (4 per output = the 4 k values)\n");
    for (int lo = 0; lo < 64; lo++) {
        printf("D lane %2d:", lo);
        for (int la = 0; la < 64; la++) for (int lb = 0; lb < 64; lb++)
            if (h[((long)la * 64 + lb) * 64 + lo] != 0.0) printf(" (%d,%d)", la, lb);
        printf("\n");
    }
    const int iters = 20000;
    for (int kind = 0; kind < 2; kind++) {
        if (kind == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(256), dim3(512), 0, 0, d, iters, cyc);
        else hipLaunchKernelGGL(rate_kernel<1>, dim3(256), dim3(512), 0, 0, d, iters, cyc);
        long long hc[256];
        hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
        double mean = 0; for (int i = 0; i < 256; i++) mean += hc[i]; mean /= 256;
        // 8 waves per CU = 2 per SIMD, each issued iters * 8 instructions: per SIMD 2 * iters * 8 instructions in `mean` ticks
        // (s_memtime ticks at 100 MHz on this part: convert with the shader clock separately) -- report ticks per instruction
        printf("%s: %.3f memtime ticks per instruction per SIMD (2 waves/SIMD)\n", kind == 0 ? "mfma_f64_4x4x4_4b " : "mfma_f64_16x16x4   ",
               mean / (2.0 * iters * 8));
    }
    return 0;
}
