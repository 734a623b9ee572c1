// mfma_probe.hip -- what keeps v_mfma_f64_16x16x4_f64 from issuing back to back on gfx950: adds the
// ingredients of the block-matvec kernels one at a time (conversions, LDS operand reads, barriers,
// streaming global loads) and reports the matrix-pipe rate of each.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o tools/mfma_probe && tools/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int V, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void probe(const float *src, double *out, long iters, long stride) {
    __shared__ double lds[2][32 * 36];
    extern __shared__ double pad_lds[];       // only limits the workgroups per CU
    if (iters < 0) pad_lds[threadIdx.x] = 0;
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    double4_t acc[8];
    for (int i = 0; i < 8; i++) acc[i] = (double4_t){0, 0, 0, 0};
    for (int i = threadIdx.x; i < 2 * 32 * 36; i += blockDim.x) (&lds[0][0])[i] = 1.0 + i * 1e-3;
    __syncthreads();
    const float *p = src + ((long)blockIdx.x * (WAVES * 64) + threadIdx.x) * 4;
    float4 cur = *reinterpret_cast<const float4 *>(p), nxt = cur;
    double b0 = 1.0 + lane, b1 = 2.0 + lane;
    for (long it = 0; it < iters; it++) {
        if (V >= 4) nxt = *reinterpret_cast<const float4 *>(p + ((it + 8) % 64) * stride);
        if (V >= 2) {
            b0 = lds[it & 1][(4 * (it & 7) + g) * 36 + c];
            b1 = lds[it & 1][(4 * (it & 7) + g) * 36 + 16 + c];
        }
        double a0, a1, a2, a3;
        if (V >= 1) { a0 = (double)cur.x; a1 = (double)cur.y; a2 = (double)cur.z; a3 = (double)cur.w; }
        else { a0 = b0; a1 = b1; a2 = b0; a3 = b1; }
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[3], 0, 0, 0);
        acc[4] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b0, acc[4], 0, 0, 0);
        acc[5] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b1, acc[5], 0, 0, 0);
        acc[6] = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b0, acc[6], 0, 0, 0);
        acc[7] = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b1, acc[7], 0, 0, 0);
        if (V >= 3 && (it & 7) == 7) __syncthreads();
        if (V >= 4) cur = nxt;
        else if (V >= 1) { cur.x += 1.f; }
    }
    double s = 0;
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(long)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V, int WAVES>
void run(const float *src, double *out, int blocks_per_cu) {
    const size_t dyn = (size_t)(120 * 1024) / blocks_per_cu;
    (void)hipFuncSetAttribute((const void *)probe<V, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    const long iters = 4096, stride = 1 << 20;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * blocks_per_cu;
    hipLaunchKernelGGL((probe<V, WAVES>), dim3(blocks), dim3(WAVES * 64), dyn, 0, src, out, iters, stride);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<V, WAVES>), dim3(blocks), dim3(WAVES * 64), dyn, 0, src, out, iters, stride);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)blocks * WAVES * iters * 8;
    printf("variant %d, %d waves/wg x %d wg/cu: %.3f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", V, WAVES,
           blocks_per_cu, ms, mfmas * 2048 / ms / 1e9, ms * 1e-3 * 2.4e9 * 1024 / mfmas);
}

int main() {
    float *src; double *out;
    (void)hipMalloc(&src, (size_t)80 << 20 << 2);
    (void)hipMemset(src, 0, (size_t)80 << 20 << 2);
    (void)hipMalloc(&out, 256 * 8 * 1024 * sizeof(double));
    run<0, 4>(src, out, 1); run<0, 8>(src, out, 1); run<0, 4>(src, out, 2);
    run<1, 8>(src, out, 1); run<2, 8>(src, out, 1); run<3, 8>(src, out, 1); run<4, 8>(src, out, 1);
    run<3, 4>(src, out, 2); run<4, 4>(src, out, 2); run<4, 4>(src, out, 1);
    run<0, 16>(src, out, 1); run<4, 16>(src, out, 1); run<4, 8>(src, out, 2); run<4, 4>(src, out, 4); run<4, 4>(src, out, 3);
    run<4, 12>(src, out, 1);
    return 0;
}
