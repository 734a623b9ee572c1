// store_probe.hip -- what write rate does a kernel shaped like the feature operator's output stage reach?  (round 4)
//   hipcc -O3 --offload-arch=gfx950 tools/store_probe.hip -o tools/store_probe && tools/store_probe
// Each wave writes "tiles" of 16 x 1 KiB (sixteen global_store_dwordx4, 16 B per lane = one tile of the float64 feature
// row) and idles `sleep` x 64 cycles between tiles (standing in for the transform).  Variants: waves per workgroup and
// workgroups per CU (occupancy), persistent (grid = CUs, tiles strided) or one tile per wave (grid = tiles / waves).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int SLEEP>
__global__ void persistent_kernel(double2 *out, long ntiles) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * (blockDim.x >> 6);
    for (long t = wave; t < ntiles; t += nwaves) {
        double2 *p = out + t * 1024 + lane;
        #pragma unroll
        for (int r = 0; r < 16; r++) p[64 * r] = make_double2((double)t, (double)r);
        #pragma unroll 1
        for (int s = 0; s < SLEEP; s++) __builtin_amdgcn_s_sleep(1);
    }
}
template <int SLEEP>
__global__ void flat_kernel(double2 *out, long ntiles) {
    const int lane = threadIdx.x & 63;
    const long t = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= ntiles) return;
    #pragma unroll 1
    for (int s = 0; s < SLEEP; s++) __builtin_amdgcn_s_sleep(1);
    double2 *p = out + t * 1024 + lane;
    #pragma unroll
    for (int r = 0; r < 16; r++) p[64 * r] = make_double2((double)t, (double)r);
}

template <typename F>
float timeit(F f) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    f(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < 5; i++) f();
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 5;
}

int main() {
    const long rows = 131072, M = 8192;
    const long bytes = rows * M * 8, ntiles = bytes / 16384;
    double2 *out; CHECK(hipMalloc(&out, bytes));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("buffer %.1f GB, %ld tiles of 16 KiB, %d CUs\n", bytes / 1e9, ntiles, cus);
    float ms = timeit([&] { CHECK(hipMemsetAsync(out, 0, bytes, 0)); });
    printf("hipMemsetAsync                                   %.3f ms  %.0f GB/s\n", ms, bytes / ms / 1e6);
#define RUN_P(SL, WAVES, WGPC) { ms = timeit([&] { hipLaunchKernelGGL(persistent_kernel<SL>, dim3(cus * WGPC), dim3(64 * WAVES), 0, 0, out, ntiles); }); \
    printf("persistent sleep %3d  %2d waves/WG x %d WG/CU        %.3f ms  %.0f GB/s\n", SL, WAVES, WGPC, ms, bytes / ms / 1e6); }
#define RUN_F(SL, WAVES) { ms = timeit([&] { hipLaunchKernelGGL(flat_kernel<SL>, dim3((unsigned)((ntiles + WAVES - 1) / WAVES)), dim3(64 * WAVES), 0, 0, out, ntiles); }); \
    printf("one tile per wave sleep %3d  %2d waves/WG           %.3f ms  %.0f GB/s\n", SL, WAVES, ms, bytes / ms / 1e6); }
    RUN_P(0, 12, 1) RUN_P(0, 16, 1) RUN_P(0, 16, 2) RUN_P(0, 4, 8)
    RUN_P(16, 12, 1) RUN_P(16, 16, 2) RUN_P(16, 4, 8)
    RUN_P(48, 12, 1) RUN_P(48, 16, 2) RUN_P(48, 4, 8)
    RUN_F(0, 4) RUN_F(16, 4) RUN_F(48, 4)
    CHECK(hipFree(out));
    return 0;
}
