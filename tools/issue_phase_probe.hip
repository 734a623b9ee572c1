// issue_phase_probe.hip -- how a SIMD of gfx950 shares its vector issue between W resident waves.
// One workgroup of 4 W waves per CU (W per SIMD); every wave runs the same stream of independent v_add_f32 and stamps
// s_memtime around it; prints, per W, the per-wave durations of workgroup 0 grouped by SIMD (HW_ID) and the launch time.
//   hipcc -O3 --offload-arch=gfx950 tools/issue_phase_probe.hip -o tools/issue_phase_probe && tools/issue_phase_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#define ITERS 4000
#define REP8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)

__global__ __launch_bounds__(1024) void probe(float *out, unsigned long long *stamps, int mode) {
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    float q0 = 1.5f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
#define I(k) "v_add_f32 %" #k ", %" #k ", %8\n\t"
        asm volatile(REP8(I) REP8(I) REP8(I) REP8(I) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(q0));
#undef I
        if (mode == 1 && (it & 63) == 63) __syncthreads();          // a workgroup barrier every 2048 instructions
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        stamps[3 * w] = t0; stamps[3 * w + 1] = t1; stamps[3 * w + 2] = hwid;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
}

int main() {
    float *out; unsigned long long *st;
    (void)hipMalloc(&out, sizeof(float) * 256 * 1024);
    (void)hipMalloc(&st, sizeof(unsigned long long) * 3 * 256 * 16);
    for (int mode = 0; mode < 2; mode++)
    for (int W = 1; W <= 4; W++) {
        const int threads = 256 * W;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        probe<<<256, threads>>>(out, st, mode);
        (void)hipEventRecord(e0);
        probe<<<256, threads>>>(out, st, mode);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(3 * 4 * W);
        (void)hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
        printf("%s W=%d waves/SIMD: launch %.3f ms = %.2f ns per wave-instruction per SIMD\n", mode ? "barrier/2048" : "free-running", W, ms,
               ms * 1e6 / ((double)ITERS * 32 * W));
        for (int w = 0; w < 4 * W; w++) {
            const unsigned hw = (unsigned)h[3 * w + 2];
            printf("   wave %2d  simd %u  wave_slot %2u  cycles/instr %.2f  (start +%llu)\n", w, (hw >> 4) & 3, hw & 15,
                   (double)(h[3 * w + 1] - h[3 * w]) / (ITERS * 32.0), h[3 * w] - h[0]);
        }
    }
    return 0;
}
