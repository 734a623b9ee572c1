// sorf_variants.hip -- throughput of alternative wave-tile SORF (3 x [sign flip, FHT-1024]) cores on
// gfx950, all bit-identical.  V0: DPP + permlane swaps (no LDS).  V1: DPP for strides 1,2,8 and the
// LDS crossbar (ds_swizzle / ds_bpermute) for 4,16,32.  V4: two register layouts with a wave-private
// LDS transpose in between, so 8 of 10 stages are register-local and the other two are quad_perm DPP.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/sorf_variants.hip -o tools/sorf_variants
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

__device__ __forceinline__ float as_f(int x) { return __builtin_bit_cast(float, x); }
__device__ __forceinline__ int as_i(float x) { return __builtin_bit_cast(int, x); }
__device__ __forceinline__ void bfly(float &a, float &b) { float s = a + b, d = a - b; a = s; b = d; }

template <int H> __device__ __forceinline__ void xstage_dpp(float (&v)[16], int lane) {
    const int sm = (lane & H) ? (int)0x80000000 : 0;
    #pragma unroll
    for (int r = 0; r < 16; r++) {
        int xi = as_i(v[r]); int p;
        if constexpr (H == 1) p = __builtin_amdgcn_mov_dpp(xi, 0xB1, 0xf, 0xf, true);
        else if constexpr (H == 2) p = __builtin_amdgcn_mov_dpp(xi, 0x4E, 0xf, 0xf, true);
        else p = __builtin_amdgcn_mov_dpp(xi, 0x128, 0xf, 0xf, true);
        v[r] = as_f(xi ^ sm) + as_f(p);
    }
}
__device__ __forceinline__ void xstage4_asm(float (&v)[16]) {
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
}
template <int H> __device__ __forceinline__ void xstage_swap(float (&v)[16]) {
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
template <int H> __device__ __forceinline__ void xstage_lds(float (&v)[16], int lane) {
    const int sm = (lane & H) ? (int)0x80000000 : 0;
    #pragma unroll
    for (int r = 0; r < 16; r++) {
        int xi = as_i(v[r]); int p;
        if constexpr (H == 4) p = __builtin_amdgcn_ds_swizzle(xi, 0x101F);        // xor 4
        else if constexpr (H == 16) p = __builtin_amdgcn_ds_swizzle(xi, 0x401F);  // xor 16
        else p = __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, xi);
        v[r] = as_f(xi ^ sm) + as_f(p);
    }
}
__device__ __forceinline__ void local_stages(float (&v)[16]) {
    #pragma unroll
    for (int q = 1; q < 16; q <<= 1) {
        #pragma unroll
        for (int r = 0; r < 16; r++) if (!(r & q)) bfly(v[r], v[r + q]);
    }
}
__device__ __forceinline__ void signs_sgpr(float (&v)[16], const uint64_t *__restrict__ mk) {
    #pragma unroll
    for (int r = 0; r < 16; r++) v[r] = __builtin_amdgcn_inverse_ballot_w64(mk[r]) ? -v[r] : v[r];
}

template <int V> __device__ __forceinline__ void fht_l1(float (&v)[16], int lane) {
    xstage_dpp<1>(v, lane); xstage_dpp<2>(v, lane);
    if constexpr (V == 0) { xstage4_asm(v); xstage_dpp<8>(v, lane); xstage_swap<16>(v); xstage_swap<32>(v); }
    else { xstage_lds<4>(v, lane); xstage_dpp<8>(v, lane); xstage_lds<16>(v, lane); xstage_lds<32>(v, lane); }
    local_stages(v);
}

// ---- V4: L2 layout lane = e[9:4], reg = e[3:0]; L1 layout lane = e[5:0], reg = e[9:6]
#define PADIDX(e) ((((e) >> 4) * 20) + ((e) & 15))
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void l1_to_l2(float (&v)[16], float *lds, int lane) {
    #pragma unroll
    for (int r = 0; r < 16; r++) lds[PADIDX(r * 64 + lane)] = v[r];
    wave_sync_lds();
    const float4 *p = reinterpret_cast<const float4 *>(lds + lane * 20);
    #pragma unroll
    for (int q = 0; q < 4; q++) { float4 t = p[q]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
    wave_sync_lds();
}
__device__ __forceinline__ void l2_to_l1(float (&v)[16], float *lds, int lane) {
    float4 *p = reinterpret_cast<float4 *>(lds + lane * 20);
    #pragma unroll
    for (int q = 0; q < 4; q++) p[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    wave_sync_lds();
    #pragma unroll
    for (int r = 0; r < 16; r++) v[r] = lds[PADIDX(r * 64 + lane)];
    wave_sync_lds();
}
__device__ __forceinline__ void signs_l2(float (&v)[16], uint32_t sw) {
    #pragma unroll
    for (int j = 0; j < 16; j++) v[j] = as_f(as_i(v[j]) ^ (int)((sw << (31 - j)) & 0x80000000u));
}
__device__ __forceinline__ void fht_low_l2(float (&v)[16], int lane) {   // strides 1..8 local, 16/32 = lane xor 1/2
    local_stages(v);
    xstage_dpp<1>(v, lane); xstage_dpp<2>(v, lane);
}

template <int V>
__global__ __launch_bounds__(256) void sorf_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                   const uint64_t *__restrict__ masks, const uint32_t *__restrict__ sw,
                                                   int ntiles, int reps) {
    __shared__ __attribute__((aligned(16))) float lds_all[4 * 1280];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *lds = lds_all + wave * 1280;
    const int nw = gridDim.x * 4;
    for (int t = blockIdx.x * 4 + wave; t < ntiles; t += nw) {
        float v[16];
        #pragma unroll
        for (int r = 0; r < 16; r++) v[r] = x[(long)t * 1024 + r * 64 + lane];
        for (int rep = 0; rep < reps; rep++) {
            if constexpr (V == 4) {
                uint32_t s0 = sw[lane], s1 = sw[64 + lane], s2 = sw[128 + lane];
                l1_to_l2(v, lds, lane);
                signs_l2(v, s0); fht_low_l2(v, lane); l2_to_l1(v, lds, lane); local_stages(v);
                l1_to_l2(v, lds, lane);
                signs_l2(v, s1); fht_low_l2(v, lane); l2_to_l1(v, lds, lane); local_stages(v);
                l1_to_l2(v, lds, lane);
                signs_l2(v, s2); fht_low_l2(v, lane); l2_to_l1(v, lds, lane); local_stages(v);
            } else {
                #pragma unroll
                for (int s = 0; s < 3; s++) { signs_sgpr(v, masks + s * 16); fht_l1<V>(v, lane); }
            }
            #pragma unroll
            for (int r = 0; r < 16; r++) v[r] *= 9.313225746154785e-10f;   // 2^-30: keep magnitudes bounded (exact)
        }
        #pragma unroll
        for (int r = 0; r < 16; r++) y[(long)t * 1024 + r * 64 + lane] = v[r];
    }
}

template <int V> double run(const float *x, float *y, const uint64_t *masks, const uint32_t *sw, int ntiles, int reps, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    sorf_kernel<V><<<blocks, 256>>>(x, y, masks, sw, ntiles, reps);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    sorf_kernel<V><<<blocks, 256>>>(x, y, masks, sw, ntiles, reps);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    const int ntiles = 1 << 16, reps = 16;
    std::vector<float> hx((size_t)ntiles * 1024);
    srand(3);
    for (auto &f : hx) f = (float)rand() / RAND_MAX - 0.5f;
    std::vector<int8_t> radem(3 * 1024);
    for (auto &r : radem) r = (rand() & 1) ? 1 : -1;
    std::vector<uint64_t> hm(48, 0);
    std::vector<uint32_t> hs(192, 0);
    for (int s = 0; s < 3; s++)
        for (int e = 0; e < 1024; e++)
            if (radem[s * 1024 + e] < 0) {
                hm[s * 16 + e / 64] |= 1ull << (e % 64);         // L1: reg e/64, lane e%64
                hs[s * 64 + e / 16] |= 1u << (e % 16);           // L2: lane e/16, bit e%16
            }
    float *x, *y0, *y1, *y4; uint64_t *m; uint32_t *sw;
    size_t bytes = hx.size() * 4;
    hipMalloc(&x, bytes); hipMalloc(&y0, bytes); hipMalloc(&y1, bytes); hipMalloc(&y4, bytes);
    hipMalloc(&m, 48 * 8); hipMalloc(&sw, 192 * 4);
    hipMemcpy(x, hx.data(), bytes, hipMemcpyHostToDevice);
    hipMemcpy(m, hm.data(), 48 * 8, hipMemcpyHostToDevice);
    hipMemcpy(sw, hs.data(), 192 * 4, hipMemcpyHostToDevice);
    for (int blocks : {256, 512, 1024, 2048}) {
        double t0 = run<0>(x, y0, m, sw, ntiles, reps, blocks);
        double t1 = run<1>(x, y1, m, sw, ntiles, reps, blocks);
        double t4 = run<4>(x, y4, m, sw, ntiles, reps, blocks);
        double n = (double)ntiles * reps;
        printf("blocks %4d (waves/SIMD %d): V0 %.3f ms (%.1f ns/tile-SORF/SIMD)  V1 %.3f ms (%.1f)  V4 %.3f ms (%.1f)\n", blocks,
               blocks / 256, t0, t0 * 1e6 / n * 1024, t1, t1 * 1e6 / n * 1024, t4, t4 * 1e6 / n * 1024);
    }
    std::vector<float> h0(hx.size()), h1(hx.size()), h4(hx.size());
    hipMemcpy(h0.data(), y0, bytes, hipMemcpyDeviceToHost);
    hipMemcpy(h1.data(), y1, bytes, hipMemcpyDeviceToHost);
    hipMemcpy(h4.data(), y4, bytes, hipMemcpyDeviceToHost);
    printf("V1 == V0 bitwise: %s;  V4 == V0 bitwise: %s\n", memcmp(h0.data(), h1.data(), bytes) ? "NO" : "yes",
           memcmp(h0.data(), h4.data(), bytes) ? "NO" : "yes");
    // CPU check of V0 on the first tile
    {
        std::vector<float> b(hx.begin(), hx.begin() + 1024);
        for (int rep = 0; rep < reps; rep++) {
            for (int s = 0; s < 3; s++) {
                for (int e = 0; e < 1024; e++) if (radem[s * 1024 + e] < 0) b[e] = -b[e];
                for (int h = 1; h < 1024; h <<= 1)
                    for (int i = 0; i < 1024; i += 2 * h)
                        for (int j = i; j < i + h; j++) { float a = b[j], c = b[j + h]; b[j] = a + c; b[j + h] = a - c; }
            }
            for (auto &f : b) f *= 9.313225746154785e-10f;
        }
        printf("V0 == CPU on tile 0: %s\n", memcmp(b.data(), h0.data(), 4096) ? "NO" : "yes");
    }
    return 0;
}
