// valu_price.hip -- issue time of single vector instructions on gfx950 at three waves per SIMD (the fused matvec's
// occupancy): 32 independent copies of one instruction per loop trip, one workgroup of 12 waves per CU.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_price.hip -o tools/valu_price && tools/valu_price
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 2000
#define REP8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)

template <int T>
__global__ __launch_bounds__(768) void bench(float *out) {
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    float q0 = 1.5f;
    unsigned u0 = threadIdx.x * 2654435761u;
    double d0 = r0, d1 = r1, d2 = r2, d3 = r3, d4 = r4, d5 = r5, d6 = r6, d7 = r7, e0 = 1.000001;
    for (int it = 0; it < ITERS; it++) {
#define F32(STR) asm volatile(REP8(STR) REP8(STR) REP8(STR) REP8(STR) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(q0), "v"(u0) : "vcc", "s20", "s21")
#define F64(STR) asm volatile(REP8(STR) REP8(STR) REP8(STR) REP8(STR) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(e0), "v"(q0))
        if (T == 0) {
#define I(k) "v_add_f32 %" #k ", %" #k ", %8\n\t"
            F32(I);
#undef I
        } else if (T == 1) {
#define I(k) "v_xor_b32 %" #k ", %" #k ", %9\n\t"
            F32(I);
#undef I
        } else if (T == 2) {
#define I(k) "v_lshlrev_b32 %" #k ", 3, %" #k "\n\t"
            F32(I);
#undef I
        } else if (T == 3) {
#define I(k) "v_bitop3_b32 %" #k ", %" #k ", %9, %8 bitop3:0x6c\n\t"
            F32(I);
#undef I
        } else if (T == 4) {
#define I(k) "v_fmac_f32_dpp %" #k ", %" #k ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            F32(I);
#undef I
        } else if (T == 5) {
#define I(k) "v_add_f32_dpp %" #k ", %" #k ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            F32(I);
#undef I
        } else if (T == 6) {
#define I(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n\t"
            F32(I);
#undef I
        } else if (T == 7) {
#define I(k) "v_cndmask_b32_e64 %" #k ", %" #k ", -%" #k ", s[20:21]\n\t"
            F32(I);
#undef I
        } else if (T == 8) {
#define I(k) "v_mul_f32 %" #k ", %" #k ", %8\n\t"
            F32(I);
#undef I
        } else if (T == 9) {
#define I(k) "v_rndne_f32 %" #k ", %" #k "\n\t"
            F32(I);
#undef I
        } else if (T == 10) {
#define I(k) "v_cmp_lt_f32_e64 s[20:21], |%" #k "|, %8\n\t"
            F32(I);
#undef I
        } else if (T == 11) {
#define I(k) "v_max3_f32 %" #k ", |%" #k "|, |%8|, |%8|\n\t"
            F32(I);
#undef I
        } else if (T == 12) {
#define I(k) "v_sin_f32 %" #k ", %" #k "\n\t"
            F32(I);
#undef I
        } else if (T == 13) {
#define I(k) "v_fmac_f64 %" #k ", %8, %8\n\t"
            F64(I);
#undef I
        } else if (T == 14) {
#define I(k) "v_cvt_f64_f32 %" #k ", %9\n\t"
            F64(I);
#undef I
        } else if (T == 15) {
#define I(k) "v_and_b32 %" #k ", %" #k ", %9\n\t"
            F32(I);
#undef I
        } else if (T == 16) {
#define I(k) "v_sub_f32 %" #k ", %" #k ", %8\n\t"
            F32(I);
#undef I
        } else if (T == 17) {
#define I(k) "v_fma_f32 %" #k ", %" #k ", %8, %8\n\t"
            F32(I);
#undef I
        } else if (T == 18) {
#define I(k) "v_add_f32_e64 %" #k ", %" #k ", -%8\n\t"
            F32(I);
#undef I
        } else if (T == 19) {
#define I(k) "v_mov_b32 %" #k ", %8\n\t"
            F32(I);
#undef I
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + u0;
}

static float base_ns = 0;
template <int T> void run(const char *name) {
    float *out;
    (void)hipMalloc(&out, sizeof(float) * 256 * 768);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    bench<T><<<256, 768>>>(out);
    (void)hipEventRecord(e0);
    bench<T><<<256, 768>>>(out);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / ((double)ITERS * 32 * 3);
    if (T == 0) base_ns = ns;
    printf("%-34s %.2f ns per wave-instruction per SIMD = %.2f x v_add_f32\n", name, ns, ns / base_ns);
    (void)hipFree(out);
}

int main() {
    run<0>("v_add_f32"); run<16>("v_sub_f32"); run<8>("v_mul_f32"); run<17>("v_fma_f32"); run<18>("v_add_f32_e64 (neg modifier)");
    run<19>("v_mov_b32"); run<1>("v_xor_b32"); run<15>("v_and_b32"); run<2>("v_lshlrev_b32"); run<3>("v_bitop3_b32");
    run<6>("v_cndmask_b32 (vcc)"); run<7>("v_cndmask_b32_e64 (sgpr, neg)"); run<10>("v_cmp_lt_f32_e64 -> sgpr"); run<11>("v_max3_f32 (abs)");
    run<9>("v_rndne_f32"); run<12>("v_sin_f32"); run<5>("v_add_f32_dpp quad_perm"); run<4>("v_fmac_f32_dpp quad_perm");
    run<13>("v_fmac_f64"); run<14>("v_cvt_f64_f32");
    return 0;
}
