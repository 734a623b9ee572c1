// valu_cost.hip -- issue cost of the vector instructions the fused matvec is built from, on gfx950: ns per
// wave-instruction per SIMD with W = 2, 3, 4 waves resident per SIMD (one workgroup of 4 W waves per CU, every wave the
// same stream of 32 instructions per trip over 8 independent registers), best of 5 launches after a warm-up.
// Output: one JSON object (profiles/rN_valu_cost.json); tools/count_loop_insts.py prices the kernel's loop with it.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_cost.hip -o tools/valu_cost && tools/valu_cost > profiles/r3_valu_cost.json
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#define ITERS 3000
#define REP8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define REP4D(I) I(0) I(1) I(2) I(3)
typedef float v2f __attribute__((ext_vector_type(2)));

#define F32BODY(STR) asm volatile(REP8(STR) REP8(STR) REP8(STR) REP8(STR) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(q0), "v"(u0), "s"(sm) : "vcc")
#define F64BODY(STR) asm volatile(REP8(STR) REP8(STR) REP8(STR) REP8(STR) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(e0), "v"(q0))
#define P32BODY(STR) asm volatile(REP8(STR) REP8(STR) REP8(STR) REP8(STR) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc))

template <int T>
__global__ __launch_bounds__(1024) void bench(float *out, unsigned long long sm) {
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    float q0 = 1.5f;
    unsigned u0 = threadIdx.x * 2654435761u;
    double d0 = r0, d1 = r1, d2 = r2, d3 = r3, d4 = r4, d5 = r5, d6 = r6, d7 = r7, e0 = 1.000001;
    v2f p0 = {r0, r1}, p1 = {r1, r2}, p2 = {r2, r3}, p3 = {r3, r4}, p4 = {r4, r5}, p5 = {r5, r6}, p6 = {r6, r7}, p7 = {r7, r0}, pc = {1.5f, 0.5f};
    for (int it = 0; it < ITERS; it++) {
#define CASE(N, BODY, STR) if constexpr (T == N) { BODY(STR); }
#define I(k) "v_add_f32 %" #k ", %" #k ", %8\n\t"
        CASE(0, F32BODY, I)
#undef I
#define I(k) "v_pk_add_f32 %" #k ", %" #k ", %8\n\t"
        CASE(1, P32BODY, I)
#undef I
#define I(k) "v_pk_add_f32 %" #k ", %" #k ", %" #k " op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
        CASE(2, P32BODY, I)
#undef I
#define I(k) "v_fmac_f32_dpp %" #k ", %" #k ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        CASE(3, F32BODY, I)
#undef I
#define I(k) "v_add_u32 %" #k ", %" #k ", %" #k "\n\t"
        CASE(4, F32BODY, I)
#undef I
#define I(k) "v_bitop3_b32 %" #k ", %9, %" #k ", %8 bitop3:0x6c\n\t"
        CASE(5, F32BODY, I)
#undef I
#define I(k) "v_cvt_f64_f32 %" #k ", %9\n\t"
        CASE(6, F64BODY, I)
#undef I
#define I(k) "v_fmac_f64 %" #k ", %8, %8\n\t"
        CASE(7, F64BODY, I)
#undef I
#define I(k) "v_add_f64 %" #k ", %" #k ", %8\n\t"
        CASE(8, F64BODY, I)
#undef I
#define I(k) "v_mul_f32 %" #k ", %" #k ", %8\n\t"
        CASE(9, F32BODY, I)
#undef I
#define I(k) "v_rndne_f32 %" #k ", %" #k "\n\t"
        CASE(10, F32BODY, I)
#undef I
#define I(k) "v_pk_fma_f32 %" #k ", %" #k ", %8, %8\n\t"
        CASE(11, P32BODY, I)
#undef I
#define I(k) "v_pk_mul_f32 %" #k ", %" #k ", %8\n\t"
        CASE(12, P32BODY, I)
#undef I
#define I(k) "v_sin_f32 %" #k ", %" #k "\n\t"
        CASE(13, F32BODY, I)
#undef I
#define I(k) "v_cos_f32 %" #k ", %" #k "\n\t"
        CASE(14, F32BODY, I)
#undef I
#define I(k) "v_max3_f32 %" #k ", |%" #k "|, |%8|, |%8|\n\t"
        CASE(15, F32BODY, I)
#undef I
#define I(k) "v_mov_b32 %" #k ", %8\n\t"
        CASE(16, F32BODY, I)
#undef I
#define I(k) "v_mov_b32_dpp %" #k ", %8 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        CASE(17, F32BODY, I)
#undef I
        if constexpr (T == 18) {
            asm volatile("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                         "v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                         "v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                         "v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                         "v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                         "v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                         "v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                         "v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
        }
#define I(k) "v_fma_f32 %" #k ", %" #k ", %8, %8\n\t"
        CASE(19, F32BODY, I)
#undef I
#define I(k) "v_cndmask_b32_e64 %" #k ", %" #k ", -%" #k ", %10\n\t"
        CASE(20, F32BODY, I)
#undef I
#define I(k) "s_nop 0\n\t"
        CASE(21, F32BODY, I)
#undef I
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + u0
        + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

static bool first = true;
template <int T> void run(const char *name) {
    float *out;
    (void)hipMalloc(&out, sizeof(float) * 256 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("%s\n  \"%s\": {", first ? "" : ",", name); first = false;
    for (int W = 2; W <= 4; W++) {
        float best = 1e30f;
        for (int rep = 0; rep < 6; rep++) {
            (void)hipEventRecord(e0);
            bench<T><<<256, 256 * W>>>(out, 0xAAAAAAAAAAAAAAAAull);
            (void)hipEventRecord(e1);
            (void)hipDeviceSynchronize();
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        printf("%s\"W%d\": %.3f", W == 2 ? "" : ", ", W, best * 1e6 / ((double)ITERS * 32 * W));
    }
    printf("}");
    (void)hipFree(out);
}

int main() {
    // warm the clocks up
    { float *o; (void)hipMalloc(&o, sizeof(float) * 256 * 1024); for (int i = 0; i < 200; i++) bench<0><<<256, 1024>>>(o, 0); (void)hipDeviceSynchronize(); (void)hipFree(o); }
    printf("{\"_unit\": \"ns per wave-instruction per SIMD, W waves resident per SIMD, independent instructions (tools/valu_cost.hip)\"");
    first = false;
    run<0>("v_add_f32"); run<9>("v_mul_f32"); run<19>("v_fma_f32"); run<16>("v_mov_b32"); run<4>("v_add_u32"); run<5>("v_bitop3_b32");
    run<1>("v_pk_add_f32"); run<2>("v_pk_add_f32 op_sel"); run<12>("v_pk_mul_f32"); run<11>("v_pk_fma_f32");
    run<3>("v_fmac_f32_dpp"); run<17>("v_mov_b32_dpp"); run<18>("v_permlane32_swap_b32"); run<20>("v_cndmask_b32_e64");
    run<10>("v_rndne_f32"); run<15>("v_max3_f32"); run<13>("v_sin_f32"); run<14>("v_cos_f32");
    run<6>("v_cvt_f64_f32"); run<7>("v_fmac_f64"); run<8>("v_add_f64"); run<21>("s_nop");
    printf("\n}\n");
    return 0;
}
