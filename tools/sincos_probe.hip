// Accuracy of the two float32 sincos candidates for the feature kernels against double precision libm:
//   poly: Cody-Waite by pi/2 in three fmas + Cephes kernels (xgpr_amd/csrc/common.inc, the shipped one)
//   hw:   Cody-Waite by 2 pi in two fmas, then v_sin_f32 / v_cos_f32 (inputs in revolutions)
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/sincos_probe.hip -o tools/ablate/sincos_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include <random>

__device__ __forceinline__ float as_f(int x) { return __builtin_bit_cast(float, x); }
__device__ __forceinline__ int as_i(float x) { return __builtin_bit_cast(int, x); }

__device__ void sincos_poly(float v, float &s, float &c) {
    float kf = __builtin_rintf(v * 0.6366197466850281f);
    float r = __builtin_fmaf(kf, -1.5707963705062866f, v);
    r = __builtin_fmaf(kf, 4.371138828673793e-08f, r);
    r = __builtin_fmaf(kf, 1.7151245100058819e-15f, r);
    int q = (int)kf;
    float r2 = r * r;
    float ps = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f), r2 * r, r);
    float pc = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f), r2 * r2, __builtin_fmaf(-0.5f, r2, 1.0f));
    float ss = (q & 1) ? pc : ps, cc = (q & 1) ? ps : pc;
    s = as_f(as_i(ss) ^ ((q & 2) << 30));
    c = as_f(as_i(cc) ^ (((q + 1) & 2) << 30));
}

__device__ void sincos_hw(float v, float &s, float &c) {
    float kf = __builtin_rintf(v * 0.15915494309189535f);
    float r = __builtin_fmaf(kf, -6.2831854820251465f, v);
    r = __builtin_fmaf(kf, 1.7484555314695172e-07f, r);
    float t = r * 0.15915494309189535f;
    s = __builtin_amdgcn_sinf(t);
    c = __builtin_amdgcn_cosf(t);
}

// the same with the reduction done directly in revolutions: t = v / (2 pi) - rint(v / (2 pi)) from a two-term 1 / (2 pi)
__device__ void sincos_hw3(float v, float &s, float &c) {
    const float c_hi = 0.15915494f, c_lo = (float)(0.15915494309189535 - (double)0.15915494f);
    float kf = __builtin_rintf(v * c_hi);
    float t = __builtin_fmaf(v, c_hi, -kf);
    t = __builtin_fmaf(v, c_lo, t);
    s = __builtin_amdgcn_sinf(t);
    c = __builtin_amdgcn_cosf(t);
}

__global__ void probe(const float *v, float *out, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s, c;
    sincos_poly(v[i], s, c); out[6 * i] = s; out[6 * i + 1] = c;
    sincos_hw(v[i], s, c); out[6 * i + 2] = s; out[6 * i + 3] = c;
    sincos_hw3(v[i], s, c); out[6 * i + 4] = s; out[6 * i + 5] = c;
}

int main() {
    const long n = 1L << 24;
    std::vector<float> v(n);
    std::mt19937_64 rng(7);
    const double ranges[] = {1.0, 8.0, 64.0, 1024.0, 65536.0, 262143.0};
    float *dv, *dout;
    hipMalloc(&dv, n * 4); hipMalloc(&dout, n * 24);
    std::vector<float> out(6 * n);
    for (double R : ranges) {
        std::uniform_real_distribution<double> u(-R, R);
        for (long i = 0; i < n; i++) v[i] = (float)u(rng);
        hipMemcpy(dv, v.data(), n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dv, dout, n);
        hipMemcpy(out.data(), dout, n * 24, hipMemcpyDeviceToHost);
        double ep = 0, eh = 0, e3 = 0;
        for (long i = 0; i < n; i++) {
            const double s = sin((double)v[i]), c = cos((double)v[i]);
            ep = fmax(ep, fmax(fabs(out[6 * i] - s), fabs(out[6 * i + 1] - c)));
            eh = fmax(eh, fmax(fabs(out[6 * i + 2] - s), fabs(out[6 * i + 3] - c)));
            e3 = fmax(e3, fmax(fabs(out[6 * i + 4] - s), fabs(out[6 * i + 5] - c)));
        }
        printf("|v| < %-9g max abs error: poly %.3e   hw(2 pi) %.3e   hw(revolutions) %.3e\n", R, ep, eh, e3);
    }
    return 0;
}
