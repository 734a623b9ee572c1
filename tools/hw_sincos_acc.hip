// accuracy of the hardware v_sin_f32 / v_cos_f32 (argument in revolutions) on the reduced range
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>
__global__ void k(const float* r, float* s, float* c, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float t = r[i] * 0.15915494309189535f; s[i] = __builtin_amdgcn_sinf(t); c[i] = __builtin_amdgcn_cosf(t); }
}
int main() {
    const int n = 1 << 22;
    std::vector<float> h(n);
    for (int i = 0; i < n; i++) h[i] = (float)(-0.7853981633974483 + 1.5707963267948966 * (i + 0.5) / n);
    float *r, *s, *c; hipMalloc(&r, n * 4); hipMalloc(&s, n * 4); hipMalloc(&c, n * 4);
    hipMemcpy(r, h.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(r, s, c, n);
    std::vector<float> hs(n), hc(n);
    hipMemcpy(hs.data(), s, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hc.data(), c, n * 4, hipMemcpyDeviceToHost);
    double ms = 0, mc = 0;
    for (int i = 0; i < n; i++) { ms = fmax(ms, fabs(hs[i] - sin((double)h[i]))); mc = fmax(mc, fabs(hc[i] - cos((double)h[i]))); }
    printf("v_sin_f32 max abs err on [-pi/4,pi/4]: %.3e   v_cos_f32: %.3e\n", ms, mc);
    return 0;
}
