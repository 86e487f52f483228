// Accuracy (against fp64) and throughput of sin variants for SnakeBeta: libm sinf, __sinf, the hardware
// sine with a two-float FMA range reduction (conv_tm.hip sin_rev), a degree-11 polynomial.
//   hipcc --offload-arch=gfx950 -O3 -o sin_probe.bin sin_probe.hip && ./sin_probe.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__device__ __forceinline__ float sin_a(float x) { return sinf(x); }
__device__ __forceinline__ float sin_b(float x) { return __sinf(x); }
__device__ __forceinline__ float sin_c(float x) {
    const float inv2pi = 0.15915494309189535f, lo = (float)(0.15915494309189535 - (double)0.15915494309189535f);
    float r = x * inv2pi;
    float k = rintf(r);
    float f = fmaf(x, inv2pi, -k);
    f = fmaf(x, lo, f);
    return __builtin_amdgcn_sinf(f);
}
// sin^2 via polynomial on reduced argument: sin(pi*f') with f' in [-0.5, 0.5], odd minimax deg 9
__device__ __forceinline__ float sin_d(float x) {
    const float invpi = 0.3183098861837907f, lo = (float)(0.3183098861837907 - (double)0.3183098861837907f);
    float r = x * invpi;
    float k = rintf(r);
    float f = fmaf(x, invpi, -k);
    f = fmaf(x, lo, f);           // x/pi - k in [-0.5, 0.5]
    float f2 = f * f;
    // sin(pi f) = f * P(f^2), Taylor-ish/minimax coefficients
    float p = -0.0073704309f * 1.0f;                  // placeholder for deg-11 term (approx)
    p = fmaf(p, f2, 0.082145887f);
    p = fmaf(p, f2, -0.59926453f);
    p = fmaf(p, f2, 2.5501640f);
    p = fmaf(p, f2, -5.1677128f);
    p = fmaf(p, f2, 3.1415927f);
    float s = f * p;
    int ki = (int)k;
    return (ki & 1) ? -s : s;
}
template <int V>
__global__ void eval(const float* x, float* y, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = x[i];
    y[i] = V == 0 ? sin_a(v) : V == 1 ? sin_b(v) : V == 2 ? sin_c(v) : sin_d(v);
}
template <int V>
__global__ void thru(const float* x, float* y, int n, int reps) {
    int i = blockIdx.x * 256 + threadIdx.x;
    float v = x[i % n], acc = 0.f;
    for (int r = 0; r < reps; ++r) {
        float s = V == 0 ? sin_a(v) : V == 1 ? sin_b(v) : V == 2 ? sin_c(v) : sin_d(v);
        acc += s * s;
        v += 0.37f;
    }
    y[i] = acc;
}
int main() {
    const int n = 1 << 22;
    for (float range : {4.f, 50.f, 400.f}) {
        std::vector<float> hx(n);
        unsigned s = 12345;
        for (int i = 0; i < n; ++i) {
            s = s * 1664525u + 1013904223u;
            hx[i] = ((s >> 8) / 16777216.0f * 2.f - 1.f) * range;
        }
        float *dx, *dy;
        hipMalloc(&dx, n * 4);
        hipMalloc(&dy, n * 4);
        hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
        std::vector<float> hy(n);
        for (int v = 0; v < 4; ++v) {
            if (v == 0) hipLaunchKernelGGL(eval<0>, dim3(n / 256), dim3(256), 0, 0, dx, dy, n);
            if (v == 1) hipLaunchKernelGGL(eval<1>, dim3(n / 256), dim3(256), 0, 0, dx, dy, n);
            if (v == 2) hipLaunchKernelGGL(eval<2>, dim3(n / 256), dim3(256), 0, 0, dx, dy, n);
            if (v == 3) hipLaunchKernelGGL(eval<3>, dim3(n / 256), dim3(256), 0, 0, dx, dy, n);
            hipMemcpy(hy.data(), dy, n * 4, hipMemcpyDeviceToHost);
            double me = 0;
            for (int i = 0; i < n; ++i) {
                double e = fabs((double)hy[i] - sin((double)hx[i]));
                me = e > me ? e : me;
            }
            printf("range %.0f variant %d max abs err %.3e\n", range, v, me);
        }
        hipFree(dx);
        hipFree(dy);
    }
    float *dx, *dy;
    const int m = 256 * 1024 * 4;
    hipMalloc(&dx, m * 4);
    hipMalloc(&dy, m * 4);
    hipMemset(dx, 0, m * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int v = 0; v < 4; ++v) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (v == 0) hipLaunchKernelGGL(thru<0>, dim3(m / 256), dim3(256), 0, 0, dx, dy, m, 256);
            if (v == 1) hipLaunchKernelGGL(thru<1>, dim3(m / 256), dim3(256), 0, 0, dx, dy, m, 256);
            if (v == 2) hipLaunchKernelGGL(thru<2>, dim3(m / 256), dim3(256), 0, 0, dx, dy, m, 256);
            if (v == 3) hipLaunchKernelGGL(thru<3>, dim3(m / 256), dim3(256), 0, 0, dx, dy, m, 256);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("variant %d: %.3f ms for %.1f G sin -> %.1f Gsin/s\n", v, ms, m * 256.0 / 1e9, m * 256.0 / ms / 1e6);
        }
    }
    return 0;
}
