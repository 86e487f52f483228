// calibrate s_memtime ticks vs wall time under MFMA load, with and without LDS traffic
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int LDSR>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* ticks, int iters, float a, float b) {
    __shared__ float4 sm[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) sm[i] = make_float4(a, b, a, b);
    __syncthreads();
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
    unsigned long long t0 = __builtin_readcyclecounter();
    float4 va = make_float4(a, a, a, a), vb = make_float4(b, b, b, b);
    for (int it = 0; it < iters; ++it) {
        if (LDSR) {
            va = sm[(threadIdx.x + it) & 1023];
            vb = sm[(threadIdx.x * 3 + it) & 1023];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(va.x, vb.x, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(va.y, vb.y, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(va.z, vb.z, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(va.w, vb.w, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    asm volatile("s_nop 0" ::"v"(s));
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int L>
void run(const char* name, float* out, unsigned long long* ticks, int blocks, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<L>, dim3(blocks), dim3(256), 0, 0, out, ticks, iters, 1.f, 2.f); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<L>, dim3(blocks), dim3(256), 0, 0, out, ticks, iters, 1.f, 2.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8]; (void)hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
    double fl = 16.0 * 2048 * iters * blocks * 4.0;
    printf("%s blocks=%d: %.3f ms  %.1f TFLOP/s  ticks/WG=%llu -> tick rate %.1f MHz, ticks per MFMA %.2f\n", name, blocks, ms,
           fl / ms / 1e9, h[0], h[0] / (ms * 1e3), (double)h[0] / (16.0 * iters));
}
int main() {
    float* out; unsigned long long* ticks;
    (void)hipMalloc(&out, 4096 * 256 * 4); (void)hipMalloc(&ticks, 4096 * 8);
    for (int bpc : {1, 2}) {
        run<0>("mfma only  ", out, ticks, 256 * bpc, 20000);
        run<1>("mfma + lds ", out, ticks, 256 * bpc, 20000);
    }
    return 0;
}
