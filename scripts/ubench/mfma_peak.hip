// fp32 MFMA peak calibration: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int ACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a, float b) {
    f32x4 acc[ACC];
    for (int i = 0; i < ACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < ACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
void run(const char* name, F launch, double flop_per_iter_per_wave, int blocks, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = flop_per_iter_per_wave * iters * blocks * 4.0;
    printf("%s: %.3f ms  %.1f TFLOP/s\n", name, ms, fl / ms / 1e9);
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 20000;
    for (int bpc : {1, 2, 4}) {
        int blocks = 256 * bpc;
        printf("blocks=%d (%d per CU)\n", blocks, bpc);
        run("16x16x4 acc=1", [&] { hipLaunchKernelGGL(k16<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 8 * 1 * 2048.0, blocks, iters);
        run("16x16x4 acc=2", [&] { hipLaunchKernelGGL(k16<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 8 * 2 * 2048.0, blocks, iters);
        run("16x16x4 acc=4", [&] { hipLaunchKernelGGL(k16<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 8 * 4 * 2048.0, blocks, iters);
        run("32x32x2 acc=2", [&] { hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 8 * 2 * 4096.0, blocks, iters);
    }
    return 0;
}
