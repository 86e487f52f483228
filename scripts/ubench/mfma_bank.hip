// Does v_mfma_f32_16x16x4_f32 slow down when srcA and srcB sit in the same VGPR bank (index mod 4)?
// hipcc --offload-arch=gfx950 -O3 mfma_bank.hip -o mfma_bank.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int SHIFT>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
    f32x4 av = {a, a + 1, a + 2, a + 3}, bv = {b, b + 1, b + 2, b + 3};
    asm volatile("" : "+v"(av), "+v"(bv));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c], bv[(c + SHIFT) & 3], acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; (void)hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 20000, blocks = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int sh = 0; sh < 4; ++sh) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            if (sh == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
            if (sh == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
            if (sh == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
            if (sh == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double fl = 32.0 * 2048.0 * iters * blocks * 4.0;
        printf("B component shift %d: %.3f ms  %.1f TFLOP/s\n", sh, ms, fl / ms / 1e9);
    }
    return 0;
}
