// How much other work can one wave issue behind each v_mfma_f32_16x16x4_f32 before the matrix pipe
// starves?  1 wave per SIMD (256 blocks x 256 threads), 8 independent accumulators.
// hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form mfma_coissue.hip -o mfma_coissue.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND, int CNT>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    f32x4 d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
    const unsigned laddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds + (threadIdx.x & 63) * 16;
    lds[threadIdx.x] = a;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < CNT; ++c) {
                if (KIND == 0) {  // VALU
                    if ((c & 3) == 0) asm volatile("v_add_u32 %0, %0, 1" : "+v"(x0));
                    if ((c & 3) == 1) asm volatile("v_add_u32 %0, %0, 1" : "+v"(x1));
                    if ((c & 3) == 2) asm volatile("v_add_u32 %0, %0, 1" : "+v"(x2));
                    if ((c & 3) == 3) asm volatile("v_add_u32 %0, %0, 1" : "+v"(x3));
                } else if (KIND == 1) {  // SALU
                    asm volatile("s_add_u32 s40, s40, 1" ::: "s40");
                } else if (KIND == 2) {  // LDS read b128
                    if ((c & 3) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(d0) : "v"(laddr));
                    if ((c & 3) == 1) asm volatile("ds_read_b128 %0, %1" : "=v"(d1) : "v"(laddr));
                    if ((c & 3) == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(d2) : "v"(laddr));
                    if ((c & 3) == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(d3) : "v"(laddr));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = x0 + x1 + x2 + x3 + d0[0] + d1[0] + d2[0] + d3[0];
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND, int CNT>
void run(float* out) {
    const int iters = 4000, blocks = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, CNT>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, CNT>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mf = 8.0 * iters;  // MFMAs per wave
    printf("%s x%d per MFMA: %.3f ms  -> %.1f TFLOP/s, %.1f ns per MFMA\n",
           KIND == 0 ? "VALU" : KIND == 1 ? "SALU" : "ds_read_b128", CNT, ms,
           mf * 2048.0 * blocks * 4 / ms / 1e9, ms * 1e6 / mf);
}
int main() {
    float* out; (void)hipMalloc(&out, 4096 * 256 * 4);
    run<0, 0>(out); run<0, 1>(out); run<0, 2>(out); run<0, 4>(out); run<0, 6>(out); run<0, 8>(out); run<0, 12>(out);
    run<1, 2>(out); run<1, 8>(out); run<1, 16>(out);
    run<2, 1>(out); run<2, 2>(out); run<2, 4>(out);
    return 0;
}
