// Issue rate of the split-bf16 inner loop of sample_seg_kernel in isolation: per 32-deep k-block 54 v_mfma_f32_16x16x32_bf16
// (3 column tiles x 3 row blocks x 6 plane products, 9 independent accumulators) with operands already in registers --
// (a) MFMAs only, (b) with the in-register three-way split of the three weight fragments in front of each block (the VALU
// work of seg_split8) -- for 1 and 2 waves per SIMD (256 / 512 threads).  Cycles per MFMA from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form mfma_split_loop.hip -o mfma_split_loop.bin
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split4(float x0, float x1, float x2, float x3, uint2& h, uint2& m, uint2& l) {
    const f2 v0 = {x0, x1}, v1 = {x2, x3};
    h.x = __builtin_bit_cast(unsigned, __builtin_convertvector(v0, bf2));
    h.y = __builtin_bit_cast(unsigned, __builtin_convertvector(v1, bf2));
    const f2 r0 = {x0 - __uint_as_float(h.x << 16), x1 - __uint_as_float(h.x & 0xFFFF0000u)};
    const f2 r1 = {x2 - __uint_as_float(h.y << 16), x3 - __uint_as_float(h.y & 0xFFFF0000u)};
    m.x = __builtin_bit_cast(unsigned, __builtin_convertvector(r0, bf2));
    m.y = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf2));
    const f2 t0 = {r0[0] - __uint_as_float(m.x << 16), r0[1] - __uint_as_float(m.x & 0xFFFF0000u)};
    const f2 t1 = {r1[0] - __uint_as_float(m.y << 16), r1[1] - __uint_as_float(m.y & 0xFFFF0000u)};
    l.x = __builtin_bit_cast(unsigned, __builtin_convertvector(t0, bf2));
    l.y = __builtin_bit_cast(unsigned, __builtin_convertvector(t1, bf2));
}

// RB: row blocks a weight fragment serves (3: one row half per wave, the shipped kernel; 6: both halves in ONE wave per SIMD --
// the "fat wave" layout: every fragment fetched and split once, 108 MFMAs per k-block behind one split)
template <int SPLIT, int RB = 3>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, float seed, int iters) {
    constexpr int WP[6] = {2, 0, 1, 1, 0, 0}, AP[6] = {0, 2, 1, 0, 1, 0};
    f32x4 acc[3 * RB];
#pragma unroll
    for (int p = 0; p < 3 * RB; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 ap[RB][3], wp[3][3];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) ap[i][p] = u32x4{threadIdx.x + i, 0x3f803f80u, 0x3f803f80u + p, 0x3f803f80u};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) wp[i][p] = u32x4{0x3f803f80u, threadIdx.x + p, 0x3f803f80u, 0x3f803f80u + i};
    f32x4 wr[3][2];
#pragma unroll
    for (int j = 0; j < 3; ++j) wr[j][0] = wr[j][1] = f32x4{seed + j, seed * 2.f, seed * 3.f, (float)threadIdx.x};
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (SPLIT) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                uint2 h0, m0, l0, h1, m1, l1;
                split4(wr[j][0][0], wr[j][0][1], wr[j][0][2], wr[j][0][3], h0, m0, l0);
                split4(wr[j][1][0], wr[j][1][1], wr[j][1][2], wr[j][1][3], h1, m1, l1);
                wp[j][0] = u32x4{h0.x, h0.y, h1.x, h1.y};
                wp[j][1] = u32x4{m0.x, m0.y, m1.x, m1.y};
                wp[j][2] = u32x4{l0.x, l0.y, l1.x, l1.y};
                wr[j][0] += 1.f;  // (a new fragment every k-block)
                wr[j][1] += 1.f;
            }
        }
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int i = 0; i < RB; ++i)
                    acc[j * RB + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wp[j][WP[p]]),
                                                                              __builtin_bit_cast(bf16x8, ap[i][AP[p]]), acc[j * RB + i], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    f32x4 s = acc[0];
#pragma unroll
    for (int p = 1; p < 3 * RB; ++p) s += acc[p];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    float* out;
    long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMalloc(&cyc, 8);
    const int iters = 64;
    for (int split = 0; split < 2; ++split)
        for (int threads : {256, 512}) {
            for (int rep = 0; rep < 2; ++rep) {
                if (split) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.f, iters);
                else hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.f, iters);
                (void)hipDeviceSynchronize();
            }
            long long c;
            (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const int wps = threads / 256;  // waves per SIMD
            printf("{\"ubench\": \"mfma_split_loop\", \"weight_split\": %d, \"waves_per_simd\": %d, \"cycles_per_mfma_per_wave\": %.1f, "
                   "\"cycles_per_mfma_per_simd\": %.1f}\n", split, wps, (double)c / (iters * 54), (double)c / (iters * 54 * wps));
        }
    for (int rep = 0; rep < 2; ++rep) {  // the fat wave: one wave per SIMD, six row blocks behind one split
        hipLaunchKernelGGL((k<1, 6>), dim3(256), dim3(256), 0, 0, out, cyc, 1.f, iters);
        (void)hipDeviceSynchronize();
    }
    long long c;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("{\"ubench\": \"mfma_split_loop\", \"weight_split\": 1, \"waves_per_simd\": 1, \"row_blocks\": 6, \"cycles_per_mfma_per_simd\": %.1f}\n",
           (double)c / (iters * 108));
    return 0;
}
