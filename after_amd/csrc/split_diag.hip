// Test / measurement entry, not on any product path: C = A W^T for fp32 A [M][K], W [N][K] through the matrix pipes in the three
// arithmetics the denoiser's Linears use, one wave per 16 x 16 output block, operands split on the fly --
//   mode 0: v_mfma_f32_16x16x4_f32 (the exact fp32 fma chain of gemm.hip)
//   mode 1: three bf16 planes per operand, six v_mfma_f32_16x16x32_bf16 per 32-deep block (gemm_x6_pipe.h)
//   mode 2: two fp16 pieces per operand scaled by exact powers of two, three v_mfma_f32_16x16x32_f16 per block (gemm_h3_pipe.h)
// so that tests/test_gemm_gpu.py can hold the split forms' error against fp64 to the fp32 chain's on the same data.
#include <hip/hip_runtime.h>

#include "common.h"
#include "gemm_h3_pipe.h"

namespace after {
namespace {

__global__ __launch_bounds__(64) void split_diag_kernel(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ C, int M, int N,
                                                        int K, int mode, float sa, float sw) {
    const int lane = threadIdx.x, r = lane & 15, kq = lane >> 4;
    const int bm = blockIdx.x, bn = blockIdx.y;
    const float* arow = A + (size_t)(16 * bm + r) * K;
    const float* wrow = W + (size_t)(16 * bn + r) * K;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (mode == 0) {
        for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[k + kq], wrow[k + kq], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 32) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(arow + k + 8 * kq), a1 = *reinterpret_cast<const f32x4*>(arow + k + 8 * kq + 4);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wrow + k + 8 * kq), w1 = *reinterpret_cast<const f32x4*>(wrow + k + 8 * kq + 4);
            if (mode == 1) {
                uint2 ah[2], am[2], al[2], wh[2], wm[2], wl[2];
                x6_split4(a0[0], a0[1], a0[2], a0[3], ah[0], am[0], al[0]);
                x6_split4(a1[0], a1[1], a1[2], a1[3], ah[1], am[1], al[1]);
                x6_split4(w0[0], w0[1], w0[2], w0[3], wh[0], wm[0], wl[0]);
                x6_split4(w1[0], w1[1], w1[2], w1[3], wh[1], wm[1], wl[1]);
                auto f = [](const uint2 (&p)[2]) { return __builtin_bit_cast(bf16x8, u32x4{p[0].x, p[0].y, p[1].x, p[1].y}); };
                // (the product order of gemm_x6_pipe.h: smallest terms first)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f(ah), f(wl), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f(al), f(wh), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f(am), f(wm), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f(ah), f(wm), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f(am), f(wh), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f(ah), f(wh), acc, 0, 0, 0);
            } else {
                uint2 ah[2], al[2], wh[2], wl[2];
                h3_split4(a0[0] * sa, a0[1] * sa, a0[2] * sa, a0[3] * sa, ah[0], al[0]);
                h3_split4(a1[0] * sa, a1[1] * sa, a1[2] * sa, a1[3] * sa, ah[1], al[1]);
                h3_split4(w0[0] * sw, w0[1] * sw, w0[2] * sw, w0[3] * sw, wh[0], wl[0]);
                h3_split4(w1[0] * sw, w1[1] * sw, w1[2] * sw, w1[3] * sw, wh[1], wl[1]);
                auto f = [](const uint2 (&p)[2]) { return __builtin_bit_cast(f16x8, u32x4{p[0].x, p[0].y, p[1].x, p[1].y}); };
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(f(al), f(wh), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(f(ah), f(wh), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(f(ah), f(wl), acc, 0, 0, 0);
            }
        }
        if (mode == 2) acc = acc * (1.0f / (sa * sw));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) C[(size_t)(16 * bm + 4 * kq + i) * N + 16 * bn + r] = acc[i];
}

}  // namespace
}  // namespace after

extern "C" int after_diag_split_gemm(const float* A, const float* W, float* C, int M, int N, int K, int mode, float a_bound, float w_bound,
                                     void* stream) {
    AFTER_REQUIRE(A && W && C && M > 0 && N > 0 && K > 0 && M % 16 == 0 && N % 16 == 0 && K % 32 == 0, AFTER_E_INVALID,
                  "diag_split_gemm: M, N multiples of 16, K of 32");
    AFTER_REQUIRE(mode >= 0 && mode <= 2, AFTER_E_INVALID, "diag_split_gemm: mode 0 / 1 / 2");
    const float sa = after::h3_scale_for(a_bound), sw = after::h3_scale_for(w_bound);
    hipLaunchKernelGGL(after::split_diag_kernel, dim3(M / 16, N / 16), dim3(64), 0, (hipStream_t)stream, A, W, C, M, N, K, mode, sa, sw);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}
