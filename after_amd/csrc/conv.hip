// Create-time satellites of the 1-D conv path (conv_tm.hip): weight-norm folding and packing of
// conv / transposed-conv weights, BatchNorm(eval) -> per-channel affine, 1 / (beta + eps) of
// SnakeBeta.  (The convs themselves: conv_tm.hip.)
//
// Reference ops: weight_norm'd Conv1d / ConvTranspose1d of the codec
// (after/autoencoder/networks/SimpleNetsStream.py:32-70,150-194), BatchNorm1d of V2ConvBlock1D /
// TDNNBlock (after/diffusion/networks/encoder.py:25-71, ecapa_encoder.py:139), SnakeBeta
// (core.py:217-260).
#include <cstdio>
#include <cstdlib>

#include "conv.h"

namespace after {
namespace {

__global__ void bn_affine_kernel(const float* w, const float* b, const float* rm, const float* rv,
                                 float* scale, float* shift, int C, int B, float eps) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= C * B) return;
    const int c = i % C;
    const float sc = w[c] / sqrtf(rv[c] + eps);
    scale[i] = sc;
    shift[i] = b[c] - rm[c] * sc;
}

// scale[row] = g[row] / ||v[row, :]||   (torch weight_norm, dim 0)
__global__ __launch_bounds__(256) void wn_scale_kernel(const float* v, const float* g, float* scale,
                                                       int rows, int inner) {
    __shared__ float sh[4];
    const int r = blockIdx.x;
    float q = 0.f;
    for (int i = threadIdx.x; i < inner; i += 256) {
        const float x = v[(size_t)r * inner + i];
        q += x * x;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = q;
    __syncthreads();
    if (threadIdx.x == 0) scale[r] = g[r] / sqrtf(sh[0] + sh[1] + sh[2] + sh[3]);
}

__global__ void pack_conv_kernel(const float* v, const float* scale, float* out, int Cout, int Cin,
                                 int k, int Cin_pad) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)Cout * k * Cin_pad;
    if (idx >= total) return;
    const int ci = idx % Cin_pad;
    const int tap = (idx / Cin_pad) % k;
    const int co = idx / ((size_t)Cin_pad * k);
    float val = 0.f;
    if (ci < Cin) val = v[((size_t)co * Cin + ci) * k + tap] * (scale ? scale[co] : 1.f);
    out[idx] = val;
}

// ConvTranspose1d(k = 2f, stride f, padding p = f/2) as f phase convolutions with 2 taps:
//   y[co, n*f + r] = sum_ci w[ci, co, phi] x[ci, n + c] + w[ci, co, phi + f] x[ci, n + c - 1]
//   with q = r + p, phi = q % f, c = q / f.      out[r][co][tap][ci]: tap 0 <-> x[n + c - 1]
__global__ void pack_convT_kernel(const float* v, const float* scale, float* out, int Cin, int Cout,
                                  int f, int Cin_pad, int pad) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)f * Cout * 2 * Cin_pad;
    if (idx >= total) return;
    const int ci = idx % Cin_pad;
    const int tap = (idx / Cin_pad) % 2;
    const int co = (idx / ((size_t)Cin_pad * 2)) % Cout;
    const int r = idx / ((size_t)Cin_pad * 2 * Cout);
    const int q = r + pad, phi = q % f;
    const int kidx = tap == 0 ? phi + f : phi;
    float val = 0.f;
    if (ci < Cin) val = v[((size_t)ci * Cout + co) * (2 * f) + kidx] * (scale ? scale[ci] : 1.f);
    out[idx] = val;
}

__global__ void inv_beta_kernel(const float* beta, float* out, int C) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < C) out[i] = 1.0f / (beta[i] + 0.000000001f);
}

}  // namespace

int launch_bn_affine(const float* w, const float* b, const float* rm, const float* rv, float* scale,
                     float* shift, int C, int B, float eps, hipStream_t s) {
    hipLaunchKernelGGL(bn_affine_kernel, dim3(cdiv(C * B, 256)), dim3(256), 0, s, w, b, rm, rv, scale,
                       shift, C, B, eps);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

static int wn_scale(const float* v, const float* g, float** scale_out, int rows, int inner,
                    hipStream_t s) {
    *scale_out = nullptr;
    if (!g) return AFTER_OK;
    float* sc = nullptr;
    AFTER_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&sc), rows * sizeof(float)));
    hipLaunchKernelGGL(wn_scale_kernel, dim3(rows), dim3(256), 0, s, v, g, sc, rows, inner);
    *scale_out = sc;
    return AFTER_OK;
}

int pack_conv_weight(const float* v, const float* g, float* out, int Cout, int Cin, int k,
                     int Cin_pad, hipStream_t s) {
    float* sc = nullptr;
    AFTER_TRY(wn_scale(v, g, &sc, Cout, Cin * k, s));
    const size_t total = (size_t)Cout * k * Cin_pad;
    hipLaunchKernelGGL(pack_conv_kernel, dim3((unsigned)cdivll(total, 256)), dim3(256), 0, s, v, sc,
                       out, Cout, Cin, k, Cin_pad);
    AFTER_HIP_CHECK(hipGetLastError());
    AFTER_HIP_CHECK(hipStreamSynchronize(s));
    if (sc) (void)hipFree(sc);
    return AFTER_OK;
}

int pack_convT_weight(const float* v, const float* g, float* out, int Cin, int Cout, int f,
                      int Cin_pad, hipStream_t s, int pad) {
    if (pad < 0) pad = f / 2;
    float* sc = nullptr;
    AFTER_TRY(wn_scale(v, g, &sc, Cin, Cout * 2 * f, s));
    const size_t total = (size_t)f * Cout * 2 * Cin_pad;
    hipLaunchKernelGGL(pack_convT_kernel, dim3((unsigned)cdivll(total, 256)), dim3(256), 0, s, v, sc,
                       out, Cin, Cout, f, Cin_pad, pad);
    AFTER_HIP_CHECK(hipGetLastError());
    AFTER_HIP_CHECK(hipStreamSynchronize(s));
    if (sc) (void)hipFree(sc);
    return AFTER_OK;
}

int snake_inv_beta(const float* beta, float* out, int C, hipStream_t s) {
    hipLaunchKernelGGL(inv_beta_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, beta, out, C);
    AFTER_HIP_CHECK(hipGetLastError());
    return AFTER_OK;
}

}  // namespace after
