// fp32 products through the bf16 matrix pipe: (a) issue rate of the bf16 MFMAs against the fp32 one,
// (b) accuracy of a K = 512 dot product with each fp32 operand split into three bf16 pieces and 3 / 6 / 9
// of the piece products accumulated in fp32, against fp64, next to the plain fp32 MFMA.
//   hipcc --offload-arch=gfx950 -O3 bf16_split.hip -o bf16_split.bin && ./bf16_split.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int ACC>
__global__ __launch_bounds__(256) void rate_f32(float* out, int iters, float a, float b) {
    f32x4 acc[ACC];
    for (int i = 0; i < ACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < ACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int ACC>
__global__ __launch_bounds__(256) void rate_bf16_k16(float* out, int iters, short a, short b) {
    f32x4 acc[ACC];
    for (int i = 0; i < ACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    s16x4 av = {a, a, a, a}, bv = {b, b, b, b};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, bv, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < ACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int ACC>
__global__ __launch_bounds__(256) void rate_bf16_k32(float* out, int iters, float a, float b) {
    f32x4 acc[ACC];
    for (int i = 0; i < ACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    bf16x8 av, bv;
    for (int i = 0; i < 8; ++i) {
        av[i] = (__bf16)a;
        bv[i] = (__bf16)b;
    }
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < ACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__device__ __forceinline__ short bf16_rne(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (short)(u >> 16);
}
__device__ __forceinline__ float bf16_f(short h) { return __uint_as_float(((unsigned)(unsigned short)h) << 16); }
__device__ __forceinline__ void split3(float x, short& h, short& m, short& l) {
    h = bf16_rne(x);
    const float r1 = x - bf16_f(h);
    m = bf16_rne(r1);
    const float r2 = r1 - bf16_f(m);
    l = bf16_rne(r2);
}

// One wave computes a 16 x 16 block of C = A[16][K] * B[K][16].  mode 0: fp32 MFMA; 3 / 6 / 9: number of
// bf16 piece products (hh | + hm, mh | + mm, hl, lh | + ml, lm, ll), smallest terms accumulated first.
__global__ __launch_bounds__(64) void dot_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                 float* __restrict__ C, int K, int mode) {
    const int lane = threadIdx.x, row = lane & 15, kq = lane >> 4;
    const int bm = blockIdx.x, bn = blockIdx.y, N = gridDim.y * 16;
    f32x4 acc = {0, 0, 0, 0};
    if (mode == 0) {
        for (int k = 0; k < K; k += 4) {
            const float a = A[(size_t)(bm * 16 + row) * K + k + kq];
            const float b = B[(size_t)(k + kq) * N + bn * 16 + row];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        }
    } else if (mode < 12) {
        f32x4 lo = {0, 0, 0, 0}, mid = {0, 0, 0, 0};
        for (int k = 0; k < K; k += 16) {
            s16x4 ah, am, al, bh, bm_, bl;
            for (int j = 0; j < 4; ++j) {
                short h, m, l;
                split3(A[(size_t)(bm * 16 + row) * K + k + 4 * kq + j], h, m, l);
                ah[j] = h; am[j] = m; al[j] = l;
                split3(B[(size_t)(k + 4 * kq + j) * N + bn * 16 + row], h, m, l);
                bh[j] = h; bm_[j] = m; bl[j] = l;
            }
            if (mode >= 9) {
                lo = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bl, lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(am, bl, lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bm_, lo, 0, 0, 0);
            }
            if (mode >= 6) {
                mid = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(am, bm_, mid, 0, 0, 0);
                mid = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bl, mid, 0, 0, 0);
                mid = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bh, mid, 0, 0, 0);
            }
            mid = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bm_, mid, 0, 0, 0);
            mid = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(am, bh, mid, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, acc, 0, 0, 0);
        }
        acc += mid + lo;
    }
    if (mode >= 12) {  // fp16 x 2 pieces: 13 = hh + hl + lh, 14 = + ll; operands pre-scaled by sa / sb (powers of two)
        typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
        acc = f32x4{0, 0, 0, 0};
        f32x4 mid = {0, 0, 0, 0};
        const float sa = 16.f, sb = 512.f;
        for (int k = 0; k < K; k += 16) {
            h16x4 ah, al, bh, bl;
            for (int j = 0; j < 4; ++j) {
                const float x = A[(size_t)(bm * 16 + row) * K + k + 4 * kq + j] * sa;
                const _Float16 h = (_Float16)x;
                ah[j] = h;
                al[j] = (_Float16)(x - (float)h);
                const float y = B[(size_t)(k + 4 * kq + j) * N + bn * 16 + row] * sb;
                const _Float16 g = (_Float16)y;
                bh[j] = g;
                bl[j] = (_Float16)(y - (float)g);
            }
            if (mode >= 14) mid = __builtin_amdgcn_mfma_f32_16x16x16f16(al, bl, mid, 0, 0, 0);
            mid = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bl, mid, 0, 0, 0);
            mid = __builtin_amdgcn_mfma_f32_16x16x16f16(al, bh, mid, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, acc, 0, 0, 0);
        }
        acc = (acc + mid) * (1.0f / (sa * sb));
    }
    // D layout of the 16x16 MFMAs: lane l holds D[4 (l >> 4) + r][l & 15]
    for (int r = 0; r < 4; ++r) C[(size_t)(bm * 16 + 4 * kq + r) * N + bn * 16 + row] = acc[r];
}

template <int ACC>
__global__ __launch_bounds__(256) void rate_f16_k32(float* out, int iters, float a, float b) {
    typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
    f32x4 acc[ACC];
    for (int i = 0; i < ACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    h16x8 av, bv;
    for (int i = 0; i < 8; ++i) {
        av[i] = (_Float16)a;
        bv[i] = (_Float16)b;
    }
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < ACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
void run(const char* name, F launch, double flop_per_iter_per_wave, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.3f ms  %.1f TFLOP/s\n", name, ms, flop_per_iter_per_wave * iters * blocks * 4.0 / ms / 1e9);
}

int main() {
    float* out;
    hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 10000, blocks = 1024;
    run("fp32 16x16x4   acc=4", [&] { hipLaunchKernelGGL(rate_f32<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 8 * 4 * 2048.0, blocks, iters);
    run("bf16 16x16x16  acc=4", [&] { hipLaunchKernelGGL(rate_bf16_k16<4>, dim3(blocks), dim3(256), 0, 0, out, iters, (short)0x3f80, (short)0x4000); }, 8 * 4 * 8192.0, blocks, iters);
    run("bf16 16x16x32  acc=4", [&] { hipLaunchKernelGGL(rate_bf16_k32<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 8 * 4 * 16384.0, blocks, iters);
    run("fp16 16x16x32  acc=4", [&] { hipLaunchKernelGGL(rate_f16_k32<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 8 * 4 * 16384.0, blocks, iters);
    // accuracy
    const int M = 256, N = 256, K = 512;
    std::vector<float> hA((size_t)M * K), hB((size_t)K * N), hC((size_t)M * N);
    unsigned s = 7;
    auto rnd = [&]() {
        s = s * 1664525u + 1013904223u;
        return ((s >> 8) / 16777216.0f) * 2.f - 1.f;
    };
    auto gauss = [&]() {
        float v = 0;
        for (int i = 0; i < 6; ++i) v += rnd();
        return v * 0.7071f;
    };
    for (auto& v : hA) v = gauss() * 1.3f;
    for (size_t i = 0; i < hA.size(); i += 97) hA[i] *= 40.f;   // outliers (|x| up to ~150)
    for (size_t i = 5; i < hA.size(); i += 31) hA[i] *= 1e-4f;  // tiny entries
    for (auto& v : hB) v = gauss() * 0.044f;  // ~ N(0, 1 / sqrt(K))
    float *dA, *dB, *dC;
    hipMalloc(&dA, hA.size() * 4);
    hipMalloc(&dB, hB.size() * 4);
    hipMalloc(&dC, hC.size() * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    std::vector<double> ref((size_t)M * N);
    double rms = 0;
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < N; ++j) {
            double a = 0;
            for (int k = 0; k < K; ++k) a += (double)hA[(size_t)i * K + k] * (double)hB[(size_t)k * N + j];
            ref[(size_t)i * N + j] = a;
            rms += a * a;
        }
    rms = sqrt(rms / ref.size());
    for (int mode : {0, 3, 6, 9, 13, 14}) {
        hipLaunchKernelGGL(dot_kernel, dim3(M / 16, N / 16), dim3(64), 0, 0, dA, dB, dC, K, mode);
        hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
        double mx = 0, sq = 0;
        for (size_t i = 0; i < ref.size(); ++i) {
            const double e = fabs((double)hC[i] - ref[i]);
            mx = e > mx ? e : mx;
            sq += e * e;
        }
        printf("K=%d %-22s max abs err %.3e  rms err %.3e  (rms of C %.3f -> relative %.2e / %.2e)\n", K,
               mode == 0 ? "fp32 MFMA" : mode == 3 ? "bf16 x 3 (hh,hm,mh)" : mode == 6 ? "bf16 x 6" : mode == 9 ? "bf16 x 9" : mode == 13 ? "fp16 x 3 (hh,hl,lh)" : "fp16 x 4", mx,
               sqrt(sq / ref.size()), rms, mx / rms, sqrt(sq / ref.size()) / rms);
    }
    return 0;
}
