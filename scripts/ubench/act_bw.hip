// What an elementwise pass of the codec's size can reach on this chip: read an fp32 tensor [rows][C], write 6 bytes per element
// (three bf16 planes in 1-KB pieces, as act_pad_x6 does), with and without the pass's arithmetic (SnakeBeta + three-way split).
// Access pattern of act_pad_x6 (lane = row l / 4, eight channels 8 (l % 4)) against a fully coalesced one (lane = 16 bytes of a
// 1-KB run).  hipcc --offload-arch=gfx950 -O3 act_bw.hip -o act_bw.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned bf16_rne(float x) {
    unsigned u = __builtin_bit_cast(unsigned, x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
}
__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
    h = bf16_rne(x);
    const float r1 = x - __builtin_bit_cast(float, h);
    m = bf16_rne(r1);
    const float r2 = r1 - __builtin_bit_cast(float, m);
    l = bf16_rne(r2);
}
__device__ __forceinline__ float snake(float x, float a, float ib) {
    const float s = __sinf(x * a);
    return x + ib * s * s;
}

// MODE 0: copy only (pattern of act_pad_x6); 1: + arithmetic; 2: coalesced copy (lane = 16 contiguous bytes); 3: coalesced + arithmetic
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ x, unsigned short* __restrict__ y, int rows, int C, int iters) {
    const int lane = threadIdx.x & 63;
    const int nkb = C / 32, nrb = rows / 16;
    const long long nitems = (long long)nkb * nrb;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    for (long long it = wave; it < nitems; it += nwaves) {
        const int kb = (int)(it % nkb);
        const long long rb = it / nkb;
        f32x4 v0, v1;
        if (MODE < 2) {
            const int r = lane >> 2, c0 = kb * 32 + 8 * (lane & 3);
            const float* p = x + (rb * 16 + r) * C + c0;
            v0 = *reinterpret_cast<const f32x4*>(p);
            v1 = *reinterpret_cast<const f32x4*>(p + 4);
        } else {  // rows of 128 bytes: lanes 0 .. 7 one row, two passes of eight rows
            const int r = lane >> 3, c0 = kb * 32 + 4 * (lane & 7);
            v0 = *reinterpret_cast<const f32x4*>(x + (rb * 16 + r) * C + c0);
            v1 = *reinterpret_cast<const f32x4*>(x + (rb * 16 + 8 + r) * C + c0);
        }
        unsigned h[8], m[8], l[8];
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) {
            float xv = k2 < 4 ? v0[k2] : v1[k2 - 4];
            if (MODE & 1) {
                xv = snake(xv * 1.01f + 0.1f, 1.3f, 0.7f);
                split3(xv, h[k2], m[k2], l[k2]);
            } else {
                h[k2] = __builtin_bit_cast(unsigned, xv);
                m[k2] = h[k2] ^ 0x5555u;
                l[k2] = h[k2] + 3u;
            }
        }
        unsigned short* bp = y + ((rb * nkb + kb) * 3) * 512 + lane * 8;
        auto pk = [&](unsigned (&a)[8]) { return uint4{(a[0] >> 16) | (a[1] & 0xffff0000u), (a[2] >> 16) | (a[3] & 0xffff0000u), (a[4] >> 16) | (a[5] & 0xffff0000u), (a[6] >> 16) | (a[7] & 0xffff0000u)}; };
        *reinterpret_cast<uint4*>(bp) = pk(h);
        *reinterpret_cast<uint4*>(bp + 512) = pk(m);
        *reinterpret_cast<uint4*>(bp + 1024) = pk(l);
    }
}

template <int MODE>
void run(const char* name, const float* x, unsigned short* y, int rows, int C, int nwg) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(nwg), dim3(256), 0, 0, x, y, rows, C, 1);
    hipEventRecord(e0, 0);
    const int reps = 20;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k<MODE>, dim3(nwg), dim3(256), 0, 0, x, y, rows, C, 1);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, bytes = (double)rows * C * 10.0;
    printf("{\"ubench\": \"act_bw\", \"mode\": \"%s\", \"rows\": %d, \"C\": %d, \"workgroups\": %d, \"us\": %.1f, \"TBps\": %.2f}\n", name, rows, C, nwg, us,
           bytes / us * 1e-6);
}

int main() {
    const int C = 384, rows = 8 * 4128;  // eight clips x (4096 + halo) frames x 384 channels: 12.7 M elements
    float* x;
    unsigned short* y;
    hipMalloc(&x, (size_t)rows * C * 4);
    hipMalloc(&y, (size_t)rows * C * 6);
    hipMemset(x, 0, (size_t)rows * C * 4);
    for (int nwg : {512, 1024, 2048, 6192}) {
        run<0>("copy, act_pad_x6 pattern", x, y, rows, C, nwg);
        run<1>("arithmetic, act_pad_x6 pattern", x, y, rows, C, nwg);
        run<2>("copy, coalesced rows", x, y, rows, C, nwg);
        run<3>("arithmetic, coalesced rows", x, y, rows, C, nwg);
    }
    return 0;
}
