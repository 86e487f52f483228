// How many 1-KB global -> LDS DMA requests (global_load_lds_dwordx4) can a wave / a CU have in flight before the ISSUE stalls?
// One 512-thread workgroup per CU (256), `nw` of its waves issue `n` requests each, back to back, from memory no other CU
// touches (cold: the data comes through the fabric); wave-level stamps of the 100 MHz wall clock: after the last issue, after
// s_waitcnt vmcnt(0).  Also the same with plain register loads (global_load_dwordx4) for comparison.
//   hipcc --offload-arch=gfx950 -O3 dma_issue.hip -o dma_issue.bin && ./dma_issue.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__device__ __forceinline__ unsigned long long wall() { return __builtin_amdgcn_s_memrealtime(); }
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N, bool DMA>
__global__ __launch_bounds__(512) void k(const float* src, int nw, unsigned long long* out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    __syncthreads();
    if (w >= nw) return;
    const float* base = src + ((size_t)blockIdx.x * 8 + w) * N * 256;  // N KB per wave, private
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 v[DMA ? 1 : N];
    const unsigned long long t0 = wall();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (DMA) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(base + i * 256) + lane * 16),
                                             (lds_ptr_t)(__attribute__((address_space(3))) char*)(unsigned long)(lds0 + (unsigned)((w * N + i) * 1024)), 16, 0, 0);
        } else {
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v[DMA ? 0 : i]) : "v"((unsigned)lane * 16u), "s"(base + i * 256) : "memory");
        }
    }
    const unsigned long long t1 = wall();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = wall();
    if (!DMA) {
#pragma unroll
        for (int i = 0; i < N; ++i) acc += v[i];
        if (acc[0] == 12345.f) sink[0] = acc[1];
    }
    if (lane == 0) {
        out[((size_t)blockIdx.x * 8 + w) * 2] = t1 - t0;
        out[((size_t)blockIdx.x * 8 + w) * 2 + 1] = t2 - t0;
    }
}

template <int N, bool DMA>
void run(const float* src, unsigned long long* out, float* sink, int nw) {
    std::vector<unsigned long long> h(256 * 8 * 2);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<N, DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    for (int rep = 0; rep < 2; ++rep) {  // (different memory each repetition would be colder still; the 256-MB cache holds these 50 MB)
        (void)hipMemset(out, 0, h.size() * 8);
        hipLaunchKernelGGL((k<N, DMA>), dim3(256), dim3(512), DMA ? 8 * N * 1024 : 1024, 0, src, nw, out, sink);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> iss, done;
    for (int b = 0; b < 256; ++b)
        for (int w = 0; w < nw; ++w) iss.push_back(h[(b * 8 + w) * 2] / 100.0), done.push_back(h[(b * 8 + w) * 2 + 1] / 100.0);
    std::sort(iss.begin(), iss.end());
    std::sort(done.begin(), done.end());
    printf("{\"ubench\": \"dma_issue\", \"kind\": \"%s\", \"waves\": %d, \"requests_per_wave\": %d, \"KB_per_CU\": %d, \"issue_us_median\": %.2f, "
           "\"issue_us_max\": %.2f, \"landed_us_median\": %.2f, \"landed_us_max\": %.2f}\n",
           DMA ? "lds dma" : "register load", nw, N, nw * N, iss[iss.size() / 2], iss.back(), done[done.size() / 2], done.back());
}

int main() {
    float *src, *sink;
    unsigned long long* out;
    (void)hipMalloc(&src, (size_t)256 * 8 * 32 * 1024 + 4096);
    (void)hipMalloc(&sink, 64);
    (void)hipMalloc(&out, 256 * 8 * 2 * 8);
    (void)hipMemset(src, 0, (size_t)256 * 8 * 32 * 1024);
    for (int nw : {1, 4, 8}) {
        run<4, true>(src, out, sink, nw);
        run<8, true>(src, out, sink, nw);
        run<12, true>(src, out, sink, nw);
        run<16, true>(src, out, sink, nw);
        if (nw <= 4) run<24, true>(src, out, sink, nw);
        run<8, false>(src, out, sink, nw);
        run<16, false>(src, out, sink, nw);
        if (nw <= 4) run<24, false>(src, out, sink, nw);
    }
    return 0;
}
