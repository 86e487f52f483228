// Round-trip times of the operations an XCD-local barrier is made of, one thread, dependent chain, uncontended line:
// returning atomic add at agent / workgroup / wavefront scope, loads with the sc bits (0, sc0, sc1, sc0 sc1) of a line that
// another workgroup of the same XCD keeps storing to (so an L1 hit would show as a stale value: `stale` counts reads that
// never change).  hipcc --offload-arch=gfx950 -O3 l2_rtt.hip -o l2_rtt.bin
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ unsigned long long wall() { return __builtin_amdgcn_s_memrealtime(); }

template <int SCOPE>
__global__ void k_atomic(unsigned* w, int n, unsigned long long* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned v = 0;
    const unsigned long long t0 = wall();
    for (int i = 0; i < n; ++i) v += __hip_atomic_fetch_add(w + (v & 1u) * 0, 1u, __ATOMIC_RELAXED, SCOPE);
    const unsigned long long t1 = wall();
    out[0] = t1 - t0;
    out[1] = v;
}

template <int AUX>
__global__ void k_load(unsigned* w, int n, unsigned long long* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned v = 0, acc = 0;
    const unsigned long long t0 = wall();
    for (int i = 0; i < n; ++i) {
        unsigned* p = w + (v & 1u) * 0;
        if (AUX == 0) asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        if (AUX == 1) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        if (AUX == 16) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        if (AUX == 17) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        if (AUX == 2) asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        acc += v;
    }
    const unsigned long long t1 = wall();
    out[0] = t1 - t0;
    out[1] = acc;
}

// no-return atomic followed by a dependent sc1 load of the same word (what a poller-less arrival + first poll costs)
__global__ void k_noret_then_load(unsigned* w, int n, unsigned long long* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned v = 0, acc = 0;
    const unsigned one = 1u;
    const unsigned long long t0 = wall();
    for (int i = 0; i < n; ++i) {
        unsigned* p = w + (v & 1u) * 0;
        asm volatile("global_atomic_add %1, %2, off\n\tglobal_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(one) : "memory");
        acc += v;
    }
    const unsigned long long t1 = wall();
    out[0] = t1 - t0;
    out[1] = acc;
}

// a plain store followed by a dependent sc1 load of the same word
__global__ void k_store_then_load(unsigned* w, int n, unsigned long long* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned v = 0, acc = 0;
    const unsigned long long t0 = wall();
    for (int i = 0; i < n; ++i) {
        unsigned* p = w + (v & 1u) * 0;
        asm volatile("global_store_dword %1, %2, off sc1\n\tglobal_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(acc) : "memory");
        acc += v;
    }
    const unsigned long long t1 = wall();
    out[0] = t1 - t0;
    out[1] = acc;
}

int main() {
    unsigned* w;
    unsigned long long* out;
    (void)hipMalloc(&w, 4096);
    (void)hipMalloc(&out, 64);
    (void)hipMemset(w, 0, 4096);
    const int n = 2000;
    unsigned long long h[2];
#define RUN(name, kern)                                                                      \
    for (int rep = 0; rep < 2; ++rep) {                                                       \
        hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, w, n, out);                         \
        (void)hipDeviceSynchronize();                                                         \
        (void)hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);                                   \
        if (rep) printf("{\"ubench\": \"l2_rtt\", \"op\": \"%s\", \"ns_per_op\": %.1f}\n", name, (double)h[0] * 10.0 / n); \
    }
    RUN("atomic add, returning, agent scope", (k_atomic<__HIP_MEMORY_SCOPE_AGENT>));
    RUN("atomic add, returning, workgroup scope", (k_atomic<__HIP_MEMORY_SCOPE_WORKGROUP>));
    RUN("atomic add, returning, wavefront scope", (k_atomic<__HIP_MEMORY_SCOPE_WAVEFRONT>));
    RUN("atomic add, returning, system scope", (k_atomic<__HIP_MEMORY_SCOPE_SYSTEM>));
    RUN("load, no sc bits", (k_load<0>));
    RUN("load, sc0", (k_load<1>));
    RUN("load, sc1", (k_load<16>));
    RUN("load, sc0 sc1", (k_load<17>));
    RUN("load, nt", (k_load<2>));
    RUN("no-return atomic + dependent sc1 load", k_noret_then_load);
    RUN("sc1 store + dependent sc1 load", k_store_then_load);
    return 0;
}
