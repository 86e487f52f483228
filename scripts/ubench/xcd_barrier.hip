// XCD-hierarchical device-wide barrier inside one kernel (MI355X_MICROARCH.md price list, row "barrier-xcd"):
// per-XCC arrival counter -> the XCC's last arriver (its leader for this round) releases the XCD's L2 once
// (buffer_wbl2: every workgroup of the XCD has drained its stores into that L2 before arriving) -> top counter ->
// the leader polls the top counter, acquires, and publishes the round in its XCC's generation word; every other
// workgroup polls that one word (relaxed, sc1) and acquires once.  Against grid_barrier.hip's single agent-scope
// counter (18.5 us on this part): one release per XCD instead of one per workgroup, 8 pollers on the top word
// instead of 256.  Each round every workgroup publishes a 128-byte record that a workgroup of ANOTHER XCD reads
// after the barrier (visibility check, the guide's "re-reading a 128-B record" variant) -- or nothing (bare cost).
//   hipcc --offload-arch=gfx950 -O3 xcd_barrier.hip -o xcd_barrier.bin && ./xcd_barrier.bin
#include <hip/hip_runtime.h>

#include <cstdio>

typedef __attribute__((address_space(1))) unsigned gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct BarrierState {
    unsigned arrive[8][32];  // [xcc][0]: arrivals of this XCC (own 128-byte line each)
    unsigned gen[8][32];     // [xcc][0]: last completed round, published by the XCC's leader
    unsigned top[32];        // [0]: XCC leaders that arrived
    unsigned pop[8][32];     // [xcc][0]: workgroups resident on the XCC (census)
    unsigned census[32];     // [0]: workgroups counted
    unsigned timeout[32];    // [0]: set when a spin gave up
};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}

__device__ __forceinline__ bool spin_until(unsigned* word, unsigned want, unsigned* timeout) {
    for (unsigned spins = 0;; ++spins) {
        if (__hip_atomic_load(word, RLX_AGENT) >= want) return true;
        __builtin_amdgcn_s_sleep(1);
        if (spins > (1u << 22)) {
            __hip_atomic_store(timeout, 1u, RLX_AGENT);
            return false;
        }
    }
}

// round = 1, 2, ...; n_xcc = workgroups on this XCC; all threads of the workgroup call it
__device__ __forceinline__ bool xcd_barrier(BarrierState* st, unsigned xcc, unsigned n_xcc, unsigned n_xccs, unsigned round) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its stores have reached the XCD's L2
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(&st->arrive[xcc][0], 1u, RLX_AGENT);
        if (ticket == round * n_xcc - 1) {  // last arriver of this XCC: the round's leader
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // write back the XCD's L2 once
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(&st->top[0], 1u, RLX_AGENT);
            ok = spin_until(&st->top[0], round * n_xccs, &st->timeout[0]);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(&st->gen[xcc][0], round, RLX_AGENT);
        } else {
            ok = spin_until(&st->gen[xcc][0], round, &st->timeout[0]);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    __syncthreads();
    return ok;
}

template <bool PUBLISH>
__global__ __launch_bounds__(256) void k_barrier(BarrierState* st, float* buf, int rounds, int* errs) {
    __shared__ unsigned s_n, s_nx;
    const unsigned xcc = xcc_id(), nb = gridDim.x;
    // census: how many workgroups live on each XCC (dispatch is observed round-robin, not promised)
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&st->pop[xcc][0], 1u, RLX_AGENT);
        __hip_atomic_fetch_add(&st->census[0], 1u, RLX_AGENT);
        spin_until(&st->census[0], nb, &st->timeout[0]);
        unsigned nx = 0;
        for (int x = 0; x < 8; ++x) nx += __hip_atomic_load(&st->pop[x][0], RLX_AGENT) ? 1u : 0u;
        s_n = __hip_atomic_load(&st->pop[xcc][0], RLX_AGENT);
        s_nx = nx;
    }
    __syncthreads();
    const unsigned n_xcc = s_n, n_xccs = s_nx;
    unsigned round = 0;
    for (int r = 0; r < rounds; ++r) {
        if (PUBLISH && threadIdx.x < 32) buf[(size_t)blockIdx.x * 32 + threadIdx.x] = (float)(r * 1000 + blockIdx.x);
        if (!xcd_barrier(st, xcc, n_xcc, n_xccs, ++round)) return;
        if (PUBLISH && threadIdx.x < 32) {
            const unsigned peer = (blockIdx.x + 37) % nb;  // 37 is odd: another XCD under round-robin dispatch
            const float v = buf[(size_t)peer * 32 + threadIdx.x];
            if (v != (float)(r * 1000 + peer)) atomicAdd(errs, 1);
        }
        if (!xcd_barrier(st, xcc, n_xcc, n_xccs, ++round)) return;  // the record may be overwritten next round
    }
}

int main() {
    BarrierState* st;
    float* buf;
    int* errs;
    (void)hipMalloc(&st, sizeof(BarrierState));
    (void)hipMalloc(&buf, 1024 * 32 * 4);
    (void)hipMalloc(&errs, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int rounds = 2000;
    for (int blocks : {256, 512}) {
        for (int publish = 0; publish < 2; ++publish) {
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipMemset(st, 0, sizeof(BarrierState));
                (void)hipMemset(errs, 0, 4);
                (void)hipEventRecord(e0);
                if (publish) hipLaunchKernelGGL(k_barrier<true>, dim3(blocks), dim3(256), 0, 0, st, buf, rounds, errs);
                else hipLaunchKernelGGL(k_barrier<false>, dim3(blocks), dim3(256), 0, 0, st, buf, rounds, errs);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                int h;
                BarrierState hs;
                (void)hipMemcpy(&h, errs, 4, hipMemcpyDeviceToHost);
                (void)hipMemcpy(&hs, st, sizeof(hs), hipMemcpyDeviceToHost);
                printf("{\"ubench\": \"xcd_barrier\", \"workgroups\": %d, \"record_128B\": %d, \"us_per_barrier\": %.2f, "
                       "\"visibility_errors\": %d, \"timeout\": %u, \"pop\": [%u,%u,%u,%u,%u,%u,%u,%u]}\n",
                       blocks, publish, ms * 1e3 / (2 * rounds), h, hs.timeout[0], hs.pop[0][0], hs.pop[1][0], hs.pop[2][0],
                       hs.pop[3][0], hs.pop[4][0], hs.pop[5][0], hs.pop[6][0], hs.pop[7][0]);
            }
        }
    }
    return 0;
}
