// XCD-local barrier + visibility: the 32 workgroups resident on one XCD share that XCD's L2, so data exchanged ONLY
// between them needs no L2 write-back / invalidate (the cross-XCD cost of an agent-scope release / acquire) -- stores
// are written through the CU's vector L1 into the L2, a reader drops its L1 (buffer_inv sc0) and reads the L2.
// Measures, per XCD concurrently (8 independent groups of 32 workgroups):
//   barrier   us per XCD-local barrier (arrival counter + generation word, relaxed agent-scope atomics on one line each)
//   record    the same with every workgroup publishing 128 B that a peer ON THE SAME XCD re-reads (visibility check)
//   tile      the same with a 32-KB tile per workgroup re-read by the peer (a GEMM phase's activation hand-over)
//   hipcc --offload-arch=gfx950 -O3 xcd_local.hip -o xcd_local.bin && ./xcd_local.bin
#include <hip/hip_runtime.h>

#include <cstdio>

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct LocalState {
    unsigned arrive[8][32];  // [xcc][0]
    unsigned gen[8][32];
    unsigned pop[8][32];
    unsigned census[32];
    unsigned timeout[32];
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

// all threads call; n = workgroups of this XCC
// no-return arrival atomic + SCALAR polling of the counter (s_load_dword glc: misses the scalar cache, served by the L2):
// no vector-memory result is waited for, so loads a wave issued before the barrier stay in flight across it
__device__ __forceinline__ bool scalar_barrier(LocalState* st, unsigned xcc, unsigned n, unsigned round) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&st->arrive[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned* word = &st->arrive[xcc][0];
        const unsigned want = round * n;
        unsigned spins = 0;
        for (;; ++spins) {
            unsigned v;
            asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(word) : "memory");
            if (v >= want) break;
            __builtin_amdgcn_s_sleep(1);
            if (spins > (1u << 22)) {
                __hip_atomic_store(&st->timeout[0], 1u, RLX_AGENT);
                ok = false;
                break;
            }
        }
    }
    __builtin_amdgcn_s_barrier();
    return ok;
}

// flag barrier: every workgroup stores the round into its own word of ONE 128-byte line per XCC (no atomics), and polls
// the whole line with two scalar 16-dword loads until the minimum reaches the round
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ bool flag_barrier(LocalState* st, unsigned xcc, unsigned rank, unsigned round) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_store(&st->gen[xcc][rank], round, RLX_AGENT);
        const unsigned* line = &st->gen[xcc][0];
        unsigned spins = 0;
        for (;; ++spins) {
            u32x16 a, b;
            asm volatile("s_load_dwordx16 %0, %2, 0x0 glc\n\ts_load_dwordx16 %1, %2, 0x40 glc\n\ts_waitcnt lgkmcnt(0)"
                         : "=s"(a), "=s"(b) : "s"(line) : "memory");
            unsigned m = 0xffffffffu;
#pragma unroll
            for (int i = 0; i < 16; ++i) m = min(m, min(a[i], b[i]));
            if (m >= round) break;
            if (spins > (1u << 22)) {
                __hip_atomic_store(&st->timeout[0], 1u, RLX_AGENT);
                ok = false;
                break;
            }
        }
    }
    __builtin_amdgcn_s_barrier();
    return ok;
}

// INV 4: ticket + every non-last workgroup polls the ARRIVAL counter itself (no generation word: one hop less after the last ticket)
__device__ __forceinline__ bool ticket_barrier(LocalState* st, unsigned xcc, unsigned n, unsigned round) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(&st->arrive[xcc][0], 1u, RLX_AGENT);
        if (ticket != round * n - 1) ok = spin_until(&st->arrive[xcc][0], round * n, &st->timeout[0]);
    }
    __builtin_amdgcn_s_barrier();
    return ok;
}

// INV 5: no-return arrival atomic, vector polls of the counter
__device__ __forceinline__ bool noret_barrier(LocalState* st, unsigned xcc, unsigned n, unsigned round) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bool ok = true;
    if (threadIdx.x == 0) {
        unsigned* w = &st->arrive[xcc][0];
        unsigned one = 1u;
        asm volatile("global_atomic_add %0, %1, off" : : "v"(w), "v"(one) : "memory");
        ok = spin_until(w, round * n, &st->timeout[0]);
    }
    __builtin_amdgcn_s_barrier();
    return ok;
}

// INV 6: flag line, vector polled: workgroup `rank` stores the round into word rank of ONE 128-byte line per XCC (sc1 store,
// no read-modify-write), lanes 0 .. n - 1 of wave 0 poll the line with ONE sc1 load per poll
__device__ __forceinline__ bool vflag_barrier(LocalState* st, unsigned xcc, unsigned n, unsigned rank, unsigned round) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bool ok = true;
    if (threadIdx.x < 64) {
        const unsigned lane = threadIdx.x;
        if (lane == 0) __hip_atomic_store(&st->gen[xcc][rank], round, RLX_AGENT);
        unsigned spins = 0;
        for (;; ++spins) {
            const unsigned v = lane < n ? __hip_atomic_load(&st->gen[xcc][lane], RLX_AGENT) : round;
            if (__builtin_amdgcn_ballot_w64(v < round) == 0) break;
            __builtin_amdgcn_s_sleep(1);
            if (spins > (1u << 22)) {
                if (lane == 0) __hip_atomic_store(&st->timeout[0], 1u, RLX_AGENT);
                ok = false;
                break;
            }
        }
    }
    __builtin_amdgcn_s_barrier();
    return ok;
}

template <int INV>
__device__ __forceinline__ bool local_barrier(LocalState* st, unsigned xcc, unsigned n, unsigned round, unsigned rank) {
    if (INV == 6) return vflag_barrier(st, xcc, n, rank, round);
    if (INV == 5) return noret_barrier(st, xcc, n, round);
    if (INV == 4) return ticket_barrier(st, xcc, n, round);
    if (INV == 3) return flag_barrier(st, xcc, rank, round);
    if (INV == 2) return scalar_barrier(st, xcc, n, round);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // stores are in the XCD's L2 (the vector L1 is write-through)
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(&st->arrive[xcc][0], 1u, RLX_AGENT);
        if (ticket == round * n - 1) __hip_atomic_store(&st->gen[xcc][0], round, RLX_AGENT);
        else ok = spin_until(&st->gen[xcc][0], round, &st->timeout[0]);
    }
    __syncthreads();
    if (INV == 1) asm volatile("buffer_inv sc1" ::: "memory");  // drop this CU's vector L1 (and clean non-coherent L2 lines)
    return ok;
}

__device__ __forceinline__ float4 load_sc1(const float4* p) {  // agent-scope load: misses the vector L1, served by the L2
    float4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int MODE, int INV>  // 0 bare, 1 128-B record, 2 32-KB tile; INV 1: buffer_inv sc1 after the barrier, 0: sc1 loads
__global__ __launch_bounds__(512) void k_local(LocalState* st, float* buf, int rounds, int* errs) {
    __shared__ unsigned s_n, s_rank;
    const unsigned xcc = xcc_id(), nb = gridDim.x;
    if (threadIdx.x == 0) {
        s_rank = __hip_atomic_fetch_add(&st->pop[xcc][0], 1u, RLX_AGENT);
        __hip_atomic_fetch_add(&st->census[0], 1u, RLX_AGENT);
        spin_until(&st->census[0], nb, &st->timeout[0]);
        s_n = __hip_atomic_load(&st->pop[xcc][0], RLX_AGENT);
    }
    __syncthreads();
    const unsigned n = s_n, rank = s_rank;
    constexpr int REC = MODE == 2 ? 8192 : 32;  // floats per workgroup record
    float* mine = buf + ((size_t)xcc * 64 + rank) * REC;
    const unsigned peer = (rank + 13) % n;
    const float* theirs = buf + ((size_t)xcc * 64 + peer) * REC;
    unsigned round = 0;
    int bad = 0;
    for (int r = 0; r < rounds; ++r) {
        if (MODE == 1 && threadIdx.x < 32) mine[threadIdx.x] = (float)(r * 1000 + rank);
        if (MODE == 2)
            for (int i = threadIdx.x; i < REC / 4; i += 512)
                reinterpret_cast<float4*>(mine)[i] = make_float4((float)(r * 1000 + rank), 1.f, 2.f, 3.f);
        if (!local_barrier<INV>(st, xcc, n, ++round, rank)) return;
        if (MODE == 1 && threadIdx.x < 8) bad += (INV == 1 ? reinterpret_cast<const float4*>(theirs)[threadIdx.x] : load_sc1(reinterpret_cast<const float4*>(theirs) + threadIdx.x)).x != (float)(r * 1000 + peer);
        if (MODE == 2)
            for (int i = threadIdx.x; i < REC / 4; i += 512)
                bad += (INV == 1 ? reinterpret_cast<const float4*>(theirs)[i] : load_sc1(reinterpret_cast<const float4*>(theirs) + i)).x != (float)(r * 1000 + peer);
        if (!local_barrier<INV>(st, xcc, n, ++round, rank)) return;
    }
    if (bad) atomicAdd(errs, bad);
}

int main() {
    LocalState* st;
    float* buf;
    int* errs;
    (void)hipMalloc(&st, sizeof(LocalState));
    (void)hipMalloc(&buf, (size_t)8 * 64 * 8192 * 4);
    (void)hipMalloc(&errs, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int rounds = 2000;
    for (int mode = 0; mode < 3; ++mode) {
        for (int inv = 0; inv < 7; ++inv) {
            (void)hipMemset(st, 0, sizeof(LocalState));
            (void)hipMemset(errs, 0, 4);
            (void)hipEventRecord(e0);
            if (mode == 0 && inv == 1) hipLaunchKernelGGL((k_local<0, 1>), dim3(256), dim3(512), 0, 0, st, buf, rounds, errs);
            if (mode == 1 && inv == 1) hipLaunchKernelGGL((k_local<1, 1>), dim3(256), dim3(512), 0, 0, st, buf, rounds, errs);
            if (mode == 2 && inv == 1) hipLaunchKernelGGL((k_local<2, 1>), dim3(256), dim3(512), 0, 0, st, buf, rounds, errs);
            if (mode == 0 && !inv) hipLaunchKernelGGL((k_local<0, 0>), dim3(256), dim3(512), 0, 0, st, buf, rounds, errs);
            if (mode == 1 && !inv) hipLaunchKernelGGL((k_local<1, 0>), dim3(256), dim3(512), 0, 0, st, buf, rounds, errs);
            if (mode == 2 && !inv) hipLaunchKernelGGL((k_local<2, 0>), dim3(256), dim3(512), 0, 0, st, buf, rounds, errs);
            if (mode == 0 && inv == 2) hipLaunchKernelGGL((k_local<0, 2>), dim3(256), dim3(512), 0, 0, st, buf, rounds, errs);
            if (mode == 1 && inv == 2) hipLaunchKernelGGL((k_local<1, 2>), dim3(256), dim3(512), 0, 0, st, buf, rounds, errs);
            if (mode == 2 && inv == 2) hipLaunchKernelGGL((k_local<2, 2>), dim3(256), dim3(512), 0, 0, st, buf, rounds, errs);
            if (mode == 0 && inv == 3) hipLaunchKernelGGL((k_local<0, 3>), dim3(256), dim3(512), 0, 0, st, buf, rounds, errs);
            if (mode == 1 && inv == 3) hipLaunchKernelGGL((k_local<1, 3>), dim3(256), dim3(512), 0, 0, st, buf, rounds, errs);
            if (mode == 2 && inv == 3) hipLaunchKernelGGL((k_local<2, 3>), dim3(256), dim3(512), 0, 0, st, buf, rounds, errs);
            if (mode == 0 && inv == 4) hipLaunchKernelGGL((k_local<0, 4>), dim3(256), dim3(512), 0, 0, st, buf, rounds, errs);
            if (mode == 1 && inv == 4) hipLaunchKernelGGL((k_local<1, 4>), dim3(256), dim3(512), 0, 0, st, buf, rounds, errs);
            if (mode == 2 && inv == 4) hipLaunchKernelGGL((k_local<2, 4>), dim3(256), dim3(512), 0, 0, st, buf, rounds, errs);
            if (mode == 0 && inv == 5) hipLaunchKernelGGL((k_local<0, 5>), dim3(256), dim3(512), 0, 0, st, buf, rounds, errs);
            if (mode == 1 && inv == 5) hipLaunchKernelGGL((k_local<1, 5>), dim3(256), dim3(512), 0, 0, st, buf, rounds, errs);
            if (mode == 2 && inv == 5) hipLaunchKernelGGL((k_local<2, 5>), dim3(256), dim3(512), 0, 0, st, buf, rounds, errs);
            if (mode == 0 && inv == 6) hipLaunchKernelGGL((k_local<0, 6>), dim3(256), dim3(512), 0, 0, st, buf, rounds, errs);
            if (mode == 1 && inv == 6) hipLaunchKernelGGL((k_local<1, 6>), dim3(256), dim3(512), 0, 0, st, buf, rounds, errs);
            if (mode == 2 && inv == 6) hipLaunchKernelGGL((k_local<2, 6>), dim3(256), dim3(512), 0, 0, st, buf, rounds, errs);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            int h;
            LocalState hs;
            (void)hipMemcpy(&h, errs, 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(&hs, st, sizeof(hs), hipMemcpyDeviceToHost);
            printf("{\"ubench\": \"xcd_local_barrier\", \"mode\": \"%s\", \"reader\": \"%s\", \"us_per_barrier\": %.2f, \"visibility_errors\": %d, "
                   "\"timeout\": %u, \"pop\": [%u,%u,%u,%u,%u,%u,%u,%u]}\n",
                   mode == 0 ? "bare" : (mode == 1 ? "record_128B" : "tile_32KB"), inv == 6 ? "sc1 loads, flag-line barrier (one vector poll)" : inv == 5 ? "sc1 loads, no-return arrival + counter polls" : inv == 4 ? "sc1 loads, ticket + counter polls" : inv == 3 ? "sc1 loads, flag-line barrier (scalar polls)" : inv == 2 ? "sc1 loads, scalar-polled barrier" : (inv ? "buffer_inv sc1" : "sc1 loads"), ms * 1e3 / (2 * rounds), h, hs.timeout[0],
                   hs.pop[0][0], hs.pop[1][0], hs.pop[2][0], hs.pop[3][0], hs.pop[4][0], hs.pop[5][0], hs.pop[6][0], hs.pop[7][0]);
        }
    }
    return 0;
}
