// XCD-local barrier, second look (round 4): what the persistent samplers pay is not the back-to-back barrier rate of xcd_local.hip
// but the time from the LAST workgroup's arrival to the exits when the arrivals are spread over ~0.5 us (a GEMM phase's tail):
// early arrivers poll while the late ones are still to draw their tickets.  Every workgroup stamps the 100 MHz wall clock at
// its arrival and at its exit; reported per variant: mean over rounds and XCDs of (last exit - last arrival) and of
// (first exit - last arrival), with arrivals staggered by a rank-dependent delay of 0 .. `spread` us.
//   hipcc --offload-arch=gfx950 -O3 xcd_barrier2.hip -o xcd_barrier2.bin && ./xcd_barrier2.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct State {
    unsigned arrive[8][1024];  // [xcc][0]: one 4-KB page per XCC
    unsigned gen[8][1024];     // [xcc][0] (variant 0), [xcc][32 rank] (variant 3: one 128-byte line per workgroup)
    unsigned pop[8][32];
    unsigned census[32];
    unsigned timeout[32];
};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}

__device__ __forceinline__ unsigned long long wall() { return __builtin_amdgcn_s_memrealtime(); }

template <int SLEEP>
__device__ __forceinline__ bool spin_until(unsigned* word, unsigned want, unsigned* timeout) {
    for (unsigned spins = 0;; ++spins) {
        if (__hip_atomic_load(word, RLX_AGENT) >= want) return true;
        __builtin_amdgcn_s_sleep(SLEEP);
        if (spins > (1u << 22)) {
            __hip_atomic_store(timeout, 1u, RLX_AGENT);
            return false;
        }
    }
}

// V 0: ticket, the last arriver stores a generation word, the others poll it        (rounds 3 - 4a of the samplers)
// V 1: ticket, the others poll the arrival counter
// V 2: no-return arrival, everybody polls the arrival counter
// V 3: ticket, the last arriver stores 32 release words on 32 lines with ONE store instruction, each workgroup polls its own
// V 4: V 2 with s_sleep(8) between polls        V 5: V 2 with s_sleep(20)
// V 6: V 0 with s_sleep(8)
// V 7: per-workgroup flag words on ONE line, plain stores, one 32-lane poll      V 8: the same, one line per workgroup
template <int V>
__device__ __forceinline__ bool barrier(State* st, unsigned xcc, unsigned n, unsigned rank, unsigned round) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bool ok = true;
    if (V == 7 || V == 8) {  // flags, no read-modify-write: word `rank` of one line (7) / of 32 lines (8); one 32-lane load per poll
        if (threadIdx.x < 64) {
            const unsigned lane = threadIdx.x, stride = V == 7 ? 1u : 32u;
            if (lane == 0) __hip_atomic_store(&st->gen[xcc][stride * rank], round, RLX_AGENT);
            for (unsigned spins = 0;; ++spins) {
                const unsigned v = lane < n ? __hip_atomic_load(&st->gen[xcc][stride * lane], RLX_AGENT) : round;
                if (__builtin_amdgcn_ballot_w64(v < round) == 0) break;
                __builtin_amdgcn_s_sleep(1);
                if (spins > (1u << 22)) {
                    ok = false;
                    break;
                }
            }
        }
    } else if (V == 3) {
        if (threadIdx.x < 64) {
            const unsigned lane = threadIdx.x;
            unsigned ticket = 0;
            if (lane == 0) ticket = __hip_atomic_fetch_add(&st->arrive[xcc][0], 1u, RLX_AGENT);
            ticket = __builtin_amdgcn_readfirstlane(ticket);
            if (ticket == round * n - 1) {
                if (lane < n) __hip_atomic_store(&st->gen[xcc][32 * lane], round, RLX_AGENT);
            } else if (lane == 0) {
                ok = spin_until<1>(&st->gen[xcc][32 * rank], round, &st->timeout[0]);
            }
        }
    } else if (threadIdx.x == 0) {
        unsigned* const word = &st->arrive[xcc][0];
        if (V == 0 || V == 6) {
            const unsigned ticket = __hip_atomic_fetch_add(word, 1u, RLX_AGENT);
            if (ticket == round * n - 1) __hip_atomic_store(&st->gen[xcc][0], round, RLX_AGENT);
            else ok = V == 6 ? spin_until<8>(&st->gen[xcc][0], round, &st->timeout[0]) : spin_until<1>(&st->gen[xcc][0], round, &st->timeout[0]);
        } else if (V == 1) {
            const unsigned ticket = __hip_atomic_fetch_add(word, 1u, RLX_AGENT);
            if (ticket != round * n - 1) ok = spin_until<1>(word, round * n, &st->timeout[0]);
        } else {
            const unsigned one = 1u;
            asm volatile("global_atomic_add %0, %1, off" : : "v"(word), "v"(one) : "memory");
            ok = V == 4 ? spin_until<8>(word, round * n, &st->timeout[0]) : V == 5 ? spin_until<20>(word, round * n, &st->timeout[0])
                                                                                   : spin_until<1>(word, round * n, &st->timeout[0]);
        }
    }
    __builtin_amdgcn_s_barrier();
    return ok;
}

template <int V>
__global__ __launch_bounds__(512) void k(State* st, int rounds, int spread_units, unsigned long long* stamps) {
    __shared__ unsigned s_n, s_rank;
    const unsigned xcc = xcc_id(), nb = gridDim.x;
    if (threadIdx.x == 0) {
        s_rank = __hip_atomic_fetch_add(&st->pop[xcc][0], 1u, RLX_AGENT);
        __hip_atomic_fetch_add(&st->census[0], 1u, RLX_AGENT);
        spin_until<1>(&st->census[0], nb, &st->timeout[0]);
        s_n = __hip_atomic_load(&st->pop[xcc][0], RLX_AGENT);
    }
    __syncthreads();
    const unsigned n = s_n, rank = s_rank;
    unsigned round = 0;
    for (int r = 0; r < rounds; ++r) {
        // arrivals spread over 0 .. spread: a rank-dependent delay that changes every round (s_sleep(1) = 64 clocks ~ 27 ns)
        const int d = (int)(((rank * 11u + (unsigned)r * 7u) % 32u) * (unsigned)spread_units) / 32;
        for (int i = 0; i < d; ++i) __builtin_amdgcn_s_sleep(1);
        unsigned long long t0 = 0, t1 = 0;
        if (threadIdx.x == 0) t0 = wall();
        if (!barrier<V>(st, xcc, n, rank, ++round)) return;
        if (threadIdx.x == 0) {
            t1 = wall();
            unsigned long long* s = stamps + (((size_t)r * 8 + xcc) * 32 + rank) * 2;
            s[0] = t0, s[1] = t1;
        }
    }
}

int main() {
    State* st;
    unsigned long long* stamps;
    const int rounds = 400;
    (void)hipMalloc(&st, sizeof(State));
    (void)hipMalloc(&stamps, (size_t)rounds * 8 * 32 * 2 * 8);
    std::vector<unsigned long long> h((size_t)rounds * 8 * 32 * 2);
    for (int spread = 0; spread <= 32; spread += 16) {  // 0, 16, 32 sleep units ~ 0, 0.43, 0.86 us
        for (int v = 0; v < 9; ++v) {
            (void)hipMemset(st, 0, sizeof(State));
            (void)hipMemset(stamps, 0, h.size() * 8);
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
#define L(V) if (v == V) hipLaunchKernelGGL((k<V>), dim3(256), dim3(512), 0, 0, st, rounds, spread, stamps)
            L(0); L(1); L(2); L(3); L(4); L(5); L(6); L(7); L(8);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            (void)hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
            State hs;
            (void)hipMemcpy(&hs, st, sizeof(hs), hipMemcpyDeviceToHost);
            double last = 0, first = 0, sp = 0;
            int cnt = 0;
            for (int r = 50; r < rounds; ++r)
                for (int x = 0; x < 8; ++x) {
                    unsigned long long amax = 0, amin = ~0ull, emax = 0, emin = ~0ull;
                    for (int q = 0; q < 32; ++q) {
                        const unsigned long long a = h[(((size_t)r * 8 + x) * 32 + q) * 2], e = h[(((size_t)r * 8 + x) * 32 + q) * 2 + 1];
                        amax = a > amax ? a : amax, amin = a < amin ? a : amin, emax = e > emax ? e : emax, emin = e < emin ? e : emin;
                    }
                    last += (double)(emax - amax) / 100.0, first += (double)((long long)emin - (long long)amax) / 100.0, sp += (double)(amax - amin) / 100.0;
                    ++cnt;
                }
            printf("{\"ubench\": \"xcd_barrier2\", \"variant\": %d, \"arrival_spread_us\": %.2f, \"last_arrival_to_last_exit_us\": %.2f, "
                   "\"last_arrival_to_first_exit_us\": %.2f, \"us_per_round\": %.2f, \"timeout\": %u}\n",
                   v, sp / cnt, last / cnt, first / cnt, ms * 1e3 / rounds, hs.timeout[0]);
        }
    }
    return 0;
}
