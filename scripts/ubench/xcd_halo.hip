// Cross-XCD hand-over of a small halo WITHOUT a device-wide barrier or any L2 write-back / invalidate: the producer
// workgroup (on XCD x) stores the record with system-scope stores (sc0 sc1: written through to memory), waits for them
// (s_waitcnt vmcnt(0)), publishes a sequence number with a system-scope store; a consumer workgroup on XCD (x + 1) % 8
// polls that word with system-scope loads and then reads the record with system-scope loads (sc0 sc1: not served by its
// own L2, which may hold the previous round's lines).  Every round rewrites the SAME addresses with new values, so a
// stale line anywhere shows up as an error.  Also timed: the same with plain stores / sc1 loads (expected to fail).
//   hipcc --offload-arch=gfx950 -O3 xcd_halo.hip -o xcd_halo.bin && ./xcd_halo.bin
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct HaloState {
    unsigned seq[8][32];  // [xcc][0]: rounds published by that XCC's producer
    unsigned ack[8][32];  // [xcc][0]: rounds consumed from that XCC's record
    unsigned pop[8][32];
    unsigned census[32];
    unsigned timeout[32];
};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}

template <int AUX>
__device__ __forceinline__ unsigned ld32(const unsigned* p) {
    return __builtin_amdgcn_raw_buffer_load_b32(rsrc(p), 0, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ void st32(unsigned* p, unsigned v) {
    __builtin_amdgcn_raw_buffer_store_b32(v, rsrc(p), 0, 0, AUX);
}

template <int AUX>
__device__ __forceinline__ bool spin(const unsigned* word, unsigned want, unsigned* timeout) {
    for (unsigned spins = 0;; ++spins) {
        if (ld32<AUX>(word) >= want) return true;
        __builtin_amdgcn_s_sleep(1);
        if (spins > (1u << 20)) {
            *timeout = 1;
            return false;
        }
    }
}

// SYS = 17 (sc0 | sc1): system scope; 16: agent scope (sc1); 0: plain
template <int ST_AUX, int LD_AUX>
__global__ __launch_bounds__(256) void k_halo(HaloState* st, float* buf, int rounds, int* errs, float* lat_us) {
    __shared__ unsigned s_rank;
    const unsigned xcc = xcc_id(), nb = gridDim.x;
    if (threadIdx.x == 0) {
        s_rank = atomicAdd(&st->pop[xcc][0], 1u);
        atomicAdd(&st->census[0], 1u);
        spin<17>(&st->census[0], nb, &st->timeout[0]);
    }
    __syncthreads();
    if (s_rank != 0) return;  // one workgroup per XCC: producer for its own record, consumer of the previous XCC's
    constexpr int REC = 7 * 3 * 1024 / 4;  // 7 frames x (k, v ...) ~ 21 KB of floats
    float* mine = buf + (size_t)xcc * REC;
    const unsigned prev = (xcc + 7) & 7;
    const float* theirs = buf + (size_t)prev * REC;
    const __amdgpu_buffer_rsrc_t mr = rsrc(mine), tr = rsrc(theirs);
    int bad = 0;
    long long t_wait = 0;
    for (int r = 1; r <= rounds; ++r) {
        // WAR guard: the consumer of MY record has finished reading round r - 1
        if (threadIdx.x == 0 && r > 1) spin<17>(&st->ack[xcc][0], (unsigned)(r - 1), &st->timeout[0]);
        __syncthreads();
        for (int i = threadIdx.x; i < REC / 4; i += 256) {
            const f32x4 v = {(float)(r * 16 + xcc), 1.f, 2.f, (float)i};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), mr, i * 16, 0, ST_AUX);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) st32<17>(&st->seq[xcc][0], (unsigned)r);
        // consume the neighbour's record of the same round
        if (threadIdx.x == 0) {
            const long long t0 = wall_clock64();
            spin<17>(&st->seq[prev][0], (unsigned)r, &st->timeout[0]);
            t_wait += wall_clock64() - t0;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < REC / 4; i += 256) {
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(tr, i * 16, 0, LD_AUX));
            bad += v[0] != (float)(r * 16 + prev) || v[3] != (float)i;
        }
        __syncthreads();
        if (threadIdx.x == 0) st32<17>(&st->ack[prev][0], (unsigned)r);
    }
    if (bad) atomicAdd(errs, bad);
    if (threadIdx.x == 0) lat_us[xcc] = (float)t_wait / 100.f / rounds;
}

int main() {
    HaloState* st;
    float *buf, *lat;
    int* errs;
    (void)hipMalloc(&st, sizeof(HaloState));
    (void)hipMalloc(&buf, (size_t)8 * 8192 * 4);
    (void)hipMalloc(&errs, 4);
    (void)hipMalloc(&lat, 32);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int rounds = 2000;
    for (int mode = 0; mode < 3; ++mode) {
        (void)hipMemset(st, 0, sizeof(HaloState));
        (void)hipMemset(errs, 0, 4);
        (void)hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL((k_halo<17, 17>), dim3(256), dim3(256), 0, 0, st, buf, rounds, errs, lat);
        if (mode == 1) hipLaunchKernelGGL((k_halo<0, 17>), dim3(256), dim3(256), 0, 0, st, buf, rounds, errs, lat);
        if (mode == 2) hipLaunchKernelGGL((k_halo<17, 16>), dim3(256), dim3(256), 0, 0, st, buf, rounds, errs, lat);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        int h;
        float hl[8];
        HaloState hs;
        (void)hipMemcpy(&h, errs, 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hl, lat, 32, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&hs, st, sizeof(hs), hipMemcpyDeviceToHost);
        printf("{\"ubench\": \"xcd_halo\", \"stores\": \"%s\", \"loads\": \"%s\", \"us_per_round\": %.2f, \"flag_wait_us\": %.2f, "
               "\"errors\": %d, \"timeout\": %u}\n",
               mode == 1 ? "plain" : "sc0 sc1", mode == 2 ? "sc1" : "sc0 sc1", ms * 1e3 / rounds, hl[1], h, hs.timeout[0]);
    }
    return 0;
}
