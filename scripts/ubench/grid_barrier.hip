// Cost of a device-wide barrier inside one kernel (256 workgroups, one per CU) on MI355X:
// hand-rolled sense-reversing barrier on an agent-scope atomic + cooperative-groups grid.sync().
// Each round every workgroup also writes a line that another workgroup (different XCD) reads after
// the barrier, to check cross-XCD visibility.
// hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier.bin
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned nblocks, unsigned& phase) {
    __syncthreads();
    if (threadIdx.x == 0) {
        phase += nblocks;
        __threadfence();  // release this workgroup's global writes at agent scope
        atomicAdd(counter, 1u);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < phase) __builtin_amdgcn_s_sleep(1);
        __threadfence();
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_manual(unsigned* counter, float* buf, int rounds, int* errs) {
    unsigned phase = 0;
    const unsigned nb = gridDim.x;
    for (int r = 0; r < rounds; ++r) {
        buf[(size_t)blockIdx.x * 256 + threadIdx.x] = (float)(r * 1000 + blockIdx.x);
        grid_barrier(counter, nb, phase);
        const unsigned peer = (blockIdx.x + 37) % nb;  // a block on another XCD
        const float v = __builtin_nontemporal_load(&buf[(size_t)peer * 256 + threadIdx.x]);
        if (v != (float)(r * 1000 + peer)) atomicAdd(errs, 1);
        grid_barrier(counter, nb, phase);
    }
}
__global__ __launch_bounds__(256) void k_coop(float* buf, int rounds, int* errs) {
    cg::grid_group g = cg::this_grid();
    for (int r = 0; r < rounds; ++r) {
        buf[(size_t)blockIdx.x * 256 + threadIdx.x] = (float)(r * 1000 + blockIdx.x);
        g.sync();
        const unsigned peer = (blockIdx.x + 37) % gridDim.x;
        const float v = buf[(size_t)peer * 256 + threadIdx.x];
        if (v != (float)(r * 1000 + peer)) atomicAdd(errs, 1);
        g.sync();
    }
}
int main() {
    unsigned* counter; float* buf; int* errs;
    (void)hipMalloc(&counter, 4); (void)hipMalloc(&buf, 256 * 256 * 4); (void)hipMalloc(&errs, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int rounds = 2000, blocks = 256;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipMemset(counter, 0, 4); (void)hipMemset(errs, 0, 4);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_manual, dim3(blocks), dim3(256), 0, 0, counter, buf, rounds, errs);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        int h; (void)hipMemcpy(&h, errs, 4, hipMemcpyDeviceToHost);
        printf("manual barrier: %.3f ms for %d barriers -> %.2f us each, visibility errors %d\n", ms, 2 * rounds, ms * 1e3 / (2 * rounds), h);
    }
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipMemset(errs, 0, 4);
        int r = rounds;
        void* args[] = {&buf, &r, &errs};
        (void)hipEventRecord(e0);
        hipError_t e = hipLaunchCooperativeKernel((void*)k_coop, dim3(blocks), dim3(256), args, 0, 0);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        int h; (void)hipMemcpy(&h, errs, 4, hipMemcpyDeviceToHost);
        printf("cooperative grid.sync (%s): %.3f ms -> %.2f us each, visibility errors %d\n", hipGetErrorString(e), ms, ms * 1e3 / (2 * rounds), h);
    }
    return 0;
}
