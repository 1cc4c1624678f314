// Microbenchmark: cost of a grid-wide barrier on MI355X -- cooperative_groups::grid_group::sync() and a hand-written sense-reversal
// barrier (one atomic counter per barrier instance, agent-scope acquire / release) -- for a persistent-kernel CG on mid-size grids.
//   hipcc --offload-arch=gfx950 -O3 -o gridsync tools/micro/gridsync.hip && ./gridsync
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
namespace cg = cooperative_groups;

__global__ __launch_bounds__(256) void k_cg(int iters, float* out) {
    cg::grid_group g = cg::this_grid();
    float acc = 0;
    for (int i = 0; i < iters; ++i) {
        acc += 1.0f;
        g.sync();
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = acc;
}

// sense-reversal barrier: counter counts arrivals, generation flips when the last workgroup arrives
__device__ __forceinline__ void grid_barrier(unsigned* counter, volatile unsigned* generation, unsigned nblocks, unsigned& local_gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        local_gen ^= 1u;
        __threadfence();
        const unsigned prev = atomicAdd(counter, 1u);
        if (prev == nblocks - 1) {
            *counter = 0;
            __threadfence();
            __hip_atomic_store((unsigned*)generation, local_gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load((unsigned*)generation, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != local_gen) __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_own(int iters, unsigned* counter, unsigned* generation, float* out) {
    __shared__ unsigned gen_s;
    unsigned gen = 0;
    float acc = 0;
    for (int i = 0; i < iters; ++i) {
        acc += 1.0f;
        grid_barrier(counter, generation, gridDim.x, gen);
    }
    (void)gen_s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = acc;
}

int main() {
    float* out;
    unsigned* sync;
    hipMalloc(&out, 4);
    hipMalloc(&sync, 256);
    hipMemset(sync, 0, 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000;
    for (int nblk : {64, 256, 512, 1024, 2048}) {
        int it = iters;
        void* args[] = {&it, &out};
        hipError_t err = hipLaunchCooperativeKernel((void*)k_cg, dim3(nblk), dim3(256), args, 0, 0);
        if (err != hipSuccess) { printf("cg   nblk=%d launch failed: %s\n", nblk, hipGetErrorString(err)); (void)hipGetLastError(); }
        else {
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchCooperativeKernel((void*)k_cg, dim3(nblk), dim3(256), args, 0, 0);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("cg::grid.sync  nblk=%4d  %.3f us per barrier\n", nblk, ms * 1e3 / iters);
        }
        unsigned* counter = sync;
        unsigned* generation = sync + 32;
        hipMemset(sync, 0, 256);
        void* args2[] = {&it, &counter, &generation, &out};
        err = hipLaunchCooperativeKernel((void*)k_own, dim3(nblk), dim3(256), args2, 0, 0);   // cooperative launch only for the co-residency guarantee
        if (err != hipSuccess) { printf("own  nblk=%d launch failed: %s\n", nblk, hipGetErrorString(err)); (void)hipGetLastError(); continue; }
        hipDeviceSynchronize();
        hipMemset(sync, 0, 256);
        hipEventRecord(e0);
        hipLaunchCooperativeKernel((void*)k_own, dim3(nblk), dim3(256), args2, 0, 0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("own barrier    nblk=%4d  %.3f us per barrier\n", nblk, ms * 1e3 / iters);
    }
    return 0;
}
