// Microbenchmark for the latency floor of mid-size CG: what does ONE grid-wide synchronisation cost on MI355X?
//   (a) XCD-hierarchical barrier (MI355X_MICROARCH.md "barrier-xcd"): per-group arrival counter (group = blockIdx & 7, the observed
//       workgroup -> XCD placement; only contention depends on it, never correctness), the last arriver of a group bumps a top counter,
//       the last group publishes the epoch to 8 per-group generation words; relaxed agent-scope polling + s_sleep, one release fence
//       before arriving and one acquire fence after;
//   (b) the same barrier carrying an all-reduce of one double per workgroup (what a persistent CG needs twice per iteration: every
//       workgroup must see sum_i partial_i): partials are published before the barrier and re-reduced by every workgroup after it;
//   (c) a dependent kernel boundary: back-to-back launches of a kernel in which every workgroup reads all partials of its predecessor
//       and writes its own (the structure of the shipped two-launch CG, stencil_march.hpp cg_prologue).
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/xcd_barrier tools/micro/xcd_barrier.hip && tools/micro/xcd_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>

#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct Sync {
    unsigned group_cnt[8][16];   // one 64-byte line per word
    unsigned group_gen[8][16];
    unsigned top_cnt[16];
};

__device__ __forceinline__ void xcd_barrier(Sync* s, unsigned epoch, unsigned nblk) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned ngroups = nblk < 8 ? nblk : 8;
        const unsigned grp = blockIdx.x & 7;
        const unsigned in_group = (nblk + 7 - grp) / 8;                    // blocks with blockIdx & 7 == grp
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned prev = __hip_atomic_fetch_add(&s->group_cnt[grp][0], 1u, RLX);
        if (prev == epoch * in_group - 1) {
            const unsigned prev2 = __hip_atomic_fetch_add(&s->top_cnt[0], 1u, RLX);
            if (prev2 == epoch * ngroups - 1)
                for (unsigned g = 0; g < ngroups; ++g) __hip_atomic_store(&s->group_gen[g][0], epoch, RLX);
        }
        unsigned spins = 0;      // bounded: a workgroup that is not resident must not hang the GPU (the result is then garbage, and says so)
        while (__hip_atomic_load(&s->group_gen[grp][0], RLX) < epoch) {
            if (++spins > 4000000u) { s->top_cnt[8] = 0xdeadu; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_barrier(Sync* s, int iters, float* out) {
    float acc = 0;
    for (int i = 1; i <= iters; ++i) {
        acc += 1.0f;
        xcd_barrier(s, (unsigned)i, gridDim.x);
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = acc;
}

__device__ __forceinline__ double block_reduce(double v, double* red) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// barrier + all-reduce of one double per workgroup, double-buffered partials (sc1 stores so that no release fence has to flush them)
__global__ __launch_bounds__(256) void k_allreduce(Sync* s, double* part, int iters, double* out) {
    __shared__ double red[4];
    double total = 0;
    for (int i = 1; i <= iters; ++i) {
        double* p = part + (size_t)(i & 1) * gridDim.x;
        if (threadIdx.x == 0) __hip_atomic_store(&p[blockIdx.x], 1.0 + total * 1e-9, RLX);
        xcd_barrier(s, (unsigned)i, gridDim.x);
        double v = 0;
        for (unsigned k = threadIdx.x; k < gridDim.x; k += 256) v += __hip_atomic_load(&p[k], RLX);
        total = block_reduce(v, red);
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = total;
}

__global__ __launch_bounds__(256) void k_dependent(const double* pin, double* pout, int nblk) {
    __shared__ double red[4];
    double v = 0;
    for (int k = threadIdx.x; k < nblk; k += 256) v += pin[k];
    const double t = block_reduce(v, red);
    if (threadIdx.x == 0) pout[blockIdx.x] = 1.0 + t * 1e-9;
}

int main() {
    Sync* s;
    float* out;
    double *part, *dout;
    hipMalloc(&s, sizeof(Sync));
    hipMalloc(&out, 4);
    hipMalloc(&dout, 8);
    hipMalloc(&part, 2 * 4096 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000;
    for (int nblk : {64, 256, 512, 1024}) {
        float ms;
        for (int rep = 0; rep < 2; ++rep) {      // rep 0 warms up
            hipMemset(s, 0, sizeof(Sync));
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_barrier, dim3(nblk), dim3(256), 0, 0, s, iters, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        unsigned flag = 0;
        (void)hipMemcpy(&flag, &s->top_cnt[8], 4, hipMemcpyDeviceToHost);
        printf("xcd barrier            nblk=%4d  %.3f us per barrier%s\n", nblk, ms * 1e3 / iters, flag == 0xdeadu ? "  (SPIN LIMIT HIT: invalid)" : "");
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(s, 0, sizeof(Sync));
            hipMemset(part, 0, 2 * 4096 * sizeof(double));
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_allreduce, dim3(nblk), dim3(256), 0, 0, s, part, iters, dout);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        double total = 0;
        hipMemcpy(&total, dout, 8, hipMemcpyDeviceToHost);
        printf("xcd barrier + allreduce nblk=%4d  %.3f us per step   (sum %.1f, expect ~%d)\n", nblk, ms * 1e3 / iters, total, nblk);
        hipMemset(part, 0, 2 * 4096 * sizeof(double));
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int i = 0; i < iters; ++i)
                hipLaunchKernelGGL(k_dependent, dim3(nblk), dim3(256), 0, 0, part + (size_t)(i & 1) * 4096, part + (size_t)((i + 1) & 1) * 4096, nblk);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("dependent launches      nblk=%4d  %.3f us per launch (each workgroup re-reduces %d partials of its predecessor)\n", nblk, ms * 1e3 / iters, nblk);
    }
    return 0;
}
