// Microbenchmark: barrier + all-reduce inside GROUPS of workgroups (one group = the workgroups that would share one batch entry of a
// persistent multi-workgroup CG), several groups running concurrently. Flat arrival counter per group, monotonic (target = epoch * G),
// relaxed agent-scope polling, one release fence before arriving and one acquire fence after. Two placements of a group's workgroups:
//   "xcd"    group = blockIdx % ngroups  (ngroups = 8: the workgroups of a group share an XCD under round-robin dispatch)
//   "spread" group = blockIdx / G        (a group's workgroups are dealt over all XCDs)
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/group_barrier tools/micro/group_barrier.hip && tools/micro/group_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>

#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct GSync {
    unsigned cnt[64][16];   // one 64-byte line per group
    unsigned bad;
};

__device__ __forceinline__ void group_barrier(GSync* s, unsigned grp, unsigned epoch, unsigned G) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(&s->cnt[grp][0], 1u, RLX);
        unsigned spins = 0;
        while (__hip_atomic_load(&s->cnt[grp][0], RLX) < epoch * G) {
            if (++spins > 4000000u) { s->bad = 0xdeadu; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

__device__ __forceinline__ double block_reduce(double v, double* red) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void k_group(GSync* s, double* part, int iters, int G, int ngroups, int xcd_layout, double* out) {
    __shared__ double red[4];
    const int grp = xcd_layout ? blockIdx.x % ngroups : blockIdx.x / G;
    const int member = xcd_layout ? blockIdx.x / ngroups : blockIdx.x % G;
    double total = 0;
    for (int i = 1; i <= iters; ++i) {
        double* p = part + ((size_t)(i & 1) * ngroups + grp) * 64;
        if (threadIdx.x == 0) __hip_atomic_store(&p[member], 1.0 + total * 1e-9, RLX);
        group_barrier(s, grp, (unsigned)i, (unsigned)G);
        double v = 0;
        for (int k = threadIdx.x; k < G; k += 256) v += __hip_atomic_load(&p[k], RLX);
        total = block_reduce(v, red);
    }
    if (threadIdx.x == 0 && member == 0) out[grp] = total;
}

int main() {
    GSync* s;
    double *part, *dout;
    hipMalloc(&s, sizeof(GSync));
    hipMalloc(&dout, 64 * 8);
    hipMalloc(&part, 2 * 64 * 64 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000;
    const int cfg[][2] = {{8, 1}, {8, 8}, {16, 8}, {32, 8}, {32, 1}, {64, 4}, {16, 16}};   // (G, groups): at most 256 workgroups, all resident
    for (auto& c : cfg)
        for (int layout = 1; layout >= 0; --layout) {
            const int G = c[0], ng = c[1];
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipMemset(s, 0, sizeof(GSync));
                hipMemset(part, 0, 2 * 64 * 64 * sizeof(double));
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_group, dim3(G * ng), dim3(256), 0, 0, s, part, iters, G, ng, layout, dout);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            unsigned bad = 0;
            double total = 0;
            hipMemcpy(&bad, &s->bad, 4, hipMemcpyDeviceToHost);
            hipMemcpy(&total, dout, 8, hipMemcpyDeviceToHost);
            printf("group barrier + allreduce  G=%2d groups=%2d layout=%-6s  %.3f us per step (sum %.1f, expect ~%d)%s\n", G, ng, layout ? "xcd" : "spread",
                   ms * 1e3 / iters, total, G, bad == 0xdeadu ? "  SPIN LIMIT HIT" : "");
        }
    return 0;
}
