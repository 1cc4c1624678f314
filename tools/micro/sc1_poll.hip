// Microbenchmark / probe (r4): does a POLLING reader on another XCD see a tagged value that a producer stores later -- for the four combinations
// of {8-byte agent-scope atomic store, 16-byte `global_store_dwordx4 sc1`} x {8-byte agent-scope atomic load, 16-byte `global_load_dwordx4 sc1`}?
// The reader first loads the slot (so that its own L2 may hold the line), the producer (block 0) writes after a delay, the readers (blocks 1..)
// poll with a bound. Prints per combination how many readers saw the update and the mean number of polls.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/sc1_poll tools/micro/sc1_poll.hip && tools/micro/sc1_poll
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned long long u64;
typedef unsigned u4 __attribute__((vector_size(16)));
#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__global__ void probe(u64* slot, int store16, int load16, unsigned tag, unsigned* seen, unsigned* polls, long long delay) {
    if (threadIdx.x != 0) return;
    u64* p = slot + 64 * 0;
    if (blockIdx.x == 0) {
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < delay) __builtin_amdgcn_s_sleep(8);
        if (store16) {
            const u4 g = {7u, tag, 9u, tag};
            asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(g) : "memory");
        } else {
            __hip_atomic_store(p, ((u64)tag << 32) | 7u, RLX);
            __hip_atomic_store(p + 1, ((u64)tag << 32) | 9u, RLX);
        }
        return;
    }
    unsigned n = 0, ok = 0;
    for (; n < 2000000u; ++n) {
        unsigned t0, t1;
        if (load16) {
            u4 q;
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(q) : "v"(p) : "memory");
            t0 = q[1]; t1 = q[3];
        } else {
            t0 = (unsigned)(__hip_atomic_load(p, RLX) >> 32);
            t1 = (unsigned)(__hip_atomic_load(p + 1, RLX) >> 32);
        }
        if (t0 == tag && t1 == tag) { ok = 1; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    seen[blockIdx.x] = ok;
    polls[blockIdx.x] = n;
}

int main() {
    u64* slot; unsigned *seen, *polls;
    hipMalloc(&slot, 4096); hipMalloc(&seen, 64 * 4); hipMalloc(&polls, 64 * 4);
    const int readers = 15;     // blocks 1..15: XCDs 1..7, 0..7
    unsigned tag = 100;
    for (int rep = 0; rep < 3; ++rep)
        for (int store16 = 0; store16 < 2; ++store16)
            for (int load16 = 0; load16 < 2; ++load16) {
                ++tag;
                hipMemset(seen, 0, 64 * 4); hipMemset(polls, 0, 64 * 4);
                hipLaunchKernelGGL(probe, dim3(readers + 1), dim3(64), 0, 0, slot, store16, load16, tag, seen, polls, 2000000LL);   // ~20 us at 100 MHz
                hipDeviceSynchronize();
                unsigned hs[64], hp[64];
                hipMemcpy(hs, seen, 64 * 4, hipMemcpyDeviceToHost); hipMemcpy(hp, polls, 64 * 4, hipMemcpyDeviceToHost);
                int ok = 0; double mp = 0;
                for (int b = 1; b <= readers; ++b) { ok += hs[b]; mp += hp[b]; }
                printf("rep %d store %s load %s: %d / %d readers saw the tag, mean polls %.0f\n", rep, store16 ? "16B sc1" : "8B atomic", load16 ? "16B sc1" : "8B atomic", ok, readers, mp / readers);
            }
    return 0;
}
