// Does the order in which consecutive streaming kernels walk their vectors matter to the 256 MiB Infinity Cache?  (round 5)
//
// The CG of the pressure path is a chain of streaming kernels that share two of their three vectors with the kernel before. Every workgroup
// marches through its own chunk of planes front to back, in EVERY kernel: what kernel k touched last (the ends of all chunks) is what kernel
// k + 1 touches last as well. If the Infinity Cache replaces roughly in LRU order, a kernel that marches its chunks BACK to front (a sawtooth)
// would start on the bytes that are still on the die. This microbenchmark measures exactly that, with nothing else in the way:
//   vectors v0 .. v3 of n^3 floats; kernel k reads v[k % 4] and v[(k + 1) % 4] and writes v[(k + 2) % 4]  (16-byte accesses, 12 B per cell);
//   B workgroups, each owns a contiguous chunk and marches through it in steps of 256 lanes x 16 B;
//   modes: ff = every kernel front to back, fb = direction alternates from kernel to kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/mall_sawtooth tools/micro/mall_sawtooth.hip        Run: tools/micro/mall_sawtooth
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void triad(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ c, long long vecs_per_block,
                                             long long total_vecs, int reverse, float s) {
    const long long lo = (long long)blockIdx.x * vecs_per_block;
    long long hi = lo + vecs_per_block;
    if (hi > total_vecs) hi = total_vecs;
    const long long steps = (hi - lo + 255) / 256;
    for (long long k = 0; k < steps; ++k) {
        const long long kk = reverse ? steps - 1 - k : k;
        const long long i = lo + kk * 256 + threadIdx.x;
        if (i < hi) {
            const float4 x = a[i], y = b[i];
            c[i] = make_float4(x.x + s * y.x, x.y + s * y.y, x.z + s * y.z, x.w + s * y.w);
        }
    }
}

// second question (same kernel): does the RELATIVE placement of the vectors matter? n^3 fp32 vectors of 256^3 / 512^3 cells are exact powers of two long, so
// vectors allocated back to back present every channel-interleave granule with the same phase in all three streams of a kernel. `skew`: vector i starts
// i * skew bytes after where back-to-back placement would put it (one allocation, carved).      tools/micro/mall_sawtooth <reps> skew
static void skew_scan(int reps) {
    const int sizes[] = {256, 384, 512};
    const long long skews[] = {0, 256, 1024, 4096, 4096 + 256, 16384, 65536, 65536 + 4096, 1 << 20, (1 << 20) + 4096 + 256, 2 << 20, (2 << 20) + 65536 + 4096};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int n : sizes) {
        const long long cells = (long long)n * n * n, vecs = cells / 4, bytes = cells * 4;
        char* base = nullptr;
        CHECK(hipMalloc(&base, 4 * bytes + 4 * (4 << 20)));
        CHECK(hipMemset(base, 0, 4 * bytes + 4 * (4 << 20)));
        for (int round = 0; round < 2; ++round)
            for (long long skew : skews) {
                float4* v[4];
                for (int i = 0; i < 4; ++i) v[i] = (float4*)(base + i * (bytes + skew));
                const int B = 1024;
                const long long per = (vecs + B - 1) / B;
                for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(triad, dim3(B), dim3(256), 0, 0, v[k % 4], v[(k + 1) % 4], v[(k + 2) % 4], per, vecs, k & 1, 0.5f);
                CHECK(hipEventRecord(e0, 0));
                for (int k = 0; k < reps; ++k) hipLaunchKernelGGL(triad, dim3(B), dim3(256), 0, 0, v[k % 4], v[(k + 1) % 4], v[(k + 2) % 4], per, vecs, k & 1, 0.5f);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                const double us = ms * 1000.0 / reps;
                if (round == 1) printf("{\"n\": %d, \"skew_bytes\": %lld, \"us_per_kernel\": %.2f, \"TBs_moved\": %.3f}\n", n, skew, us, cells * 12.0 / us * 1e-6);
            }
        CHECK(hipFree(base));
    }
}

int main(int argc, char** argv) {
    if (argc > 2) { skew_scan(atoi(argv[1])); return 0; }
    const int sizes[] = {192, 256, 288, 320, 384, 448, 512};
    const int blocks[] = {512, 1024, 2048, 8192};
    const int reps = argc > 1 ? atoi(argv[1]) : 120;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("{\"note\": \"triad over 4 rotating vectors of n^3 floats, 12 B per cell and kernel; ff = all kernels march their chunks front to back, fb = alternate\"}\n");
    for (int n : sizes) {
        const long long cells = (long long)n * n * n, vecs = cells / 4;
        float4* v[4];
        for (int i = 0; i < 4; ++i) { CHECK(hipMalloc(&v[i], cells * sizeof(float))); CHECK(hipMemset(v[i], 0, cells * sizeof(float))); }
        for (int B : blocks) {
            const long long per = (vecs + B - 1) / B;
            double us[2];
            for (int round = 0; round < 2; ++round)
                for (int mode = 0; mode < 2; ++mode) {
                    for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(triad, dim3(B), dim3(256), 0, 0, v[k % 4], v[(k + 1) % 4], v[(k + 2) % 4], per, vecs, mode ? (k & 1) : 0, 0.5f);
                    CHECK(hipEventRecord(e0, 0));
                    for (int k = 0; k < reps; ++k) hipLaunchKernelGGL(triad, dim3(B), dim3(256), 0, 0, v[k % 4], v[(k + 1) % 4], v[(k + 2) % 4], per, vecs, mode ? (k & 1) : 0, 0.5f);
                    CHECK(hipEventRecord(e1, 0));
                    CHECK(hipEventSynchronize(e1));
                    float ms = 0;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    us[mode] = ms * 1000.0 / reps;
                    if (round == 1)
                        printf("{\"n\": %d, \"workgroups\": %d, \"mode\": \"%s\", \"us_per_kernel\": %.2f, \"TBs_moved\": %.3f}\n", n, B, mode ? "fb" : "ff", us[mode],
                               cells * 12.0 / us[mode] * 1e-6);
                }
            printf("{\"n\": %d, \"workgroups\": %d, \"fb_over_ff\": %.3f}\n", n, B, us[1] / us[0]);
        }
        for (int i = 0; i < 4; ++i) CHECK(hipFree(v[i]));
    }
    return 0;
}
