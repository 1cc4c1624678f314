// Microbenchmark (r5): the rates the advection kernels' instruction budget is priced with, measured instead of recalled.
//   (1) VALU issue: cycles per wave64 instruction of v_fma_f32, v_pk_fma_f32, v_add_u32, v_floor_f32 / v_fract_f32, v_cvt_i32_f32, v_med3_f32, v_cndmask,
//       v_mul_i32_i24 with every SIMD saturated (8 independent chains per lane, 4 / 8 waves per SIMD);
//   (2) LDS reads: ds_read_b32, ds_read2_b32 (two dwords P2 apart), ds_read_b64 at 4-byte alignment (lane stride 4 B: the tap pairs of a multilinear lookup);
//   (3) LDS-DMA: global_load_lds_dwordx4 whose per-lane SOURCE is only 4-byte aligned (rows of n - 1 / n + 1 faces): does it arrive, intact?
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/issue_rates tools/micro/issue_rates.hip && tools/micro/issue_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define CHAINS 8
// one instruction per statement, pinned by inline assembly (the compiler packs scalar fp32 chains into v_pk_* by itself and folds constants)
template <int OP>
__global__ __launch_bounds__(256) void valu(float* out, int iters, float seed) {
    float a[CHAINS];
    f2 p[CHAINS];
    int q[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) { a[i] = seed + threadIdx.x * 1e-3f + i; p[i] = f2{a[i], a[i] + 1.f}; q[i] = (int)threadIdx.x + i; }
    float m = 1.0001f, c = 0.5f;
    f2 mm = {m, m}, cc = {c, c};
    int k3 = 3;
    asm volatile("" : "+v"(m), "+v"(c), "+v"(mm), "+v"(cc), "+v"(k3));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(mm), "v"(cc));
            if (OP == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(q[i]) : "v"(k3));
            if (OP == 3) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
            if (OP == 4) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
            if (OP == 5) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(q[i]) : "v"(a[i]));
            if (OP == 6) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (OP == 7) asm volatile("v_cmp_gt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(q[i]) : "v"(k3) : "vcc");
            if (OP == 8) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(q[i]) : "v"(k3));
            if (OP == 9) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 10) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(mm));
            if (OP == 11) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(cc));
            if (OP == 12) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(q[i]) : "v"(k3));
            if (OP == 13) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(q[i]) : "v"(k3));
            if (OP == 14) asm volatile("v_max3_f32 %0, |%0|, |%1|, |%2|" : "+v"(a[i]) : "v"(m), "v"(c));
        }
    }
    float s = 0;
    int t = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) { s += a[i] + p[i].x + p[i].y; t += q[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)t;
}

// LDS read forms: every lane reads at base + 4 lane (+ P2 rows), like the tap pairs of a lookup
template <int FORM>
__global__ __launch_bounds__(256) void ldsread(float* out, int iters) {
    __shared__ float L[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) L[i] = (float)i;
    __syncthreads();
    float acc[4] = {0, 0, 0, 0};
    int base = (threadIdx.x & 63) + (threadIdx.x >> 6) * 66 * 4;
    for (int it = 0; it < iters; ++it) {
        if (FORM == 3) {      // eight ds_read_b64 at 4-byte alignment in flight, ONE wait (the compiler does not emit the form itself: it cannot assume it may)
            f2 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned addr = (unsigned)(size_t)(&L[base + k * 66 + (it & 3)]);
                asm volatile("ds_read_b64 %0, %1" : "=v"(v[k]) : "v"(addr) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : : "memory");
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k & 3] += v[k].x + v[k].y;
            continue;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int b = base + k * 66 + (it & 3);
            if (FORM == 0) { acc[k & 3] += L[b]; }
            if (FORM == 1) { acc[k & 3] += L[b] + L[b + 66]; }        // -> ds_read2_b32 offset1 = 66
            if (FORM == 2) { acc[k & 3] += L[b] + L[b + 1]; }         // -> ds_read2_b32 offset1 = 1
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

// LDS-DMA with a source that is only 4-byte aligned
__global__ __launch_bounds__(64) void dma_probe(const float* src, int shift, float* out) {
    __shared__ __attribute__((aligned(16))) float L[256];
    for (int i = threadIdx.x; i < 256; i += 64) L[i] = -1.f;
    __syncthreads();
    const float* g = src + shift + 4 * threadIdx.x;       // lane l: 16 bytes at element shift + 4 l
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)L, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = L[i];
}

static double time_ms(hipEvent_t a, hipEvent_t b) { float ms; hipEventElapsedTime(&ms, a, b); return ms; }

int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const double ghz = pr.clockRate * 1e-6;
    const int cus = pr.multiProcessorCount;
    printf("device %s, %d CUs, %.2f GHz (clockRate)\n", pr.name, cus, ghz);
    float* out; hipMalloc(&out, 1 << 26);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"v_fma_f32", "v_pk_fma_f32 (2 fma per lane)", "v_add_u32", "v_floor_f32", "v_fract_f32", "v_cvt_i32_f32",
                           "v_med3_f32", "v_cmp_gt_i32 + v_cndmask_b32 (2)", "v_mul_i32_i24", "v_mul_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_lshl_add_u32", "v_add3_u32", "v_max3_f32 |.|"};
    for (int wps : {4, 8}) {        // waves per SIMD
        const int blocks = cus * wps;      // 256 threads = 4 waves = one per SIMD
        const int iters = 4096;
        for (int op = 0; op <= 14; ++op) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                switch (op) {
                    case 0: hipLaunchKernelGGL(valu<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f); break;
                    case 1: hipLaunchKernelGGL(valu<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f); break;
                    case 2: hipLaunchKernelGGL(valu<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f); break;
                    case 3: hipLaunchKernelGGL(valu<3>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f); break;
                    case 4: hipLaunchKernelGGL(valu<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f); break;
                    case 5: hipLaunchKernelGGL(valu<5>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f); break;
                    case 6: hipLaunchKernelGGL(valu<6>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f); break;
                    case 7: hipLaunchKernelGGL(valu<7>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f); break;
                    case 8: hipLaunchKernelGGL(valu<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f); break;
                    case 9: hipLaunchKernelGGL(valu<9>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f); break;
                    case 10: hipLaunchKernelGGL(valu<10>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f); break;
                    case 11: hipLaunchKernelGGL(valu<11>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f); break;
                    case 12: hipLaunchKernelGGL(valu<12>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f); break;
                    case 13: hipLaunchKernelGGL(valu<13>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f); break;
                    case 14: hipLaunchKernelGGL(valu<14>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            const double ms = time_ms(e0, e1);
            // per SIMD: wps waves x iters x CHAINS source statements
            const double stmts = (double)wps * iters * CHAINS;
            printf("VALU  %d waves/SIMD  %-36s %8.3f ms  %6.2f cycles per statement and wave on its SIMD (at clockRate)\n", wps, names[op], ms, ms * 1e-3 * ghz * 1e9 / stmts);
        }
    }
    const char* lnames[] = {"ds_read_b32", "2 dwords 66 apart (read2 offset1=66)", "2 adjacent dwords (read2 offset1=1)", "ds_read_b64 at 4-byte alignment (asm)"};
    for (int wps : {4, 8}) {
        const int blocks = cus * wps / 1;
        const int iters = 2048;
        for (int f = 0; f < 4; ++f) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                switch (f) {
                    case 0: hipLaunchKernelGGL(ldsread<0>, dim3(blocks), dim3(256), 0, 0, out, iters); break;
                    case 1: hipLaunchKernelGGL(ldsread<1>, dim3(blocks), dim3(256), 0, 0, out, iters); break;
                    case 2: hipLaunchKernelGGL(ldsread<2>, dim3(blocks), dim3(256), 0, 0, out, iters); break;
                    case 3: hipLaunchKernelGGL(ldsread<3>, dim3(blocks), dim3(256), 0, 0, out, iters); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            const double ms = time_ms(e0, e1);
            const double per_cu = (double)wps * 4 * iters * 8;       // wave-level read statements per CU
            float chk = 0; hipMemcpy(&chk, out + 77, 4, hipMemcpyDeviceToHost);
            printf("LDS   %d waves/SIMD  %-44s %8.3f ms  %6.2f cycles per statement and CU   (checksum of thread 77: %.0f)\n", wps, lnames[f], ms, ms * 1e-3 * ghz * 1e9 / per_cu, chk);
        }
    }
    // LDS-DMA from 4-byte-aligned sources
    float* src; hipMalloc(&src, 4096 * 4);
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (int shift : {0, 1, 2, 3, 5, 255}) {
        hipMemset(out, 0, 1024);
        hipLaunchKernelGGL(dma_probe, dim3(1), dim3(64), 0, 0, src, shift, out);
        hipError_t e = hipDeviceSynchronize();
        std::vector<float> r(256);
        hipMemcpy(r.data(), out, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; ++i) bad += r[i] != (float)(shift + i);
        printf("LDS-DMA dwordx4, source shifted by %3d elements (%s): %s, %d of 256 words wrong (first: %.0f %.0f %.0f %.0f %.0f)\n", shift, shift % 4 ? "4-byte aligned only" : "16-byte aligned",
               hipGetErrorString(e), bad, r[0], r[1], r[2], r[3], r[4]);
    }
    return 0;
}
