// project.hip -- the non-iterative stencils of fluid.make_incompressible (/root/reference phi/physics/fluid.py:94-162):
// divergence of the staggered velocity (+ active mask, + mean balance), pressure-gradient subtraction, obstacle flags,
// soft obstacle scaling, and the explicit diffusion stencil (phi/physics/diffuse.py:13-60).
// All are single-pass HBM-bound kernels: one thread per output sample, fast axis on consecutive lanes.
#include <stdlib.h>

#include "common.hpp"

namespace phihip {

template <typename T>
struct Comp3 {
    T* p[3];
};
template <typename T>
struct CComp3 {
    const T* p[3];
};
// V elements at 16-byte alignment: 16 bytes (fp32 V = 4, fp64 V = 2) or 32 bytes (fp64 V = 4: two dwordx4 -- r4: with V = 2 the fp64
// instantiations of the vector kernels below issue one address per 16 bytes and are bound by the address units, not by HBM)
template <typename T, int V>
struct alignas(16) VecA {
    T v[V];
};
template <typename T, int V>
__device__ __forceinline__ VecA<T, V> veca_load(const T* p) { return *reinterpret_cast<const VecA<T, V>*>(p); }
template <typename T, int V>
__device__ __forceinline__ void veca_store(T* p, const VecA<T, V>& x) { *reinterpret_cast<VecA<T, V>*>(p) = x; }
template <typename T, int V>
__device__ __forceinline__ VecA<T, V> veca_zero() {
    VecA<T, V> r;
#pragma unroll
    for (int i = 0; i < V; ++i) r.v[i] = T(0);
    return r;
}
// cells per thread of the vector kernels: 16 bytes (fp32 4, fp64 2). The 32-byte fp64 form (4 cells: two dwordx4 per lane at a lane stride
// of 32 bytes, so each instruction touches every other 16 bytes of its lines) was the default for a few commits of r4 and LOST on the same
// box against the 16-byte form: 384^3 fp64 closed gradient subtraction 0.64 -> 0.885 ms, divergence 0.384 -> 0.432, resample 0.56 -> 0.78
// (profiles/r04_time_frow_session_h.jsonl). It stays instantiated behind PHIHIP_F64_VEC_CELLS=4 for A/B runs only.
static inline int f64_vec_cells_setting() {
    static const int v = [] { const char* e = getenv("PHIHIP_F64_VEC_CELLS"); return (e && e[0] == '4') ? 4 : 2; }();
    return v;
}
template <typename T>
static inline int vec_cells(int n2) { return sizeof(T) == 4 ? 4 : ((f64_vec_cells_setting() == 4 && n2 % 4 == 0) ? 4 : 2); }
// V elements at ELEMENT alignment (rows of n2 - 1 / n2 + 1 faces do not start on 16-byte boundaries; gfx950 takes dwordx4 at any 4-byte address)
template <typename T, int V>
struct __attribute__((packed, aligned(sizeof(T)))) VecU {
    T v[V];
};

// Value of velocity component `ca` at stored index (i0,i1,i2) with the velocity extrapolation applied outside the array.
// Mixed boundaries follow PhiML's sequential padding: the LAST axis that lies outside a constant side decides.
template <typename T>
__device__ __forceinline__ T fetch_comp(const T* C, const VelGrid& g, int ca, long long bbase, int i0, int i1, int i2) {
    int idx[3] = {i0, i1, i2};
#pragma unroll
    for (int ax = 2; ax >= 0; --ax) {
        if (ax < g.ax0) { idx[ax] = 0; continue; }
        const int n = g.cn[ca][ax];
        int i = idx[ax];
        if (i < 0) {
            const int code = g.bc[ax][0];
            if (code == PHIHIP_BC_PERIODIC) { i %= n; if (i < 0) i += n; }
            else if (code == PHIHIP_BC_CLOSED) return (T)g.bcv[ax][0][ca];
            else i = 0;
        } else if (i >= n) {
            const int code = g.bc[ax][1];
            if (code == PHIHIP_BC_PERIODIC) i %= n;
            else if (code == PHIHIP_BC_CLOSED) return (T)g.bcv[ax][1][ca];
            else i = n - 1;
        }
        idx[ax] = i;
    }
    return C[bbase + ((long long)idx[0] * g.cn[ca][1] + idx[1]) * g.cn[ca][2] + idx[2]];
}

// ---------------------------------------------------------------------------------------------------------------------
// divergence (phi/field/_field_math.py:617-626 with bake_extrapolation :20-39)
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kMaxPartialBlocks = 4096;   // patch-stride kernels: bounded number of per-block partial sums
constexpr int kPatchCols = 64, kPatchRows = kBlock / kPatchCols;   // a workgroup visits (4 rows x 64 columns) patches of one a0 plane

// uniform decode of a patch number into (plane, first row, first column): the per-thread div / mod of a linear cell index used to be
// a third of this kernel's instructions
__device__ __forceinline__ void decode_patch(int patch, int patches1, int patches2, int& i0, int& r0, int& c0) {
    const int t = patch / patches2;
    c0 = (patch - t * patches2) * kPatchCols;
    i0 = t / patches1;
    r0 = (t - i0 * patches1) * kPatchRows;
}

// one face index of component `ax` along its own axis under the velocity extrapolation: stored index to read, or -1 / -2 when the
// lower / upper CONSTANT side supplies the value (bake_extrapolation, phi/field/_field_math.py:20-39)
__device__ __forceinline__ int face_index(int i, int n, int code_lo, int code_hi) {
    if (i < 0) return code_lo == PHIHIP_BC_PERIODIC ? i + n : (code_lo == PHIHIP_BC_CLOSED ? -1 : 0);
    if (i >= n) return code_hi == PHIHIP_BC_PERIODIC ? i - n : (code_hi == PHIHIP_BC_CLOSED ? -2 : n - 1);
    return i;
}

// A workgroup owns a (4 rows x 64 columns) column of cells and marches over a chunk of a0 planes: the in-plane face offsets (and the
// boundary rule of the a1 / a2 faces) are resolved once per thread, the upper a0 face of one plane is the lower face of the next
// (5 loads + 1 store per cell, no integer division anywhere), and the per-workgroup partial sums of div and of the active cells
// feed fluid._balance_divergence.
template <typename T, int DIM>
__global__ __launch_bounds__(kBlock) void divergence_kernel(VelGrid g, CComp3<T> v, const uint8_t* flags, int flags_per_batch,
                                                            T* __restrict__ div, double* part_sum, double* part_act, int nblk, int tiles1,
                                                            int tiles2, int chunk, int finite_guard) {
    __shared__ double red[kBlock / kWave];
    const int b = blockIdx.y;
    const int n0 = g.n[0], n1 = g.n[1], n2 = g.n[2];
    const int tx = threadIdx.x & (kPatchCols - 1), ty = threadIdx.x / kPatchCols;
    const int t2 = blockIdx.x % tiles2;
    const int t1 = (blockIdx.x / tiles2) % tiles1;
    const int c0 = blockIdx.x / (tiles2 * tiles1);
    const int i1 = t1 * kPatchRows + ty, i2 = t2 * kPatchCols + tx;
    const bool inside = i1 < n1 && i2 < n2;
    const int p0 = DIM == 3 ? c0 * chunk : 0, p1 = DIM == 3 ? min(p0 + chunk, n0) : 1;
    const T r0 = (T)g.rdx[0], r1 = (T)g.rdx[1], r2 = (T)g.rdx[2];   // (a true division is ~10 instructions; 3 per cell made the kernel VALU-heavy)
    // in-plane taps: component 1 at rows (i1 - off1, +1), component 2 at columns (i2 - off2, +1)
    int o1[2] = {0, 0}, o2[2] = {0, 0};
    T k1[2] = {T(0), T(0)}, k2[2] = {T(0), T(0)};
    bool c1f[2] = {false, false}, c2f[2] = {false, false};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int f1 = face_index(i1 - g.off[1] + k, g.cn[1][1], g.bc[1][0], g.bc[1][1]);
        const int f2 = face_index(i2 - g.off[2] + k, g.cn[2][2], g.bc[2][0], g.bc[2][1]);
        c1f[k] = f1 < 0; k1[k] = (T)(f1 == -1 ? g.bcv[1][0][1] : g.bcv[1][1][1]); o1[k] = (f1 < 0 ? 0 : f1) * g.cn[1][2] + i2;
        c2f[k] = f2 < 0; k2[k] = (T)(f2 == -1 ? g.bcv[2][0][2] : g.bcv[2][1][2]); o2[k] = i1 * g.cn[2][2] + (f2 < 0 ? 0 : f2);
    }
    const int o0 = i1 * g.cn[0][2] + i2;                                   // component 0: same (i1, i2) in every plane
    const long long ps0 = (long long)g.cn[0][1] * g.cn[0][2], ps1 = (long long)g.cn[1][1] * g.cn[1][2], ps2 = (long long)g.cn[2][1] * g.cn[2][2];
    const T* __restrict__ C0 = DIM == 3 ? v.p[0] + (long long)b * g.ccells[0] : nullptr;
    const T* __restrict__ C1 = v.p[1] + (long long)b * g.ccells[1];
    const T* __restrict__ C2 = v.p[2] + (long long)b * g.ccells[2];
    const uint8_t* F = flags ? flags + (flags_per_batch ? (long long)b * g.cells : 0) : nullptr;
    T* __restrict__ D = div + (long long)b * g.cells;
    auto face0 = [&](int phys) -> T {      // component 0 at physical face `phys` of this thread's column (uniform boundary decision)
        const int f = face_index(phys - g.off[0], g.cn[0][0], g.bc[0][0], g.bc[0][1]);
        if (f < 0) return (T)(f == -1 ? g.bcv[0][0][0] : g.bcv[0][1][0]);
        return C0[(long long)f * ps0 + o0];
    };
    T acc_val = T(0), acc_act = T(0);
    if (inside && p0 < p1) {
        T lo0 = DIM == 3 ? face0(p0) : T(0);
        for (int p = p0; p < p1; ++p) {
            T sum = T(0);
            if (DIM == 3) {
                const T hi0 = face0(p + 1);
                sum += (hi0 - lo0) * r0;
                lo0 = hi0;
            }
            const T a = c1f[0] ? k1[0] : C1[(long long)p * ps1 + o1[0]], bb = c1f[1] ? k1[1] : C1[(long long)p * ps1 + o1[1]];
            sum += (bb - a) * r1;
            const T c = c2f[0] ? k2[0] : C2[(long long)p * ps2 + o2[0]], d = c2f[1] ? k2[1] : C2[(long long)p * ps2 + o2[1]];
            sum += (d - c) * r2;
            const long long cell = ((long long)p * n1 + i1) * n2 + i2;
            T act = T(1);
            if (F) {
                const unsigned f = F[cell];
                act = (f & 64u) ? T(1) : T(0);
                sum = (f & 64u) ? sum : T(0);   // div * active, and non-finite values of inactive cells do not leak (fluid.py:139-144)
            }
            if (finite_guard) sum = __builtin_isfinite(sum) ? sum : T(0);   // field.where(field.is_finite(div), div, 0) on EVERY cell (fluid.py:143-144)
            D[cell] = sum;
            acc_val += sum;
            acc_act += act;
        }
    }
    const double s1 = block_sum((double)acc_val, red);
    const double s2 = block_sum((double)acc_act, red);
    if (threadIdx.x == 0) {
        part_sum[(long long)b * nblk + blockIdx.x] = s1;
        part_act[(long long)b * nblk + blockIdx.x] = s2;
    }
}

// The same pass with one 16-byte vector of the fast axis per thread (r4; the scalar kernel above issues 5 dword loads and 1 dword store per
// cell and ran at 0.50 of the HBM rate, bound by the address units like every one-dword-per-lane kernel here): per V cells
//   a0: ONE aligned vector of component 0 per plane (the upper faces of this plane are the lower faces of the next: registers),
//   a1: the two rows of component 1 around the cells (aligned vectors; the second is an L1 / L2 hit of the neighbouring thread row),
//   a2: the V + 1 faces around the cells = one vector of component 2 at ELEMENT alignment (its rows hold n2 - 1 / n2 / n2 + 1 faces) plus
//       the face at the open end as a scalar with the velocity's boundary rule (wrap / wall constant),
//   flags as one V-byte load, div as one aligned vector store. Rows must be whole vectors (n2 % V == 0), else the scalar kernel runs.
// `ltpr`: log2 of the threads along a row (narrow grids put more rows into a workgroup instead of idle lanes).
template <typename T, int DIM, int V>
__global__ __launch_bounds__(kBlock) void divergence_vec_kernel(VelGrid g, CComp3<T> v, const uint8_t* flags, int flags_per_batch,
                                                                T* __restrict__ div, double* part_sum, double* part_act, int nblk, int tiles1,
                                                                int tiles2, int chunk, int ltpr, int finite_guard) {
    using VT = VecA<T, V>;
    using VU = VecU<T, V>;
    using VF = Vec<uint8_t, V>;
    __shared__ double red[kBlock / kWave];
    const int b = blockIdx.y;
    const int n0 = g.n[0], n1 = g.n[1], n2 = g.n[2];
    const int tx = threadIdx.x & ((1 << ltpr) - 1), ty = threadIdx.x >> ltpr;
    const int bid = xcd_order(blockIdx.x, nblk);                // neighbouring columns share an XCD's L2 (the a1 rows between them)
    const int t2 = bid % tiles2;
    const int t1 = (bid / tiles2) % tiles1;
    const int c0 = bid / (tiles2 * tiles1);
    const int i1 = t1 * (kBlock >> ltpr) + ty, i2 = ((t2 << ltpr) + tx) * V;
    const bool inside = i1 < n1 && i2 < n2;
    const int p0 = DIM == 3 ? c0 * chunk : 0, p1 = DIM == 3 ? min(p0 + chunk, n0) : 1;
    const T r0 = (T)g.rdx[0], r1 = (T)g.rdx[1], r2 = (T)g.rdx[2];
    const int cn2 = g.cn[2][2];
    // component 1: rows (i1 - off1, + 1); component 2: main vector at stored columns i2 .. i2 + V - 1 (faces k = 0 .. V - 1 of the V + 1 when
    // the lower face of a cell is stored, off2 = 0; faces k = 1 .. V when the wall face is not, off2 = 1) + the face at the other end
    int o1[2] = {0, 0};
    T k1[2] = {T(0), T(0)};
    bool c1f[2] = {false, false};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int f1 = face_index(i1 - g.off[1] + k, g.cn[1][1], g.bc[1][0], g.bc[1][1]);
        c1f[k] = f1 < 0; k1[k] = (T)(f1 == -1 ? g.bcv[1][0][1] : g.bcv[1][1][1]); o1[k] = (f1 < 0 ? 0 : f1) * g.cn[1][2] + i2;
    }
    const int off2 = g.off[2];
    const bool full2 = i2 + V <= cn2;                                      // (false only for the last vector of a row between two CLOSED sides)
    const int fe = face_index(off2 ? i2 - 1 : i2 + V, cn2, g.bc[2][0], g.bc[2][1]);
    const bool cef = fe < 0;
    const T ke = (T)(fe == -1 ? g.bcv[2][0][2] : g.bcv[2][1][2]);
    const int o2 = i1 * cn2 + i2, oe = i1 * cn2 + (fe < 0 ? 0 : fe);
    int ft[V];                                                             // tail vector: per-element face rule
    T kt[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        const int f = face_index(i2 + e, cn2, g.bc[2][0], g.bc[2][1]);
        ft[e] = f < 0 ? -1 : i1 * cn2 + f;
        kt[e] = (T)(f == -1 ? g.bcv[2][0][2] : g.bcv[2][1][2]);
    }
    const int o0 = i1 * g.cn[0][2] + i2;
    const long long ps0 = (long long)g.cn[0][1] * g.cn[0][2], ps1 = (long long)g.cn[1][1] * g.cn[1][2], ps2 = (long long)g.cn[2][1] * cn2;
    const T* __restrict__ C0 = DIM == 3 ? v.p[0] + (long long)b * g.ccells[0] : nullptr;
    const T* __restrict__ C1 = v.p[1] + (long long)b * g.ccells[1];
    const T* __restrict__ C2 = v.p[2] + (long long)b * g.ccells[2];
    const uint8_t* F = flags ? flags + (flags_per_batch ? (long long)b * g.cells : 0) : nullptr;
    T* __restrict__ D = div + (long long)b * g.cells;
    auto splat = [](T x) { VT r;
#pragma unroll
        for (int e = 0; e < V; ++e) r.v[e] = x;
        return r; };
    auto face0 = [&](int phys) -> VT {
        const int f = face_index(phys - g.off[0], g.cn[0][0], g.bc[0][0], g.bc[0][1]);
        if (f < 0) return splat((T)(f == -1 ? g.bcv[0][0][0] : g.bcv[0][1][0]));
        return veca_load<T, V>(C0 + (long long)f * ps0 + o0);
    };
    T acc_val = T(0), acc_act = T(0);
    if (inside && p0 < p1) {
        VT lo0 = DIM == 3 ? face0(p0) : splat(T(0));
        for (int p = p0; p < p1; ++p) {
            VT hi0 = lo0;
            if (DIM == 3) hi0 = face0(p + 1);
            const VT a = c1f[0] ? splat(k1[0]) : veca_load<T, V>(C1 + (long long)p * ps1 + o1[0]);
            const VT bb = c1f[1] ? splat(k1[1]) : veca_load<T, V>(C1 + (long long)p * ps1 + o1[1]);
            const T* R2 = C2 + (long long)p * ps2;
            VT m;
            if (full2) {
                const VU u = *reinterpret_cast<const VU*>(R2 + o2);
#pragma unroll
                for (int e = 0; e < V; ++e) m.v[e] = u.v[e];
            } else {
#pragma unroll
                for (int e = 0; e < V; ++e) m.v[e] = ft[e] < 0 ? kt[e] : R2[ft[e]];
            }
            const T edge = cef ? ke : R2[oe];
            const long long cell = ((long long)p * n1 + i1) * n2 + i2;
            VF fl;
            if (F) fl = *reinterpret_cast<const VF*>(F + cell);
            VT out;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const T c = off2 ? (e > 0 ? m.v[e > 0 ? e - 1 : 0] : edge) : m.v[e];
                const T d = off2 ? m.v[e] : (e < V - 1 ? m.v[e < V - 1 ? e + 1 : e] : edge);
                T sum = T(0);
                if (DIM == 3) sum += (hi0.v[e] - lo0.v[e]) * r0;
                sum += (bb.v[e] - a.v[e]) * r1;
                sum += (d - c) * r2;
                T act = T(1);
                if (F) {
                    const unsigned f = fl.v[e];
                    act = (f & 64u) ? T(1) : T(0);
                    sum = (f & 64u) ? sum : T(0);
                }
                if (finite_guard) sum = __builtin_isfinite(sum) ? sum : T(0);
                out.v[e] = sum;
                acc_val += sum;
                acc_act += act;
            }
            veca_store<T, V>(D + cell, out);
            lo0 = hi0;
        }
    }
    const double s1 = block_sum((double)acc_val, red);
    const double s2 = block_sum((double)acc_act, red);
    if (threadIdx.x == 0) {
        part_sum[(long long)b * nblk + blockIdx.x] = s1;
        part_act[(long long)b * nblk + blockIdx.x] = s2;
    }
}

// shift[b] = sum(div) / sum(active)   (= mean(div) / mean(active), fluid._balance_divergence)
__global__ __launch_bounds__(kBlock) void balance_scalar_kernel(const double* part_sum, const double* part_act, int nblk, double* shift) {
    __shared__ double red[kBlock / kWave];
    const int b = blockIdx.x;
    double s = 0, a = 0;
    for (int i = threadIdx.x; i < nblk; i += kBlock) {
        s += part_sum[(long long)b * nblk + i];
        a += part_act[(long long)b * nblk + i];
    }
    s = block_sum(s, red);
    a = block_sum(a, red);
    if (threadIdx.x == 0) shift[b] = a != 0 ? s / a : 0;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void balance_apply_kernel(T* __restrict__ div, const uint8_t* flags, int flags_per_batch, const double* shift,
                                                               long long cells) {
    const int b = blockIdx.y;
    const T sh = (T)shift[b];
    for (long long cell = (long long)blockIdx.x * kBlock + threadIdx.x; cell < cells; cell += (long long)gridDim.x * kBlock) {
        T a = T(1);
        if (flags) a = (flags[(flags_per_batch ? (long long)b * cells : 0) + cell] & 64u) ? T(1) : T(0);
        div[(long long)b * cells + cell] -= a * sh;
    }
}

// standalone balance: x -= active * sum(x) / sum(active)   (fluid._balance_divergence on an arbitrary cell field)
template <typename T>
__global__ __launch_bounds__(kBlock) void sum_cells_kernel(const T* __restrict__ x, const uint8_t* flags, int flags_per_batch, long long cells,
                                                           double* part_sum, double* part_act, int nblk) {
    __shared__ double red[kBlock / kWave];
    const int b = blockIdx.y;
    double sx = 0, sa = 0;
    for (long long c = (long long)blockIdx.x * kBlock + threadIdx.x; c < cells; c += (long long)gridDim.x * kBlock) {
        sx += (double)x[(long long)b * cells + c];
        sa += flags ? ((flags[(flags_per_batch ? (long long)b * cells : 0) + c] & 64u) ? 1.0 : 0.0) : 1.0;
    }
    const double s1 = block_sum(sx, red);
    const double s2 = block_sum(sa, red);
    if (threadIdx.x == 0) {
        part_sum[(long long)b * nblk + blockIdx.x] = s1;
        part_act[(long long)b * nblk + blockIdx.x] = s2;
    }
}

int run_balance(phihip_ctx* ctx, const GridView& v, const uint8_t* flags, int mask_batch, void* x, hipStream_t s) {
    const int nblk = ceil_div(v.cells, kBlock) < kMaxPartialBlocks ? ceil_div(v.cells, kBlock) : kMaxPartialBlocks;
    PHIHIP_TRY(ensure_buffer(ctx->ws_div, (size_t)2 * v.batch * nblk * sizeof(double)));
    PHIHIP_TRY(ensure_buffer(ctx->ws_scalars, (size_t)v.batch * sizeof(double)));
    double* part_sum = (double*)ctx->ws_div.ptr;
    double* part_act = part_sum + (size_t)v.batch * nblk;
    double* shift = (double*)ctx->ws_scalars.ptr;
    const int fpb = mask_batch > 1 ? 1 : 0;
    LaunchScope ls(ctx, PHIHIP_K_DIVERGENCE, s);
    const int nb2 = ceil_div(v.cells, kBlock) < 8192 ? ceil_div(v.cells, kBlock) : 8192;
    if (v.dtype == PHIHIP_F64) {
        hipLaunchKernelGGL(sum_cells_kernel<double>, dim3(nblk, v.batch), dim3(kBlock), 0, s, (const double*)x, flags, fpb, v.cells, part_sum, part_act, nblk);
        hipLaunchKernelGGL(balance_scalar_kernel, dim3(v.batch), dim3(kBlock), 0, s, (const double*)part_sum, (const double*)part_act, nblk, shift);
        hipLaunchKernelGGL(balance_apply_kernel<double>, dim3(nb2, v.batch), dim3(kBlock), 0, s, (double*)x, flags, fpb, (const double*)shift, v.cells);
    } else {
        hipLaunchKernelGGL(sum_cells_kernel<float>, dim3(nblk, v.batch), dim3(kBlock), 0, s, (const float*)x, flags, fpb, v.cells, part_sum, part_act, nblk);
        hipLaunchKernelGGL(balance_scalar_kernel, dim3(v.batch), dim3(kBlock), 0, s, (const double*)part_sum, (const double*)part_act, nblk, shift);
        hipLaunchKernelGGL(balance_apply_kernel<float>, dim3(nb2, v.batch), dim3(kBlock), 0, s, (float*)x, flags, fpb, (const double*)shift, v.cells);
    }
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

// vector path: rows of whole 16-byte vectors, 16-byte aligned div / a0 / a1 components (their rows have the cells' length), V-byte aligned
// flags; the a2 component only needs element alignment
template <typename T>
static bool divergence_vec_ok(const GridView& v, const void* const vel[3], const uint8_t* flags, const void* div, int V) {
    bool ok = v.n[2] % V == 0 && ((uintptr_t)div & 15u) == 0 && (!flags || ((uintptr_t)flags & (V - 1)) == 0);
    for (int ca = v.ax0; ca < 2; ++ca) ok = ok && ((uintptr_t)vel[ca] & 15u) == 0;
    return ok && ((uintptr_t)vel[2] & (sizeof(T) - 1)) == 0;
}

template <typename T, int DIM>
static int launch_divergence(phihip_ctx* ctx, const GridView& v, const VelGrid& g, const void* const vel[3], const uint8_t* flags, int fpb, void* div,
                             int finite_guard, int* nblk_out, hipStream_t s) {
    CComp3<T> c{{(const T*)vel[0], (const T*)vel[1], (const T*)vel[2]}};
    const int V = vec_cells<T>(v.n[2]);
    const bool vec = divergence_vec_ok<T>(v, vel, flags, div, V);
    int ltpr = 6;                                                 // threads along a row: 64, fewer on narrow grids (whole workgroup rows instead of idle lanes)
    if (vec) while (ltpr > 0 && (1 << (ltpr - 1)) * V >= v.n[2]) --ltpr;
    const int rows = vec ? kBlock >> ltpr : kPatchRows, cols = vec ? (1 << ltpr) * V : kPatchCols;
    // (rows x cols)-cell columns x chunks of planes: ~4096 workgroups per batch entry when the grid allows (2 rounds of 8 per CU)
    const int tiles1 = ceil_div(v.n[1], rows), tiles2 = ceil_div(v.n[2], cols);
    const long long tiles = (long long)tiles1 * tiles2;
    PHIHIP_REQUIRE(tiles <= (1 << 24), "divergence: grid too large");
    int chunks = v.rank == 3 ? (int)((4096 + tiles - 1) / tiles) : 1;
    chunks = chunks > v.n[0] ? v.n[0] : (chunks < 1 ? 1 : chunks);
    const int chunk_planes = ceil_div(v.n[0], chunks);
    chunks = ceil_div(v.n[0], chunk_planes);
    const int nblk = (int)tiles * chunks;
    PHIHIP_TRY(ensure_buffer(ctx->ws_div, (size_t)2 * v.batch * nblk * sizeof(double)));
    double* part_sum = (double*)ctx->ws_div.ptr;
    double* part_act = part_sum + (size_t)v.batch * nblk;
    if (vec && V == 4)
        hipLaunchKernelGGL((divergence_vec_kernel<T, DIM, 4>), dim3(nblk, v.batch), dim3(kBlock), 0, s, g, c, flags, fpb, (T*)div, part_sum, part_act, nblk,
                           tiles1, tiles2, chunk_planes, ltpr, finite_guard);
    else if (vec)
        hipLaunchKernelGGL((divergence_vec_kernel<T, DIM, 2>), dim3(nblk, v.batch), dim3(kBlock), 0, s, g, c, flags, fpb, (T*)div, part_sum, part_act, nblk,
                           tiles1, tiles2, chunk_planes, ltpr, finite_guard);
    else
        hipLaunchKernelGGL((divergence_kernel<T, DIM>), dim3(nblk, v.batch), dim3(kBlock), 0, s, g, c, flags, fpb, (T*)div, part_sum, part_act, nblk,
                           tiles1, tiles2, chunk_planes, finite_guard);
    *nblk_out = nblk;
    return PHIHIP_OK;
}

// balance: bit 0 = subtract the active-weighted mean (fluid._balance_divergence); value 2 (internal) = leave the shift in ctx->ws_scalars for the
// solver's first residual; PHIHIP_DIV_FINITE_GUARD (4) = non-finite divergence -> 0 on every cell (fluid.py:143-144: the user passed `active`)
int run_divergence(phihip_ctx* ctx, const GridView& v, const void* const vel[3], const uint8_t* flags, int mask_batch, int balance,
                   void* div, hipStream_t s) {
    if (v.cells >= (1LL << 31)) {
        set_error("divergence: more than 2^31 cells per batch entry are not supported");
        return PHIHIP_ERR_UNSUPPORTED;
    }
    const int finite_guard = (balance & PHIHIP_DIV_FINITE_GUARD) ? 1 : 0;
    balance &= ~PHIHIP_DIV_FINITE_GUARD;
    const VelGrid g = make_velgrid(v);
    PHIHIP_TRY(ensure_buffer(ctx->ws_scalars, (size_t)v.batch * sizeof(double)));
    const int fpb = mask_batch > 1 ? 1 : 0;
    int nblk = 0;
    {
        LaunchScope ls(ctx, PHIHIP_K_DIVERGENCE, s);
        if (v.dtype == PHIHIP_F64) {
            if (v.rank == 3) PHIHIP_TRY((launch_divergence<double, 3>(ctx, v, g, vel, flags, fpb, div, finite_guard, &nblk, s)));
            else PHIHIP_TRY((launch_divergence<double, 2>(ctx, v, g, vel, flags, fpb, div, finite_guard, &nblk, s)));
        } else {
            if (v.rank == 3) PHIHIP_TRY((launch_divergence<float, 3>(ctx, v, g, vel, flags, fpb, div, finite_guard, &nblk, s)));
            else PHIHIP_TRY((launch_divergence<float, 2>(ctx, v, g, vel, flags, fpb, div, finite_guard, &nblk, s)));
        }
    }
    double* part_sum = (double*)ctx->ws_div.ptr;
    double* part_act = part_sum + (size_t)v.batch * nblk;
    double* shift = (double*)ctx->ws_scalars.ptr;
    if (balance) {
        LaunchScope ls(ctx, PHIHIP_K_DIVERGENCE, s);
        hipLaunchKernelGGL(balance_scalar_kernel, dim3(v.batch), dim3(kBlock), 0, s, (const double*)part_sum, (const double*)part_act,
                           nblk, shift);
        if (balance == 2) {   // the caller folds the shift (ctx->ws_scalars) into the solver's initial residual
            PHIHIP_CHECK_HIP(hipGetLastError());
            return PHIHIP_OK;
        }
        const int nb2 = ceil_div(v.cells, kBlock) < 8192 ? ceil_div(v.cells, kBlock) : 8192;
        if (v.dtype == PHIHIP_F64)
            hipLaunchKernelGGL(balance_apply_kernel<double>, dim3(nb2, v.batch), dim3(kBlock), 0, s, (double*)div, flags, fpb,
                               (const double*)shift, v.cells);
        else
            hipLaunchKernelGGL(balance_apply_kernel<float>, dim3(nb2, v.batch), dim3(kBlock), 0, s, (float*)div, flags, fpb,
                               (const double*)shift, v.cells);
    }
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// v_d[f] -= h_f (p_R - p_L) / dx_d   (phi/physics/fluid.py:158-161; stagger :535-581)
// ---------------------------------------------------------------------------------------------------------------------
// ONE launch for all components: a thread owns the stored index (i0, i1, i2) in every component's array (they differ by at most one
// sample per axis), so the pressure cell it shares between the D faces is fetched once and the D neighbours come from lines the
// neighbouring lanes / rows touch anyway; the per-component launches read p three times.
template <typename T, int DIM>
__global__ __launch_bounds__(kBlock) void grad_subtract_kernel(VelGrid g, Comp3<T> vc, const T* __restrict__ p, const uint8_t* flags, int flags_per_batch,
                                                               int nmax0, int patches1, int patches2, int comps) {
    constexpr int A0 = 3 - DIM;
    const int b = blockIdx.y;
    const T* __restrict__ P = p + (long long)b * g.cells;
    const uint8_t* F = flags ? flags + (flags_per_batch ? (long long)b * g.cells : 0) : nullptr;
    const int tx = threadIdx.x & (kPatchCols - 1), ty = threadIdx.x / kPatchCols;
    const int npatch = nmax0 * patches1 * patches2;
    for (int patch = blockIdx.x; patch < npatch; patch += gridDim.x) {
        int idx[3], r0, c0;
        decode_patch(patch, patches1, patches2, idx[0], r0, c0);
        idx[1] = r0 + ty;
        idx[2] = c0 + tx;
#pragma unroll
        for (int ca = A0; ca < 3; ++ca) {
            if (!((comps >> ca) & 1)) continue;     // (the vector kernel above took the others)
            if (idx[0] >= g.cn[ca][0] || idx[1] >= g.cn[ca][1] || idx[2] >= g.cn[ca][2]) continue;
            const int n = g.n[ca];
            const int pstride = ca == 0 ? g.n[1] * g.n[2] : (ca == 1 ? g.n[2] : 1);
            const int phys = idx[ca] + g.off[ca];
            int l = phys - 1, r = phys;
            bool zl = false, zr = false;
            const bool l_in = l >= 0, r_in = r < n;
            if (!l_in) { if (g.bc[ca][0] == PHIHIP_BC_PERIODIC) l += n; else { zl = true; l = 0; } }
            if (!r_in) { if (g.bc[ca][1] == PHIHIP_BC_PERIODIC) r -= n; else { zr = true; r = n - 1; } }
            const int rest = (idx[0] * g.n[1] + idx[1]) * g.n[2] + idx[2] - idx[ca] * pstride;   // other axes coincide with cell indices
            const int offL = rest + l * pstride, offR = rest + r * pstride;
            const T pl = zl ? T(0) : P[offL];
            const T pr = zr ? T(0) : P[offR];
            T h = T(1);
            if (F) {
                // the face is the lower face of cell R (if R exists in the domain or by wrap) else the upper face of cell L
                if (r_in || g.bc[ca][1] == PHIHIP_BC_PERIODIC) h = (F[offR] >> (2 * ca)) & 1u ? T(1) : T(0);
                else if (l_in) h = (F[offL] >> (2 * ca + 1)) & 1u ? T(1) : T(0);
            }
            T* __restrict__ V = vc.p[ca] + (long long)b * g.ccells[ca];
            const int f = (idx[0] * g.cn[ca][1] + idx[1]) * g.cn[ca][2] + idx[2];
            V[f] = V[f] - h * ((pr - pl) * (T)g.rdx[ca]);
        }
    }
}

// The same update with one 16-byte vector of the fast axis per thread (V = 4 fp32 / 2 fp64 cells of a row); `comps` = bit mask of the
// components this launch updates. The scalar kernel above issues 9 dword loads (3 of them redundant) and 3 dword stores per cell.
//   a0 / a1: the two cells of a face are whole rows -- same columns, neighbouring plane / row: two aligned vector loads of p.
//   a2: the faces j = c .. c + V - 1 of a row lie between the cells of ONE aligned vector pc = p[c .. c + V - 1] and that vector shifted by
//       one cell; the cell that enters at the open end (p[c - 1] when the lower face of a cell is stored, off = 0; p[c + V] when the wall
//       face is not, off = 1) is one scalar load with the pressure's boundary rule (wrap / zero ghost). Rows of this component hold
//       n2 - 1 / n2 / n2 + 1 faces (closed / periodic or mixed / open), so its own vector is addressed with element alignment only
//       (VecU: the hardware takes dwordx4 at any 4-byte address); the last, partial vector of a row and the extra face of an OPEN upper
//       side (j = n2) are scalar. r3: closed and open boxes -- and with them every obstacle scenario -- used to send this component
//       through the scalar kernel in a second launch that read p and the flags again.
template <typename T, int DIM, int V>
__global__ __launch_bounds__(kBlock) void grad_subtract_vec_kernel(VelGrid g, Comp3<T> vc, const T* __restrict__ p, const uint8_t* flags, int flags_per_batch,
                                                                   int nmax0, int patches1, int patches2, int comps) {
    constexpr int A0 = 3 - DIM;
    using VT = VecA<T, V>;
    using VU = VecU<T, V>;
    using VF = Vec<uint8_t, V>;
    const int b = blockIdx.y;
    const T* __restrict__ P = p + (long long)b * g.cells;
    const uint8_t* F = flags ? flags + (flags_per_batch ? (long long)b * g.cells : 0) : nullptr;
    const int tx = threadIdx.x & (kPatchCols - 1), ty = threadIdx.x / kPatchCols;
    const int npatch = nmax0 * patches1 * patches2;
    const int n2 = g.n[2];
    const int cn2 = g.cn[2][2], off2 = g.off[2];
    const bool per2 = g.bc[2][0] == PHIHIP_BC_PERIODIC;
    for (int patch = blockIdx.x; patch < npatch; patch += gridDim.x) {
        int idx[3], r0, c0;
        decode_patch(patch, patches1, patches2, idx[0], r0, c0);
        idx[1] = r0 + ty;
        idx[2] = (c0 + tx) * V;          // (decode_patch counts columns in threads: kPatchCols threads x V cells)
        if (idx[2] >= n2) continue;
#pragma unroll
        for (int ca = A0; ca < 2; ++ca) {
            if (!((comps >> ca) & 1)) continue;
            if (idx[0] >= g.cn[ca][0] || idx[1] >= g.cn[ca][1]) continue;
            VT pl, pr;
            VF fl;
            bool use_f = false;
            int fshift = 2 * ca;
            // the neighbours along a0 / a1 are whole rows: same columns, other plane / row
            const int n = g.n[ca];
            const int pstride = ca == 0 ? g.n[1] * n2 : n2;
            const int phys = idx[ca] + g.off[ca];
            int l = phys - 1, r = phys;
            bool zl = false, zr = false;
            const bool l_in = l >= 0, r_in = r < n;
            if (!l_in) { if (g.bc[ca][0] == PHIHIP_BC_PERIODIC) l += n; else { zl = true; l = 0; } }
            if (!r_in) { if (g.bc[ca][1] == PHIHIP_BC_PERIODIC) r -= n; else { zr = true; r = n - 1; } }
            const int rest = (idx[0] * g.n[1] + idx[1]) * n2 + idx[2] - idx[ca] * pstride;
            const int offL = rest + l * pstride, offR = rest + r * pstride;
            pl = zl ? veca_zero<T, V>() : veca_load<T, V>(P + offL);
            pr = zr ? veca_zero<T, V>() : veca_load<T, V>(P + offR);
            if (F) {
                if (r_in || g.bc[ca][1] == PHIHIP_BC_PERIODIC) { fl = *reinterpret_cast<const VF*>(F + offR); use_f = true; }
                else if (l_in) { fl = *reinterpret_cast<const VF*>(F + offL); use_f = true; fshift = 2 * ca + 1; }
            }
            T* __restrict__ Vp = vc.p[ca] + (long long)b * g.ccells[ca] + ((long long)(idx[0] * g.cn[ca][1] + idx[1]) * n2 + idx[2]);
            VT u = veca_load<T, V>(Vp);
            const T rd = (T)g.rdx[ca];
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const T h = use_f ? (((fl.v[e] >> fshift) & 1u) ? T(1) : T(0)) : T(1);
                u.v[e] = u.v[e] - h * ((pr.v[e] - pl.v[e]) * rd);
            }
            veca_store<T, V>(Vp, u);
        }
        if (((comps >> 2) & 1) && idx[0] < g.cn[2][0] && idx[1] < g.cn[2][1]) {
            const int c = idx[2];
            const int row = (idx[0] * g.n[1] + idx[1]) * n2;
            const VT pc = veca_load<T, V>(P + row + c);
            VF fc;
            if (F) fc = *reinterpret_cast<const VF*>(F + row + c);
            // the cell beyond the open end of the vector, with the pressure's boundary rule (periodic wrap; ghost 0 outside an open / closed side:
            // a closed side stores no face there, the value is not used)
            T edge = T(0);
            if (off2 == 0) { if (c > 0) edge = P[row + c - 1]; else if (per2) edge = P[row + n2 - 1]; }
            else { if (c + V < n2) edge = P[row + c + V]; else if (per2) edge = P[row]; }
            VT d;        // p_R - p_L per face, h per face from the cell of the vector that owns the face's flag bit
            T hh[V];
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const T left = off2 == 0 ? (e > 0 ? pc.v[e > 0 ? e - 1 : 0] : edge) : pc.v[e];
                const T right = off2 == 0 ? pc.v[e] : (e < V - 1 ? pc.v[e < V - 1 ? e + 1 : e] : edge);
                d.v[e] = right - left;
                // off = 0: face j is the LOWER face of cell j (bit 4); off = 1: the UPPER face of cell j (bit 5) -- cellflags are symmetric
                hh[e] = F ? (((fc.v[e] >> (off2 == 0 ? 4 : 5)) & 1u) ? T(1) : T(0)) : T(1);
            }
            const T rd = (T)g.rdx[2];
            T* __restrict__ Vp = vc.p[2] + (long long)b * g.ccells[2] + ((long long)(idx[0] * g.cn[2][1] + idx[1]) * cn2 + c);
            if (c + V <= cn2) {
                VU u = *reinterpret_cast<const VU*>(Vp);
#pragma unroll
                for (int e = 0; e < V; ++e) u.v[e] = u.v[e] - hh[e] * (d.v[e] * rd);
                *reinterpret_cast<VU*>(Vp) = u;
            } else {
#pragma unroll
                for (int e = 0; e < V; ++e)
                    if (c + e < cn2) Vp[e] = Vp[e] - hh[e] * (d.v[e] * rd);
            }
            if (cn2 > n2 && c + V == n2) {     // OPEN upper side: the extra face j = n2 between the last cell and the zero ghost
                const T h = F ? (((fc.v[V - 1] >> 5) & 1u) ? T(1) : T(0)) : T(1);
                Vp[V] = Vp[V] - h * ((T(0) - pc.v[V - 1]) * rd);
            }
        }
    }
}

template <typename T, int DIM>
static void launch_grad_subtract(const GridView& v, const VelGrid& g, const uint8_t* flags, int fpb, const void* p, void* const vel[3],
                                 hipStream_t s) {
    int nmax[3] = {1, 1, 1};
    for (int a = 0; a < 3; ++a)
        for (int c = v.ax0; c < 3; ++c) nmax[a] = v.cn[c][a] > nmax[a] ? v.cn[c][a] : nmax[a];
    Comp3<T> c{{(T*)vel[0], (T*)vel[1], (T*)vel[2]}};
    const int V = vec_cells<T>(v.n[2]);
    int scalar_comps = 7;
    // vector path: rows of the pressure are whole vectors and the buffers of p, of the flags and of the a0 / a1 components are 16-byte
    // aligned (flags: V bytes); the a2 component only needs element alignment
    bool vec_ok = v.n[2] % V == 0 && ((uintptr_t)p & 15u) == 0 && (!flags || ((uintptr_t)flags & (V - 1)) == 0);
    for (int ca = v.ax0; ca < 2; ++ca) vec_ok = vec_ok && ((uintptr_t)vel[ca] & 15u) == 0;
    vec_ok = vec_ok && ((uintptr_t)vel[2] & (sizeof(T) - 1)) == 0;
    if (vec_ok) {
        const int vec_comps = 7;
        const int patches1 = ceil_div(nmax[1], kPatchRows), patches2 = ceil_div(v.n[2], kPatchCols * V);
        const long long npatch = (long long)nmax[0] * patches1 * patches2;
        const int nblk = npatch < 16384 ? (int)npatch : 16384;
        if (V == 4)
            hipLaunchKernelGGL((grad_subtract_vec_kernel<T, DIM, 4>), dim3(nblk, v.batch), dim3(kBlock), 0, s, g, c, (const T*)p, flags, fpb, nmax[0], patches1,
                               patches2, vec_comps);
        else
            hipLaunchKernelGGL((grad_subtract_vec_kernel<T, DIM, 2>), dim3(nblk, v.batch), dim3(kBlock), 0, s, g, c, (const T*)p, flags, fpb, nmax[0], patches1,
                               patches2, vec_comps);
        scalar_comps = 7 & ~vec_comps;
    }
    if (scalar_comps) {
        const int patches1 = ceil_div(nmax[1], kPatchRows), patches2 = ceil_div(nmax[2], kPatchCols);
        const long long npatch = (long long)nmax[0] * patches1 * patches2;
        const int nblk = npatch < 16384 ? (int)npatch : 16384;
        hipLaunchKernelGGL((grad_subtract_kernel<T, DIM>), dim3(nblk, v.batch), dim3(kBlock), 0, s, g, c, (const T*)p, flags, fpb, nmax[0], patches1, patches2,
                           scalar_comps);
    }
}

int run_grad_subtract(phihip_ctx* ctx, const GridView& v, const uint8_t* flags, int mask_batch, const void* p, void* const vel[3],
                      hipStream_t s) {
    for (int ca = v.ax0; ca < 3; ++ca)
        if (v.ccells[ca] >= (1LL << 31)) {
            set_error("grad_subtract: more than 2^31 samples per component and batch entry are not supported");
            return PHIHIP_ERR_UNSUPPORTED;
        }
    const VelGrid g = make_velgrid(v);
    const int fpb = mask_batch > 1 ? 1 : 0;
    LaunchScope ls(ctx, PHIHIP_K_GRAD_SUBTRACT, s);
    if (v.dtype == PHIHIP_F64) {
        if (v.rank == 3) launch_grad_subtract<double, 3>(v, g, flags, fpb, p, vel, s);
        else launch_grad_subtract<double, 2>(v, g, flags, fpb, p, vel, s);
    } else {
        if (v.rank == 3) launch_grad_subtract<float, 3>(v, g, flags, fpb, p, vel, s);
        else launch_grad_subtract<float, 2>(v, g, flags, fpb, p, vel, s);
    }
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// soft obstacle mask: v_d *= m_d  (apply_boundary_conditions for stationary obstacles, fluid.py:231-233; m = 1 - mask)
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void scale_kernel(T* v, const T* m, long long total, int mask_batched) {
    const int b = blockIdx.y;
    for (long long f = (long long)blockIdx.x * kBlock + threadIdx.x; f < total; f += (long long)gridDim.x * kBlock) {
        const T mv = m[(mask_batched ? (long long)b * total : 0) + f];
        const T x = v[(long long)b * total + f];
        v[(long long)b * total + f] = mv == T(0) ? T(0) : mv * x;   // safe_mul: 0 * nan = 0
    }
}

int run_scale_faces(phihip_ctx* ctx, const GridView& v, void* const vel[3], const void* const m[3], hipStream_t s) {
    LaunchScope ls(ctx, PHIHIP_K_OTHER, s);
    for (int ca = v.ax0; ca < 3; ++ca) {
        const int nblk = ceil_div(v.ccells[ca], kBlock) < 8192 ? ceil_div(v.ccells[ca], kBlock) : 8192;
        if (v.dtype == PHIHIP_F64)
            hipLaunchKernelGGL(scale_kernel<double>, dim3(nblk, v.batch), dim3(kBlock), 0, s, (double*)vel[ca], (const double*)m[ca],
                               v.ccells[ca], 0);
        else
            hipLaunchKernelGGL(scale_kernel<float>, dim3(nblk, v.batch), dim3(kBlock), 0, s, (float*)vel[ca], (const float*)m[ca],
                               v.ccells[ca], 0);
    }
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// resample(s * vector, to=velocity): centred scalar times a constant vector sampled at the stored faces
// (sample_grid_at_faces, phi/field/_resample.py:272-276): mean of the two adjacent cells, outside cells from the scalar's
// extrapolation. accumulate != 0: out += value (buoyancy: v + resample(smoke * (0, 0.1), to=v), Smoke_Plume.ipynb cell 5)
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void centered_to_staggered_kernel(VelGrid g, ScalarBc sb, int ca, const T* __restrict__ sfield,
                                                                       T* __restrict__ out, T scale, int accumulate) {
    const int b = blockIdx.y;
    const int total = (int)g.ccells[ca];
    const int c1 = g.cn[ca][1], c2 = g.cn[ca][2];
    const int n = g.n[ca];
    const int pstride = ca == 0 ? g.n[1] * g.n[2] : (ca == 1 ? g.n[2] : 1);
    const T* __restrict__ S = sfield + (long long)b * g.cells;
    T* __restrict__ O = out + (long long)b * total;
    for (int f = blockIdx.x * kBlock + threadIdx.x; f < total; f += gridDim.x * kBlock) {
        int idx[3];
        idx[2] = f % c2;
        const int t = f / c2;
        idx[1] = t % c1;
        idx[0] = t / c1;
        const int phys = idx[ca] + g.off[ca];
        int l = phys - 1, r = phys;
        bool cl = false, cr = false;
        if (l < 0) { if (sb.bc[ca][0] == PHIHIP_BC_PERIODIC) l += n; else { cl = sb.bc[ca][0] == PHIHIP_BC_CLOSED; l = 0; } }
        if (r >= n) { if (sb.bc[ca][1] == PHIHIP_BC_PERIODIC) r -= n; else { cr = sb.bc[ca][1] == PHIHIP_BC_CLOSED; r = n - 1; } }
        const int rest = (idx[0] * g.n[1] + idx[1]) * g.n[2] + idx[2] - idx[ca] * pstride;
        const T sl = (cl ? (T)sb.val[ca][0] : S[rest + l * pstride]) * scale;
        const T sr = (cr ? (T)sb.val[ca][1] : S[rest + r * pstride]) * scale;
        const T val = sl * T(0.5) + sr * T(0.5);
        O[f] = accumulate ? O[f] + val : val;
    }
}

// ONE launch for all components with one 16-byte vector of the fast axis per thread (r4; the per-component kernel above re-derives three
// indices per sample with integer divisions and reads the scalar once per component with dword loads: 69 us per component at 256^3, 0.36
// of the HBM rate). Same structure as grad_subtract_vec_kernel -- both are "face value from the two adjacent cells":
//   a0 / a1: the two cells of a face are whole rows (same columns, neighbouring plane / row): two aligned vector loads of s;
//   a2: the faces of a row lie between the cells of ONE aligned vector and that vector shifted by one cell; the cell beyond the open end
//       comes from one scalar load or from the scalar's extrapolation (wrap / edge cell / constant). Rows of this component hold
//       n2 - 1 / n2 / n2 + 1 faces: element-aligned vector access, masked tail, the extra face of an OPEN upper side as a scalar.
// `comps`: bit mask of the components to write (accumulate: those with a non-zero vector entry).
template <typename T, int DIM, int V>
__global__ __launch_bounds__(kBlock) void centered_to_staggered_vec_kernel(VelGrid g, ScalarBc sb, Comp3<T> oc, const T* __restrict__ sfield, T sc0, T sc1,
                                                                           T sc2, int accumulate, int comps, int nmax0, int patches1, int patches2,
                                                                           int ltpr) {
    constexpr int A0 = 3 - DIM;
    using VT = VecA<T, V>;
    using VU = VecU<T, V>;
    const int b = blockIdx.y;
    const T* __restrict__ S = sfield + (long long)b * g.cells;
    const int tx = threadIdx.x & ((1 << ltpr) - 1), ty = threadIdx.x >> ltpr;
    const int rows = kBlock >> ltpr;
    const int npatch = nmax0 * patches1 * patches2;
    const int n2 = g.n[2];
    const int cn2 = g.cn[2][2], off2 = g.off[2];
    const T scale[3] = {sc0, sc1, sc2};
    for (int patch = blockIdx.x; patch < npatch; patch += gridDim.x) {
        int idx[3];
        const int t = patch / patches2;
        idx[0] = t / patches1;
        idx[1] = (t - idx[0] * patches1) * rows + ty;
        idx[2] = (((patch - t * patches2) << ltpr) + tx) * V;
        if (idx[2] >= n2) continue;
#pragma unroll
        for (int ca = A0; ca < 2; ++ca) {
            if (!((comps >> ca) & 1)) continue;
            if (idx[0] >= g.cn[ca][0] || idx[1] >= g.cn[ca][1]) continue;
            const int n = g.n[ca];
            const int pstride = ca == 0 ? g.n[1] * n2 : n2;
            const int phys = idx[ca] + g.off[ca];
            int l = phys - 1, r = phys;
            bool cl = false, cr = false;
            if (l < 0) { if (sb.bc[ca][0] == PHIHIP_BC_PERIODIC) l += n; else { cl = sb.bc[ca][0] == PHIHIP_BC_CLOSED; l = 0; } }
            if (r >= n) { if (sb.bc[ca][1] == PHIHIP_BC_PERIODIC) r -= n; else { cr = sb.bc[ca][1] == PHIHIP_BC_CLOSED; r = n - 1; } }
            const int rest = (idx[0] * g.n[1] + idx[1]) * n2 + idx[2] - idx[ca] * pstride;
            VT sl = veca_load<T, V>(S + rest + l * pstride), sr = veca_load<T, V>(S + rest + r * pstride);
            T* __restrict__ Op = oc.p[ca] + (long long)b * g.ccells[ca] + ((long long)(idx[0] * g.cn[ca][1] + idx[1]) * n2 + idx[2]);
            VT o;
            if (accumulate) o = veca_load<T, V>(Op);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const T a = (cl ? (T)sb.val[ca][0] : sl.v[e]) * scale[ca], c = (cr ? (T)sb.val[ca][1] : sr.v[e]) * scale[ca];
                const T val = a * T(0.5) + c * T(0.5);
                o.v[e] = accumulate ? o.v[e] + val : val;
            }
            veca_store<T, V>(Op, o);
        }
        if (((comps >> 2) & 1) && idx[0] < g.cn[2][0] && idx[1] < g.cn[2][1]) {
            const int c = idx[2];
            const int row = (idx[0] * g.n[1] + idx[1]) * n2;
            const VT pc = veca_load<T, V>(S + row + c);
            // the cell beyond the open end of the vector under the scalar's extrapolation: wrap, the edge cell itself (zero-gradient), or the constant
            T edge;
            if (off2 == 0) {
                if (c > 0) edge = S[row + c - 1];
                else edge = sb.bc[2][0] == PHIHIP_BC_PERIODIC ? S[row + n2 - 1] : (sb.bc[2][0] == PHIHIP_BC_CLOSED ? (T)sb.val[2][0] : pc.v[0]);
            } else {
                if (c + V < n2) edge = S[row + c + V];
                else edge = sb.bc[2][1] == PHIHIP_BC_PERIODIC ? S[row] : (sb.bc[2][1] == PHIHIP_BC_CLOSED ? (T)sb.val[2][1] : pc.v[V - 1]);
            }
            T val[V];
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const T left = off2 == 0 ? (e > 0 ? pc.v[e > 0 ? e - 1 : 0] : edge) : pc.v[e];
                const T right = off2 == 0 ? pc.v[e] : (e < V - 1 ? pc.v[e < V - 1 ? e + 1 : e] : edge);
                val[e] = (left * scale[2]) * T(0.5) + (right * scale[2]) * T(0.5);
            }
            T* __restrict__ Op = oc.p[2] + (long long)b * g.ccells[2] + ((long long)(idx[0] * g.cn[2][1] + idx[1]) * cn2 + c);
            if (c + V <= cn2) {
                VU o;
                if (accumulate) o = *reinterpret_cast<const VU*>(Op);
#pragma unroll
                for (int e = 0; e < V; ++e) o.v[e] = accumulate ? o.v[e] + val[e] : val[e];
                *reinterpret_cast<VU*>(Op) = o;
            } else {
#pragma unroll
                for (int e = 0; e < V; ++e)
                    if (c + e < cn2) Op[e] = accumulate ? Op[e] + val[e] : val[e];
            }
            if (off2 == 0 && cn2 > n2 && c + V == n2) {     // OPEN upper side of the velocity: the extra face j = n2 between the last cell and the scalar's outside value
                const T outside = sb.bc[2][1] == PHIHIP_BC_PERIODIC ? S[row] : (sb.bc[2][1] == PHIHIP_BC_CLOSED ? (T)sb.val[2][1] : pc.v[V - 1]);
                const T v2 = (pc.v[V - 1] * scale[2]) * T(0.5) + (outside * scale[2]) * T(0.5);
                Op[V] = accumulate ? Op[V] + v2 : v2;
            }
        }
    }
}

template <typename T, int DIM>
static bool launch_c2s_vec(const GridView& v, const VelGrid& g, const ScalarBc& sb, const void* sfield, const double vector[3], int accumulate,
                           void* const out[3], hipStream_t s) {
    const int V = vec_cells<T>(v.n[2]);
    bool ok = v.n[2] % V == 0 && ((uintptr_t)sfield & 15u) == 0;
    for (int ca = v.ax0; ca < 2; ++ca) ok = ok && ((uintptr_t)out[ca] & 15u) == 0;
    ok = ok && ((uintptr_t)out[2] & (sizeof(T) - 1)) == 0;
    if (!ok) return false;
    int comps = 0, nmax[3] = {1, 1, 1};
    for (int ca = v.ax0; ca < 3; ++ca) {
        if (accumulate && vector[ca] == 0.0) continue;   // adding 0 * s leaves the component as it is
        comps |= 1 << ca;
        for (int a = 0; a < 3; ++a) nmax[a] = v.cn[ca][a] > nmax[a] ? v.cn[ca][a] : nmax[a];
    }
    if (!comps) return true;
    int ltpr = 6;
    while (ltpr > 0 && (1 << (ltpr - 1)) * V >= v.n[2]) --ltpr;
    const int patches1 = ceil_div(nmax[1], kBlock >> ltpr), patches2 = ceil_div(v.n[2], (1 << ltpr) * V);
    const long long npatch = (long long)nmax[0] * patches1 * patches2;
    const int nblk = npatch < 16384 ? (int)npatch : 16384;
    Comp3<T> oc{{(T*)out[0], (T*)out[1], (T*)out[2]}};
    if (V == 4)
        hipLaunchKernelGGL((centered_to_staggered_vec_kernel<T, DIM, 4>), dim3(nblk, v.batch), dim3(kBlock), 0, s, g, sb, oc, (const T*)sfield, (T)vector[0],
                           (T)vector[1], (T)vector[2], accumulate, comps, nmax[0], patches1, patches2, ltpr);
    else
        hipLaunchKernelGGL((centered_to_staggered_vec_kernel<T, DIM, 2>), dim3(nblk, v.batch), dim3(kBlock), 0, s, g, sb, oc, (const T*)sfield, (T)vector[0],
                           (T)vector[1], (T)vector[2], accumulate, comps, nmax[0], patches1, patches2, ltpr);
    return true;
}

int run_centered_to_staggered(phihip_ctx* ctx, const GridView& v, const void* sfield, const int32_t s_bc[3][2], const double s_val[3][2],
                              const double vector[3], int accumulate, void* const out[3], hipStream_t s) {
    for (int ca = v.ax0; ca < 3; ++ca)
        if (v.ccells[ca] >= (1LL << 31) || v.cells >= (1LL << 31)) {
            set_error("centered_to_staggered: more than 2^31 samples per batch entry are not supported");
            return PHIHIP_ERR_UNSUPPORTED;
        }
    const VelGrid g = make_velgrid(v);
    const ScalarBc sb = make_scalar_bc(v, s_bc, s_val);
    LaunchScope ls(ctx, PHIHIP_K_OTHER, s);
    {
        bool done;
        if (v.dtype == PHIHIP_F64) done = v.rank == 3 ? launch_c2s_vec<double, 3>(v, g, sb, sfield, vector, accumulate, out, s) : launch_c2s_vec<double, 2>(v, g, sb, sfield, vector, accumulate, out, s);
        else done = v.rank == 3 ? launch_c2s_vec<float, 3>(v, g, sb, sfield, vector, accumulate, out, s) : launch_c2s_vec<float, 2>(v, g, sb, sfield, vector, accumulate, out, s);
        if (done) {
            PHIHIP_CHECK_HIP(hipGetLastError());
            return PHIHIP_OK;
        }
    }
    for (int ca = v.ax0; ca < 3; ++ca) {      // rows that are not whole vectors / unaligned buffers: one scalar launch per component
        if (accumulate && vector[ca] == 0.0) continue;   // adding 0 * s leaves the component as it is
        const int nblk = ceil_div(v.ccells[ca], kBlock) < 16384 ? ceil_div(v.ccells[ca], kBlock) : 16384;
        if (v.dtype == PHIHIP_F64)
            hipLaunchKernelGGL(centered_to_staggered_kernel<double>, dim3(nblk, v.batch), dim3(kBlock), 0, s, g, sb, ca, (const double*)sfield,
                               (double*)out[ca], vector[ca], accumulate);
        else
            hipLaunchKernelGGL(centered_to_staggered_kernel<float>, dim3(nblk, v.batch), dim3(kBlock), 0, s, g, sb, ca, (const float*)sfield,
                               (float*)out[ca], (float)vector[ca], accumulate);
    }
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// obstacles rasterised on the device: hard cell mask and apply_boundary_conditions with moving obstacles.
// The reference evaluates these on NumPy every step (`with NUMPY:` fluid.py:132); here the obstacle list travels as kernel
// arguments, nothing is rasterised on the host. Positions and distances are computed in double.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kObstaclesPerLaunch = 16;

struct ObstacleSet {
    int count;
    int kind[kObstaclesPerLaunch];
    double center[kObstaclesPerLaunch][3];   // internal axis order
    double half[kObstaclesPerLaunch][3];
    double vel[kObstaclesPerLaunch][3];
    double ang[kObstaclesPerLaunch][3];
    double rot[kObstaclesPerLaunch][3][3];   // box frame -> world, internal axis order
    int moving[kObstaclesPerLaunch];
    int rotated[kObstaclesPerLaunch];
    int group[kObstaclesPerLaunch];          // > 0: consecutive entries with the same value are ONE obstacle (union of the members)
    int skip[kObstaclesPerLaunch];           // bit a (internal axis): the geometry is infinite along a (embed): coordinate ignored
};

static ObstacleSet make_obstacle_set(const GridView& v, const phihip_obstacle* obs, int first, int count) {
    ObstacleSet s;
    memset(&s, 0, sizeof(s));
    s.count = count;
    for (int k = 0; k < count; ++k) {
        const phihip_obstacle& o = obs[first + k];
        s.kind[k] = o.kind;
        s.group[k] = o.group;
        for (int d = 0; d < v.rank; ++d)
            if (o.embed_mask & (1 << d)) s.skip[k] |= 1 << (d + v.ax0);
        bool moving = false;
        for (int d = 0; d < v.rank; ++d) {
            s.center[k][d + v.ax0] = o.center[d];
            s.half[k][d + v.ax0] = o.kind == PHIHIP_OBSTACLE_SPHERE ? o.half_size[0] : o.half_size[d];
            s.vel[k][d + v.ax0] = o.velocity[d];
            moving = moving || o.velocity[d] != 0.0;
        }
        if (v.rank == 2) {
            s.ang[k][0] = o.angular_velocity[0];   // rotation about the missing axis a0
            moving = moving || o.angular_velocity[0] != 0.0;
        } else {
            for (int d = 0; d < 3; ++d) {
                s.ang[k][d] = o.angular_velocity[d];
                moving = moving || o.angular_velocity[d] != 0.0;
            }
        }
        s.moving[k] = moving ? 1 : 0;
        bool any = false;
        for (int i = 0; i < 9; ++i) any = any || o.rotation[i] != 0.0;
        bool identity = true;
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) s.rot[k][a][c] = a == c ? 1.0 : 0.0;
        if (any)
            for (int a = 0; a < v.rank; ++a)
                for (int c = 0; c < v.rank; ++c) {
                    s.rot[k][a + v.ax0][c + v.ax0] = o.rotation[a * 3 + c];
                    identity = identity && o.rotation[a * 3 + c] == (a == c ? 1.0 : 0.0);
                }
        s.rotated[k] = (any && !identity && o.kind == PHIHIP_OBSTACLE_BOX) ? 1 : 0;
    }
    return s;
}

// box-frame coordinates of x - center: R^T (x - c)  (Box.global_to_local(scale=False, origin='center'), phi/geom/_box.py:134-152)
__device__ __forceinline__ void obstacle_local(const ObstacleSet& s, int k, const double (&x)[3], int ax0, double (&loc)[3]) {
    double r[3] = {0, 0, 0};
    for (int a = ax0; a < 3; ++a) r[a] = (s.skip[k] >> a) & 1 ? 0.0 : x[a] - s.center[k][a];   // embedded axes: always "at the centre"
    if (!s.rotated[k]) {
        for (int a = 0; a < 3; ++a) loc[a] = r[a];
        return;
    }
    for (int a = 0; a < 3; ++a) {
        loc[a] = 0;
        for (int c = ax0; c < 3; ++c) loc[a] += s.rot[k][c][a] * r[c];
    }
}

// lies_inside: box |x - c| <= half (inclusive, phi/geom/_box.py:174-185); sphere |x - c|^2 <= r^2
__device__ __forceinline__ bool obstacle_inside(const ObstacleSet& s, int k, const double (&x)[3], int ax0) {
    if (s.kind[k] == PHIHIP_OBSTACLE_SPHERE) {
        double d2 = 0;
        for (int a = ax0; a < 3; ++a)
            if (!((s.skip[k] >> a) & 1)) d2 += (x[a] - s.center[k][a]) * (x[a] - s.center[k][a]);
        return d2 <= s.half[k][ax0] * s.half[k][ax0];
    }
    double loc[3];
    obstacle_local(s, k, x, ax0, loc);
    bool in = true;
    for (int a = ax0; a < 3; ++a) in = in && (((s.skip[k] >> a) & 1) || fabs(loc[a]) <= s.half[k][a]);
    return in;
}

// approximate_signed_distance: box = L-infinity distance to the surface (phi/geom/_box.py:217-236), sphere = |x - c| - r
__device__ __forceinline__ double obstacle_sdf(const ObstacleSet& s, int k, const double (&x)[3], int ax0) {
    if (s.kind[k] == PHIHIP_OBSTACLE_SPHERE) {
        double d2 = 0;
        for (int a = ax0; a < 3; ++a)
            if (!((s.skip[k] >> a) & 1)) d2 += (x[a] - s.center[k][a]) * (x[a] - s.center[k][a]);
        return sqrt(d2) - s.half[k][ax0];
    }
    double loc[3];
    obstacle_local(s, k, x, ax0, loc);
    double dist = -1e300;
    for (int a = ax0; a < 3; ++a) {
        if ((s.skip[k] >> a) & 1) continue;
        const double da = fabs(loc[a]) - s.half[k][a];
        dist = da > dist ? da : dist;
    }
    return dist;
}

// Bounding boxes of the obstacles of one launch in LDS (half extents, the bounding radius for a rotated box, unbounded along embedded
// axes), built once per workgroup: the per-patch "is this obstacle anywhere near" test then reads LDS instead of walking the by-value
// obstacle table in the kernel arguments (dependent scalar loads, ~0.2 us each, per obstacle and patch made these kernels latency-bound).
__device__ __forceinline__ void obstacle_boxes(const ObstacleSet& s, int ax0, double (*box)[6]) {
    const int k = threadIdx.x;
    if (k < s.count) {
        double rad = 0;
        if (s.rotated[k]) {
            for (int a = ax0; a < 3; ++a) rad += s.half[k][a] * s.half[k][a];
            rad = sqrt(rad);
        }
        for (int a = 0; a < 3; ++a) {
            const bool skip = a < ax0 || ((s.skip[k] >> a) & 1);
            const double h = s.rotated[k] ? rad : (s.kind[k] == PHIHIP_OBSTACLE_SPHERE ? s.half[k][ax0] : s.half[k][a]);
            box[k][2 * a] = skip ? -1e300 : s.center[k][a] - h;
            box[k][2 * a + 1] = skip ? 1e300 : s.center[k][a] + h;
        }
    }
    __syncthreads();
}

// bit k set: obstacle k can reach the axis-aligned box [lo, hi] grown by `margin` (uniform per patch: whole workgroups skip the others)
__device__ __forceinline__ unsigned obstacles_near(const double (*box)[6], int count, const double (&lo)[3], const double (&hi)[3], double margin) {
    unsigned near = 0;
    for (int k = 0; k < count; ++k) {
        bool hit = true;
#pragma unroll
        for (int a = 0; a < 3; ++a) hit = hit && !(lo[a] - margin > box[k][2 * a + 1] || hi[a] + margin < box[k][2 * a]);
        near |= hit ? (1u << k) : 0u;
    }
    return near;
}

// r5: the patches a launch visits are those of the obstacles' INDEX-SPACE bounding box (all obstacles of the launch together), per lattice
// (slot ca = the faces of component ca; the cell lattice uses slot 2). Until r4 every launch walked every patch of the grid and rejected
// them one by one -- 65 536 patches per component at 256^3 for two small obstacles: apply_boundary_conditions cost 3 x 0.13 ms of which 6 %
// of the samples were touched; the cell mask 0.11 ms. A moving obstacle pays this every step (Moving_Obstacles / Rotating_Bar notebooks).
struct PatchBox {
    int i0[3], n0[3];        // first plane, planes
    int p1[3], np1[3];       // first patch row, patch rows
    int p2[3], np2[3];       // first patch column, patch columns
    int first[4];            // running patch count: slot l owns [first[l], first[l + 1])
};

// slot `l`: samples of component `ca` (ca >= 0) or cell centres (ca < 0) within `margin` of any obstacle of the set; conservative by one
// sample per side (the kernels repeat the exact per-patch test)
static void patch_box_slot(const GridView& v, const ObstacleSet& set, int ca, double margin, int l, PatchBox* pb) {
    int lo_i[3] = {0, 0, 0}, hi_i[3] = {0, 0, 0};
    bool empty = set.count == 0;
    for (int a = 0; a < 3 && !empty; ++a) {
        const int n = ca >= 0 ? v.cn[ca][a] : v.n[a];
        if (a < v.ax0) { lo_i[a] = 0; hi_i[a] = 0; continue; }
        double lo = 1e300, hi = -1e300;
        for (int k = 0; k < set.count; ++k) {
            if ((set.skip[k] >> a) & 1) { lo = -1e300; hi = 1e300; break; }
            double h = set.kind[k] == PHIHIP_OBSTACLE_SPHERE ? set.half[k][v.ax0] : set.half[k][a];
            if (set.rotated[k]) {
                double r2 = 0;
                for (int c = v.ax0; c < 3; ++c) r2 += set.half[k][c] * set.half[k][c];
                h = sqrt(r2);
            }
            lo = fmin(lo, set.center[k][a] - h);
            hi = fmax(hi, set.center[k][a] + h);
        }
        const double shift = (ca >= 0 && a == ca) ? (double)v.off[a] : 0.5;     // position of sample i: lower + (i + shift) dx
        const double flo = (lo - margin - v.lower[a]) / v.dx[a] - shift, fhi = (hi + margin - v.lower[a]) / v.dx[a] - shift;
        long long il = flo < -1e9 ? 0 : (flo > 1e9 ? (long long)n : (long long)floor(flo) - 1);
        long long ih = fhi > 1e9 ? (long long)n - 1 : (fhi < -1e9 ? -1 : (long long)ceil(fhi) + 1);
        il = il < 0 ? 0 : il;
        ih = ih > n - 1 ? n - 1 : ih;
        if (ih < il) { empty = true; break; }
        lo_i[a] = (int)il; hi_i[a] = (int)ih;
    }
    if (empty) {
        pb->i0[l] = pb->p1[l] = pb->p2[l] = 0;
        pb->n0[l] = pb->np1[l] = pb->np2[l] = 0;
    } else {
        pb->i0[l] = lo_i[0]; pb->n0[l] = hi_i[0] - lo_i[0] + 1;
        pb->p1[l] = lo_i[1] / kPatchRows; pb->np1[l] = hi_i[1] / kPatchRows - pb->p1[l] + 1;
        pb->p2[l] = lo_i[2] / kPatchCols; pb->np2[l] = hi_i[2] / kPatchCols - pb->p2[l] + 1;
    }
}

// patch number -> (slot, plane, first row, first column); uniform
__device__ __forceinline__ void decode_box_patch(const PatchBox& pb, int patch, int& l, int& i0, int& r0, int& c0) {
    l = patch >= pb.first[2] ? 2 : (patch >= pb.first[1] ? 1 : 0);
    const int q = patch - pb.first[l];
    const int np2 = pb.np2[l], np1 = pb.np1[l];
    const int t = q / np2;
    c0 = (pb.p2[l] + (q - t * np2)) * kPatchCols;
    const int i = t / np1;
    r0 = (pb.p1[l] + (t - i * np1)) * kPatchRows;
    i0 = pb.i0[l] + i;
}

// r3: (4 x 64)-cell patches decoded without integer division (the 64-bit div / mod per cell and the fp64 geometry of EVERY obstacle for
// EVERY cell made the obstacle kernels run at 1-8 % of the HBM rate); a patch evaluates only the obstacles whose bounding box it touches.
__global__ __launch_bounds__(kBlock) void obstacle_accessible_kernel(VelGrid g, double lower0, double lower1, double lower2, ObstacleSet s,
                                                                     PatchBox pb, uint8_t* accessible) {
    const double lower[3] = {lower0, lower1, lower2};
    __shared__ double box[kObstaclesPerLaunch][6];
    obstacle_boxes(s, g.ax0, box);
    const int tx = threadIdx.x & (kPatchCols - 1), ty = threadIdx.x / kPatchCols;
    const int npatch = pb.first[3];
    for (int patch = blockIdx.x; patch < npatch; patch += gridDim.x) {
        int l, i0, r0, c0;
        decode_box_patch(pb, patch, l, i0, r0, c0);
        const int first[3] = {i0, r0, c0};
        const int last[3] = {i0, min(r0 + kPatchRows, g.n[1]) - 1, min(c0 + kPatchCols, g.n[2]) - 1};
        double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
        for (int a = g.ax0; a < 3; ++a) { lo[a] = lower[a] + (first[a] + 0.5) * g.dx[a]; hi[a] = lower[a] + (last[a] + 0.5) * g.dx[a]; }
        const unsigned near = obstacles_near(box, s.count, lo, hi, 1e-9 * (hi[2] - lo[2] + g.dx[2]));
        const int idx[3] = {i0, r0 + ty, c0 + tx};
        if (!near || idx[1] >= g.n[1] || idx[2] >= g.n[2]) continue;            // nothing to change in this patch (the array starts as all ones)
        const long long cell = ((long long)i0 * g.n[1] + idx[1]) * g.n[2] + idx[2];
        bool inside = false;
        double x[3] = {0, 0, 0};
        for (int a = g.ax0; a < 3; ++a) x[a] = lower[a] + (idx[a] + 0.5) * g.dx[a];
        for (int k = 0; k < s.count; ++k)
            if ((near >> k) & 1u) inside = inside || obstacle_inside(s, k, x, g.ax0);
        if (inside) accessible[cell] = (uint8_t)0;
    }
}

int run_obstacle_accessible(phihip_ctx* ctx, const GridView& v, const phihip_obstacle* obs, int count, uint8_t* accessible, hipStream_t s) {
    const VelGrid g = make_velgrid(v);
    PHIHIP_REQUIRE((long long)v.n[0] * ceil_div(v.n[1], kPatchRows) * ceil_div(v.n[2], kPatchCols) < (1LL << 31), "obstacle_accessible: grid too large");
    LaunchScope ls(ctx, PHIHIP_K_OTHER, s);
    // accessible = ~union(geometries) (fluid.py:130-133): everything is accessible, then every launch clears the cells inside its obstacles --
    // and visits the patches of their bounding box only (r5)
    PHIHIP_CHECK_HIP(hipMemsetAsync(accessible, 1, (size_t)v.cells, s));
    for (int first = 0; first < count;) {
        const int n = count - first < kObstaclesPerLaunch ? count - first : kObstaclesPerLaunch;
        const ObstacleSet set = make_obstacle_set(v, obs, first, n);
        PatchBox pb;
        memset(&pb, 0, sizeof(pb));
        patch_box_slot(v, set, -1, 1e-6 * (v.dx[2] + v.dx[1]), 2, &pb);
        const long long np = (long long)pb.n0[2] * pb.np1[2] * pb.np2[2];
        pb.first[0] = pb.first[1] = pb.first[2] = 0;
        pb.first[3] = (int)np;
        if (np > 0) {
            const int nblk = np < 4096 ? (int)np : 4096;
            hipLaunchKernelGGL(obstacle_accessible_kernel, dim3(nblk), dim3(kBlock), 0, s, g, v.lower[0], v.lower[1], v.lower[2], set, pb, accessible);
        }
        first += n;
    }
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void apply_obstacles_kernel(VelGrid g, double lower0, double lower1, double lower2, ObstacleSet s, PatchBox pb,
                                                                 Comp3<T> vel) {
    const double lower[3] = {lower0, lower1, lower2};
    const int b = blockIdx.y;
    double r2 = 0;   // bounding radius of a face cell: |half size of a grid cell|
    for (int a = g.ax0; a < 3; ++a) r2 += 0.25 * g.dx[a] * g.dx[a];
    const double radius = sqrt(r2);
    __shared__ double box[kObstaclesPerLaunch][6];
    obstacle_boxes(s, g.ax0, box);
    const int tx = threadIdx.x & (kPatchCols - 1), ty = threadIdx.x / kPatchCols;
    const int npatch = pb.first[3];
    for (int patch = blockIdx.x; patch < npatch; patch += gridDim.x) {
        int ca, i0, r0, cc0;
        decode_box_patch(pb, patch, ca, i0, r0, cc0);          // slot = component (uniform)
        const int c1 = g.cn[ca][1], c2 = g.cn[ca][2];
        T* __restrict__ V = (ca == 0 ? vel.p[0] : (ca == 1 ? vel.p[1] : vel.p[2])) + (long long)b * g.ccells[ca];
        // position of sample i along axis a: faces of the component's own axis, cell centres otherwise
        auto pos = [&](int a, int i) -> double { return a == ca ? lower[a] + (double)(i + g.off[a]) * g.dx[a] : lower[a] + (i + 0.5) * g.dx[a]; };
        const int first[3] = {i0, r0, cc0};
        const int last[3] = {i0, min(r0 + kPatchRows, c1) - 1, min(cc0 + kPatchCols, c2) - 1};
        double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
        for (int a = g.ax0; a < 3; ++a) { lo[a] = pos(a, first[a]); hi[a] = pos(a, last[a]); }
        // the soft mask m = clip(1 - sdf / radius, 0, 1) vanishes farther than `radius` from the surface: such patches are left untouched
        const unsigned near = obstacles_near(box, s.count, lo, hi, radius * 1.000001);
        if (!near) continue;
        const int idx[3] = {i0, r0 + ty, cc0 + tx};
        if (idx[1] >= c1 || idx[2] >= c2) continue;
        const long long f = ((long long)i0 * c1 + idx[1]) * c2 + idx[2];
        double x[3] = {0, 0, 0};
        for (int a = g.ax0; a < 3; ++a) x[a] = pos(a, idx[a]);
        T val = V[f];
        double m_union = 0.0;
        for (int k = 0; k < s.count; ++k) {
            double m = 0.0;
            if ((near >> k) & 1u) {
                m = 1.0 - obstacle_sdf(s, k, x, g.ax0) / radius;
                m = m < 0.0 ? 0.0 : (m > 1.0 ? 1.0 : m);
            }
            if (s.group[k] != 0) {   // union: sdf = min over the members <=> mask = max; applied once, after the last member
                const bool cont = k > 0 && s.group[k - 1] == s.group[k];
                m_union = cont && m_union > m ? m_union : m;
                if (k + 1 < s.count && s.group[k + 1] == s.group[k]) continue;
                m = m_union;
            }
            const T mt = (T)m, keep = T(1) - mt;
            val = keep == T(0) ? T(0) : keep * val;   // safe_mul(1 - mask, velocity)
            if (s.moving[k]) {
                double r[3] = {0, 0, 0};
                for (int a = g.ax0; a < 3; ++a) r[a] = x[a] - s.center[k][a];
                double u;
                if (g.ax0 == 1)   // rank 2: cross(w, r) = (-w r_y, w r_x)
                    u = ca == 1 ? -s.ang[k][0] * r[2] : s.ang[k][0] * r[1];
                else
                    u = ca == 0 ? s.ang[k][1] * r[2] - s.ang[k][2] * r[1]
                                : (ca == 1 ? s.ang[k][2] * r[0] - s.ang[k][0] * r[2] : s.ang[k][0] * r[1] - s.ang[k][1] * r[0]);
                u += s.vel[k][ca];
                if (mt != T(0)) val += mt * (T)u;     // safe_mul(mask, angular_velocity + obstacle.velocity)
            }
        }
        V[f] = val;
    }
}

int run_apply_obstacles(phihip_ctx* ctx, const GridView& v, const phihip_obstacle* obs, int count, void* const vel[3], hipStream_t s) {
    const VelGrid g = make_velgrid(v);
    LaunchScope ls(ctx, PHIHIP_K_OTHER, s);
    double r2 = 0;
    for (int a = v.ax0; a < 3; ++a) r2 += 0.25 * v.dx[a] * v.dx[a];
    const double radius = sqrt(r2);
    for (int first = 0, n = 0; first < count; first += n) {
        n = count - first < kObstaclesPerLaunch ? count - first : kObstaclesPerLaunch;
        // a union is applied by ONE launch: do not cut inside a group
        while (n > 0 && first + n < count && obs[first + n].group != 0 && obs[first + n].group == obs[first + n - 1].group) --n;
        if (n == 0) {
            set_error("apply_obstacles: a union (group %d) has more than %d members", obs[first].group, kObstaclesPerLaunch);
            return PHIHIP_ERR_UNSUPPORTED;
        }
        const ObstacleSet set = make_obstacle_set(v, obs, first, n);
        // r5: ONE launch for all components (fluid.py:212-240 is one call), over the patches of the obstacles' bounding box only
        PatchBox pb;
        memset(&pb, 0, sizeof(pb));
        long long total = 0;
        for (int ca = 0; ca < 3; ++ca) {
            pb.first[ca] = (int)total;
            if (ca >= v.ax0) {
                patch_box_slot(v, set, ca, radius * 1.00001, ca, &pb);
                total += (long long)pb.n0[ca] * pb.np1[ca] * pb.np2[ca];
            }
            PHIHIP_REQUIRE(total < (1LL << 31), "apply_obstacles: grid too large");
        }
        pb.first[3] = (int)total;
        if (total == 0) continue;
        const int nblk = total < 4096 ? (int)total : 4096;
        if (v.dtype == PHIHIP_F64) {
            Comp3<double> c{{(double*)vel[0], (double*)vel[1], (double*)vel[2]}};
            hipLaunchKernelGGL(apply_obstacles_kernel<double>, dim3(nblk, v.batch), dim3(kBlock), 0, s, g, v.lower[0], v.lower[1], v.lower[2], set, pb, c);
        } else {
            Comp3<float> c{{(float*)vel[0], (float*)vel[1], (float*)vel[2]}};
            hipLaunchKernelGGL(apply_obstacles_kernel<float>, dim3(nblk, v.batch), dim3(kBlock), 0, s, g, v.lower[0], v.lower[1], v.lower[2], set, pb, c);
        }
    }
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// obstacle flags (fluid.py:130-137: accessible, hard_bcs = stagger(accessible, minimum), active)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void cellflags_kernel(VelGrid g, const uint8_t* accessible, const uint8_t* active, int per_batch,
                                                           uint8_t* flags, int patches1, int patches2) {
    const int b = blockIdx.y;
    const long long mb = per_batch ? (long long)b * g.cells : 0;
    const uint8_t* __restrict__ A = accessible ? accessible + mb : nullptr;
    const int tx = threadIdx.x & (kPatchCols - 1), ty = threadIdx.x / kPatchCols;
    const int npatch = g.n[0] * patches1 * patches2;
    const int stride[3] = {g.n[1] * g.n[2], g.n[2], 1};
    for (int patch = blockIdx.x; patch < npatch; patch += gridDim.x) {
        int idx[3], r0, c0;
        decode_patch(patch, patches1, patches2, idx[0], r0, c0);
        idx[1] = r0 + ty;
        idx[2] = c0 + tx;
        if (idx[1] >= g.n[1] || idx[2] >= g.n[2]) continue;
        const int cell = (idx[0] * g.n[1] + idx[1]) * g.n[2] + idx[2];
        const unsigned self = A ? (A[cell] ? 1u : 0u) : 1u;
        unsigned f = 0;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            if (ax < g.ax0) continue;
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const int j = idx[ax] + (side ? 1 : -1);
                unsigned other;
                if (j < 0 || j >= g.n[ax]) {
                    const int code = g.bc[ax][side];
                    if (code == PHIHIP_BC_PERIODIC) other = A ? (A[cell + (j < 0 ? g.n[ax] - 1 : 1 - g.n[ax]) * stride[ax]] ? 1u : 0u) : 1u;
                    else other = code == PHIHIP_BC_OPEN ? 1u : 0u;   // _accessible_extrapolation: BOUNDARY -> ONE, constant -> ZERO
                } else {
                    other = A ? (A[cell + (side ? stride[ax] : -stride[ax])] ? 1u : 0u) : 1u;
                }
                if (self & other) f |= 1u << (2 * ax + side);
            }
        }
        const unsigned act = self & (active ? (active[mb + cell] ? 1u : 0u) : 1u);
        if (act) f |= 64u;
        flags[mb + cell] = (uint8_t)f;
    }
}

// r5: W dwords = 4 W consecutive cells of a row per thread. The r4 kernel read seven single bytes per cell and wrote one: 64-byte requests per
// wavefront, the a1 / a0 neighbour rows fetched by other wavefronts again -- PMC 3.0x the bytes, 120 us at 256^3 (0.03 of the HBM rate). Here a thread
// loads the five byte vectors (self, a0 -+, a1 -+) as W dwords each + the two bytes beyond the row ends of its vector, forms the six face bits
// and the active bit of its cells with byte-parallel integer operations and stores W dwords. Bytes are normalised to 0 / 1 first (a user mask
// may hold any non-zero value).
__device__ __forceinline__ unsigned nz_bytes(unsigned v) {      // 0x01 in every byte of v that is non-zero
    return ((((v & 0x7f7f7f7fu) + 0x7f7f7f7fu) | v) & 0x80808080u) >> 7;
}

template <int W>
struct ByteVec {
    unsigned w[W];
};
struct alignas(16) U4 {
    unsigned x, y, z, w;
};
template <int W>
__device__ __forceinline__ ByteVec<W> load_bytes(const uint8_t* p) {
    ByteVec<W> r;
    if (W == 4) {
        const U4 q = *reinterpret_cast<const U4*>(p);
        r.w[0] = q.x; r.w[1 % W] = q.y; r.w[2 % W] = q.z; r.w[3 % W] = q.w;
    } else {
#pragma unroll
        for (int i = 0; i < W; ++i) r.w[i] = reinterpret_cast<const unsigned*>(p)[i];
    }
    return r;
}

template <int W>
__global__ __launch_bounds__(kBlock) void cellflags_vec_kernel(VelGrid g, const uint8_t* accessible, const uint8_t* active, int per_batch,
                                                               uint8_t* flags, int patches1, int patches2, int ltpr) {
    constexpr int V = 4 * W;
    const int b = blockIdx.y;
    const long long mb = per_batch ? (long long)b * g.cells : 0;
    const uint8_t* __restrict__ A = accessible ? accessible + mb : nullptr;
    // a patch = (256 >> ltpr) rows x (1 << ltpr) threads of V cells: a 256-cell row is 16 threads of 16 cells -- with the fixed 64-thread rows of
    // the other patch kernels three quarters of the lanes had no cells
    const int tx = threadIdx.x & ((1 << ltpr) - 1), ty = threadIdx.x >> ltpr;
    const int prow = kBlock >> ltpr;
    const int npatch = g.n[0] * patches1 * patches2;
    const int n0 = g.n[0], n1 = g.n[1], n2 = g.n[2];
    const long long stride0 = (long long)n1 * n2;
    // outside value per (axis, side) for non-periodic sides: _accessible_extrapolation -- BOUNDARY (open) -> ONE, constant (closed) -> ZERO
    auto outside = [&](int ax, int side) -> unsigned { return g.bc[ax][side] == PHIHIP_BC_OPEN ? 0x01010101u : 0u; };
    for (int patch = blockIdx.x; patch < npatch; patch += gridDim.x) {
        const int t = patch / patches2;
        const int i0 = t / patches1;
        const int i1 = (t - i0 * patches1) * prow + ty, i2 = (((patch - t * patches2) << ltpr) + tx) * V;
        if (i1 >= n1 || i2 >= n2) continue;
        const long long cell = (long long)i0 * stride0 + (long long)i1 * n2 + i2;
        ByteVec<W> self, nb[3][2];
        unsigned endl = 0, endr = 0;          // the bytes left of the first / right of the last cell of this vector
        if (A) {
            self = load_bytes<W>(A + cell);
#pragma unroll
            for (int i = 0; i < W; ++i) self.w[i] = nz_bytes(self.w[i]);
            // a0 and a1 neighbours: whole vectors of the neighbouring plane / row
#pragma unroll
            for (int ax = 0; ax < 2; ++ax) {
                if (ax < g.ax0) continue;
                const int idx = ax == 0 ? i0 : i1, n = ax == 0 ? n0 : n1;
                const long long st = ax == 0 ? stride0 : (long long)n2;
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    const int j = idx + (side ? 1 : -1);
                    if (j < 0 || j >= n) {
                        if (g.bc[ax][side] == PHIHIP_BC_PERIODIC) {
                            nb[ax][side] = load_bytes<W>(A + cell + (j < 0 ? (long long)(n - 1) : (long long)(1 - n)) * st);
#pragma unroll
                            for (int i = 0; i < W; ++i) nb[ax][side].w[i] = nz_bytes(nb[ax][side].w[i]);
                        } else {
#pragma unroll
                            for (int i = 0; i < W; ++i) nb[ax][side].w[i] = outside(ax, side);
                        }
                    } else {
                        nb[ax][side] = load_bytes<W>(A + cell + (side ? st : -st));
#pragma unroll
                        for (int i = 0; i < W; ++i) nb[ax][side].w[i] = nz_bytes(nb[ax][side].w[i]);
                    }
                }
            }
            const long long row = cell - i2;
            if (i2 > 0) endl = A[cell - 1] ? 1u : 0u;
            else endl = g.bc[2][0] == PHIHIP_BC_PERIODIC ? (A[row + n2 - 1] ? 1u : 0u) : (outside(2, 0) & 1u);
            if (i2 + V < n2) endr = A[cell + V] ? 1u : 0u;
            else endr = g.bc[2][1] == PHIHIP_BC_PERIODIC ? (A[row] ? 1u : 0u) : (outside(2, 1) & 1u);
        } else {
            // no obstacle mask: every cell accessible; only the domain boundary shapes the bits
#pragma unroll
            for (int i = 0; i < W; ++i) self.w[i] = 0x01010101u;
#pragma unroll
            for (int ax = 0; ax < 2; ++ax) {
                const int idx = ax == 0 ? i0 : i1, n = ax == 0 ? n0 : n1;
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    const int j = idx + (side ? 1 : -1);
                    const unsigned val = (j < 0 || j >= n) && g.bc[ax][side] != PHIHIP_BC_PERIODIC ? outside(ax, side) : 0x01010101u;
#pragma unroll
                    for (int i = 0; i < W; ++i) nb[ax][side].w[i] = val;
                }
            }
            endl = i2 > 0 || g.bc[2][0] == PHIHIP_BC_PERIODIC ? 1u : (outside(2, 0) & 1u);
            endr = i2 + V < n2 || g.bc[2][1] == PHIHIP_BC_PERIODIC ? 1u : (outside(2, 1) & 1u);
        }
        // a2 neighbours: the vector shifted by one byte either way
#pragma unroll
        for (int i = 0; i < W; ++i) {
            const unsigned prev = i > 0 ? self.w[i - 1] >> 24 : endl;
            const unsigned next = i + 1 < W ? self.w[i + 1] & 0xffu : endr;
            nb[2][0].w[i] = (self.w[i] << 8) | prev;
            nb[2][1].w[i] = (self.w[i] >> 8) | (next << 24);
        }
        ByteVec<W> act;
        if (active) {
            act = load_bytes<W>(active + mb + cell);
#pragma unroll
            for (int i = 0; i < W; ++i) act.w[i] = nz_bytes(act.w[i]) & self.w[i];
        } else {
            act = self;
        }
        unsigned out[W];
#pragma unroll
        for (int i = 0; i < W; ++i) {
            unsigned f = act.w[i] << 6;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                if (ax < g.ax0) continue;
                f |= (self.w[i] & nb[ax][0].w[i]) << (2 * ax);
                f |= (self.w[i] & nb[ax][1].w[i]) << (2 * ax + 1);
            }
            out[i] = f;
        }
        if (W == 4) {
            *reinterpret_cast<U4*>(flags + mb + cell) = U4{out[0], out[1 % W], out[2 % W], out[3 % W]};
        } else {
#pragma unroll
            for (int i = 0; i < W; ++i) reinterpret_cast<unsigned*>(flags + mb + cell)[i] = out[i];
        }
    }
}

int run_build_cellflags(phihip_ctx* ctx, const GridView& v, const uint8_t* accessible, const uint8_t* active, int mask_batch,
                        uint8_t* flags, hipStream_t s) {
    const VelGrid g = make_velgrid(v);
    PHIHIP_REQUIRE(v.cells < (1LL << 31), "build_cellflags: more than 2^31 cells per batch entry are not supported");
    LaunchScope ls(ctx, PHIHIP_K_OTHER, s);
    const dim3 gridy(1, mask_batch > 1 ? mask_batch : 1);
    auto aligned = [](const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
    // rows of whole 16-byte (4-byte) vectors at aligned addresses: the byte-parallel kernel; anything else the one-byte-per-thread kernel
    const int W = (v.n[2] % 16 == 0 && aligned(accessible, 16) && aligned(active, 16) && aligned(flags, 16)) ? 4
                : ((v.n[2] % 4 == 0 && aligned(accessible, 4) && aligned(active, 4) && aligned(flags, 4)) ? 1 : 0);
    if (W) {
        int ltpr = 0;
        while ((1 << ltpr) < 64 && (1 << ltpr) * 4 * W < v.n[2]) ++ltpr;          // threads per row: enough for the row, at most a wavefront
        const int patches1 = ceil_div(v.n[1], kBlock >> ltpr), patches2 = ceil_div(v.n[2], (4 * W) << ltpr);
        const long long npatch = (long long)v.n[0] * patches1 * patches2;
        const int nblk = npatch < 8192 ? (int)npatch : 8192;
        if (W == 4)
            hipLaunchKernelGGL(cellflags_vec_kernel<4>, dim3(nblk, gridy.y), dim3(kBlock), 0, s, g, accessible, active, mask_batch > 1 ? 1 : 0, flags, patches1, patches2, ltpr);
        else
            hipLaunchKernelGGL(cellflags_vec_kernel<1>, dim3(nblk, gridy.y), dim3(kBlock), 0, s, g, accessible, active, mask_batch > 1 ? 1 : 0, flags, patches1, patches2, ltpr);
    } else {
        const int patches1 = ceil_div(v.n[1], kPatchRows), patches2 = ceil_div(v.n[2], kPatchCols);
        const long long npatch = (long long)v.n[0] * patches1 * patches2;
        const int nblk = npatch < 16384 ? (int)npatch : 16384;
        hipLaunchKernelGGL(cellflags_kernel, dim3(nblk, gridy.y), dim3(kBlock), 0, s, g, accessible, active, mask_batch > 1 ? 1 : 0, flags, patches1, patches2);
    }
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// diffuse.explicit, order 2: v_d += k dt * laplace(v_d) with the velocity's own padding (phi/physics/diffuse.py:13-60)
// ---------------------------------------------------------------------------------------------------------------------
// A workgroup owns a (4 rows x 64 columns) column of samples of one component and marches over a chunk of a0 planes (like divergence_kernel):
// the a0 neighbours of a sample are the values the thread read one plane earlier / reads one plane ahead -- every plane is read from HBM
// once, where the one-thread-per-sample form of rounds 1-2 pulled each plane into three different L2s (FETCH_SIZE 1.8x the array) and spent
// a 64-bit div / mod per sample: 2-word kernel at 10 % of the HBM rate. The in-plane taps (row +-1, column +-1) and their boundary rule are
// resolved once per thread: periodic wrap, OPEN = the edge sample itself (zero gradient), CLOSED = the constant wall value.
// ADJ: the adjoint (I + k dt L)^T as a GATHER (no atomics): the stencil is symmetric in the interior; a clamped tap returns its weight to
// the edge sample itself, a constant tap contributes nothing, a wrapped tap is the wrapped neighbour.  gin += (...)^T gout.
template <typename T, bool ADJ>
__global__ __launch_bounds__(kBlock) void diffuse_kernel(VelGrid g, int ca, const T* __restrict__ vin, T* __restrict__ vout, T kdt, int tiles1, int tiles2, int chunk) {
    const int b = blockIdx.y;
    const int c0n = g.cn[ca][0], c1 = g.cn[ca][1], c2 = g.cn[ca][2];
    const long long bb = (long long)b * g.ccells[ca];
    const T* __restrict__ I = vin + bb;
    T* __restrict__ O = vout + bb;
    const int tx = threadIdx.x & (kPatchCols - 1), ty = threadIdx.x / kPatchCols;
    const int bid = xcd_order(blockIdx.x, gridDim.x);       // neighbouring columns share an XCD's L2 (their halo rows / columns)
    const int t2 = bid % tiles2;
    const int t1 = (bid / tiles2) % tiles1;
    const int ch = bid / (tiles2 * tiles1);
    const int i1 = t1 * kPatchRows + ty, i2 = t2 * kPatchCols + tx;
    if (i1 >= c1 || i2 >= c2) return;
    const int p0 = ch * chunk, p1 = min(p0 + chunk, c0n);
    if (p0 >= p1) return;
    T w[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) w[ax] = ax < g.ax0 ? T(0) : kdt / (T)(g.dx[ax] * g.dx[ax]);
    // in-plane taps: offset within the plane, or the rule that replaces the sample outside the array (1: the centre value / nothing for
    // the adjoint, 2: the wall constant / nothing); `self` counts the clamped taps of this sample (adjoint: weight returned to the centre)
    int off[4];
    int rule[4];
    T cval[4];
    T centre = T(1);
    const int o_c = i1 * c2 + i2;
    {
        const int idx[2] = {i1, i2}, n[2] = {c1, c2}, st[2] = {c2, 1};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const int k = 2 * a + side, ax = a + 1;
                const int j = idx[a] + (side ? 1 : -1);
                off[k] = o_c + (side ? st[a] : -st[a]);
                rule[k] = 0;
                cval[k] = T(0);
                if (ax >= g.ax0 && (j < 0 || j >= n[a])) {
                    const int code = g.bc[ax][side];
                    if (code == PHIHIP_BC_PERIODIC) off[k] = o_c + (side ? -(n[a] - 1) * st[a] : (n[a] - 1) * st[a]);
                    else if (code == PHIHIP_BC_OPEN) { rule[k] = 1; off[k] = o_c; if (ADJ) centre += w[ax]; }
                    else { rule[k] = 2; off[k] = o_c; cval[k] = (T)g.bcv[ax][side][ca]; }
                }
                if (ax < g.ax0) { rule[k] = 2; off[k] = o_c; }      // (unused axis of a 2-D grid: weight 0 anyway)
            }
    }
#pragma unroll
    for (int ax = 0; ax < 3; ++ax)
        if (ax >= g.ax0) centre -= T(2) * w[ax];
    const long long ps = (long long)c1 * c2;
    const bool has0 = g.ax0 == 0;
    // value of the a0 neighbour plane `p` of this thread's column under the boundary rule (uniform decision); `c_edge` = the edge sample
    auto plane_val = [&](int p, T c_edge, T& centre_adj) -> T {
        if (p >= 0 && p < c0n) return I[(long long)p * ps + o_c];
        const int side = p < 0 ? 0 : 1;
        const int code = g.bc[0][side];
        if (code == PHIHIP_BC_PERIODIC) return I[(long long)(p < 0 ? p + c0n : p - c0n) * ps + o_c];
        if (code == PHIHIP_BC_OPEN) { if (ADJ) centre_adj += w[0]; return ADJ ? T(0) : c_edge; }
        return ADJ ? T(0) : (T)g.bcv[0][side][ca];
    };
    T cur = I[(long long)p0 * ps + o_c];
    T dummy = T(0);
    T prev = has0 ? plane_val(p0 - 1, cur, dummy) : T(0);
    for (int p = p0; p < p1; ++p) {
        const long long po = (long long)p * ps;
        T cadj = T(0);
        T next = T(0);
        if (has0) {
            next = plane_val(p + 1, cur, cadj);
            if (ADJ && p == 0 && g.bc[0][0] == PHIHIP_BC_OPEN) cadj += w[0];      // (the lower clamped tap of plane 0: prev was formed before the loop)
        }
        T tap[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const T v = I[po + off[k]];
            tap[k] = rule[k] == 0 ? v : (ADJ ? T(0) : (rule[k] == 1 ? cur : cval[k]));
        }
        T acc;
        if (ADJ) acc = (centre + cadj) * cur + w[0] * (prev + next) + w[1] * (tap[0] + tap[1]) + w[2] * (tap[2] + tap[3]);
        else acc = cur + (w[0] * ((prev - cur) + (next - cur)) + w[1] * ((tap[0] - cur) + (tap[1] - cur)) + w[2] * ((tap[2] - cur) + (tap[3] - cur)));
        if (ADJ) O[po + o_c] += acc;
        else O[po + o_c] = acc;
        prev = cur;
        cur = next;
    }
}

template <typename T, bool ADJ>
static void launch_diffuse(const VelGrid& g, int ca, int batch, const void* in, void* out, double kdt, hipStream_t s) {
    const int tiles1 = ceil_div(g.cn[ca][1], kPatchRows), tiles2 = ceil_div(g.cn[ca][2], kPatchCols);
    const long long tiles = (long long)tiles1 * tiles2;
    int chunks = (int)((4096 + tiles - 1) / tiles);                     // ~4096 workgroups per batch entry when the grid allows
    chunks = chunks > g.cn[ca][0] ? g.cn[ca][0] : (chunks < 1 ? 1 : chunks);
    const int chunk = ceil_div(g.cn[ca][0], chunks);
    chunks = ceil_div(g.cn[ca][0], chunk);
    hipLaunchKernelGGL((diffuse_kernel<T, ADJ>), dim3((unsigned)(tiles * chunks), batch), dim3(kBlock), 0, s, g, ca, (const T*)in, (T*)out, (T)kdt, tiles1, tiles2, chunk);
}

// The lattice of one staggered component (or of a centred scalar described as component `ca` of g) as a grid of the marching kernels with the
// operator ident * I + scale * L and the field's own extrapolation as the neighbour rule: PERIODIC -> wrap, OPEN (zero-gradient) -> clamp,
// CLOSED (constant c) -> zero ghost + the affine part scale * c / dx_a^2 in the samples next to that side (`affine`: some c != 0).
static GridView lattice_view(const GridView& v, const VelGrid& g, int ca, double ident, double scale, bool* affine) {
    GridView w = v;
    *affine = false;
    for (int a = 0; a < 3; ++a) {
        w.n[a] = g.cn[ca][a];
        for (int side = 0; side < 2; ++side) {
            const int code = a < v.ax0 ? PHIHIP_BC_PERIODIC : g.bc[a][side];
            w.op_rule[a][side] = code == PHIHIP_BC_PERIODIC ? NB_WRAP : (code == PHIHIP_BC_OPEN ? NB_CLAMP : NB_ZERO);
            if (a >= v.ax0 && code == PHIHIP_BC_CLOSED && g.bcv[a][side][ca] != 0.0) *affine = true;
        }
    }
    w.cells = g.ccells[ca];
    w.halo[0] = w.halo[1] = 0;
    w.op_custom = 1;
    w.op_ident = ident;
    w.op_scale = scale;
    return w;
}

// out[sample next to a CLOSED side with constant c] += kdt * c / dx_a^2: the constant's share of the stencil (the marching kernels see a
// zero ghost there). One thread per sample of the boundary plane(s); only launched for sides with c != 0 (the lid of a cavity).
template <typename T>
__global__ __launch_bounds__(kBlock) void affine_walls_kernel(VelGrid g, int ca, int axis, int side, T* __restrict__ out, T add) {
    const int b = blockIdx.y;
    const int n[3] = {g.cn[ca][0], g.cn[ca][1], g.cn[ca][2]};
    const int u = axis == 0 ? 1 : 0, w = axis == 2 ? 1 : 2;       // the two other axes, w the faster one
    const long long total = (long long)n[u] * n[w];
    for (long long f = (long long)blockIdx.x * kBlock + threadIdx.x; f < total; f += (long long)gridDim.x * kBlock) {
        int idx[3];
        idx[axis] = side ? n[axis] - 1 : 0;
        idx[w] = (int)(f % n[w]);
        idx[u] = (int)(f / n[w]);
        out[(long long)b * g.ccells[ca] + ((long long)idx[0] * n[1] + idx[1]) * n[2] + idx[2]] += add;
    }
}

// the wall constants' share of diffuse.explicit on one lattice (only sides with c != 0 launch anything: the lid of a cavity)
static int diffuse_affine_walls(phihip_ctx* ctx, const GridView& v, const VelGrid& g, int ca, void* out, double kdt, hipStream_t s) {
    LaunchScope ls(ctx, PHIHIP_K_OTHER, s);
    for (int a = v.ax0; a < 3; ++a)
        for (int side = 0; side < 2; ++side) {
            if (g.bc[a][side] != PHIHIP_BC_CLOSED || g.bcv[a][side][ca] == 0.0) continue;
            const double add = kdt * g.bcv[a][side][ca] * g.rdx[a] * g.rdx[a];
            const long long total = g.ccells[ca] / g.cn[ca][a];
            const dim3 grid((unsigned)((total + kBlock - 1) / kBlock < 4096 ? (total + kBlock - 1) / kBlock : 4096), v.batch);
            if (v.dtype == PHIHIP_F64) hipLaunchKernelGGL(affine_walls_kernel<double>, grid, dim3(kBlock), 0, s, g, ca, a, side, (double*)out, add);
            else hipLaunchKernelGGL(affine_walls_kernel<float>, grid, dim3(kBlock), 0, s, g, ca, a, side, (float*)out, (float)add);
        }
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

// diffuse.explicit on one lattice = ONE pass of the marching kernels (MODE_APPLY with the operator I + k dt L: 2 words per sample, the tuned
// tiles of the pressure operator; r4 -- the one-dword-per-lane kernel it replaces ran at 0.32 of the HBM rate)
static int diffuse_explicit_lattice(phihip_ctx* ctx, const GridView& v, const VelGrid& g, int ca, const void* in, void* out, double kdt, hipStream_t s) {
    bool affine;
    const GridView w = lattice_view(v, g, ca, 1.0, kdt, &affine);
    if (w.cells >= (1LL << 31)) { set_error("diffuse: more than 2^31 samples per component and batch entry are not supported"); return PHIHIP_ERR_UNSUPPORTED; }
    PHIHIP_TRY(run_laplace_apply(ctx, w, nullptr, 1, in, out, s));
    if (affine) PHIHIP_TRY(diffuse_affine_walls(ctx, v, g, ca, out, kdt, s));
    return PHIHIP_OK;
}

// diffuse.explicit of a staggered field (phi/physics/diffuse.py:13-60 -- ONE call in the reference): r6 all components in ONE launch where their lattices share a
// tile configuration (cg.hip laplace_apply_multi_t: every periodic box; the components with whole rows of a closed box), then the wall constants
int run_diffuse(phihip_ctx* ctx, const GridView& v, const void* const vin[3], void* const vout[3], double kdt, hipStream_t s) {
    const VelGrid g = make_velgrid(v);
    GridView w[3];
    const void* in[3];
    void* out[3];
    bool affine[3] = {false, false, false};
    int count = 0;
    for (int ca = v.ax0; ca < 3; ++ca) {
        w[count] = lattice_view(v, g, ca, 1.0, kdt, &affine[ca]);
        if (w[count].cells >= (1LL << 31)) { set_error("diffuse: more than 2^31 samples per component and batch entry are not supported"); return PHIHIP_ERR_UNSUPPORTED; }
        in[count] = vin[ca];
        out[count] = vout[ca];
        ++count;
    }
    PHIHIP_TRY(run_laplace_apply_multi(ctx, w, count, in, out, s));
    for (int ca = v.ax0; ca < 3; ++ca)
        if (affine[ca]) PHIHIP_TRY(diffuse_affine_walls(ctx, v, g, ca, vout[ca], kdt, s));
    return PHIHIP_OK;
}

int run_diffuse_bwd(phihip_ctx* ctx, const GridView& v, const void* const gout[3], void* const gin[3], double kdt, hipStream_t s) {
    const VelGrid g = make_velgrid(v);
    LaunchScope ls(ctx, PHIHIP_K_OTHER, s);
    for (int ca = v.ax0; ca < 3; ++ca) {
        if (v.ccells[ca] >= (1LL << 31)) { set_error("diffuse: more than 2^31 samples per component and batch entry are not supported"); return PHIHIP_ERR_UNSUPPORTED; }
        if (v.dtype == PHIHIP_F64) launch_diffuse<double, true>(g, ca, v.batch, gout[ca], gin[ca], kdt, s);
        else launch_diffuse<float, true>(g, ca, v.batch, gout[ca], gin[ca], kdt, s);
    }
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

// a centred scalar with its own extrapolation viewed as "component 2" of a layout whose stored shape is the cell grid
static VelGrid scalar_as_component(const GridView& v, const ScalarBc& sb) {
    VelGrid g = make_velgrid(v);
    for (int a = 0; a < 3; ++a) {
        g.cn[2][a] = v.n[a];
        for (int side = 0; side < 2; ++side) {
            g.bc[a][side] = a < v.ax0 ? PHIHIP_BC_PERIODIC : sb.bc[a][side];
            g.bcv[a][side][2] = sb.val[a][side];
        }
    }
    g.ccells[2] = v.cells;
    g.off[2] = 0;
    return g;
}

// diffuse.explicit on a CenteredGrid (direction 1 = forward: out = s + k dt laplace(s); -1 = adjoint: gin += (...)^T gout)
int run_diffuse_centered(phihip_ctx* ctx, const GridView& v, const void* sfield, const int32_t s_bc[3][2], const double s_val[3][2], void* out,
                         double kdt, int adjoint, hipStream_t s) {
    const ScalarBc sb = make_scalar_bc(v, s_bc, s_val);
    const VelGrid g = scalar_as_component(v, sb);
    if (v.cells >= (1LL << 31)) { set_error("diffuse: more than 2^31 cells per batch entry are not supported"); return PHIHIP_ERR_UNSUPPORTED; }
    if (!adjoint) return diffuse_explicit_lattice(ctx, v, g, 2, sfield, out, kdt, s);
    LaunchScope ls(ctx, PHIHIP_K_OTHER, s);
    if (v.dtype == PHIHIP_F64) launch_diffuse<double, true>(g, 2, v.batch, sfield, out, kdt, s);
    else launch_diffuse<float, true>(g, 2, v.batch, sfield, out, kdt, s);
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

// ---- diffuse.implicit (phi/physics/diffuse.py:63-92): solve_linear(sharpen, y = field, x0 = field) with sharpen(x) = explicit(x, k, -dt) ---------
// The CG of the pressure path runs on the field's own lattice with the operator I - k dt L (GridView::op_*: ident = 1, scale = -k dt) and the
// field's own extrapolation as the neighbour rule: PERIODIC -> wrap, OPEN (zero-gradient) -> clamp, CLOSED (constant c) -> zero ghost, the
// constant's contribution being the affine part of `sharpen`, which solve_linear moves to the right-hand side (A x = y - sharpen(0)):
// rhs = field + k dt c / dx_a^2 in the samples next to such a side.
template <typename T>
__global__ __launch_bounds__(kBlock) void implicit_rhs_kernel(VelGrid g, int ca, const T* __restrict__ in, T* __restrict__ rhs, T kdt) {
    const int b = blockIdx.y;
    const long long total = g.ccells[ca];
    const int n1 = g.cn[ca][1], n2 = g.cn[ca][2];
    for (long long f = (long long)blockIdx.x * kBlock + threadIdx.x; f < total; f += (long long)gridDim.x * kBlock) {
        const int i2 = (int)(f % n2);
        const long long t = f / n2;
        const int idx[3] = {(int)(t / n1), (int)(t % n1), i2};
        T add = T(0);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (a < g.ax0) continue;
            const T w = kdt * (T)(g.rdx[a] * g.rdx[a]);
            if (idx[a] == 0 && g.bc[a][0] == PHIHIP_BC_CLOSED) add += w * (T)g.bcv[a][0][ca];
            if (idx[a] == g.cn[ca][a] - 1 && g.bc[a][1] == PHIHIP_BC_CLOSED) add += w * (T)g.bcv[a][1][ca];
        }
        rhs[(long long)b * total + f] = in[(long long)b * total + f] + add;
    }
}

// one lattice (a centred scalar or one staggered component, described as component `ca` of g): out = (I - k dt L)^-1 in, CG from x0 = in
static int diffuse_implicit_lattice(phihip_ctx* ctx, const GridView& v, const VelGrid& g, int ca, const void* in, void* out, double kdt,
                                    const phihip_solve* solve, phihip_solve_info* info, hipStream_t s) {
    bool affine;
    const GridView w = lattice_view(v, g, ca, 1.0, -kdt, &affine);
    if (w.cells >= (1LL << 31)) { set_error("diffuse_implicit: more than 2^31 samples per batch entry are not supported"); return PHIHIP_ERR_UNSUPPORTED; }
    const size_t esize = v.dtype == PHIHIP_F64 ? 8 : 4;
    const size_t bytes = (size_t)v.batch * w.cells * esize;
    const void* rhs = in;
    if (affine) {
        PHIHIP_TRY(ensure_buffer(ctx->ws_adj_q, bytes));
        LaunchScope ls(ctx, PHIHIP_K_OTHER, s);
        const long long nb = (w.cells + kBlock - 1) / kBlock;
        const dim3 grid((unsigned)(nb < 65536 ? nb : 65536), v.batch);
        if (v.dtype == PHIHIP_F64) hipLaunchKernelGGL(implicit_rhs_kernel<double>, grid, dim3(kBlock), 0, s, g, ca, (const double*)in, (double*)ctx->ws_adj_q.ptr, kdt);
        else hipLaunchKernelGGL(implicit_rhs_kernel<float>, grid, dim3(kBlock), 0, s, g, ca, (const float*)in, (float*)ctx->ws_adj_q.ptr, (float)kdt);
        PHIHIP_CHECK_HIP(hipGetLastError());
        rhs = ctx->ws_adj_q.ptr;
    }
    PHIHIP_CHECK_HIP(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, s));      // x0 = field (diffuse.py:90-91)
    return run_cg(ctx, w, nullptr, 1, rhs, out, solve, info, s);
}

int run_diffuse_implicit(phihip_ctx* ctx, const GridView& v, const void* const vin[3], void* const vout[3], double kdt, const phihip_solve* solve,
                         phihip_solve_info* info, hipStream_t s) {
    const VelGrid g = make_velgrid(v);
    for (int ca = v.ax0; ca < 3; ++ca)
        PHIHIP_TRY(diffuse_implicit_lattice(ctx, v, g, ca, vin[ca], vout[ca], kdt, solve, info ? info + (size_t)(ca - v.ax0) * v.batch : nullptr, s));
    return PHIHIP_OK;
}

int run_diffuse_implicit_centered(phihip_ctx* ctx, const GridView& v, const void* sfield, const int32_t s_bc[3][2], const double s_val[3][2], void* out,
                                  double kdt, const phihip_solve* solve, phihip_solve_info* info, hipStream_t s) {
    const ScalarBc sb = make_scalar_bc(v, s_bc, s_val);
    const VelGrid g = scalar_as_component(v, sb);
    return diffuse_implicit_lattice(ctx, v, g, 2, sfield, out, kdt, solve, info, s);
}

}  // namespace phihip
