// project.hip -- the non-iterative stencils of fluid.make_incompressible (/root/reference phi/physics/fluid.py:94-162):
// divergence of the staggered velocity (+ active mask, + mean balance), pressure-gradient subtraction, obstacle flags,
// soft obstacle scaling, and the explicit diffusion stencil (phi/physics/diffuse.py:13-60).
// All are single-pass HBM-bound kernels: one thread per output sample, fast axis on consecutive lanes.
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
template <typename T>
__global__ __launch_bounds__(kBlock) void divergence_kernel(VelGrid g, CComp3<T> v, const uint8_t* flags, int flags_per_batch,
                                                            T* div, double* part_sum, double* part_act, int nblk) {
    __shared__ double red[kBlock / kWave];
    const int b = blockIdx.y;
    const long long cell = (long long)blockIdx.x * kBlock + threadIdx.x;
    T val = T(0);
    T act = T(0);
    if (cell < g.cells) {
        const int i2 = (int)(cell % g.n[2]);
        const int i1 = (int)((cell / g.n[2]) % g.n[1]);
        const int i0 = (int)(cell / ((long long)g.n[2] * g.n[1]));
        const int idx[3] = {i0, i1, i2};
        T sum = T(0);
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            if (ax < g.ax0) continue;
            int lo[3] = {i0, i1, i2}, hi[3] = {i0, i1, i2};
            lo[ax] = idx[ax] - g.off[ax];
            hi[ax] = idx[ax] + 1 - g.off[ax];
            const long long bb = (long long)b * g.ccells[ax];
            const T vl = fetch_comp<T>(v.p[ax], g, ax, bb, lo[0], lo[1], lo[2]);
            const T vh = fetch_comp<T>(v.p[ax], g, ax, bb, hi[0], hi[1], hi[2]);
            sum += (vh - vl) / (T)g.dx[ax];
        }
        act = T(1);
        if (flags) {
            const unsigned f = flags[(flags_per_batch ? (long long)b * g.cells : 0) + cell];
            act = (f & 64u) ? T(1) : T(0);
            sum *= act;
        }
        div[(long long)b * g.cells + cell] = sum;
        val = sum;
    }
    const double s1 = block_sum((double)val, red);
    const double s2 = block_sum((double)act, red);
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
__global__ __launch_bounds__(kBlock) void balance_apply_kernel(T* div, const uint8_t* flags, int flags_per_batch, const double* shift,
                                                               long long cells) {
    const int b = blockIdx.y;
    const T sh = (T)shift[b];
    for (long long cell = (long long)blockIdx.x * kBlock + threadIdx.x; cell < cells; cell += (long long)gridDim.x * kBlock) {
        T a = T(1);
        if (flags) a = (flags[(flags_per_batch ? (long long)b * cells : 0) + cell] & 64u) ? T(1) : T(0);
        div[(long long)b * cells + cell] -= a * sh;
    }
}

int run_divergence(phihip_ctx* ctx, const GridView& v, const void* const vel[3], const uint8_t* flags, int mask_batch, int balance,
                   void* div, hipStream_t s) {
    const VelGrid g = make_velgrid(v);
    const int nblk = ceil_div(v.cells, kBlock);
    PHIHIP_TRY(ensure_buffer(ctx->ws_div, (size_t)2 * v.batch * nblk * sizeof(double)));
    PHIHIP_TRY(ensure_buffer(ctx->ws_scalars, (size_t)v.batch * sizeof(double)));
    double* part_sum = (double*)ctx->ws_div.ptr;
    double* part_act = part_sum + (size_t)v.batch * nblk;
    double* shift = (double*)ctx->ws_scalars.ptr;
    const int fpb = mask_batch > 1 ? 1 : 0;
    {
        LaunchScope ls(ctx, PHIHIP_K_DIVERGENCE, s);
        if (v.dtype == PHIHIP_F64) {
            CComp3<double> c{{(const double*)vel[0], (const double*)vel[1], (const double*)vel[2]}};
            hipLaunchKernelGGL(divergence_kernel<double>, dim3(nblk, v.batch), dim3(kBlock), 0, s, g, c, flags, fpb, (double*)div,
                               part_sum, part_act, nblk);
        } else {
            CComp3<float> c{{(const float*)vel[0], (const float*)vel[1], (const float*)vel[2]}};
            hipLaunchKernelGGL(divergence_kernel<float>, dim3(nblk, v.batch), dim3(kBlock), 0, s, g, c, flags, fpb, (float*)div,
                               part_sum, part_act, nblk);
        }
    }
    if (balance) {
        LaunchScope ls(ctx, PHIHIP_K_DIVERGENCE, s);
        hipLaunchKernelGGL(balance_scalar_kernel, dim3(v.batch), dim3(kBlock), 0, s, (const double*)part_sum, (const double*)part_act,
                           nblk, shift);
        const int nb2 = nblk < 4096 ? nblk : 4096;
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
template <typename T>
__global__ __launch_bounds__(kBlock) void grad_subtract_kernel(VelGrid g, int ca, T* vc, const T* p, const uint8_t* flags,
                                                               int flags_per_batch) {
    const int b = blockIdx.y;
    const long long total = g.ccells[ca];
    const int c1 = g.cn[ca][1], c2 = g.cn[ca][2];
    for (long long f = (long long)blockIdx.x * kBlock + threadIdx.x; f < total; f += (long long)gridDim.x * kBlock) {
        int idx[3];
        idx[2] = (int)(f % c2);
        idx[1] = (int)((f / c2) % c1);
        idx[0] = (int)(f / ((long long)c2 * c1));
        const int phys = idx[ca] + g.off[ca];
        const int n = g.n[ca];
        int L[3] = {idx[0], idx[1], idx[2]}, Rr[3] = {idx[0], idx[1], idx[2]};
        int l = phys - 1, r = phys;
        bool zl = false, zr = false, l_in = l >= 0, r_in = r < n;
        if (l < 0) { if (g.bc[ca][0] == PHIHIP_BC_PERIODIC) l += n; else { zl = true; l = 0; } }
        if (r >= n) { if (g.bc[ca][1] == PHIHIP_BC_PERIODIC) r -= n; else { zr = true; r = n - 1; } }
        L[ca] = l; Rr[ca] = r;
        const long long pb = (long long)b * g.cells;
        const long long offL = ((long long)L[0] * g.n[1] + L[1]) * g.n[2] + L[2];
        const long long offR = ((long long)Rr[0] * g.n[1] + Rr[1]) * g.n[2] + Rr[2];
        const T pl = zl ? T(0) : p[pb + offL];
        const T pr = zr ? T(0) : p[pb + offR];
        T h = T(1);
        if (flags) {
            const long long fb = flags_per_batch ? pb : 0;
            // the face is the lower face of cell R (if R exists in the domain or by wrap) else the upper face of cell L
            if (r_in || g.bc[ca][1] == PHIHIP_BC_PERIODIC) h = (flags[fb + offR] >> (2 * ca)) & 1u ? T(1) : T(0);
            else if (l_in) h = (flags[fb + offL] >> (2 * ca + 1)) & 1u ? T(1) : T(0);
        }
        const long long vo = (long long)b * total + f;
        vc[vo] = vc[vo] - h * ((pr - pl) / (T)g.dx[ca]);
    }
}

int run_grad_subtract(phihip_ctx* ctx, const GridView& v, const uint8_t* flags, int mask_batch, const void* p, void* const vel[3],
                      hipStream_t s) {
    const VelGrid g = make_velgrid(v);
    const int fpb = mask_batch > 1 ? 1 : 0;
    LaunchScope ls(ctx, PHIHIP_K_GRAD_SUBTRACT, s);
    for (int ca = v.ax0; ca < 3; ++ca) {
        const int nblk = ceil_div(v.ccells[ca], kBlock) < 8192 ? ceil_div(v.ccells[ca], kBlock) : 8192;
        if (v.dtype == PHIHIP_F64)
            hipLaunchKernelGGL(grad_subtract_kernel<double>, dim3(nblk, v.batch), dim3(kBlock), 0, s, g, ca, (double*)vel[ca],
                               (const double*)p, flags, fpb);
        else
            hipLaunchKernelGGL(grad_subtract_kernel<float>, dim3(nblk, v.batch), dim3(kBlock), 0, s, g, ca, (float*)vel[ca],
                               (const float*)p, flags, fpb);
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
// obstacle flags (fluid.py:130-137: accessible, hard_bcs = stagger(accessible, minimum), active)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void cellflags_kernel(VelGrid g, const uint8_t* accessible, const uint8_t* active, int per_batch,
                                                           uint8_t* flags) {
    const int b = blockIdx.y;
    const long long mb = per_batch ? (long long)b * g.cells : 0;
    for (long long cell = (long long)blockIdx.x * kBlock + threadIdx.x; cell < g.cells; cell += (long long)gridDim.x * kBlock) {
        const int i2 = (int)(cell % g.n[2]);
        const int i1 = (int)((cell / g.n[2]) % g.n[1]);
        const int i0 = (int)(cell / ((long long)g.n[2] * g.n[1]));
        const int idx[3] = {i0, i1, i2};
        const unsigned self = accessible ? (accessible[mb + cell] ? 1u : 0u) : 1u;
        unsigned f = 0;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            if (ax < g.ax0) continue;
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                int nbv[3] = {i0, i1, i2};
                int j = idx[ax] + (side ? 1 : -1);
                unsigned other;
                if (j < 0 || j >= g.n[ax]) {
                    const int code = g.bc[ax][side];
                    if (code == PHIHIP_BC_PERIODIC) {
                        nbv[ax] = j < 0 ? j + g.n[ax] : j - g.n[ax];
                        other = accessible ? (accessible[mb + ((long long)nbv[0] * g.n[1] + nbv[1]) * g.n[2] + nbv[2]] ? 1u : 0u) : 1u;
                    } else {
                        other = code == PHIHIP_BC_OPEN ? 1u : 0u;   // _accessible_extrapolation: BOUNDARY -> ONE, constant -> ZERO
                    }
                } else {
                    nbv[ax] = j;
                    other = accessible ? (accessible[mb + ((long long)nbv[0] * g.n[1] + nbv[1]) * g.n[2] + nbv[2]] ? 1u : 0u) : 1u;
                }
                if (self & other) f |= 1u << (2 * ax + side);
            }
        }
        const unsigned act = self & (active ? (active[mb + cell] ? 1u : 0u) : 1u);
        if (act) f |= 64u;
        flags[mb + cell] = (uint8_t)f;
    }
}

int run_build_cellflags(phihip_ctx* ctx, const GridView& v, const uint8_t* accessible, const uint8_t* active, int mask_batch,
                        uint8_t* flags, hipStream_t s) {
    const VelGrid g = make_velgrid(v);
    const int nblk = ceil_div(v.cells, kBlock) < 8192 ? ceil_div(v.cells, kBlock) : 8192;
    LaunchScope ls(ctx, PHIHIP_K_OTHER, s);
    hipLaunchKernelGGL(cellflags_kernel, dim3(nblk, mask_batch > 1 ? mask_batch : 1), dim3(kBlock), 0, s, g, accessible, active,
                       mask_batch > 1 ? 1 : 0, flags);
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// diffuse.explicit, order 2: v_d += k dt * laplace(v_d) with the velocity's own padding (phi/physics/diffuse.py:13-60)
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void diffuse_kernel(VelGrid g, int ca, const T* vin, T* vout, T kdt) {
    const int b = blockIdx.y;
    const long long total = g.ccells[ca];
    const int c1 = g.cn[ca][1], c2 = g.cn[ca][2];
    const long long bb = (long long)b * total;
    for (long long f = (long long)blockIdx.x * kBlock + threadIdx.x; f < total; f += (long long)gridDim.x * kBlock) {
        const int i2 = (int)(f % c2);
        const int i1 = (int)((f / c2) % c1);
        const int i0 = (int)(f / ((long long)c2 * c1));
        const T c = vin[bb + f];
        T lap = T(0);
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            if (ax < g.ax0) continue;
            int lo[3] = {i0, i1, i2}, hi[3] = {i0, i1, i2};
            lo[ax] -= 1; hi[ax] += 1;
            const T vl = fetch_comp<T>(vin, g, ca, bb, lo[0], lo[1], lo[2]);
            const T vh = fetch_comp<T>(vin, g, ca, bb, hi[0], hi[1], hi[2]);
            lap += (vl + vh - T(2) * c) / (T)(g.dx[ax] * g.dx[ax]);
        }
        vout[bb + f] = c + kdt * lap;
    }
}

int run_diffuse(phihip_ctx* ctx, const GridView& v, const void* const vin[3], void* const vout[3], double kdt, hipStream_t s) {
    const VelGrid g = make_velgrid(v);
    LaunchScope ls(ctx, PHIHIP_K_OTHER, s);
    for (int ca = v.ax0; ca < 3; ++ca) {
        const int nblk = ceil_div(v.ccells[ca], kBlock) < 8192 ? ceil_div(v.ccells[ca], kBlock) : 8192;
        if (v.dtype == PHIHIP_F64)
            hipLaunchKernelGGL(diffuse_kernel<double>, dim3(nblk, v.batch), dim3(kBlock), 0, s, g, ca, (const double*)vin[ca],
                               (double*)vout[ca], kdt);
        else
            hipLaunchKernelGGL(diffuse_kernel<float>, dim3(nblk, v.batch), dim3(kBlock), 0, s, g, ca, (const float*)vin[ca],
                               (float*)vout[ca], (float)kdt);
    }
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

}  // namespace phihip
