// advect.hip -- advect.semi_lagrangian with the euler back-trace (/root/reference phi/physics/advect.py:156-179, :20-24).
// Per stored face of component d:  u = (v_d, 4-point means of the other components at the face)
//                                  x* = x_f - dt u            (index space of component d's own array)
//                                  out = multilinear(field_d, x*) with the extrapolation supplying outside taps
// (phi/field/_resample.py:279-287, 341-364 for u; :257-259 + phiml grid_sample for the gather).
// Gather kernel: taps land within ~CFL cells of the face, so neighbouring lanes hit the same L1/L2 lines; the kernel
// is bound by HBM streaming of 2 x D components. One thread per face, fast axis on consecutive lanes.
#include "common.hpp"

namespace phihip {

template <typename T>
struct CComp3a {
    const T* p[3];
};

// constant per (axis, side) for a centred scalar
struct ScalarBc {
    int bc[3][2];
    double val[3][2];
};

template <typename T>
__device__ __forceinline__ T fetch_vel(const T* C, const VelGrid& g, int ca, long long bbase, const int (&idx_in)[3]) {
    int idx[3] = {idx_in[0], idx_in[1], idx_in[2]};
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

template <typename T>
__device__ __forceinline__ T fetch_scalar(const T* C, const VelGrid& g, const ScalarBc& sb, long long bbase, const int (&idx_in)[3]) {
    int idx[3] = {idx_in[0], idx_in[1], idx_in[2]};
#pragma unroll
    for (int ax = 2; ax >= 0; --ax) {
        if (ax < g.ax0) { idx[ax] = 0; continue; }
        const int n = g.n[ax];
        int i = idx[ax];
        if (i < 0) {
            const int code = sb.bc[ax][0];
            if (code == PHIHIP_BC_PERIODIC) { i %= n; if (i < 0) i += n; }
            else if (code == PHIHIP_BC_CLOSED) return (T)sb.val[ax][0];
            else i = 0;
        } else if (i >= n) {
            const int code = sb.bc[ax][1];
            if (code == PHIHIP_BC_PERIODIC) i %= n;
            else if (code == PHIHIP_BC_CLOSED) return (T)sb.val[ax][1];
            else i = n - 1;
        }
        idx[ax] = i;
    }
    return C[bbase + ((long long)idx[0] * g.n[1] + idx[1]) * g.n[2] + idx[2]];
}

// multilinear weights: prod(where(bit, frac, 1 - frac)) summed over the 2^D taps in corner order (axis a0 = lowest bit)
template <typename T, typename Fetch>
__device__ __forceinline__ T multilinear(const T (&coord)[3], int ax0, Fetch fetch) {
    int i0[3];
    T fr[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const T fl = floor(coord[ax]);
        i0[ax] = (int)fl;
        fr[ax] = coord[ax] - fl;
    }
    T out = T(0);
    const int ncorner = ax0 == 0 ? 8 : 4;
    for (int corner = 0; corner < ncorner; ++corner) {
        int idx[3] = {0, 0, 0};
        T w = T(1);
        int bitpos = 0;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            if (ax < ax0) continue;
            const int bit = (corner >> bitpos) & 1;
            ++bitpos;
            idx[ax] = i0[ax] + bit;
            w *= bit ? fr[ax] : (T(1) - fr[ax]);
        }
        out += fetch(idx) * w;
    }
    return out;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void advect_staggered_kernel(VelGrid g, int ca, CComp3a<T> field, CComp3a<T> vel, T* out, T dt) {
    const int b = blockIdx.y;
    const long long total = g.ccells[ca];
    const int c1 = g.cn[ca][1], c2 = g.cn[ca][2];
    for (long long f = (long long)blockIdx.x * kBlock + threadIdx.x; f < total; f += (long long)gridDim.x * kBlock) {
        int idx[3];
        idx[2] = (int)(f % c2);
        idx[1] = (int)((f / c2) % c1);
        idx[0] = (int)(f / ((long long)c2 * c1));
        T coord[3] = {T(0), T(0), T(0)};
#pragma unroll
        for (int cb = 0; cb < 3; ++cb) {
            if (cb < g.ax0) continue;
            T u;
            if (cb == ca) {
                u = vel.p[ca][(long long)b * total + f];
            } else {
                // component cb at this ca-face: cells (m-1, m) along ca, faces (i, i+1) along cb
                const int m = idx[ca] + g.off[ca];        // physical face number along ca
                const int s = idx[cb] - g.off[cb];        // stored index of physical face idx[cb] along cb
                const long long bb = (long long)b * g.ccells[cb];
                int t[3] = {idx[0], idx[1], idx[2]};
                T v00, v01, v10, v11;   // [ca offset][cb offset]
                t[ca] = m - 1; t[cb] = s;     v00 = fetch_vel<T>(vel.p[cb], g, cb, bb, t);
                t[ca] = m - 1; t[cb] = s + 1; v01 = fetch_vel<T>(vel.p[cb], g, cb, bb, t);
                t[ca] = m;     t[cb] = s;     v10 = fetch_vel<T>(vel.p[cb], g, cb, bb, t);
                t[ca] = m;     t[cb] = s + 1; v11 = fetch_vel<T>(vel.p[cb], g, cb, bb, t);
                // sample_subgrid lerps axis after axis in spatial order with weights (0.5, 0.5)
                if (ca < cb) {
                    const T a0 = v10 * T(0.5) + v00 * T(0.5), a1 = v11 * T(0.5) + v01 * T(0.5);
                    u = a1 * T(0.5) + a0 * T(0.5);
                } else {
                    const T a0 = v01 * T(0.5) + v00 * T(0.5), a1 = v11 * T(0.5) + v10 * T(0.5);
                    u = a1 * T(0.5) + a0 * T(0.5);
                }
            }
            coord[cb] = (T)idx[cb] - dt * u / (T)g.dx[cb];
        }
        const long long fb = (long long)b * total;
        const T* F = field.p[ca];
        out[fb + f] = multilinear<T>(coord, g.ax0, [&](const int (&t)[3]) { return fetch_vel<T>(F, g, ca, fb, t); });
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void advect_centered_kernel(VelGrid g, ScalarBc sb, const T* sfield, CComp3a<T> vel, T* out, T dt) {
    const int b = blockIdx.y;
    const long long total = g.cells;
    for (long long f = (long long)blockIdx.x * kBlock + threadIdx.x; f < total; f += (long long)gridDim.x * kBlock) {
        int idx[3];
        idx[2] = (int)(f % g.n[2]);
        idx[1] = (int)((f / g.n[2]) % g.n[1]);
        idx[0] = (int)(f / ((long long)g.n[2] * g.n[1]));
        T coord[3] = {T(0), T(0), T(0)};
#pragma unroll
        for (int cb = 0; cb < 3; ++cb) {
            if (cb < g.ax0) continue;
            // staggered velocity at the cell centre: mean of the cell's two cb-faces (missing ones from padding)
            const long long bb = (long long)b * g.ccells[cb];
            int t[3] = {idx[0], idx[1], idx[2]};
            t[cb] = idx[cb] - g.off[cb];
            const T lo = fetch_vel<T>(vel.p[cb], g, cb, bb, t);
            t[cb] += 1;
            const T hi = fetch_vel<T>(vel.p[cb], g, cb, bb, t);
            const T u = hi * T(0.5) + lo * T(0.5);
            coord[cb] = (T)idx[cb] - dt * u / (T)g.dx[cb];
        }
        const long long fb = (long long)b * total;
        out[fb + f] = multilinear<T>(coord, g.ax0, [&](const int (&t)[3]) { return fetch_scalar<T>(sfield, g, sb, fb, t); });
    }
}

int run_advect_staggered(phihip_ctx* ctx, const GridView& v, const void* const f[3], const void* const vel[3], void* const out[3],
                         double dt, hipStream_t s) {
    const VelGrid g = make_velgrid(v);
    LaunchScope ls(ctx, PHIHIP_K_ADVECT, s);
    for (int ca = v.ax0; ca < 3; ++ca) {
        const int nblk = ceil_div(v.ccells[ca], kBlock) < 16384 ? ceil_div(v.ccells[ca], kBlock) : 16384;
        if (v.dtype == PHIHIP_F64) {
            CComp3a<double> ff{{(const double*)f[0], (const double*)f[1], (const double*)f[2]}};
            CComp3a<double> vv{{(const double*)vel[0], (const double*)vel[1], (const double*)vel[2]}};
            hipLaunchKernelGGL(advect_staggered_kernel<double>, dim3(nblk, v.batch), dim3(kBlock), 0, s, g, ca, ff, vv, (double*)out[ca], dt);
        } else {
            CComp3a<float> ff{{(const float*)f[0], (const float*)f[1], (const float*)f[2]}};
            CComp3a<float> vv{{(const float*)vel[0], (const float*)vel[1], (const float*)vel[2]}};
            hipLaunchKernelGGL(advect_staggered_kernel<float>, dim3(nblk, v.batch), dim3(kBlock), 0, s, g, ca, ff, vv, (float*)out[ca],
                               (float)dt);
        }
    }
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

int run_advect_centered(phihip_ctx* ctx, const GridView& v, const void* sfield, const int32_t s_bc[3][2], const double s_val[3][2],
                        const void* const vel[3], void* out, double dt, hipStream_t s) {
    const VelGrid g = make_velgrid(v);
    ScalarBc sb;
    memset(&sb, 0, sizeof(sb));
    for (int d = 0; d < v.rank; ++d)
        for (int side = 0; side < 2; ++side) {
            sb.bc[d + v.ax0][side] = s_bc[d][side];
            sb.val[d + v.ax0][side] = s_val ? s_val[d][side] : 0.0;
        }
    const int nblk = ceil_div(v.cells, kBlock) < 16384 ? ceil_div(v.cells, kBlock) : 16384;
    LaunchScope ls(ctx, PHIHIP_K_ADVECT, s);
    if (v.dtype == PHIHIP_F64) {
        CComp3a<double> vv{{(const double*)vel[0], (const double*)vel[1], (const double*)vel[2]}};
        hipLaunchKernelGGL(advect_centered_kernel<double>, dim3(nblk, v.batch), dim3(kBlock), 0, s, g, sb, (const double*)sfield, vv,
                           (double*)out, dt);
    } else {
        CComp3a<float> vv{{(const float*)vel[0], (const float*)vel[1], (const float*)vel[2]}};
        hipLaunchKernelGGL(advect_centered_kernel<float>, dim3(nblk, v.batch), dim3(kBlock), 0, s, g, sb, (const float*)sfield, vv,
                           (float*)out, (float)dt);
    }
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

}  // namespace phihip
