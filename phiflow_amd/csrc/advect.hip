// advect.hip -- advect.semi_lagrangian with the euler back-trace (/root/reference phi/physics/advect.py:156-179, :20-24).
// Per stored face of component d:  u = (v_d, 4-point means of the other components at the face)
//                                  x* = x_f - dt u            (index space of component d's own array)
//                                  out = multilinear(field_d, x*) with the extrapolation supplying outside taps
// (phi/field/_resample.py:279-287, 341-364 for u; :257-259 + phiml grid_sample for the gather).
//
// Gather kernel, HBM-bound (reads D components, writes D): one thread per face with the fast axis on consecutive lanes so
// the own-value load, the store and (for CFL ~ 1) the taps of neighbouring lanes fall into the same cache lines.
// The boundary rule is resolved ONCE PER AXIS into (index, is-constant, constant) pairs; a tap is then three adds and a
// load, and interior wavefronts never diverge. All index math is 32-bit inside one batch entry.
#include "common.hpp"

namespace phihip {

template <typename T>
struct CComp3a {
    const T* p[3];
};

struct ScalarBc {
    int bc[3][2];
    double val[3][2];
};

// one axis of a multilinear tap pair / stencil pair: resolved indices + "outside a constant side" flags
template <typename T>
struct AxisPair {
    int off[2];     // element offset contribution (index * stride), valid when !cst
    bool cst[2];
    T cv[2];
};

template <typename T>
__device__ __forceinline__ void resolve_axis(int i, int n, int stride, int code_lo, int code_hi, T c_lo, T c_hi, int& off, bool& cst, T& cv) {
    cst = false;
    cv = T(0);
    if (i < 0) {
        if (code_lo == PHIHIP_BC_PERIODIC) { i %= n; if (i < 0) i += n; }
        else if (code_lo == PHIHIP_BC_CLOSED) { cst = true; cv = c_lo; i = 0; }
        else i = 0;
    } else if (i >= n) {
        if (code_hi == PHIHIP_BC_PERIODIC) i %= n;
        else if (code_hi == PHIHIP_BC_CLOSED) { cst = true; cv = c_hi; i = n - 1; }
        else i = n - 1;
    }
    off = i * stride;
}

template <typename T>
__device__ __forceinline__ AxisPair<T> make_pair(int i_lo, int n, int stride, int code_lo, int code_hi, T c_lo, T c_hi) {
    AxisPair<T> a;
    if (i_lo >= 0 && i_lo + 1 < n) {   // interior fast path
        a.off[0] = i_lo * stride; a.off[1] = a.off[0] + stride;
        a.cst[0] = a.cst[1] = false;
        a.cv[0] = a.cv[1] = T(0);
    } else {
        resolve_axis<T>(i_lo, n, stride, code_lo, code_hi, c_lo, c_hi, a.off[0], a.cst[0], a.cv[0]);
        resolve_axis<T>(i_lo + 1, n, stride, code_lo, code_hi, c_lo, c_hi, a.off[1], a.cst[1], a.cv[1]);
    }
    return a;
}

// multilinear interpolation from per-axis pairs; constant sides follow PhiML's sequential padding: the LAST axis that lies
// outside a constant side decides. Weights: prod(where(bit, frac, 1 - frac)) summed in corner order (a0 = lowest bit).
template <typename T, int DIM>
__device__ __forceinline__ T gather_multilinear(const T* __restrict__ F, const AxisPair<T> (&ax)[3], const T (&fr)[3]) {
    constexpr int A0 = 3 - DIM;
    const bool any_const = ax[2].cst[0] | ax[2].cst[1] | ax[1].cst[0] | ax[1].cst[1] | (DIM == 3 ? (ax[0].cst[0] | ax[0].cst[1]) : false);
    T out = T(0);
#pragma unroll
    for (int corner = 0; corner < (1 << DIM); ++corner) {
        const int b0 = DIM == 3 ? (corner & 1) : 0;
        const int b1 = DIM == 3 ? ((corner >> 1) & 1) : (corner & 1);
        const int b2 = DIM == 3 ? ((corner >> 2) & 1) : ((corner >> 1) & 1);
        T w = T(1);
        if (DIM == 3) w *= b0 ? fr[0] : (T(1) - fr[0]);
        w *= b1 ? fr[1] : (T(1) - fr[1]);
        w *= b2 ? fr[2] : (T(1) - fr[2]);
        T val;
        if (!any_const) {
            val = F[(DIM == 3 ? ax[0].off[b0] : 0) + ax[1].off[b1] + ax[2].off[b2]];
        } else if (ax[2].cst[b2]) {
            val = ax[2].cv[b2];
        } else if (ax[1].cst[b1]) {
            val = ax[1].cv[b1];
        } else if (DIM == 3 && ax[0].cst[b0]) {
            val = ax[0].cv[b0];
        } else {
            val = F[(DIM == 3 ? ax[0].off[b0] : 0) + ax[1].off[b1] + ax[2].off[b2]];
        }
        out += val * w;
    }
    (void)A0;
    return out;
}

template <typename T, int DIM, int CA>
__global__ __launch_bounds__(kBlock) void advect_staggered_kernel(VelGrid g, CComp3a<T> field, CComp3a<T> vel, T* __restrict__ out, T dt) {
    constexpr int A0 = 3 - DIM;
    constexpr int ca = CA;
    const int b = blockIdx.y;
    const int total = (int)g.ccells[ca];
    const int c1 = g.cn[ca][1], c2 = g.cn[ca][2];
    const T* __restrict__ F = field.p[ca] + (long long)b * total;
    const T* __restrict__ Vown = vel.p[ca] + (long long)b * total;
    T* __restrict__ O = out + (long long)b * total;
    for (int f = blockIdx.x * kBlock + threadIdx.x; f < total; f += gridDim.x * kBlock) {
        int idx[3];
        idx[2] = f % c2;
        const int t = f / c2;
        idx[1] = t % c1;
        idx[0] = t / c1;
        T coord[3] = {T(0), T(0), T(0)};
#pragma unroll
        for (int cb = A0; cb < 3; ++cb) {
            T u;
            if (cb == ca) {
                u = Vown[f];
            } else {
                // component cb at this ca-face: cells (m-1, m) along ca, physical faces (i, i+1) along cb
                const int m = idx[ca] + g.off[ca];
                const int s = idx[cb] - g.off[cb];
                const int n1 = g.cn[cb][1], n2 = g.cn[cb][2];
                const int stride[3] = {n1 * n2, n2, 1};
                const T* __restrict__ C = vel.p[cb] + (long long)b * g.ccells[cb];
                const AxisPair<T> pa = make_pair<T>(m - 1, g.cn[cb][ca], stride[ca], g.bc[ca][0], g.bc[ca][1], (T)g.bcv[ca][0][cb], (T)g.bcv[ca][1][cb]);
                const AxisPair<T> pb = make_pair<T>(s, g.cn[cb][cb], stride[cb], g.bc[cb][0], g.bc[cb][1], (T)g.bcv[cb][0][cb], (T)g.bcv[cb][1][cb]);
                int rest = 0;
#pragma unroll
                for (int ax = A0; ax < 3; ++ax)
                    if (ax != ca && ax != cb) rest += idx[ax] * stride[ax];
                // the later axis of (ca, cb) wins when both lie outside a constant side
                const bool a_last = ca > cb;
                T v[2][2];   // [ca offset][cb offset]
#pragma unroll
                for (int ia = 0; ia < 2; ++ia)
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib) {
                        const bool ca_c = pa.cst[ia], cb_c = pb.cst[ib];
                        if (ca_c || cb_c) {
                            if (a_last) v[ia][ib] = ca_c ? pa.cv[ia] : pb.cv[ib];
                            else v[ia][ib] = cb_c ? pb.cv[ib] : pa.cv[ia];
                        } else {
                            v[ia][ib] = C[rest + pa.off[ia] + pb.off[ib]];
                        }
                    }
                // sample_subgrid lerps axis after axis in spatial order with weights (0.5, 0.5)
                if (ca < cb) {
                    const T a0 = v[1][0] * T(0.5) + v[0][0] * T(0.5), a1 = v[1][1] * T(0.5) + v[0][1] * T(0.5);
                    u = a1 * T(0.5) + a0 * T(0.5);
                } else {
                    const T a0 = v[0][1] * T(0.5) + v[0][0] * T(0.5), a1 = v[1][1] * T(0.5) + v[1][0] * T(0.5);
                    u = a1 * T(0.5) + a0 * T(0.5);
                }
            }
            coord[cb] = (T)idx[cb] - dt * u / (T)g.dx[cb];
        }
        AxisPair<T> ax[3];
        T fr[3] = {T(0), T(0), T(0)};
        const int stride[3] = {c1 * c2, c2, 1};
#pragma unroll
        for (int a = A0; a < 3; ++a) {
            const T fl = floor(coord[a]);
            fr[a] = coord[a] - fl;
            ax[a] = make_pair<T>((int)fl, g.cn[ca][a], stride[a], g.bc[a][0], g.bc[a][1], (T)g.bcv[a][0][ca], (T)g.bcv[a][1][ca]);
        }
        if (DIM == 2) { ax[0].off[0] = ax[0].off[1] = 0; ax[0].cst[0] = ax[0].cst[1] = false; ax[0].cv[0] = ax[0].cv[1] = T(0); }
        O[f] = gather_multilinear<T, DIM>(F, ax, fr);
    }
}

template <typename T, int DIM>
__global__ __launch_bounds__(kBlock) void advect_centered_kernel(VelGrid g, ScalarBc sb, const T* __restrict__ sfield, CComp3a<T> vel, T* __restrict__ out, T dt) {
    constexpr int A0 = 3 - DIM;
    const int b = blockIdx.y;
    const int total = (int)g.cells;
    const int c1 = g.n[1], c2 = g.n[2];
    const T* __restrict__ F = sfield + (long long)b * total;
    T* __restrict__ O = out + (long long)b * total;
    for (int f = blockIdx.x * kBlock + threadIdx.x; f < total; f += gridDim.x * kBlock) {
        int idx[3];
        idx[2] = f % c2;
        const int t = f / c2;
        idx[1] = t % c1;
        idx[0] = t / c1;
        T coord[3] = {T(0), T(0), T(0)};
#pragma unroll
        for (int cb = A0; cb < 3; ++cb) {
            // staggered velocity at the cell centre: mean of the cell's two cb-faces (missing ones from padding)
            const int n1 = g.cn[cb][1], n2 = g.cn[cb][2];
            const int stride[3] = {n1 * n2, n2, 1};
            const T* __restrict__ C = vel.p[cb] + (long long)b * g.ccells[cb];
            const AxisPair<T> pb = make_pair<T>(idx[cb] - g.off[cb], g.cn[cb][cb], stride[cb], g.bc[cb][0], g.bc[cb][1], (T)g.bcv[cb][0][cb], (T)g.bcv[cb][1][cb]);
            int rest = 0;
#pragma unroll
            for (int ax = A0; ax < 3; ++ax)
                if (ax != cb) rest += idx[ax] * stride[ax];
            const T lo = pb.cst[0] ? pb.cv[0] : C[rest + pb.off[0]];
            const T hi = pb.cst[1] ? pb.cv[1] : C[rest + pb.off[1]];
            const T u = hi * T(0.5) + lo * T(0.5);
            coord[cb] = (T)idx[cb] - dt * u / (T)g.dx[cb];
        }
        AxisPair<T> ax[3];
        T fr[3] = {T(0), T(0), T(0)};
        const int stride[3] = {c1 * c2, c2, 1};
#pragma unroll
        for (int a = A0; a < 3; ++a) {
            const T fl = floor(coord[a]);
            fr[a] = coord[a] - fl;
            ax[a] = make_pair<T>((int)fl, g.n[a], stride[a], sb.bc[a][0], sb.bc[a][1], (T)sb.val[a][0], (T)sb.val[a][1]);
        }
        if (DIM == 2) { ax[0].off[0] = ax[0].off[1] = 0; ax[0].cst[0] = ax[0].cst[1] = false; ax[0].cv[0] = ax[0].cv[1] = T(0); }
        O[f] = gather_multilinear<T, DIM>(F, ax, fr);
    }
}

static inline int advect_blocks(long long total) {
    const long long nb = (total + kBlock - 1) / kBlock;
    return (int)(nb < 65536 ? nb : 65536);
}

template <typename T, int DIM>
static void launch_advect_staggered(const GridView& v, const VelGrid& g, const void* const f[3], const void* const vel[3], void* const out[3],
                                    double dt, hipStream_t s) {
    CComp3a<T> ff{{(const T*)f[0], (const T*)f[1], (const T*)f[2]}};
    CComp3a<T> vv{{(const T*)vel[0], (const T*)vel[1], (const T*)vel[2]}};
    if (DIM == 3)
        hipLaunchKernelGGL((advect_staggered_kernel<T, DIM, 0>), dim3(advect_blocks(v.ccells[0]), v.batch), dim3(kBlock), 0, s, g, ff, vv,
                           (T*)out[0], (T)dt);
    hipLaunchKernelGGL((advect_staggered_kernel<T, DIM, 1>), dim3(advect_blocks(v.ccells[1]), v.batch), dim3(kBlock), 0, s, g, ff, vv,
                       (T*)out[1], (T)dt);
    hipLaunchKernelGGL((advect_staggered_kernel<T, DIM, 2>), dim3(advect_blocks(v.ccells[2]), v.batch), dim3(kBlock), 0, s, g, ff, vv,
                       (T*)out[2], (T)dt);
}

int run_advect_staggered(phihip_ctx* ctx, const GridView& v, const void* const f[3], const void* const vel[3], void* const out[3],
                         double dt, hipStream_t s) {
    for (int ca = v.ax0; ca < 3; ++ca)
        if (v.ccells[ca] >= (1LL << 31)) {
            set_error("advect: more than 2^31 samples per component and batch entry are not supported");
            return PHIHIP_ERR_UNSUPPORTED;
        }
    const VelGrid g = make_velgrid(v);
    LaunchScope ls(ctx, PHIHIP_K_ADVECT, s);
    if (v.dtype == PHIHIP_F64) {
        if (v.rank == 3) launch_advect_staggered<double, 3>(v, g, f, vel, out, dt, s);
        else launch_advect_staggered<double, 2>(v, g, f, vel, out, dt, s);
    } else {
        if (v.rank == 3) launch_advect_staggered<float, 3>(v, g, f, vel, out, dt, s);
        else launch_advect_staggered<float, 2>(v, g, f, vel, out, dt, s);
    }
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

template <typename T, int DIM>
static void launch_advect_centered(const GridView& v, const VelGrid& g, const ScalarBc& sb, const void* sfield, const void* const vel[3],
                                   void* out, double dt, hipStream_t s) {
    CComp3a<T> vv{{(const T*)vel[0], (const T*)vel[1], (const T*)vel[2]}};
    hipLaunchKernelGGL((advect_centered_kernel<T, DIM>), dim3(advect_blocks(v.cells), v.batch), dim3(kBlock), 0, s, g, sb, (const T*)sfield, vv,
                       (T*)out, (T)dt);
}

int run_advect_centered(phihip_ctx* ctx, const GridView& v, const void* sfield, const int32_t s_bc[3][2], const double s_val[3][2],
                        const void* const vel[3], void* out, double dt, hipStream_t s) {
    if (v.cells >= (1LL << 31)) {
        set_error("advect: more than 2^31 cells per batch entry are not supported");
        return PHIHIP_ERR_UNSUPPORTED;
    }
    const VelGrid g = make_velgrid(v);
    ScalarBc sb;
    memset(&sb, 0, sizeof(sb));
    for (int d = 0; d < v.rank; ++d)
        for (int side = 0; side < 2; ++side) {
            sb.bc[d + v.ax0][side] = s_bc[d][side];
            sb.val[d + v.ax0][side] = s_val ? s_val[d][side] : 0.0;
        }
    LaunchScope ls(ctx, PHIHIP_K_ADVECT, s);
    if (v.dtype == PHIHIP_F64) {
        if (v.rank == 3) launch_advect_centered<double, 3>(v, g, sb, sfield, vel, out, dt, s);
        else launch_advect_centered<double, 2>(v, g, sb, sfield, vel, out, dt, s);
    } else {
        if (v.rank == 3) launch_advect_centered<float, 3>(v, g, sb, sfield, vel, out, dt, s);
        else launch_advect_centered<float, 2>(v, g, sb, sfield, vel, out, dt, s);
    }
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

}  // namespace phihip
