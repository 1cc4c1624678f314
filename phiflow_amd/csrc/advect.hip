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

// min / max over the 2^D taps of a lookup (Field.closest_values + math.min / math.max, advect.py:210-212); same tap
// resolution as gather_multilinear
template <typename T, int DIM>
__device__ __forceinline__ void gather_minmax(const T* __restrict__ F, const AxisPair<T> (&ax)[3], T& lo, T& hi) {
#pragma unroll
    for (int corner = 0; corner < (1 << DIM); ++corner) {
        const int b0 = DIM == 3 ? (corner & 1) : 0;
        const int b1 = DIM == 3 ? ((corner >> 1) & 1) : (corner & 1);
        const int b2 = DIM == 3 ? ((corner >> 2) & 1) : ((corner >> 1) & 1);
        T val;
        if (ax[2].cst[b2]) val = ax[2].cv[b2];
        else if (ax[1].cst[b1]) val = ax[1].cv[b1];
        else if (DIM == 3 && ax[0].cst[b0]) val = ax[0].cv[b0];
        else val = F[(DIM == 3 ? ax[0].off[b0] : 0) + ax[1].off[b1] + ax[2].off[b2]];
        lo = corner == 0 ? val : (val < lo ? val : lo);
        hi = corner == 0 ? val : (val > hi ? val : hi);
    }
}

// velocity at the stored face `idx` of component CA: own component + 4-point
// means of the others (sample(velocity, field.geometry, at='face'), phi/field/_resample.py:158-161,279-287,341-364)
template <typename T, int DIM, int CA>
__device__ __forceinline__ void face_velocity(const VelGrid& g, const CComp3a<T>& vel, int b, const int (&idx)[3], int f, T (&u)[3]) {
    constexpr int A0 = 3 - DIM;
    constexpr int ca = CA;
    u[0] = u[1] = u[2] = T(0);
#pragma unroll
    for (int cb = A0; cb < 3; ++cb) {
        if (cb == ca) {
            u[cb] = vel.p[ca][(long long)b * g.ccells[ca] + f];
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
                u[cb] = a1 * T(0.5) + a0 * T(0.5);
            } else {
                const T a0 = v[0][1] * T(0.5) + v[0][0] * T(0.5), a1 = v[1][1] * T(0.5) + v[1][0] * T(0.5);
                u[cb] = a1 * T(0.5) + a0 * T(0.5);
            }
        }
    }
}

// staggered velocity at a cell centre: mean of the cell's two cb-faces (missing ones from padding)
template <typename T, int DIM>
__device__ __forceinline__ void center_velocity(const VelGrid& g, const CComp3a<T>& vel, int b, const int (&idx)[3], T (&u)[3]) {
    constexpr int A0 = 3 - DIM;
    u[0] = u[1] = u[2] = T(0);
#pragma unroll
    for (int cb = A0; cb < 3; ++cb) {
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
        u[cb] = hi * T(0.5) + lo * T(0.5);
    }
}

// AxisPairs + fractions of a lookup at fractional index coordinates `coord` into an array of shape n[] (strides from n)
template <typename T, int DIM>
__device__ __forceinline__ void lookup_pairs(const T (&coord)[3], const int (&n)[3], const int (&bc)[3][2], const T (&cv)[3][2],
                                             AxisPair<T> (&ax)[3], T (&fr)[3]) {
    constexpr int A0 = 3 - DIM;
    const int stride[3] = {n[1] * n[2], n[2], 1};
    fr[0] = fr[1] = fr[2] = T(0);
#pragma unroll
    for (int a = A0; a < 3; ++a) {
        const T fl = floor(coord[a]);
        fr[a] = coord[a] - fl;
        ax[a] = make_pair<T>((int)fl, n[a], stride[a], bc[a][0], bc[a][1], cv[a][0], cv[a][1]);
    }
    if (DIM == 2) { ax[0].off[0] = ax[0].off[1] = 0; ax[0].cst[0] = ax[0].cst[1] = false; ax[0].cv[0] = ax[0].cv[1] = T(0); }
}

// component boundary rule as the (codes, constants) pair lookup_pairs wants
template <typename T>
__device__ __forceinline__ void comp_rule(const VelGrid& g, int comp, int (&bc)[3][2], T (&cv)[3][2]) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bc[a][s] = g.bc[a][s];
            cv[a][s] = (T)g.bcv[a][s][comp];
        }
}

template <typename T>
__device__ __forceinline__ void scalar_rule(const ScalarBc& sb, int (&bc)[3][2], T (&cv)[3][2]) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bc[a][s] = sb.bc[a][s];
            cv[a][s] = (T)sb.val[a][s];
        }
}

__device__ __forceinline__ void unravel(int f, int c1, int c2, int (&idx)[3]) {
    idx[2] = f % c2;
    const int t = f / c2;
    idx[1] = t % c1;
    idx[0] = t / c1;
}

// MODE 0: semi-Lagrangian  out = field(x - dt u)
// MODE 1: MacCormack correction pass (advect.py:203-215). `field` = original field, `fwd` = the semi-Lagrangian result:
//         out = clip(fwd + ch (field - fwd(x + dt u)), min / max of field's taps around x - dt u)
// For staggered components the limiter's lookup uses the CELL grid's frame like the reference (Field.closest_values,
// phi/field/_field.py:427-429 returns from its centred branch for every field): own-axis coordinate m - 1/2.
template <typename T, int DIM, int CA, int MODE>
__global__ __launch_bounds__(kBlock) void advect_staggered_kernel(VelGrid g, CComp3a<T> field, CComp3a<T> vel, const T* __restrict__ fwd,
                                                                  T* __restrict__ out, T dt, T ch) {
    constexpr int A0 = 3 - DIM;
    constexpr int ca = CA;
    const int b = blockIdx.y;
    const int total = (int)g.ccells[ca];
    const int n[3] = {g.cn[ca][0], g.cn[ca][1], g.cn[ca][2]};
    const T* __restrict__ F = field.p[ca] + (long long)b * total;
    T* __restrict__ O = out + (long long)b * total;
    int bc[3][2];
    T cv[3][2];
    comp_rule<T>(g, ca, bc, cv);
    for (int f = blockIdx.x * kBlock + threadIdx.x; f < total; f += gridDim.x * kBlock) {
        int idx[3];
        unravel(f, n[1], n[2], idx);
        T u[3];
        face_velocity<T, DIM, CA>(g, vel, b, idx, f, u);
        T cb_[3] = {T(0), T(0), T(0)}, cf_[3] = {T(0), T(0), T(0)};
#pragma unroll
        for (int a = A0; a < 3; ++a) {
            const T sft = dt * u[a] / (T)g.dx[a];
            cb_[a] = (T)idx[a] - sft;
            cf_[a] = (T)idx[a] + sft;
        }
        AxisPair<T> ax[3];
        T fr[3];
        if (MODE == 0) {
            lookup_pairs<T, DIM>(cb_, n, bc, cv, ax, fr);
            O[f] = gather_multilinear<T, DIM>(F, ax, fr);
        } else {
            const T* __restrict__ W = fwd + (long long)b * total;
            lookup_pairs<T, DIM>(cf_, n, bc, cv, ax, fr);
            const T bwd = gather_multilinear<T, DIM>(W, ax, fr);
            const T nv = W[f] + ch * (F[f] - bwd);
            cb_[ca] += (T)g.off[ca] - T(0.5);
            lookup_pairs<T, DIM>(cb_, n, bc, cv, ax, fr);
            T lo, hi;
            gather_minmax<T, DIM>(F, ax, lo, hi);
            O[f] = nv < lo ? lo : (nv > hi ? hi : nv);   // math.clip = minimum(maximum(x, lo), hi)
        }
    }
}

template <typename T, int DIM, int MODE>
__global__ __launch_bounds__(kBlock) void advect_centered_kernel(VelGrid g, ScalarBc sb, const T* __restrict__ sfield, CComp3a<T> vel,
                                                                 const T* __restrict__ fwd, T* __restrict__ out, T dt, T ch) {
    constexpr int A0 = 3 - DIM;
    const int b = blockIdx.y;
    const int total = (int)g.cells;
    const int n[3] = {g.n[0], g.n[1], g.n[2]};
    const T* __restrict__ F = sfield + (long long)b * total;
    T* __restrict__ O = out + (long long)b * total;
    int bc[3][2];
    T cv[3][2];
    scalar_rule<T>(sb, bc, cv);
    for (int f = blockIdx.x * kBlock + threadIdx.x; f < total; f += gridDim.x * kBlock) {
        int idx[3];
        unravel(f, n[1], n[2], idx);
        T u[3];
        center_velocity<T, DIM>(g, vel, b, idx, u);
        T cb_[3] = {T(0), T(0), T(0)}, cf_[3] = {T(0), T(0), T(0)};
#pragma unroll
        for (int a = A0; a < 3; ++a) {
            const T sft = dt * u[a] / (T)g.dx[a];
            cb_[a] = (T)idx[a] - sft;
            cf_[a] = (T)idx[a] + sft;
        }
        AxisPair<T> ax[3];
        T fr[3];
        if (MODE == 0) {
            lookup_pairs<T, DIM>(cb_, n, bc, cv, ax, fr);
            O[f] = gather_multilinear<T, DIM>(F, ax, fr);
        } else {
            const T* __restrict__ W = fwd + (long long)b * total;
            lookup_pairs<T, DIM>(cf_, n, bc, cv, ax, fr);
            const T bwd = gather_multilinear<T, DIM>(W, ax, fr);
            const T nv = W[f] + ch * (F[f] - bwd);
            lookup_pairs<T, DIM>(cb_, n, bc, cv, ax, fr);
            T lo, hi;
            gather_minmax<T, DIM>(F, ax, lo, hi);
            O[f] = nv < lo ? lo : (nv > hi ? hi : nv);
        }
    }
}

static inline int advect_blocks(long long total) {
    const long long nb = (total + kBlock - 1) / kBlock;
    return (int)(nb < 65536 ? nb : 65536);
}

template <typename T, int DIM, int MODE>
static void launch_advect_staggered(const GridView& v, const VelGrid& g, const void* const f[3], const void* const vel[3],
                                    const void* const fwd[3], void* const out[3], double dt, double ch, hipStream_t s) {
    CComp3a<T> ff{{(const T*)f[0], (const T*)f[1], (const T*)f[2]}};
    CComp3a<T> vv{{(const T*)vel[0], (const T*)vel[1], (const T*)vel[2]}};
    if (DIM == 3)
        hipLaunchKernelGGL((advect_staggered_kernel<T, DIM, 0, MODE>), dim3(advect_blocks(v.ccells[0]), v.batch), dim3(kBlock), 0, s, g, ff,
                           vv, (const T*)(fwd ? fwd[0] : nullptr), (T*)out[0], (T)dt, (T)ch);
    hipLaunchKernelGGL((advect_staggered_kernel<T, DIM, 1, MODE>), dim3(advect_blocks(v.ccells[1]), v.batch), dim3(kBlock), 0, s, g, ff, vv,
                       (const T*)(fwd ? fwd[1] : nullptr), (T*)out[1], (T)dt, (T)ch);
    hipLaunchKernelGGL((advect_staggered_kernel<T, DIM, 2, MODE>), dim3(advect_blocks(v.ccells[2]), v.batch), dim3(kBlock), 0, s, g, ff, vv,
                       (const T*)(fwd ? fwd[2] : nullptr), (T*)out[2], (T)dt, (T)ch);
}

template <int MODE>
static void dispatch_advect_staggered(const GridView& v, const VelGrid& g, const void* const f[3], const void* const vel[3],
                                      const void* const fwd[3], void* const out[3], double dt, double ch, hipStream_t s) {
    if (v.dtype == PHIHIP_F64) {
        if (v.rank == 3) launch_advect_staggered<double, 3, MODE>(v, g, f, vel, fwd, out, dt, ch, s);
        else launch_advect_staggered<double, 2, MODE>(v, g, f, vel, fwd, out, dt, ch, s);
    } else {
        if (v.rank == 3) launch_advect_staggered<float, 3, MODE>(v, g, f, vel, fwd, out, dt, ch, s);
        else launch_advect_staggered<float, 2, MODE>(v, g, f, vel, fwd, out, dt, ch, s);
    }
}

static int check_advect_sizes(const GridView& v) {
    for (int ca = v.ax0; ca < 3; ++ca)
        if (v.ccells[ca] >= (1LL << 31) || v.cells >= (1LL << 31)) {
            set_error("advect: more than 2^31 samples per component and batch entry are not supported");
            return PHIHIP_ERR_UNSUPPORTED;
        }
    return PHIHIP_OK;
}

int run_advect_staggered(phihip_ctx* ctx, const GridView& v, const void* const f[3], const void* const vel[3], void* const out[3],
                         double dt, hipStream_t s) {
    PHIHIP_TRY(check_advect_sizes(v));
    const VelGrid g = make_velgrid(v);
    LaunchScope ls(ctx, PHIHIP_K_ADVECT, s);
    dispatch_advect_staggered<0>(v, g, f, vel, nullptr, out, dt, 0.0, s);
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

// advect.mac_cormack (phi/physics/advect.py:182-215) = semi-Lagrangian pass into the context's scratch + correction pass
int run_mac_cormack_staggered(phihip_ctx* ctx, const GridView& v, const void* const f[3], const void* const vel[3], void* const out[3],
                              double dt, double strength, hipStream_t s) {
    PHIHIP_TRY(check_advect_sizes(v));
    const VelGrid g = make_velgrid(v);
    const size_t esize = v.dtype == PHIHIP_F64 ? 8 : 4;
    size_t offs[3] = {0, 0, 0}, total = 0;
    for (int ca = v.ax0; ca < 3; ++ca) {
        offs[ca] = total;
        total += (((size_t)v.batch * v.ccells[ca] * esize + 255) / 256) * 256;
    }
    PHIHIP_TRY(ensure_buffer(ctx->ws_adv, total));
    void* tmp[3] = {nullptr, nullptr, nullptr};
    for (int ca = v.ax0; ca < 3; ++ca) tmp[ca] = (char*)ctx->ws_adv.ptr + offs[ca];
    LaunchScope ls(ctx, PHIHIP_K_ADVECT, s);
    dispatch_advect_staggered<0>(v, g, f, vel, nullptr, tmp, dt, 0.0, s);
    dispatch_advect_staggered<1>(v, g, f, vel, tmp, out, dt, 0.5 * strength, s);
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

template <typename T, int DIM, int MODE>
static void launch_advect_centered(const GridView& v, const VelGrid& g, const ScalarBc& sb, const void* sfield, const void* const vel[3],
                                   const void* fwd, void* out, double dt, double ch, hipStream_t s) {
    CComp3a<T> vv{{(const T*)vel[0], (const T*)vel[1], (const T*)vel[2]}};
    hipLaunchKernelGGL((advect_centered_kernel<T, DIM, MODE>), dim3(advect_blocks(v.cells), v.batch), dim3(kBlock), 0, s, g, sb,
                       (const T*)sfield, vv, (const T*)fwd, (T*)out, (T)dt, (T)ch);
}

template <int MODE>
static void dispatch_advect_centered(const GridView& v, const VelGrid& g, const ScalarBc& sb, const void* sfield, const void* const vel[3],
                                     const void* fwd, void* out, double dt, double ch, hipStream_t s) {
    if (v.dtype == PHIHIP_F64) {
        if (v.rank == 3) launch_advect_centered<double, 3, MODE>(v, g, sb, sfield, vel, fwd, out, dt, ch, s);
        else launch_advect_centered<double, 2, MODE>(v, g, sb, sfield, vel, fwd, out, dt, ch, s);
    } else {
        if (v.rank == 3) launch_advect_centered<float, 3, MODE>(v, g, sb, sfield, vel, fwd, out, dt, ch, s);
        else launch_advect_centered<float, 2, MODE>(v, g, sb, sfield, vel, fwd, out, dt, ch, s);
    }
}

int run_advect_centered(phihip_ctx* ctx, const GridView& v, const void* sfield, const int32_t s_bc[3][2], const double s_val[3][2],
                        const void* const vel[3], void* out, double dt, hipStream_t s) {
    PHIHIP_TRY(check_advect_sizes(v));
    const VelGrid g = make_velgrid(v);
    const ScalarBc sb = make_scalar_bc(v, s_bc, s_val);
    LaunchScope ls(ctx, PHIHIP_K_ADVECT, s);
    dispatch_advect_centered<0>(v, g, sb, sfield, vel, nullptr, out, dt, 0.0, s);
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

int run_mac_cormack_centered(phihip_ctx* ctx, const GridView& v, const void* sfield, const int32_t s_bc[3][2], const double s_val[3][2],
                             const void* const vel[3], void* out, double dt, double strength, hipStream_t s) {
    PHIHIP_TRY(check_advect_sizes(v));
    const VelGrid g = make_velgrid(v);
    const ScalarBc sb = make_scalar_bc(v, s_bc, s_val);
    const size_t esize = v.dtype == PHIHIP_F64 ? 8 : 4;
    PHIHIP_TRY(ensure_buffer(ctx->ws_adv, (size_t)v.batch * v.cells * esize));
    LaunchScope ls(ctx, PHIHIP_K_ADVECT, s);
    dispatch_advect_centered<0>(v, g, sb, sfield, vel, nullptr, ctx->ws_adv.ptr, dt, 0.0, s);
    dispatch_advect_centered<1>(v, g, sb, sfield, vel, ctx->ws_adv.ptr, out, dt, 0.5 * strength, s);
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

}  // namespace phihip
