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
#include "advect_common.hpp"

namespace phihip {

// Launch geometry of the gather kernels: one sample per thread, a workgroup = 64 consecutive fast-axis samples x 4 rows; the
// (a0, row block, column block) coordinates of a workgroup are decoded from blockIdx.x with UNIFORM integer divisions -- the
// per-thread div / mod of a linear index cost ~80 of the ~370 VALU instructions per wave of the first version (the kernels
// are VALU-bound: rocprofv3 SQ_INSTS_VALU, profiles/r01_advect_pmc.json).
constexpr int kRowLanes = 64, kRowsPerBlock = kBlock / kRowLanes;

__device__ __forceinline__ bool decode_sample(int n1, int n2, int (&idx)[3], int& f) {
    const int nb2 = (n2 + kRowLanes - 1) / kRowLanes, nb1 = (n1 + kRowsPerBlock - 1) / kRowsPerBlock;
    const int bx = blockIdx.x;
    const int t = bx / nb2;
    idx[2] = (bx - t * nb2) * kRowLanes + (threadIdx.x & (kRowLanes - 1));
    idx[0] = t / nb1;
    idx[1] = (t - idx[0] * nb1) * kRowsPerBlock + (threadIdx.x / kRowLanes);
    f = (idx[0] * n1 + idx[1]) * n2 + idx[2];
    return idx[2] < n2 && idx[1] < n1;
}

static inline long long sample_blocks(const int n[3]) {
    return (long long)((n[2] + kRowLanes - 1) / kRowLanes) * ((n[1] + kRowsPerBlock - 1) / kRowsPerBlock) * n[0];
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
    {
        int idx[3], f;
        if (!decode_sample(n[1], n[2], idx, f)) return;
        T cb_[3] = {T(0), T(0), T(0)}, cf_[3];      // displacements in index units (lookup_pairs_rel: exact integer part), in the windows' arithmetic (face_disp)
        face_disp<T, DIM, CA>(g, vel, b, idx, f, dt, cf_);
#pragma unroll
        for (int a = A0; a < 3; ++a) cb_[a] = -cf_[a];
        AxisPair<T> ax[3];
        T fr[3];
        if (MODE == 0) {
            lookup_pairs_rel<T, DIM>(idx, cb_, n, bc, cv, ax, fr);
            O[f] = gather_multilinear<T, DIM>(F, ax, fr);
        } else {
            const T* __restrict__ W = fwd + (long long)b * total;
            lookup_pairs_rel<T, DIM>(idx, cf_, n, bc, cv, ax, fr);
            const T bwd = gather_multilinear<T, DIM>(W, ax, fr);
            const T nv = mc_correct(W[f], ch, F[f], bwd);
            cb_[ca] += (T)g.off[ca] - T(0.5);
            lookup_pairs_rel<T, DIM>(idx, cb_, n, bc, cv, ax, fr);
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
    {
        int idx[3], f;
        if (!decode_sample(n[1], n[2], idx, f)) return;
        T u[3];
        center_velocity<T, DIM>(g, vel, b, idx, u);
        T cb_[3] = {T(0), T(0), T(0)}, cf_[3] = {T(0), T(0), T(0)};      // displacements in index units
#pragma unroll
        for (int a = A0; a < 3; ++a) {
            const T sft = u[a] * (dt * (T)g.rdx[a]);
            cb_[a] = -sft;
            cf_[a] = sft;
        }
        AxisPair<T> ax[3];
        T fr[3];
        if (MODE == 0) {
            lookup_pairs_rel<T, DIM>(idx, cb_, n, bc, cv, ax, fr);
            O[f] = gather_multilinear<T, DIM>(F, ax, fr);
        } else {
            const T* __restrict__ W = fwd + (long long)b * total;
            lookup_pairs_rel<T, DIM>(idx, cf_, n, bc, cv, ax, fr);
            const T bwd = gather_multilinear<T, DIM>(W, ax, fr);
            const T nv = mc_correct(W[f], ch, F[f], bwd);
            lookup_pairs_rel<T, DIM>(idx, cb_, n, bc, cv, ax, fr);
            T lo, hi;
            gather_minmax<T, DIM>(F, ax, lo, hi);
            O[f] = nv < lo ? lo : (nv > hi ? hi : nv);
        }
    }
}


template <typename T, int DIM, int MODE>
static void launch_advect_staggered(const GridView& v, const VelGrid& g, const void* const f[3], const void* const vel[3],
                                    const void* const fwd[3], void* const out[3], double dt, double ch, hipStream_t s) {
    CComp3a<T> ff{{(const T*)f[0], (const T*)f[1], (const T*)f[2]}};
    CComp3a<T> vv{{(const T*)vel[0], (const T*)vel[1], (const T*)vel[2]}};
    if (DIM == 3)
        hipLaunchKernelGGL((advect_staggered_kernel<T, DIM, 0, MODE>), dim3((unsigned)sample_blocks(v.cn[0]), v.batch), dim3(kBlock), 0, s, g, ff,
                           vv, (const T*)(fwd ? fwd[0] : nullptr), (T*)out[0], (T)dt, (T)ch);
    hipLaunchKernelGGL((advect_staggered_kernel<T, DIM, 1, MODE>), dim3((unsigned)sample_blocks(v.cn[1]), v.batch), dim3(kBlock), 0, s, g, ff, vv,
                       (const T*)(fwd ? fwd[1] : nullptr), (T*)out[1], (T)dt, (T)ch);
    hipLaunchKernelGGL((advect_staggered_kernel<T, DIM, 2, MODE>), dim3((unsigned)sample_blocks(v.cn[2]), v.batch), dim3(kBlock), 0, s, g, ff, vv,
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

// reach of an LDS-staged pass of `kind`: the user's fixed setting (phihip_set_advect_halo 0 / 1 / 2 / 3), or the adaptive choice (-1, default)
static long long grid_fingerprint(const GridView& v) {
    unsigned long long h = 1469598103934665603ULL;      // FNV-1a in unsigned arithmetic (wraps by definition)
    const long long parts[6] = {v.n[0], v.n[1], v.n[2], v.batch, v.dtype, v.rank};
    for (long long x : parts) h = (h ^ (unsigned long long)x) * 1099511628211ULL;
    return h ? (long long)h : 1;
}
static int pass_reach(phihip_ctx* ctx, const GridView& v, int kind, bool has_wide, hipStream_t s) {
    const int reach = ctx->adv_halo >= 0 ? ((!has_wide && ctx->adv_halo > 1) ? 1 : ctx->adv_halo) : adv_choose(ctx, kind, has_wide, grid_fingerprint(v), s);
    ctx->adv_reach_now = reach >= 2 ? 2 : (reach == 1 ? 1 : 0);       // (3 = the experimental 16-row tile: reach 1... tagged 1 below)
    if (reach == 3) ctx->adv_reach_now = 1;
    return reach;
}
static int pass_done(phihip_ctx* ctx, int kind, int reach, hipStream_t s) {
    return ctx->adv_halo < 0 ? adv_record(ctx, kind, reach, s) : PHIHIP_OK;
}

int run_advect_staggered(phihip_ctx* ctx, const GridView& v, const void* const f[3], const void* const vel[3], void* const out[3],
                         double dt, hipStream_t s) {
    PHIHIP_TRY(check_advect_sizes(v));
    bool self = true;
    for (int ca = v.ax0; ca < 3; ++ca) self = self && f[ca] == vel[ca];
    if (self) {   // one launch, taps from LDS (advect_tile.hip); axes with fewer than 4 samples keep the gather kernels
        const int reach = pass_reach(ctx, v, AK_SL_SELF, true, s);
        if (reach > 0) {
            const int st = run_advect_self_tiled(ctx, v, vel, out, dt, reach, AK_SL_SELF, s);
            if (st == PHIHIP_OK) return pass_done(ctx, AK_SL_SELF, reach, s);
            if (st != PHIHIP_ERR_UNSUPPORTED) return st;
        }
        ctx->adv_last_nblk = 0;
    }
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
    // the semi-Lagrangian pass of mac_cormack(v, v, dt) is the self-advection: one LDS-tiled launch instead of D gather launches
    bool first_done = false;
    bool self = true;
    for (int ca = v.ax0; ca < 3; ++ca) self = self && f[ca] == vel[ca];
    if (self) {
        const int reach = pass_reach(ctx, v, AK_SL_SELF, true, s);
        if (reach > 0) {
            const int st = run_advect_self_tiled(ctx, v, vel, tmp, dt, reach, AK_SL_SELF, s);
            if (st == PHIHIP_OK) { first_done = true; PHIHIP_TRY(pass_done(ctx, AK_SL_SELF, reach, s)); }
            else if (st != PHIHIP_ERR_UNSUPPORTED) return st;
        }
    }
    if (self) {             // ... and so is the correction pass: velocity + forward pass staged in LDS windows, all components in one launch (advect_win.hip)
        const int reach = pass_reach(ctx, v, AK_MC_STAG, false, s);
        if (reach > 0) {
            if (!first_done) {      // (the self-advection chose the gather kernels: the correction's windows still read their result)
                LaunchScope ls(ctx, PHIHIP_K_ADVECT, s);
                dispatch_advect_staggered<0>(v, g, f, vel, nullptr, tmp, dt, 0.0, s);
                first_done = true;
            }
            const int st = run_mc_correct_self_tiled(ctx, v, vel, tmp, out, dt, 0.5 * strength, AK_MC_STAG, s);
            if (st == PHIHIP_OK) return pass_done(ctx, AK_MC_STAG, reach, s);
            if (st != PHIHIP_ERR_UNSUPPORTED) return st;
        }
    }
    LaunchScope ls(ctx, PHIHIP_K_ADVECT, s);
    if (!first_done) dispatch_advect_staggered<0>(v, g, f, vel, nullptr, tmp, dt, 0.0, s);
    dispatch_advect_staggered<1>(v, g, f, vel, tmp, out, dt, 0.5 * strength, s);
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

template <typename T, int DIM, int MODE>
static void launch_advect_centered(const GridView& v, const VelGrid& g, const ScalarBc& sb, const void* sfield, const void* const vel[3],
                                   const void* fwd, void* out, double dt, double ch, hipStream_t s) {
    CComp3a<T> vv{{(const T*)vel[0], (const T*)vel[1], (const T*)vel[2]}};
    hipLaunchKernelGGL((advect_centered_kernel<T, DIM, MODE>), dim3((unsigned)sample_blocks(v.n), v.batch), dim3(kBlock), 0, s, g, sb,
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
    {   // scalar + velocity staged in LDS windows (advect_win.hip); phihip_set_advect_halo(ctx, 0) keeps the gather kernel
        const int reach = pass_reach(ctx, v, AK_SL_CEN, true, s);
        if (reach > 0) {
            const int st = run_advect_centered_tiled(ctx, v, sfield, sb, vel, out, dt, reach, AK_SL_CEN, s);
            if (st == PHIHIP_OK) return pass_done(ctx, AK_SL_CEN, reach, s);
            if (st != PHIHIP_ERR_UNSUPPORTED) return st;
        }
        ctx->adv_last_nblk = 0;
    }
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
    // both passes from LDS windows (advect_win.hip), each with the reach its own history asks for; a pass that keeps the gather kernel reads /
    // writes the same buffers
    bool first_done = false;
    {
        const int reach = pass_reach(ctx, v, AK_SL_CEN, true, s);
        if (reach > 0) {
            const int st = run_advect_centered_tiled(ctx, v, sfield, sb, vel, ctx->ws_adv.ptr, dt, reach, AK_SL_CEN, s);
            if (st == PHIHIP_OK) { first_done = true; PHIHIP_TRY(pass_done(ctx, AK_SL_CEN, reach, s)); }
            else if (st != PHIHIP_ERR_UNSUPPORTED) return st;
        }
    }
    if (!first_done) {
        LaunchScope ls(ctx, PHIHIP_K_ADVECT, s);
        dispatch_advect_centered<0>(v, g, sb, sfield, vel, nullptr, ctx->ws_adv.ptr, dt, 0.0, s);
    }
    {
        const int reach = pass_reach(ctx, v, AK_MC_CEN, true, s);
        if (reach > 0) {
            const int st = run_mc_correct_centered_tiled(ctx, v, sfield, sb, vel, ctx->ws_adv.ptr, out, dt, 0.5 * strength, reach, AK_MC_CEN, s);
            if (st == PHIHIP_OK) return pass_done(ctx, AK_MC_CEN, reach, s);
            if (st != PHIHIP_ERR_UNSUPPORTED) return st;
        }
        ctx->adv_last_nblk = 0;
    }
    LaunchScope ls(ctx, PHIHIP_K_ADVECT, s);
    dispatch_advect_centered<1>(v, g, sb, sfield, vel, ctx->ws_adv.ptr, out, dt, 0.5 * strength, s);
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// math.grid_sample: the same tap resolution + gather at coordinates given by the caller
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int DIM>
__global__ __launch_bounds__(kBlock) void grid_sample_kernel(ScalarBc sb, int n0, int n1, int n2, const T* __restrict__ values, long long vstride,
                                                             CComp3a<T> coords, long long npts, T* __restrict__ out, T* __restrict__ omin,
                                                             T* __restrict__ omax) {
    constexpr int A0 = 3 - DIM;
    const int b = blockIdx.y;
    const int n[3] = {n0, n1, n2};
    int bc[3][2];
    T cv[3][2];
    scalar_rule<T>(sb, bc, cv);
    const T* __restrict__ F = values + (long long)b * vstride;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < npts; i += (long long)gridDim.x * kBlock) {
        const long long o = (long long)b * npts + i;
        T c[3] = {T(0), T(0), T(0)};
#pragma unroll
        for (int a = A0; a < 3; ++a) c[a] = coords.p[a][o];
        AxisPair<T> ax[3];
        T fr[3];
        lookup_pairs<T, DIM>(c, n, bc, cv, ax, fr);
        if (out) out[o] = gather_multilinear_weights<T, DIM>(F, ax, fr);
        if (omin) {
            T lo, hi;
            gather_minmax<T, DIM>(F, ax, lo, hi);
            omin[o] = lo;
            omax[o] = hi;
        }
    }
}

int run_grid_sample(phihip_ctx* ctx, const GridView& v, const int32_t s_bc[3][2], const double s_val[3][2], const void* values, int values_batch, const void* const coords[3], long long npts,
                    void* out, void* out_min, void* out_max, hipStream_t s) {
    if (v.cells >= (1LL << 31)) {
        set_error("grid_sample: more than 2^31 values per batch entry are not supported");
        return PHIHIP_ERR_UNSUPPORTED;
    }
    const ScalarBc sb = make_scalar_bc(v, s_bc, s_val);
    const long long vstride = values_batch > 1 ? v.cells : 0;
    const unsigned nblk = (unsigned)((npts + kBlock - 1) / kBlock < 65536 ? (npts + kBlock - 1) / kBlock : 65536);
    LaunchScope ls(ctx, PHIHIP_K_ADVECT, s);
    if (npts > 0) {
#define PHIHIP_GS(T, DIM)                                                                                                                       \
    hipLaunchKernelGGL((grid_sample_kernel<T, DIM>), dim3(nblk, v.batch), dim3(kBlock), 0, s, sb, v.n[0], v.n[1], v.n[2], (const T*)values, vstride, \
                       (CComp3a<T>{{(const T*)coords[0], (const T*)coords[1], (const T*)coords[2]}}), npts, (T*)out, (T*)out_min, (T*)out_max)
        if (v.dtype == PHIHIP_F64) { if (v.rank == 3) PHIHIP_GS(double, 3); else PHIHIP_GS(double, 2); }
        else { if (v.rank == 3) PHIHIP_GS(float, 3); else PHIHIP_GS(float, 2); }
#undef PHIHIP_GS
    }
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

}  // namespace phihip
