// adjoint.hip -- backward passes (vector-Jacobian products) of the fluid step, SURVEY §8 f5: PhiFlow is differentiable through
// its backends' autodiff (/root/reference tests/commit/physics/test_fluid.py:55-73, tests/commit/test_colab_fluids_tutorial.py:11-34);
// here every forward kernel gets a hand-written adjoint:
//   * semi-Lagrangian advection: the transposed gather into the advected field AND, through the lookup coordinates x* = x - dt u, into the
//     advecting velocity (own component + the 4-point means of the others) -- as three gather passes without atomics (r3, below); the
//     scatter-add form (atomicAdd per tap) remains for samples whose lookup left their cell neighbourhood and for grid_sample;
//   * centred -> staggered resample: adjoint scatter into the cells;
//   * make_incompressible: the implicit-function adjoint of the linear solve (A is symmetric: one more CG solve with the
//     same matrix-free operator), divergence and gradient swap roles (G^T = -D with homogeneous boundary values).
// Every kernel recomputes the forward quantities it needs; nothing is taped on the device.
#include <stdlib.h>

#include "advect_common.hpp"
#include "march_dispatch.hpp"

namespace phihip {

template <typename T>
struct Comp3w {
    T* p[3];
};

// d(out)/d(frac_a) of the multilinear gather and scatter of g * w into the taps of the field gradient
template <typename T, int DIM>
__device__ __forceinline__ void gather_adjoint(const T* __restrict__ F, T* __restrict__ gF, const AxisPair<T> (&ax)[3], const T (&fr)[3], T g,
                                               T (&dfr)[3]) {
    dfr[0] = dfr[1] = dfr[2] = T(0);
#pragma unroll
    for (int corner = 0; corner < (1 << DIM); ++corner) {
        const int b0 = DIM == 3 ? (corner & 1) : 0;
        const int b1 = DIM == 3 ? ((corner >> 1) & 1) : (corner & 1);
        const int b2 = DIM == 3 ? ((corner >> 2) & 1) : ((corner >> 1) & 1);
        const T w0 = DIM == 3 ? (b0 ? fr[0] : (T(1) - fr[0])) : T(1);
        const T w1 = b1 ? fr[1] : (T(1) - fr[1]);
        const T w2 = b2 ? fr[2] : (T(1) - fr[2]);
        T val;
        bool is_const = true;
        if (ax[2].cst[b2]) val = ax[2].cv[b2];
        else if (ax[1].cst[b1]) val = ax[1].cv[b1];
        else if (DIM == 3 && ax[0].cst[b0]) val = ax[0].cv[b0];
        else {
            is_const = false;
            const int off = (DIM == 3 ? ax[0].off[b0] : 0) + ax[1].off[b1] + ax[2].off[b2];
            val = F[off];
            if (gF) atomicAdd(gF + off, g * (w0 * w1 * w2));
        }
        (void)is_const;
        if (DIM == 3) dfr[0] += val * (b0 ? T(1) : T(-1)) * w1 * w2;
        dfr[1] += val * (b1 ? T(1) : T(-1)) * w0 * w2;
        dfr[2] += val * (b2 ? T(1) : T(-1)) * w0 * w1;
    }
}

// the index a tap of make_pair resolves to, as ONE index: false = outside a constant side (the tap is a constant: no gradient)
__device__ __forceinline__ bool resolve_index(int i, int n, int code_lo, int code_hi, int& r) {
    if ((unsigned)i < (unsigned)n) { r = i; return true; }
    if (code_lo == PHIHIP_BC_PERIODIC) { r = wrap_index(i, n); return true; }
    if (i < 0) { r = 0; return code_lo != PHIHIP_BC_CLOSED; }
    r = n - 1;
    return code_hi != PHIHIP_BC_CLOSED;
}

// =====================================================================================================================
// Gather form of the semi-Lagrangian adjoints (r3). The scatter form above needs 17 global atomics per sample and runs at 0.08-0.10 of the
// HBM rate; accumulating them in LDS is no faster (LDS float atomics retire ~0.5 lane-operations per clock and CU: measured).
// Without atomics:
//   pass A  per sample: back-trace, store the lookup coordinate x* (per axis), g, and g d(out)/d(x*) (-dt / dx) per axis. A sample whose
//           lookup left the cell neighbourhood (|x*_a - i_a| >= 1 for some axis, or not finite) scatters its taps atomically as before and
//           stores g = 0.
//   pass B  per cell t of the advected field:  g_field[t] += sum over the 3^D neighbouring samples s of  g_s prod_a hat(x*_a(s) - t_a),
//           hat(x) = max(0, 1 - |x|) -- the multilinear weight of tap t in the lookup of s. The boundary rule is folded into the staging of the
//           neighbourhood: the slot at unresolved index i beyond the array holds, per axis, the sample of the periodic image with its
//           coordinate shifted by i - s, the EDGE sample itself with the coordinate shifted by s - i under a clamped (zero-gradient) side -- the
//           taps beyond the edge are the edge cell --, nothing beyond a constant side (those taps are constants).
//   pass C  per sample j of velocity component cb: the transposed means -- sum over the samples whose velocity lookup read j of their
//           g d(out)/d(x*_cb) with the forward weights (own component 1, 4-point means 1/4 each, cell-centre means 1/2 each); per axis the
//           unresolved positions that resolve to j -- itself and the ghost positions at the ends of the array -- name the sources.
// =====================================================================================================================
template <typename T>
struct alignas(4 * sizeof(T)) GatherSlot {
    T c0, c1, c2, g;      // lookup coordinate per axis (index space of the sample's own array) and the upstream gradient of the sample (0: handled
};                        // atomically in pass A): one 16 / 32-byte record per sample -- one store in pass A, one load per staged slot in pass B

template <typename T>
struct TraceOut {
    GatherSlot<T>* rec;
    T* du[3];     // g * d(out)/d(x*_a) * d(x*_a)/d(u_a)
};

template <typename T, int DIM, int CA, bool STAG>
__device__ __forceinline__ void advect_bwd_trace_body(const VelGrid& g, const ScalarBc& sb, const T* __restrict__ fieldp, const CComp3a<T>& vel,
                                                      const T* __restrict__ gout, T* __restrict__ gfield, const TraceOut<T>& out, int want_gvel, T dt) {
    constexpr int A0 = 3 - DIM;
    constexpr int ca = CA;
    const int b = blockIdx.y;
    const int total = STAG ? (int)g.ccells[ca] : (int)g.cells;
    const int n[3] = {STAG ? g.cn[ca][0] : g.n[0], STAG ? g.cn[ca][1] : g.n[1], STAG ? g.cn[ca][2] : g.n[2]};
    const T* __restrict__ F = fieldp + (long long)b * total;
    T* __restrict__ GF = gfield ? gfield + (long long)b * total : nullptr;
    int bc[3][2];
    T cv[3][2];
    if (STAG) comp_rule<T>(g, ca, bc, cv); else scalar_rule<T>(sb, bc, cv);
    for (int f = blockIdx.x * kBlock + threadIdx.x; f < total; f += gridDim.x * kBlock) {
        const long long o = (long long)b * total + f;
        const T go = gout[o];
        int idx[3];
        unravel(f, n[1], n[2], idx);
        T u[3];
        if (STAG) face_velocity<T, DIM, CA>(g, vel, b, idx, f, u); else center_velocity<T, DIM>(g, vel, b, idx, u);
        T coord[3] = {T(0), T(0), T(0)};
        bool near = true;
#pragma unroll
        for (int a = A0; a < 3; ++a) {
            coord[a] = (T)idx[a] - u[a] * (dt * (T)g.rdx[a]);
            near = near && fabs(coord[a] - (T)idx[a]) < T(1);          // (false for NaN)
        }
        AxisPair<T> ax[3];
        T fr[3], dfr[3];
        lookup_pairs<T, DIM>(coord, n, bc, cv, ax, fr);
        gather_adjoint<T, DIM>(F, near ? nullptr : GF, ax, fr, go, dfr);
        GatherSlot<T> rec;
        rec.c0 = coord[0]; rec.c1 = coord[1]; rec.c2 = coord[2];
        rec.g = near ? go : T(0);
        out.rec[o] = rec;
        if (want_gvel) {
#pragma unroll
            for (int a = A0; a < 3; ++a) out.du[a][o] = go * dfr[a] * -(dt * (T)g.rdx[a]);   // coord_a = idx_a - dt u_a / dx_a
        }
    }
}

template <typename T, int DIM, int CA, bool STAG>
__global__ __launch_bounds__(kBlock) void advect_bwd_trace_kernel(VelGrid g, ScalarBc sb, const T* __restrict__ fieldp, CComp3a<T> vel,
                                                                  const T* __restrict__ gout, T* __restrict__ gfield, TraceOut<T> out, int want_gvel, T dt) {
    advect_bwd_trace_body<T, DIM, CA, STAG>(g, sb, fieldp, vel, gout, gfield, out, want_gvel, dt);
}

// r6: ALL staggered components in one launch (blockIdx.z = component - first axis; every component has its own record array, so pass B runs once for all as well).
// The body is the per-component one: same arithmetic, same bits.
template <typename T>
struct TraceAll {
    const T* f[3];
    const T* gout[3];
    T* gfield[3];
    TraceOut<T> out[3];
};

template <typename T, int DIM>
__global__ __launch_bounds__(kBlock) void advect_bwd_trace_all_kernel(VelGrid g, CComp3a<T> vel, TraceAll<T> a, int want_gvel, T dt) {
    const ScalarBc none{};                                   // (staggered samples take their rule from the grid)
    const int ca = (3 - DIM) + (int)blockIdx.z;
    if (ca == 0) advect_bwd_trace_body<T, DIM, 0, true>(g, none, a.f[0], vel, a.gout[0], a.gfield[0], a.out[0], want_gvel, dt);
    else if (ca == 1) advect_bwd_trace_body<T, DIM, 1, true>(g, none, a.f[1], vel, a.gout[1], a.gfield[1], a.out[1], want_gvel, dt);
    else advect_bwd_trace_body<T, DIM, 2, true>(g, none, a.f[2], vel, a.gout[2], a.gfield[2], a.out[2], want_gvel, dt);
}

// ---- pass B ---------------------------------------------------------------------------------------------------------------
constexpr int kGatherT1 = 8, kGatherT2 = 32;
template <typename T, int DIM> constexpr int gather_t0() { return DIM == 3 ? (sizeof(T) == 4 ? 4 : 2) : 1; }
template <typename T, int DIM> constexpr int gather_cells() { return (gather_t0<T, DIM>() + (DIM == 3 ? 2 : 0)) * (kGatherT1 + 2) * (kGatherT2 + 2); }

// the sample a staging slot at unresolved index i holds along one axis, and the shift of its coordinate (see the header of this section)
template <typename T>
__device__ __forceinline__ bool gather_slot(int i, int n, int code_lo, int code_hi, int& s, T& shift) {
    s = i;
    shift = T(0);
    if ((unsigned)i < (unsigned)n) return true;
    if (code_lo == PHIHIP_BC_PERIODIC) {
        s = wrap_index(i, n);
        shift = (T)(i - s);
        return true;
    }
    const int code = i < 0 ? code_lo : code_hi;
    if (code == PHIHIP_BC_CLOSED) return false;          // constant side: the taps beyond it are constants
    s = i < 0 ? 0 : n - 1;                               // clamped side: the taps beyond it ARE the edge cell
    shift = (T)(s - i);
    return true;
}

template <typename T>
__device__ __forceinline__ T hat(T x) { return fmax(T(0), T(1) - fabs(x)); }

template <typename T, int DIM>
__device__ __forceinline__ void advect_bwd_field_gather_body(GatherSlot<T>* slots, int n0, int n1, int n2, const ScalarBc& rule, const GatherSlot<T>* __restrict__ in_rec,
                                                             T* __restrict__ gfield, int nb1, int nb2) {
    constexpr int T0 = gather_t0<T, DIM>(), T1 = kGatherT1, T2 = kGatherT2;
    constexpr int E0 = DIM == 3 ? T0 + 2 : 1, E1 = T1 + 2, E2 = T2 + 2;
    const int b = blockIdx.y;
    const long long total = (long long)n0 * n1 * n2;
    const long long base = (long long)b * total;
    const int bid = blockIdx.x;
    const int g2 = (bid % nb2) * T2, g1 = ((bid / nb2) % nb1) * T1, g0 = DIM == 3 ? (bid / (nb2 * nb1)) * T0 : 0;
    // staging: every slot of the tile grown by one cell; tiles whose grown window lies inside the array skip the boundary rule (uniform)
    const bool inside = (DIM != 3 || (g0 >= 1 && g0 + T0 + 1 <= n0)) && g1 >= 1 && g1 + T1 + 1 <= n1 && g2 >= 1 && g2 + T2 + 1 <= n2;
    constexpr int kSlots = E0 * E1 * E2, kIters = (kSlots + kBlock - 1) / kBlock;
    GatherSlot<T> rec[kIters];
    T sh[kIters][3];
    bool okv[kIters];
#pragma unroll
    for (int it = 0; it < kIters; ++it) {                // every record of the thread is requested before the first one is used
        const int e = threadIdx.x + it * kBlock;
        const int l2 = e % E2, t = e / E2;
        const int l1 = t % E1, l0 = t / E1;
        int s0 = DIM == 3 ? g0 - 1 + l0 : 0, s1 = g1 - 1 + l1, s2 = g2 - 1 + l2;
        T h0 = T(0), h1 = T(0), h2 = T(0);
        bool ok = e < kSlots;
        if (!inside && ok) {
            if (DIM == 3) ok = gather_slot<T>(g0 - 1 + l0, n0, rule.bc[0][0], rule.bc[0][1], s0, h0);
            ok = gather_slot<T>(g1 - 1 + l1, n1, rule.bc[1][0], rule.bc[1][1], s1, h1) && ok;
            ok = gather_slot<T>(g2 - 1 + l2, n2, rule.bc[2][0], rule.bc[2][1], s2, h2) && ok;
        }
        rec[it] = in_rec[ok ? base + ((long long)s0 * n1 + s1) * n2 + s2 : base];
        sh[it][0] = h0; sh[it][1] = h1; sh[it][2] = h2;
        okv[it] = ok;
    }
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
        const int e = threadIdx.x + it * kBlock;
        GatherSlot<T> v = rec[it];
        v.c0 += sh[it][0]; v.c1 += sh[it][1]; v.c2 += sh[it][2];
        if (!okv[it]) v.c0 = v.c1 = v.c2 = v.g = T(0);
        if (e < kSlots) slots[e] = v;
    }
    __syncthreads();
    // a thread owns the T0 targets (g0 .. g0 + T0 - 1, t1, t2): every staged slot is read once and serves up to three of them -- the in-plane
    // weights hat(c1 - t1) hat(c2 - t2) are the same for all, only the a0 weight differs
    const int tx = threadIdx.x % T2, ty = threadIdx.x / T2;
    const int t1 = g1 + ty, t2 = g2 + tx;
    const bool col_ok = t1 < n1 && t2 < n2;
    if (DIM == 3) {
        // plane pl of the staged window serves the targets pl - 2, pl - 1, pl (tile-local): three running sums rotate through the planes
        T a0 = T(0), a1 = T(0), a2 = T(0);
#pragma unroll 1
        for (int pl = 0; pl < E0; ++pl) {
            const T z = (T)(g0 + pl);
            T s0 = T(0), s1 = T(0), s2 = T(0);
#pragma unroll
            for (int d1 = 0; d1 < 3; ++d1)
#pragma unroll
                for (int d2 = 0; d2 < 3; ++d2) {
                    const GatherSlot<T> v = slots[(pl * E1 + (ty + d1)) * E2 + tx + d2];
                    const T w12 = v.g * hat<T>(v.c1 - (T)t1) * hat<T>(v.c2 - (T)t2);
                    s0 += w12 * hat<T>(v.c0 - (z - T(2)));          // slot plane pl sits at unresolved index g0 - 1 + pl and serves the
                    s1 += w12 * hat<T>(v.c0 - (z - T(1)));          // targets g0 + pl - 2, g0 + pl - 1, g0 + pl
                    s2 += w12 * hat<T>(v.c0 - z);
                }
            a0 += s0; a1 += s1; a2 += s2;
            const int k0 = pl - 2;                                  // the target one plane below the slot plane has seen its three planes
            if (k0 >= 0 && g0 + k0 < n0 && col_ok) gfield[base + ((long long)(g0 + k0) * n1 + t1) * n2 + t2] += a0;
            a0 = a1; a1 = a2; a2 = T(0);
        }
    } else {
        T acc = T(0);
#pragma unroll
        for (int d1 = 0; d1 < 3; ++d1)
#pragma unroll
            for (int d2 = 0; d2 < 3; ++d2) {
                const GatherSlot<T> v = slots[(ty + d1) * E2 + tx + d2];
                acc += v.g * hat<T>(v.c1 - (T)t1) * hat<T>(v.c2 - (T)t2);
            }
        if (col_ok) gfield[base + (long long)t1 * n2 + t2] += acc;
    }
}

template <typename T, int DIM>
__global__ __launch_bounds__(kBlock) void advect_bwd_field_gather_kernel(int n0, int n1, int n2, ScalarBc rule, TraceOut<T> in, T* __restrict__ gfield,
                                                                         int nb1, int nb2) {
    __shared__ GatherSlot<T> slots[gather_cells<T, DIM>()];
    advect_bwd_field_gather_body<T, DIM>(slots, n0, n1, n2, rule, in.rec, gfield, nb1, nb2);
}

// r6: the record arrays of ALL staggered components in one launch (blockIdx.z = component - first axis); a component with fewer tiles than the launch's
// blockIdx.x range lets the surplus workgroups return before the barrier
template <typename T>
struct GatherAll {
    int n[3][3];
    int bc[3][3][2];
    int nb1[3], nb2[3], blocks[3];
    const GatherSlot<T>* rec[3];
    T* gfield[3];
};

template <typename T, int DIM>
__global__ __launch_bounds__(kBlock) void advect_bwd_field_gather_all_kernel(GatherAll<T> a) {
    __shared__ GatherSlot<T> slots[gather_cells<T, DIM>()];
    const int ca = (3 - DIM) + (int)blockIdx.z;
    if ((int)blockIdx.x >= a.blocks[ca] || a.gfield[ca] == nullptr) return;
    ScalarBc rule{};
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) { rule.bc[ax][0] = a.bc[ca][ax][0]; rule.bc[ax][1] = a.bc[ca][ax][1]; }
    advect_bwd_field_gather_body<T, DIM>(slots, a.n[ca][0], a.n[ca][1], a.n[ca][2], rule, a.rec[ca], a.gfield[ca], a.nb1[ca], a.nb2[ca]);
}

// ---- pass C ---------------------------------------------------------------------------------------------------------------
// One axis of a transposed pair stencil. Forward: the source at index q reads the taps (q + d, q + d + 1) of an array of n_t samples under the
// boundary codes of the axis (d = off - 1: the cell pair (m - 1, m) of a face; d = -off: the face pair (s, s + 1)). Transposed: target j receives
// from every source q in [0, n_src) one of whose taps resolves, non-constant, to j. The unresolved positions that resolve to j are j itself and,
// at the ends of the array, the ghost position beyond a clamped side next to it or the periodic image: at most three. Every (position, tap)
// pair names one source index; a source named twice reads the target twice (both taps clamp onto the edge sample) -- entries may repeat.
struct AxisSources {
    int q[6];
    bool on[6];
};

__device__ __forceinline__ AxisSources transposed_sources(int j, int n_t, int code_lo, int code_hi, int d, int n_src) {
    const bool periodic = code_lo == PHIHIP_BC_PERIODIC;
    int p[3] = {j, 0, 0};
    bool have[3] = {true, false, false};
    if (j == 0) {                       // ... the position below the array: clamps onto sample 0; the image above the array wraps onto it
        p[1] = periodic ? n_t : -1;
        have[1] = periodic || code_lo == PHIHIP_BC_OPEN;
    }
    if (j == n_t - 1) {
        p[2] = periodic ? -1 : n_t;
        have[2] = periodic || code_hi == PHIHIP_BC_OPEN;
    }
    AxisSources r;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int q = p[k] - d - t;
            r.q[2 * k + t] = q;
            r.on[2 * k + t] = have[k] && q >= 0 && q < n_src;
        }
    return r;
}

template <typename T>
struct DuIn {
    const T* p[3];     // per SOURCE component ca (STAG) -- or [0] only (cell samples): g d(out)/d(x*_cb) of that source's samples
};

template <typename T, int DIM, int CB, bool STAG>
__device__ __forceinline__ void advect_bwd_velocity_gather_body(const VelGrid& g, const DuIn<T>& du, T* __restrict__ gvel) {
    constexpr int A0 = 3 - DIM;
    constexpr int cb = CB;
    const int b = blockIdx.y;
    const int total = (int)g.ccells[cb];
    const int n[3] = {g.cn[cb][0], g.cn[cb][1], g.cn[cb][2]};
    for (int f = blockIdx.x * kBlock + threadIdx.x; f < total; f += gridDim.x * kBlock) {
        int j[3];
        unravel(f, n[1], n[2], j);
        T acc = T(0);
        if (STAG) {
            // faces of component ca (array shape cn[ca]) whose 4-point mean of component cb reads sample j: cells (m - 1, m) along ca with
            // m = idx[ca] + off[ca], faces (s, s + 1) along cb with s = idx[cb] - off[cb]; the other axis is shared. Phase 1 requests the own
            // sample and the own-position sources of BOTH other components (entries 0, 1 of both axes: branch-free, nine loads in flight);
            // phase 2 sums them and handles the ghost positions at the ends of the array under one uniform branch per component.
            AxisSources A[3], B[3];
            int oa[3][6], ob[3][6];
            T val[3][2][2];
            const T* D[3] = {nullptr, nullptr, nullptr};
            const T own = du.p[cb][(long long)b * total + f];                   // own component: the sample reads itself with weight 1
#pragma unroll
            for (int ca = A0; ca < 3; ++ca) {
                if (ca == cb) continue;
                const int s1 = g.cn[ca][1], s2 = g.cn[ca][2];
                const int stride[3] = {s1 * s2, s2, 1};
                A[ca] = transposed_sources(j[ca], g.cn[cb][ca], g.bc[ca][0], g.bc[ca][1], g.off[ca] - 1, g.cn[ca][ca]);
                B[ca] = transposed_sources(j[cb], g.cn[cb][cb], g.bc[cb][0], g.bc[cb][1], -g.off[cb], g.cn[ca][cb]);
                int rest = 0;                                         // (32-bit offsets within one batch entry: < 2^31 samples, checked by the caller)
#pragma unroll
                for (int ax = A0; ax < 3; ++ax)
                    if (ax != ca && ax != cb) rest += j[ax] * stride[ax];
                D[ca] = du.p[ca] + (long long)b * g.ccells[ca];
#pragma unroll
                for (int k = 0; k < 6; ++k) { oa[ca][k] = rest + A[ca].q[k] * stride[ca]; ob[ca][k] = B[ca].q[k] * stride[cb]; }
#pragma unroll
                for (int ka = 0; ka < 2; ++ka)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) val[ca][ka][kb] = D[ca][(A[ca].on[ka] && B[ca].on[kb]) ? oa[ca][ka] + ob[ca][kb] : 0];
            }
            acc = own;
#pragma unroll
            for (int ca = A0; ca < 3; ++ca) {
                if (ca == cb) continue;
                T part = T(0);
#pragma unroll
                for (int ka = 0; ka < 2; ++ka)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) part += (A[ca].on[ka] && B[ca].on[kb]) ? val[ca][ka][kb] : T(0);
                bool ghost = false;
#pragma unroll
                for (int k = 2; k < 6; ++k) ghost = ghost || A[ca].on[k] || B[ca].on[k];
                if (wave_any(ghost)) {                              // wavefronts at an end of the array: the 32 remaining slots, again all loads first
                    T gval[32];
                    int m = 0;
#pragma unroll
                    for (int ka = 0; ka < 6; ++ka)
#pragma unroll
                        for (int kb = (ka < 2 ? 2 : 0); kb < 6; ++kb) gval[m++] = D[ca][(A[ca].on[ka] && B[ca].on[kb]) ? oa[ca][ka] + ob[ca][kb] : 0];
                    m = 0;
#pragma unroll
                    for (int ka = 0; ka < 6; ++ka)
#pragma unroll
                        for (int kb = (ka < 2 ? 2 : 0); kb < 6; ++kb) part += (A[ca].on[ka] && B[ca].on[kb]) ? gval[m++] : (m++, T(0));
                }
                acc += T(0.25) * part;
            }
        } else {
            // cells whose centre mean of component cb reads face j: faces (s, s + 1) along cb with s = idx[cb] - off[cb]
            const int stride[3] = {g.n[1] * g.n[2], g.n[2], 1};
            const AxisSources B = transposed_sources(j[cb], g.cn[cb][cb], g.bc[cb][0], g.bc[cb][1], -g.off[cb], g.n[cb]);
            int rest = 0;
#pragma unroll
            for (int ax = A0; ax < 3; ++ax)
                if (ax != cb) rest += j[ax] * stride[ax];
            const T* __restrict__ D = du.p[0] + (long long)b * g.cells;
            const T v0 = D[B.on[0] ? rest + B.q[0] * stride[cb] : 0], v1 = D[B.on[1] ? rest + B.q[1] * stride[cb] : 0];
            T part = (B.on[0] ? v0 : T(0)) + (B.on[1] ? v1 : T(0));
            if (wave_any(B.on[2] || B.on[3] || B.on[4] || B.on[5])) {
                T gval[4];
#pragma unroll
                for (int kb = 2; kb < 6; ++kb) gval[kb - 2] = D[B.on[kb] ? rest + B.q[kb] * stride[cb] : 0];
#pragma unroll
                for (int kb = 2; kb < 6; ++kb) part += B.on[kb] ? gval[kb - 2] : T(0);
            }
            acc = T(0.5) * part;
        }
        gvel[(long long)b * total + f] += acc;
    }
}

template <typename T, int DIM, int CB, bool STAG>
__global__ __launch_bounds__(kBlock) void advect_bwd_velocity_gather_kernel(VelGrid g, DuIn<T> du, T* __restrict__ gvel) {
    advect_bwd_velocity_gather_body<T, DIM, CB, STAG>(g, du, gvel);
}

// r6: every velocity component in one launch (blockIdx.z = component - first axis)
template <typename T>
struct DuAll {
    DuIn<T> in[3];
    T* gvel[3];
};

template <typename T, int DIM, bool STAG>
__global__ __launch_bounds__(kBlock) void advect_bwd_velocity_gather_all_kernel(VelGrid g, DuAll<T> a) {
    const int cb = (3 - DIM) + (int)blockIdx.z;
    if (cb == 0) advect_bwd_velocity_gather_body<T, DIM, 0, STAG>(g, a.in[0], a.gvel[0]);
    else if (cb == 1) advect_bwd_velocity_gather_body<T, DIM, 1, STAG>(g, a.in[1], a.gvel[1]);
    else advect_bwd_velocity_gather_body<T, DIM, 2, STAG>(g, a.in[2], a.gvel[2]);
}

// adjoint of centered_to_staggered_kernel (face = 0.5 * scale * (cell left + cell right)) as a gather: every cell sums the faces whose pair
// (phys - 1, phys) resolves to it under the scalar's extrapolation -- the transposed pair stencil of pass C, all components in one launch
template <typename T>
__global__ __launch_bounds__(kBlock) void c2s_bwd_kernel(VelGrid g, ScalarBc sb, CComp3a<T> gout, T* __restrict__ gs, T sc0, T sc1, T sc2) {
    const int b = blockIdx.y;
    const int total = (int)g.cells;
    const T scale[3] = {sc0, sc1, sc2};
    for (int f = blockIdx.x * kBlock + threadIdx.x; f < total; f += gridDim.x * kBlock) {
        int idx[3];
        unravel(f, g.n[1], g.n[2], idx);
        T acc = T(0);
#pragma unroll
        for (int ca = 0; ca < 3; ++ca) {
            if (ca < g.ax0 || scale[ca] == T(0)) continue;
            const int s1 = g.cn[ca][1], s2 = g.cn[ca][2];
            const int stride[3] = {s1 * s2, s2, 1};
            const AxisSources A = transposed_sources(idx[ca], g.n[ca], sb.bc[ca][0], sb.bc[ca][1], g.off[ca] - 1, g.cn[ca][ca]);
            int rest = 0;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax)
                if (ax != ca) rest += idx[ax] * stride[ax];
            const T* __restrict__ D = gout.p[ca] + (long long)b * g.ccells[ca];
            const T v0 = D[A.on[0] ? rest + A.q[0] * stride[ca] : 0], v1 = D[A.on[1] ? rest + A.q[1] * stride[ca] : 0];
            T part = (A.on[0] ? v0 : T(0)) + (A.on[1] ? v1 : T(0));
            if (wave_any(A.on[2] || A.on[3] || A.on[4] || A.on[5])) {
                T gval[4];
#pragma unroll
                for (int k = 2; k < 6; ++k) gval[k - 2] = D[A.on[k] ? rest + A.q[k] * stride[ca] : 0];
#pragma unroll
                for (int k = 2; k < 6; ++k) part += A.on[k] ? gval[k - 2] : T(0);
            }
            acc += T(0.5) * scale[ca] * part;
        }
        gs[(long long)b * total + f] += acc;
    }
}

// min / max over the taps with the offset of the extremal tap (-1: a constant boundary value, no gradient)
template <typename T, int DIM>
__device__ __forceinline__ void gather_minmax_arg(const T* __restrict__ F, const AxisPair<T> (&ax)[3], T& lo, T& hi, int& off_lo, int& off_hi) {
#pragma unroll
    for (int corner = 0; corner < (1 << DIM); ++corner) {
        const int b0 = DIM == 3 ? (corner & 1) : 0;
        const int b1 = DIM == 3 ? ((corner >> 1) & 1) : (corner & 1);
        const int b2 = DIM == 3 ? ((corner >> 2) & 1) : ((corner >> 1) & 1);
        T val;
        int off = -1;
        if (ax[2].cst[b2]) val = ax[2].cv[b2];
        else if (ax[1].cst[b1]) val = ax[1].cv[b1];
        else if (DIM == 3 && ax[0].cst[b0]) val = ax[0].cv[b0];
        else {
            off = (DIM == 3 ? ax[0].off[b0] : 0) + ax[1].off[b1] + ax[2].off[b2];
            val = F[off];
        }
        if (corner == 0 || val < lo) { lo = val; off_lo = off; }
        if (corner == 0 || val > hi) { hi = val; off_hi = off; }
    }
}

// Adjoint of the MacCormack correction pass (advect.hip MODE 1):  out = clip(fwd + ch (s - fwd(x + dt u)), lo, hi).
// Accumulates into g_fwd (gradient of the semi-Lagrangian intermediate; a following semi-Lagrangian backward pass takes it
// to the field and the velocity), g_field (own value / the extremal tap when clamped) and g_velocity (forward lookup).
// STAG: staggered component CA (limiter lookup in the cell frame like the forward pass); else centred scalar.
template <typename T, int DIM, int CA, bool STAG>
__device__ __forceinline__ void mac_cormack_bwd_body(const VelGrid& g, const ScalarBc& sb, const CComp3a<T>& field, const T* __restrict__ sfield,
                                                     const CComp3a<T>& vel, const T* __restrict__ fwd, const T* __restrict__ gout,
                                                     T* __restrict__ gfwd, T* __restrict__ gfield, const TraceOut<T>& out, int want_gvel,
                                                     T dt, T ch) {
    constexpr int A0 = 3 - DIM;
    constexpr int ca = CA;
    const int b = blockIdx.y;
    const int total = STAG ? (int)g.ccells[ca] : (int)g.cells;
    const int n[3] = {STAG ? g.cn[ca][0] : g.n[0], STAG ? g.cn[ca][1] : g.n[1], STAG ? g.cn[ca][2] : g.n[2]};
    const T* __restrict__ F = (STAG ? field.p[ca] : sfield) + (long long)b * total;
    const T* __restrict__ W = fwd + (long long)b * total;
    T* __restrict__ GW = gfwd + (long long)b * total;
    T* __restrict__ GF = gfield + (long long)b * total;
    int bc[3][2];
    T cv[3][2];
    if (STAG) comp_rule<T>(g, ca, bc, cv); else scalar_rule<T>(sb, bc, cv);
    for (int f = blockIdx.x * kBlock + threadIdx.x; f < total; f += gridDim.x * kBlock) {
        const T go = gout[(long long)b * total + f];
        int idx[3];
        unravel(f, n[1], n[2], idx);
        T u[3];
        if (STAG) face_velocity<T, DIM, CA>(g, vel, b, idx, f, u); else center_velocity<T, DIM>(g, vel, b, idx, u);
        T cb_[3] = {T(0), T(0), T(0)}, cf_[3] = {T(0), T(0), T(0)};
#pragma unroll
        for (int a = A0; a < 3; ++a) {
            const T sft = u[a] * (dt * (T)g.rdx[a]);
            cb_[a] = (T)idx[a] - sft;
            cf_[a] = (T)idx[a] + sft;
        }
        AxisPair<T> ax[3];
        T fr[3];
        lookup_pairs<T, DIM>(cf_, n, bc, cv, ax, fr);
        const T bwd = gather_multilinear<T, DIM>(W, ax, fr);
        const T nv = mc_correct(W[f], ch, F[f], bwd);
        AxisPair<T> axl[3];
        T frl[3];
        if (STAG) cb_[ca] += (T)g.off[ca] - T(0.5);
        lookup_pairs<T, DIM>(cb_, n, bc, cv, axl, frl);
        T lo, hi;
        int off_lo, off_hi;
        gather_minmax_arg<T, DIM>(F, axl, lo, hi, off_lo, off_hi);
        // the forward lookup's scatter (into g_fwd) and the velocity means go through the gather passes: record (x + dt u, gb), gb d(out)/d(x*)
        GatherSlot<T> rec;
        rec.c0 = cf_[0]; rec.c1 = cf_[1]; rec.c2 = cf_[2];
        rec.g = T(0);
        T du[3] = {T(0), T(0), T(0)};
        if (nv < lo) {
            if (off_lo >= 0) atomicAdd(GF + off_lo, go);
        } else if (nv > hi) {
            if (off_hi >= 0) atomicAdd(GF + off_hi, go);
        } else {
            atomicAdd(GW + f, go);
            atomicAdd(GF + f, ch * go);
            const T gb = -ch * go;
            bool near = true;
#pragma unroll
            for (int a = A0; a < 3; ++a) near = near && fabs(cf_[a] - (T)idx[a]) < T(1);
            T dfr[3];
            gather_adjoint<T, DIM>(W, near ? nullptr : GW, ax, fr, gb, dfr);
            rec.g = near ? gb : T(0);
#pragma unroll
            for (int a = A0; a < 3; ++a) du[a] = gb * dfr[a] * (dt * (T)g.rdx[a]);   // cf_a = idx_a + dt u_a / dx_a
        }
        const long long o = (long long)b * total + f;
        out.rec[o] = rec;
        if (want_gvel) {
#pragma unroll
            for (int a = A0; a < 3; ++a) out.du[a][o] = du[a];
        }
    }
}

template <typename T, int DIM, int CA, bool STAG>
__global__ __launch_bounds__(kBlock) void mac_cormack_bwd_kernel(VelGrid g, ScalarBc sb, CComp3a<T> field, const T* __restrict__ sfield,
                                                                 CComp3a<T> vel, const T* __restrict__ fwd, const T* __restrict__ gout,
                                                                 T* __restrict__ gfwd, T* __restrict__ gfield, TraceOut<T> out, int want_gvel,
                                                                 T dt, T ch) {
    mac_cormack_bwd_body<T, DIM, CA, STAG>(g, sb, field, sfield, vel, fwd, gout, gfwd, gfield, out, want_gvel, dt, ch);
}

// r6: the correction pass's adjoint for ALL staggered components in one launch (blockIdx.z = component - first axis; per-component record arrays)
template <typename T>
struct McAll {
    const T* fwd[3];
    const T* gout[3];
    T* gfwd[3];
    T* gfield[3];
    TraceOut<T> out[3];
};

template <typename T, int DIM>
__global__ __launch_bounds__(kBlock) void mac_cormack_bwd_all_kernel(VelGrid g, CComp3a<T> field, CComp3a<T> vel, McAll<T> a, int want_gvel, T dt, T ch) {
    const ScalarBc none{};
    const int ca = (3 - DIM) + (int)blockIdx.z;
    if (ca == 0) mac_cormack_bwd_body<T, DIM, 0, true>(g, none, field, (const T*)nullptr, vel, a.fwd[0], a.gout[0], a.gfwd[0], a.gfield[0], a.out[0], want_gvel, dt, ch);
    else if (ca == 1) mac_cormack_bwd_body<T, DIM, 1, true>(g, none, field, (const T*)nullptr, vel, a.fwd[1], a.gout[1], a.gfwd[1], a.gfield[1], a.out[1], want_gvel, dt, ch);
    else mac_cormack_bwd_body<T, DIM, 2, true>(g, none, field, (const T*)nullptr, vel, a.fwd[2], a.gout[2], a.gfwd[2], a.gfield[2], a.out[2], want_gvel, dt, ch);
}

// r6: the staggered adjoints run ALL components per launch (three launches per call instead of nine); PHIHIP_ADJOINT_ALL=0 = one launch per component (A/B, tests)
static inline bool adjoint_all_components() {
    static const bool on = [] { const char* e = getenv("PHIHIP_ADJOINT_ALL"); return !(e && e[0] == '0'); }();
    return on;
}

static inline int bwd_blocks(long long total) {
    const long long nb = (total + kBlock - 1) / kBlock;
    return (int)(nb < 65536 ? nb : 65536);
}

// scratch of the gather-form adjoints: [cx0 | cx1 | cx2 | gw] of one sample array at a time + the du arrays of every source (9 staggered, 3 centred)
template <typename T>
static int adjoint_scratch(phihip_ctx* ctx, size_t max_samples, int batch, int n_du, T* trace[4], T* du[9], int n_rec = 1, GatherSlot<T>** recs = nullptr) {
    const size_t slot = (((size_t)batch * max_samples * sizeof(T) + 255) / 256) * 256;
    PHIHIP_TRY(ensure_buffer(ctx->ws_adj_g, slot * (4 * n_rec + n_du)));
    for (int k = 0; k < 4; ++k) trace[k] = (T*)((char*)ctx->ws_adj_g.ptr + slot * k);
    for (int k = 0; k < n_rec && recs; ++k) recs[k] = (GatherSlot<T>*)((char*)ctx->ws_adj_g.ptr + slot * 4 * k);      // (the four words of a record array are contiguous)
    for (int k = 0; k < 9; ++k) du[k] = k < n_du ? (T*)((char*)ctx->ws_adj_g.ptr + slot * (4 * n_rec + k)) : nullptr;
    return PHIHIP_OK;
}

template <typename T, int DIM>
static void launch_field_gather(const int n[3], const int bc[3][2], int batch, const TraceOut<T>& tr, T* gfield, hipStream_t s) {
    ScalarBc rule;
    memset(&rule, 0, sizeof(rule));
    for (int a = 0; a < 3; ++a)
        for (int side = 0; side < 2; ++side) rule.bc[a][side] = bc[a][side];
    const int nb2 = ceil_div(n[2], kGatherT2), nb1 = ceil_div(n[1], kGatherT1), nb0 = DIM == 3 ? ceil_div(n[0], gather_t0<T, DIM>()) : 1;
    hipLaunchKernelGGL((advect_bwd_field_gather_kernel<T, DIM>), dim3((unsigned)nb0 * nb1 * nb2, batch), dim3(kBlock), 0, s, n[0], n[1], n[2], rule, tr, gfield,
                       nb1, nb2);
}

// pass C for every velocity component: du[3 ca + cb] = source component ca (staggered samples) resp. du[cb] (cell samples)
template <typename T, int DIM>
static void launch_field_gather_all(const GridView& v, GatherSlot<T>* const recs[3], void* const gfield[3], hipStream_t s) {
    GatherAll<T> a;
    memset(&a, 0, sizeof(a));
    int blocks = 0;
    for (int ca = v.ax0; ca < 3; ++ca) {
        const int* n = v.cn[ca];
        for (int ax = 0; ax < 3; ++ax) { a.n[ca][ax] = n[ax]; a.bc[ca][ax][0] = v.bc[ax][0]; a.bc[ca][ax][1] = v.bc[ax][1]; }
        a.nb2[ca] = ceil_div(n[2], kGatherT2); a.nb1[ca] = ceil_div(n[1], kGatherT1);
        a.blocks[ca] = (DIM == 3 ? ceil_div(n[0], gather_t0<T, DIM>()) : 1) * a.nb1[ca] * a.nb2[ca];
        a.rec[ca] = recs[ca - v.ax0]; a.gfield[ca] = gfield ? (T*)gfield[ca] : nullptr;
        blocks = a.blocks[ca] > blocks ? a.blocks[ca] : blocks;
    }
    hipLaunchKernelGGL((advect_bwd_field_gather_all_kernel<T, DIM>), dim3((unsigned)blocks, v.batch, DIM), dim3(kBlock), 0, s, a);
}

template <typename T, int DIM, bool STAG>
static void launch_velocity_gathers(const GridView& v, const VelGrid& g, T* const du[9], void* const gv[3], hipStream_t s) {
    if (adjoint_all_components()) {
        DuAll<T> a;
        memset(&a, 0, sizeof(a));
        long long most = 0;
        for (int cb = v.ax0; cb < 3; ++cb) {
            a.in[cb] = DuIn<T>{{STAG ? du[0 + cb] : du[cb], STAG ? du[3 + cb] : nullptr, STAG ? du[6 + cb] : nullptr}};
            a.gvel[cb] = (T*)gv[cb];
            most = v.ccells[cb] > most ? v.ccells[cb] : most;
        }
        hipLaunchKernelGGL((advect_bwd_velocity_gather_all_kernel<T, DIM, STAG>), dim3(bwd_blocks(most), v.batch, DIM), dim3(kBlock), 0, s, g, a);
        return;
    }
    for (int cb = v.ax0; cb < 3; ++cb) {
        DuIn<T> in{{STAG ? du[0 + cb] : du[cb], STAG ? du[3 + cb] : nullptr, STAG ? du[6 + cb] : nullptr}};
        const dim3 grid(bwd_blocks(v.ccells[cb]), v.batch);
        if (cb == 0) hipLaunchKernelGGL((advect_bwd_velocity_gather_kernel<T, DIM, 0, STAG>), grid, dim3(kBlock), 0, s, g, in, (T*)gv[0]);
        else if (cb == 1) hipLaunchKernelGGL((advect_bwd_velocity_gather_kernel<T, DIM, 1, STAG>), grid, dim3(kBlock), 0, s, g, in, (T*)gv[1]);
        else hipLaunchKernelGGL((advect_bwd_velocity_gather_kernel<T, DIM, 2, STAG>), grid, dim3(kBlock), 0, s, g, in, (T*)gv[2]);
    }
}

template <typename T, int DIM>
static int advect_staggered_bwd_t(phihip_ctx* ctx, const GridView& v, const VelGrid& g, const void* const f[3], const void* const vel[3],
                                  const void* const gout[3], void* const gf[3], void* const gv[3], double dt, hipStream_t s) {
    size_t max_samples = 0;
    for (int ca = v.ax0; ca < 3; ++ca) max_samples = (size_t)v.ccells[ca] > max_samples ? (size_t)v.ccells[ca] : max_samples;
    T *trace[4], *du[9];
    GatherSlot<T>* recs[3] = {nullptr, nullptr, nullptr};
    const bool all = adjoint_all_components();
    PHIHIP_TRY(adjoint_scratch<T>(ctx, max_samples, v.batch, gv ? 9 : 0, trace, du, all ? DIM : 1, recs));
    CComp3a<T> vv{{(const T*)vel[0], (const T*)vel[1], (const T*)vel[2]}};
    ScalarBc none;
    memset(&none, 0, sizeof(none));
    if (all) {          // r6: pass A for all components, pass B for all components, pass C for all components -- three launches instead of nine
        TraceAll<T> a;
        memset(&a, 0, sizeof(a));
        long long most = 0;
        for (int ca = v.ax0; ca < 3; ++ca) {
            a.f[ca] = (const T*)f[ca]; a.gout[ca] = (const T*)gout[ca]; a.gfield[ca] = gf ? (T*)gf[ca] : nullptr;
            a.out[ca] = TraceOut<T>{recs[ca - v.ax0], {du[3 * ca], du[3 * ca + 1], du[3 * ca + 2]}};
            most = v.ccells[ca] > most ? v.ccells[ca] : most;
        }
        hipLaunchKernelGGL((advect_bwd_trace_all_kernel<T, DIM>), dim3(bwd_blocks(most), v.batch, DIM), dim3(kBlock), 0, s, g, vv, a, gv ? 1 : 0, (T)dt);
        if (gf) launch_field_gather_all<T, DIM>(v, recs, gf, s);
        if (gv) launch_velocity_gathers<T, DIM, true>(v, g, du, gv, s);
        return PHIHIP_OK;
    }
    for (int ca = v.ax0; ca < 3; ++ca) {
        TraceOut<T> tr{(GatherSlot<T>*)trace[0], {du[3 * ca], du[3 * ca + 1], du[3 * ca + 2]}};      // (the four trace slots are contiguous)
        const dim3 grid(bwd_blocks(v.ccells[ca]), v.batch);
        T* gfield = gf ? (T*)gf[ca] : nullptr;
#define PHIHIP_TRACE(CA) hipLaunchKernelGGL((advect_bwd_trace_kernel<T, DIM, CA, true>), grid, dim3(kBlock), 0, s, g, none, (const T*)f[ca], vv, (const T*)gout[ca], gfield, tr, gv ? 1 : 0, (T)dt)
        if (ca == 0) PHIHIP_TRACE(0); else if (ca == 1) PHIHIP_TRACE(1); else PHIHIP_TRACE(2);
#undef PHIHIP_TRACE
        if (gfield) launch_field_gather<T, DIM>(v.cn[ca], v.bc, v.batch, tr, gfield, s);
    }
    if (gv) launch_velocity_gathers<T, DIM, true>(v, g, du, gv, s);
    return PHIHIP_OK;
}

int run_advect_staggered_bwd(phihip_ctx* ctx, const GridView& v, const void* const f[3], const void* const vel[3], const void* const gout[3],
                             void* const gf[3], void* const gv[3], double dt, hipStream_t s) {
    const VelGrid g = make_velgrid(v);
    LaunchScope ls(ctx, PHIHIP_K_ADVECT, s);
    if (v.dtype == PHIHIP_F64) {
        if (v.rank == 3) PHIHIP_TRY((advect_staggered_bwd_t<double, 3>(ctx, v, g, f, vel, gout, gf, gv, dt, s)));
        else PHIHIP_TRY((advect_staggered_bwd_t<double, 2>(ctx, v, g, f, vel, gout, gf, gv, dt, s)));
    } else {
        if (v.rank == 3) PHIHIP_TRY((advect_staggered_bwd_t<float, 3>(ctx, v, g, f, vel, gout, gf, gv, dt, s)));
        else PHIHIP_TRY((advect_staggered_bwd_t<float, 2>(ctx, v, g, f, vel, gout, gf, gv, dt, s)));
    }
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

template <typename T, int DIM>
static int advect_centered_bwd_t(phihip_ctx* ctx, const GridView& v, const VelGrid& g, const ScalarBc& sb, const void* sfield, const void* const vel[3],
                                 const void* gout, void* gs, void* const gv[3], double dt, hipStream_t s) {
    T *trace[4], *du[9];
    PHIHIP_TRY(adjoint_scratch<T>(ctx, (size_t)v.cells, v.batch, gv ? 3 : 0, trace, du));
    CComp3a<T> vv{{(const T*)vel[0], (const T*)vel[1], (const T*)vel[2]}};
    TraceOut<T> tr{(GatherSlot<T>*)trace[0], {du[0], du[1], du[2]}};
    hipLaunchKernelGGL((advect_bwd_trace_kernel<T, DIM, 2, false>), dim3(bwd_blocks(v.cells), v.batch), dim3(kBlock), 0, s, g, sb, (const T*)sfield, vv,
                       (const T*)gout, (T*)gs, tr, gv ? 1 : 0, (T)dt);
    if (gs) launch_field_gather<T, DIM>(v.n, sb.bc, v.batch, tr, (T*)gs, s);
    if (gv) launch_velocity_gathers<T, DIM, false>(v, g, du, gv, s);
    return PHIHIP_OK;
}

int run_advect_centered_bwd(phihip_ctx* ctx, const GridView& v, const void* sfield, const int32_t s_bc[3][2], const double s_val[3][2],
                            const void* const vel[3], const void* gout, void* gs, void* const gv[3], double dt, hipStream_t s) {
    const VelGrid g = make_velgrid(v);
    const ScalarBc sb = make_scalar_bc(v, s_bc, s_val);
    LaunchScope ls(ctx, PHIHIP_K_ADVECT, s);
    if (v.dtype == PHIHIP_F64) {
        if (v.rank == 3) PHIHIP_TRY((advect_centered_bwd_t<double, 3>(ctx, v, g, sb, sfield, vel, gout, gs, gv, dt, s)));
        else PHIHIP_TRY((advect_centered_bwd_t<double, 2>(ctx, v, g, sb, sfield, vel, gout, gs, gv, dt, s)));
    } else {
        if (v.rank == 3) PHIHIP_TRY((advect_centered_bwd_t<float, 3>(ctx, v, g, sb, sfield, vel, gout, gs, gv, dt, s)));
        else PHIHIP_TRY((advect_centered_bwd_t<float, 2>(ctx, v, g, sb, sfield, vel, gout, gs, gv, dt, s)));
    }
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

int run_centered_to_staggered_bwd(phihip_ctx* ctx, const GridView& v, const int32_t s_bc[3][2], const double vector[3],
                                  const void* const gout[3], void* gs, hipStream_t s) {
    const VelGrid g = make_velgrid(v);
    const ScalarBc sb = make_scalar_bc(v, s_bc, nullptr);
    LaunchScope ls(ctx, PHIHIP_K_OTHER, s);
    const dim3 grid(bwd_blocks(v.cells), v.batch);
    if (v.dtype == PHIHIP_F64)
        hipLaunchKernelGGL(c2s_bwd_kernel<double>, grid, dim3(kBlock), 0, s, g, sb, (CComp3a<double>{{(const double*)gout[0], (const double*)gout[1], (const double*)gout[2]}}),
                           (double*)gs, vector[0], vector[1], vector[2]);
    else
        hipLaunchKernelGGL(c2s_bwd_kernel<float>, grid, dim3(kBlock), 0, s, g, sb, (CComp3a<float>{{(const float*)gout[0], (const float*)gout[1], (const float*)gout[2]}}),
                           (float*)gs, (float)vector[0], (float)vector[1], (float)vector[2]);
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

int run_advect_staggered(phihip_ctx*, const GridView&, const void* const f[3], const void* const v[3], void* const out[3], double dt, hipStream_t);
int run_advect_centered(phihip_ctx*, const GridView&, const void* s, const int32_t s_bc[3][2], const double s_val[3][2], const void* const v[3],
                        void* out, double dt, hipStream_t);

template <typename T, int DIM>
static int launch_mc_staggered_bwd(phihip_ctx* ctx, const GridView& v, const VelGrid& g, const void* const f[3], const void* const vel[3], void* const fwd[3],
                                   const void* const gout[3], void* const gfwd[3], void* const gf[3], void* const gv[3], double dt, double ch,
                                   hipStream_t s) {
    CComp3a<T> ff{{(const T*)f[0], (const T*)f[1], (const T*)f[2]}};
    CComp3a<T> vv{{(const T*)vel[0], (const T*)vel[1], (const T*)vel[2]}};
    ScalarBc sb;
    memset(&sb, 0, sizeof(sb));
    const int want = gv ? 1 : 0;
    size_t max_samples = 0;
    for (int ca = v.ax0; ca < 3; ++ca) max_samples = (size_t)v.ccells[ca] > max_samples ? (size_t)v.ccells[ca] : max_samples;
    T *trace[4], *du[9];
    GatherSlot<T>* recs[3] = {nullptr, nullptr, nullptr};
    const bool all = adjoint_all_components();
    PHIHIP_TRY(adjoint_scratch<T>(ctx, max_samples, v.batch, gv ? 9 : 0, trace, du, all ? DIM : 1, recs));
    if (all) {
        McAll<T> a;
        memset(&a, 0, sizeof(a));
        long long most = 0;
        for (int ca = v.ax0; ca < 3; ++ca) {
            a.fwd[ca] = (const T*)fwd[ca]; a.gout[ca] = (const T*)gout[ca]; a.gfwd[ca] = (T*)gfwd[ca]; a.gfield[ca] = (T*)gf[ca];
            a.out[ca] = TraceOut<T>{recs[ca - v.ax0], {du[3 * ca], du[3 * ca + 1], du[3 * ca + 2]}};
            most = v.ccells[ca] > most ? v.ccells[ca] : most;
        }
        hipLaunchKernelGGL((mac_cormack_bwd_all_kernel<T, DIM>), dim3(bwd_blocks(most), v.batch, DIM), dim3(kBlock), 0, s, g, ff, vv, a, want, (T)dt, (T)ch);
        launch_field_gather_all<T, DIM>(v, recs, gfwd, s);                                   // the forward lookup's taps -> g_fwd
        if (gv) launch_velocity_gathers<T, DIM, true>(v, g, du, gv, s);
        return PHIHIP_OK;
    }
    for (int ca = v.ax0; ca < 3; ++ca) {
        TraceOut<T> tr{(GatherSlot<T>*)trace[0], {du[3 * ca], du[3 * ca + 1], du[3 * ca + 2]}};
        const dim3 grid(bwd_blocks(v.ccells[ca]), v.batch);
#define PHIHIP_MC(CA) hipLaunchKernelGGL((mac_cormack_bwd_kernel<T, DIM, CA, true>), grid, dim3(kBlock), 0, s, g, sb, ff, (const T*)nullptr, vv, (const T*)fwd[ca], \
                                         (const T*)gout[ca], (T*)gfwd[ca], (T*)gf[ca], tr, want, (T)dt, (T)ch)
        if (ca == 0) PHIHIP_MC(0); else if (ca == 1) PHIHIP_MC(1); else PHIHIP_MC(2);
#undef PHIHIP_MC
        launch_field_gather<T, DIM>(v.cn[ca], v.bc, v.batch, tr, (T*)gfwd[ca], s);          // the forward lookup's taps -> g_fwd
    }
    if (gv) launch_velocity_gathers<T, DIM, true>(v, g, du, gv, s);
    return PHIHIP_OK;
}

// scratch layout for the MacCormack adjoints: [fwd | g_fwd] per component, 256-byte aligned
static int mc_scratch(phihip_ctx* ctx, const GridView& v, bool staggered, void* fwd[3], void* gfwd[3]) {
    const size_t esize = v.dtype == PHIHIP_F64 ? 8 : 4;
    size_t offs[3] = {0, 0, 0}, total = 0;
    for (int ca = staggered ? v.ax0 : 2; ca < 3; ++ca) {
        offs[ca] = total;
        total += (((size_t)v.batch * (staggered ? v.ccells[ca] : v.cells) * esize + 255) / 256) * 256;
    }
    PHIHIP_TRY(ensure_buffer(ctx->ws_adv, 2 * total));
    fwd[0] = fwd[1] = fwd[2] = gfwd[0] = gfwd[1] = gfwd[2] = nullptr;
    for (int ca = staggered ? v.ax0 : 2; ca < 3; ++ca) {
        fwd[ca] = (char*)ctx->ws_adv.ptr + offs[ca];
        gfwd[ca] = (char*)ctx->ws_adv.ptr + total + offs[ca];
    }
    return PHIHIP_OK;
}

int run_mac_cormack_staggered_bwd(phihip_ctx* ctx, const GridView& v, const void* const f[3], const void* const vel[3],
                                  const void* const gout[3], void* const gf[3], void* const gv[3], double dt, double strength, hipStream_t s) {
    const VelGrid g = make_velgrid(v);
    void *fwd[3], *gfwd[3];
    PHIHIP_TRY(mc_scratch(ctx, v, true, fwd, gfwd));
    PHIHIP_TRY(run_advect_staggered(ctx, v, f, vel, fwd, dt, s));                                // recompute the forward intermediate
    const size_t esize = v.dtype == PHIHIP_F64 ? 8 : 4;
    for (int ca = v.ax0; ca < 3; ++ca) PHIHIP_CHECK_HIP(hipMemsetAsync(gfwd[ca], 0, (size_t)v.batch * v.ccells[ca] * esize, s));
    {
        LaunchScope ls(ctx, PHIHIP_K_ADVECT, s);
        if (v.dtype == PHIHIP_F64) {
            if (v.rank == 3) PHIHIP_TRY((launch_mc_staggered_bwd<double, 3>(ctx, v, g, f, vel, fwd, gout, gfwd, gf, gv, dt, 0.5 * strength, s)));
            else PHIHIP_TRY((launch_mc_staggered_bwd<double, 2>(ctx, v, g, f, vel, fwd, gout, gfwd, gf, gv, dt, 0.5 * strength, s)));
        } else {
            if (v.rank == 3) PHIHIP_TRY((launch_mc_staggered_bwd<float, 3>(ctx, v, g, f, vel, fwd, gout, gfwd, gf, gv, dt, 0.5 * strength, s)));
            else PHIHIP_TRY((launch_mc_staggered_bwd<float, 2>(ctx, v, g, f, vel, fwd, gout, gfwd, gf, gv, dt, 0.5 * strength, s)));
        }
    }
    const void* cg[3] = {gfwd[0], gfwd[1], gfwd[2]};
    PHIHIP_TRY(run_advect_staggered_bwd(ctx, v, f, vel, cg, gf, gv, dt, s));                     // g_fwd -> field, velocity
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

template <typename T, int DIM>
static int launch_mc_centered_bwd(phihip_ctx* ctx, const GridView& v, const VelGrid& g, const ScalarBc& sb, const void* sfield, const void* const vel[3],
                                  const void* fwd, const void* gout, void* gfwd, void* gs, void* const gv[3], double dt, double ch,
                                  hipStream_t s) {
    CComp3a<T> none{{nullptr, nullptr, nullptr}};
    CComp3a<T> vv{{(const T*)vel[0], (const T*)vel[1], (const T*)vel[2]}};
    T *trace[4], *du[9];
    PHIHIP_TRY(adjoint_scratch<T>(ctx, (size_t)v.cells, v.batch, gv ? 3 : 0, trace, du));
    TraceOut<T> tr{(GatherSlot<T>*)trace[0], {du[0], du[1], du[2]}};
    hipLaunchKernelGGL((mac_cormack_bwd_kernel<T, DIM, 2, false>), dim3(bwd_blocks(v.cells), v.batch), dim3(kBlock), 0, s, g, sb, none,
                       (const T*)sfield, vv, (const T*)fwd, (const T*)gout, (T*)gfwd, (T*)gs, tr, gv ? 1 : 0, (T)dt, (T)ch);
    launch_field_gather<T, DIM>(v.n, sb.bc, v.batch, tr, (T*)gfwd, s);
    if (gv) launch_velocity_gathers<T, DIM, false>(v, g, du, gv, s);
    return PHIHIP_OK;
}

int run_mac_cormack_centered_bwd(phihip_ctx* ctx, const GridView& v, const void* sfield, const int32_t s_bc[3][2], const double s_val[3][2],
                                 const void* const vel[3], const void* gout, void* gs, void* const gv[3], double dt, double strength,
                                 hipStream_t s) {
    const VelGrid g = make_velgrid(v);
    const ScalarBc sb = make_scalar_bc(v, s_bc, s_val);
    void *fwd[3], *gfwd[3];
    PHIHIP_TRY(mc_scratch(ctx, v, false, fwd, gfwd));
    PHIHIP_TRY(run_advect_centered(ctx, v, sfield, s_bc, s_val, vel, fwd[2], dt, s));
    const size_t esize = v.dtype == PHIHIP_F64 ? 8 : 4;
    PHIHIP_CHECK_HIP(hipMemsetAsync(gfwd[2], 0, (size_t)v.batch * v.cells * esize, s));
    {
        LaunchScope ls(ctx, PHIHIP_K_ADVECT, s);
        if (v.dtype == PHIHIP_F64) {
            if (v.rank == 3) PHIHIP_TRY((launch_mc_centered_bwd<double, 3>(ctx, v, g, sb, sfield, vel, fwd[2], gout, gfwd[2], gs, gv, dt, 0.5 * strength, s)));
            else PHIHIP_TRY((launch_mc_centered_bwd<double, 2>(ctx, v, g, sb, sfield, vel, fwd[2], gout, gfwd[2], gs, gv, dt, 0.5 * strength, s)));
        } else {
            if (v.rank == 3) PHIHIP_TRY((launch_mc_centered_bwd<float, 3>(ctx, v, g, sb, sfield, vel, fwd[2], gout, gfwd[2], gs, gv, dt, 0.5 * strength, s)));
            else PHIHIP_TRY((launch_mc_centered_bwd<float, 2>(ctx, v, g, sb, sfield, vel, fwd[2], gout, gfwd[2], gs, gv, dt, 0.5 * strength, s)));
        }
    }
    PHIHIP_TRY(run_advect_centered_bwd(ctx, v, sfield, s_bc, s_val, vel, gfwd[2], gs, gv, dt, s));
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// make_incompressible backward. Forward (fluid.py:138-161):  b = a (D v' + c) ;  rhs = P b ;  p = A^-1 rhs ;  v_out = v' - H G p
// with a = active, H = hard_bcs, P = balance, c = boundary-value terms. With upstream gradients g_v (of v_out), g_p (of p):
//     pbar = g_p + D0 (H g_v)                [G^T = -D0 : divergence with homogeneous boundary values]
//     lam  = A^-1 P (a pbar)                 [A symmetric; P projects onto its range when the system is singular]
//     g_v' = g_v - G0 P (a lam)              [D^T = -G0 : gradient at the stored faces, no H]
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int CA>
__global__ __launch_bounds__(kBlock) void mask_faces_kernel(VelGrid g, const T* __restrict__ in, T* __restrict__ out, const uint8_t* flags,
                                                            int flags_per_batch) {
    constexpr int ca = CA;
    const int b = blockIdx.y;
    const int total = (int)g.ccells[ca];
    const int c1 = g.cn[ca][1], c2 = g.cn[ca][2];
    const int n = g.n[ca];
    const int pstride = ca == 0 ? g.n[1] * g.n[2] : (ca == 1 ? g.n[2] : 1);
    const uint8_t* F = flags + (flags_per_batch ? (long long)b * g.cells : 0);
    for (int f = blockIdx.x * kBlock + threadIdx.x; f < total; f += gridDim.x * kBlock) {
        int idx[3];
        unravel(f, c1, c2, idx);
        const int phys = idx[ca] + g.off[ca];
        int l = phys - 1, r = phys;
        const bool l_in = l >= 0, r_in = r < n;
        if (!l_in) l = g.bc[ca][0] == PHIHIP_BC_PERIODIC ? l + n : 0;
        if (!r_in) r = g.bc[ca][1] == PHIHIP_BC_PERIODIC ? r - n : n - 1;
        const int rest = (idx[0] * g.n[1] + idx[1]) * g.n[2] + idx[2] - idx[ca] * pstride;
        T h = T(1);
        if (r_in || g.bc[ca][1] == PHIHIP_BC_PERIODIC) h = (F[rest + r * pstride] >> (2 * ca)) & 1u ? T(1) : T(0);
        else if (l_in) h = (F[rest + l * pstride] >> (2 * ca + 1)) & 1u ? T(1) : T(0);
        out[(long long)b * total + f] = h * in[(long long)b * total + f];
    }
}

// x = active * (x + add)
template <typename T>
__global__ __launch_bounds__(kBlock) void mask_cells_kernel(T* __restrict__ x, const T* __restrict__ add, const uint8_t* flags, int flags_per_batch,
                                                            long long cells) {
    const int b = blockIdx.y;
    for (long long c = (long long)blockIdx.x * kBlock + threadIdx.x; c < cells; c += (long long)gridDim.x * kBlock) {
        T val = x[(long long)b * cells + c];
        if (add) val += add[(long long)b * cells + c];
        if (flags && !(flags[(flags_per_batch ? (long long)b * cells : 0) + c] & 64u)) val = T(0);
        x[(long long)b * cells + c] = val;
    }
}

int run_balance(phihip_ctx* ctx, const GridView& v, const uint8_t* flags, int mask_batch, void* x, hipStream_t s);

template <typename T>
static int project_bwd_t(phihip_ctx* ctx, const GridView& v0, const uint8_t* flags, int mask_batch, int balance, void* const gv[3],
                         const void* gp, const phihip_solve* solve, phihip_solve_info* info, hipStream_t s) {
    GridView v = v0;   // homogeneous boundary values: the constants do not depend on the inputs
    memset(v.bcv, 0, sizeof(v.bcv));
    const VelGrid g = make_velgrid(v);
    const int fpb = mask_batch > 1 ? 1 : 0;
    const size_t cell_bytes = (size_t)v.batch * v.cells * sizeof(T);
    PHIHIP_TRY(ensure_buffer(ctx->ws_adj_q, cell_bytes));
    PHIHIP_TRY(ensure_buffer(ctx->ws_adj_l, cell_bytes));
    T* q = (T*)ctx->ws_adj_q.ptr;
    T* lam = (T*)ctx->ws_adj_l.ptr;
    const void* src[3] = {gv[0], gv[1], gv[2]};
    if (flags) {   // t = H g_v
        size_t offs[3] = {0, 0, 0}, total = 0;
        for (int ca = v.ax0; ca < 3; ++ca) {
            offs[ca] = total;
            total += (((size_t)v.batch * v.ccells[ca] * sizeof(T) + 255) / 256) * 256;
        }
        PHIHIP_TRY(ensure_buffer(ctx->ws_adv, total));
        LaunchScope ls(ctx, PHIHIP_K_OTHER, s);
        for (int ca = v.ax0; ca < 3; ++ca) {
            T* t = (T*)((char*)ctx->ws_adv.ptr + offs[ca]);
            const dim3 grid(bwd_blocks(v.ccells[ca]), v.batch);
            if (ca == 0) hipLaunchKernelGGL((mask_faces_kernel<T, 0>), grid, dim3(kBlock), 0, s, g, (const T*)gv[0], t, flags, fpb);
            if (ca == 1) hipLaunchKernelGGL((mask_faces_kernel<T, 1>), grid, dim3(kBlock), 0, s, g, (const T*)gv[1], t, flags, fpb);
            if (ca == 2) hipLaunchKernelGGL((mask_faces_kernel<T, 2>), grid, dim3(kBlock), 0, s, g, (const T*)gv[2], t, flags, fpb);
            src[ca] = t;
        }
    }
    PHIHIP_TRY(run_divergence(ctx, v, src, nullptr, 1, 0, q, s));                      // q = D0 (H g_v)
    if (gp || flags) {
        LaunchScope ls(ctx, PHIHIP_K_OTHER, s);
        hipLaunchKernelGGL(mask_cells_kernel<T>, dim3(bwd_blocks(v.cells), v.batch), dim3(kBlock), 0, s, q, (const T*)gp, flags, fpb, v.cells);
    }
    if (balance) PHIHIP_TRY(run_balance(ctx, v, flags, mask_batch, q, s));             // q = P (a pbar)
    PHIHIP_CHECK_HIP(hipMemsetAsync(lam, 0, cell_bytes, s));
    PHIHIP_TRY(run_cg(ctx, v, flags, mask_batch, q, lam, solve, info, s));             // lam = A^-1 q
    if (flags) {
        LaunchScope ls(ctx, PHIHIP_K_OTHER, s);
        hipLaunchKernelGGL(mask_cells_kernel<T>, dim3(bwd_blocks(v.cells), v.batch), dim3(kBlock), 0, s, lam, (const T*)nullptr, flags, fpb, v.cells);
    }
    if (balance) PHIHIP_TRY(run_balance(ctx, v, flags, mask_batch, lam, s));           // P (a lam)
    PHIHIP_TRY(run_grad_subtract(ctx, v, nullptr, 1, lam, gv, s));                     // g_v' = g_v - G0 (...)
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

int run_project_bwd(phihip_ctx* ctx, const GridView& v, const uint8_t* flags, int mask_batch, int balance, void* const gv[3], const void* gp,
                    const phihip_solve* solve, phihip_solve_info* info, hipStream_t s) {
    return v.dtype == PHIHIP_F64 ? project_bwd_t<double>(ctx, v, flags, mask_batch, balance, gv, gp, solve, info, s)
                                 : project_bwd_t<float>(ctx, v, flags, mask_batch, balance, gv, gp, solve, info, s);
}

// ---------------------------------------------------------------------------------------------------------------------
// VJP of math.grid_sample (advect.hip grid_sample_kernel): taps <- g * weight, coordinates <- g * d(out)/d(frac)
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int DIM>
__global__ __launch_bounds__(kBlock) void grid_sample_bwd_kernel(ScalarBc sb, int n0, int n1, int n2, const T* __restrict__ values, long long vstride,
                                                                 CComp3a<T> coords, long long npts, const T* __restrict__ gout, T* __restrict__ gvalues,
                                                                 Comp3w<T> gcoords) {
    constexpr int A0 = 3 - DIM;
    const int b = blockIdx.y;
    const int n[3] = {n0, n1, n2};
    int bc[3][2];
    T cv[3][2];
    scalar_rule<T>(sb, bc, cv);
    const T* __restrict__ F = values + (long long)b * vstride;
    T* __restrict__ GF = gvalues ? gvalues + (long long)b * vstride : nullptr;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < npts; i += (long long)gridDim.x * kBlock) {
        const long long o = (long long)b * npts + i;
        T c[3] = {T(0), T(0), T(0)};
#pragma unroll
        for (int a = A0; a < 3; ++a) c[a] = coords.p[a][o];
        AxisPair<T> ax[3];
        T fr[3], dfr[3];
        lookup_pairs<T, DIM>(c, n, bc, cv, ax, fr);
        const T g = gout[o];
        gather_adjoint<T, DIM>(F, GF, ax, fr, g, dfr);
#pragma unroll
        for (int a = A0; a < 3; ++a)
            if (gcoords.p[a]) gcoords.p[a][o] += g * dfr[a];
    }
}

int run_grid_sample_bwd(phihip_ctx* ctx, const GridView& v, const int32_t s_bc[3][2], const double s_val[3][2], const void* values, int values_batch, const void* const coords[3], long long npts,
                        const void* gout, void* gvalues, void* const gcoords[3], hipStream_t s) {
    if (v.cells >= (1LL << 31)) {
        set_error("grid_sample_backward: more than 2^31 values per batch entry are not supported");
        return PHIHIP_ERR_UNSUPPORTED;
    }
    const ScalarBc sb = make_scalar_bc(v, s_bc, s_val);
    const long long vstride = values_batch > 1 ? v.cells : 0;   // shared values: every batch entry scatters into the same gradient
    const unsigned nblk = (unsigned)((npts + kBlock - 1) / kBlock < 65536 ? (npts + kBlock - 1) / kBlock : 65536);
    LaunchScope ls(ctx, PHIHIP_K_ADVECT, s);
    if (npts > 0) {
#define PHIHIP_GSB(T, DIM)                                                                                                                           \
    hipLaunchKernelGGL((grid_sample_bwd_kernel<T, DIM>), dim3(nblk, v.batch), dim3(kBlock), 0, s, sb, v.n[0], v.n[1], v.n[2], (const T*)values, vstride, \
                       (CComp3a<T>{{(const T*)coords[0], (const T*)coords[1], (const T*)coords[2]}}), npts, (const T*)gout, (T*)gvalues,               \
                       (Comp3w<T>{{gcoords ? (T*)gcoords[0] : nullptr, gcoords ? (T*)gcoords[1] : nullptr, gcoords ? (T*)gcoords[2] : nullptr}}))
        if (v.dtype == PHIHIP_F64) { if (v.rank == 3) PHIHIP_GSB(double, 3); else PHIHIP_GSB(double, 2); }
        else { if (v.rank == 3) PHIHIP_GSB(float, 3); else PHIHIP_GSB(float, 2); }
#undef PHIHIP_GSB
    }
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

}  // namespace phihip
