// advect_common.hpp -- device helpers shared by the advection kernels (advect.hip) and their adjoints (adjoint.hip):
// boundary-resolved tap pairs, multilinear gather, velocity at faces / centres (phi/field/_resample.py:158-161,241-287,341-364).
#pragma once
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

// ---- work list of the LDS-staged advection kernels' fix-up pass (advect_tile.hip, advect_win.hip; r4) ------------------------------------
// A workgroup of a tiled kernel that meets a lookup outside its staged window appends (workgroup, PLANE) to a list in device memory; the
// fix-up launch behind it has a FIXED modest grid whose workgroups stride over the list and recompute exactly those planes with the gather
// code. Until r3 the unit was a workgroup's whole chunk of planes and the fix-up launch had one workgroup per tile workgroup: a plume that
// crosses CFL 1 in 5 % of the workgroups left 60 fix-up workgroups marching 64 planes each while 250 CUs idled -- the 0.14 ms advection of
// the smoke workload took 0.46 ms (profiles/r04_bench_smoke256_first.json). Two counters alternate between launches (common.hpp FixList): a tile
// kernel appends to one and clears the other for the launch after it -- the fix-up launch only reads, no tickets, no atomics.
__device__ __forceinline__ void fix_append(const FixList& L, int wg, int plane) {
    const int k = atomicAdd(L.count, 1);
    if (k < L.cap) L.items[k] = FixItem{wg, plane};
}
__device__ __forceinline__ int fix_count(const FixList& L) {
    const int c = *L.count;
    return c < L.cap ? c : L.cap;
}
// fix-up launch, one thread: the count of this launch goes to pinned host memory for the adaptive reach of the next pass (common.hpp AdvPolicy)
__device__ __forceinline__ void fix_publish(const FixList& L, int count) {
    if (!L.publish) return;
#ifdef __HIP_DEVICE_COMPILE__
    __hip_atomic_store(L.publish, count | L.reach_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#else
    *reinterpret_cast<volatile int*>(L.publish) = count | L.reach_tag;
#endif
}

// x - floor(x) in [0, 1) as ONE instruction (v_fract_f32 / v_fract_f64; floor + subtract are two). For a tiny negative x the plain difference
// rounds to 1.0, the instruction returns the largest value below 1 instead -- the host form restates that, so the emulation computes the same bits.
__device__ __forceinline__ float frac_part(float x) {
#ifdef __HIP_DEVICE_COMPILE__
    return __builtin_amdgcn_fractf(x);
#else
    const float f = x - floorf(x);
    return f < 1.0f ? f : (f == f ? 0x1.fffffep-1f : f);
#endif
}
__device__ __forceinline__ double frac_part(double x) {
#ifdef __HIP_DEVICE_COMPILE__
    return __builtin_amdgcn_fract(x);
#else
    const double f = x - floor(x);
    return f < 1.0 ? f : (f == f ? 0x1.fffffffffffffp-1 : f);
#endif
}

// LDS-DMA (advect_tile.hip r5, advect_win.hip r6): HBM / L2 -> LDS without a VGPR round trip. The transfer is inline assembly, invisible to the compiler's wait
// counters: the caller waits (s_waitcnt vmcnt) and meets at a workgroup barrier before any wavefront reads the landed bytes (MI355X_MICROARCH.md).
// 16 bytes per lane from `gsrc` (per lane) to LDS at `lds_dst` (wave-uniform) + 16 lane
template <typename T>
__device__ __forceinline__ void lds_dma16(const void* gsrc, T* lds_dst, int lane) {
#ifdef __HIP_DEVICE_COMPILE__
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
#else
    memcpy(reinterpret_cast<char*>(lds_dst) + 16 * lane, gsrc, 16);
#endif
}


// wrap into [0, n): one conditional +-n covers every shift below n cells; the integer modulo (~25 instructions) stays behind a
// branch that no wavefront takes at sensible CFL numbers
__device__ __forceinline__ int wrap_index(int i, int n) {
    i += i < 0 ? n : 0;
    i -= i >= n ? n : 0;
    if (__builtin_expect((unsigned)i >= (unsigned)n, 0)) {
        i %= n;
        if (i < 0) i += n;
    }
    return i;
}

// taps i_lo and i_lo + 1 of one axis under its boundary rule; every branch is wave-uniform (scalar)
template <typename T>
__device__ __forceinline__ AxisPair<T> make_pair(int i_lo, int n, int stride, int code_lo, int code_hi, T c_lo, T c_hi) {
    AxisPair<T> a;
    if (!wave_any(i_lo < 0 || i_lo + 1 >= n)) {   // no lane of the wavefront touches the boundary (scalar branch)
        a.off[0] = i_lo * stride; a.off[1] = a.off[0] + stride;
        a.cst[0] = a.cst[1] = false;
        a.cv[0] = a.cv[1] = T(0);
    } else if (code_lo == PHIHIP_BC_PERIODIC) {   // periodic is always set on both sides
        const int w0 = wrap_index(i_lo, n);
        const int w1 = w0 + 1 == n ? 0 : w0 + 1;
        a.off[0] = w0 * stride; a.off[1] = w1 * stride;
        a.cst[0] = a.cst[1] = false;
        a.cv[0] = a.cv[1] = T(0);
    } else {
        const int i_hi = i_lo + 1;
        const bool lo_c = code_lo == PHIHIP_BC_CLOSED, hi_c = code_hi == PHIHIP_BC_CLOSED;
        a.off[0] = min(max(i_lo, 0), n - 1) * stride;
        a.off[1] = min(max(i_hi, 0), n - 1) * stride;
        a.cst[0] = (i_lo < 0 && lo_c) || (i_lo >= n && hi_c);
        a.cst[1] = (i_hi < 0 && lo_c) || (i_hi >= n && hi_c);
        a.cv[0] = i_lo < 0 ? c_lo : c_hi;
        a.cv[1] = i_hi < 0 ? c_lo : c_hi;
    }
    return a;
}

// ---- r6: ONE arithmetic per advection sample -------------------------------------------------------------------------------------------
// Which kernel computes a sample of an advection pass is a matter of policy (LDS tile of reach 1 or 2, register-staged or LDS-DMA fill, the fix-up
// work list, the gather kernels; eager passes adapt their reach, captured ones keep it, a slab's window passes redo planes of the whole-slab pass).
// Until r5 these paths agreed with the oracle but not with each other in the last bits: the face means were a chain here and a tree there, the
// interpolation a sum of weight products in the gather kernels and nested fma-lerps in the windows, the fraction `x - floor(x)` or v_fract. Now every
// path evaluates the SAME expressions, defined here: sum4_chain, lerp_cell (a2, then a1, then a0, each as fma(fr, hi - lo, lo)), frac_part, and
// mc_correct for MacCormack's corrected value; min / max of the taps are order-free. Same inputs => same bits whatever the path
// (tests/parity_cases.py check_advect_paths_same_bits on the emulation and the GPU).
template <typename T>
__device__ __forceinline__ T sum4_chain(const T (&v)[2][2]) {      // [offset along the face's own axis][offset along the component's axis]
    return ((v[0][0] + v[0][1]) + v[1][0]) + v[1][1];
}
// t[k][b1][b2]: the 2^D taps (k = a0 offset; 2-D: k = 0 only)
template <typename T, int DIM>
__device__ __forceinline__ T lerp_cell(const T (&t)[2][2][2], const T (&fr)[3]) {
    T y[2];
#pragma unroll
    for (int k = 0; k < (DIM == 3 ? 2 : 1); ++k) {
        const T x0 = fma(fr[2], t[k][0][1] - t[k][0][0], t[k][0][0]), x1 = fma(fr[2], t[k][1][1] - t[k][1][0], t[k][1][0]);
        y[k] = fma(fr[1], x1 - x0, x0);
    }
    return DIM == 3 ? fma(fr[0], y[1] - y[0], y[0]) : y[0];
}
// MacCormack's corrected value before the limiter (advect.py:208): fwd + ch (field - bwd), one explicit fma in every kernel
template <typename T>
__device__ __forceinline__ T mc_correct(T fwd_here, T ch, T field_here, T bwd) {
    return fma(ch, field_here - bwd, fwd_here);
}

// the 2^D taps of a lookup from per-axis pairs; constant sides follow PhiML's sequential padding: the LAST axis that lies outside a constant side decides
template <typename T, int DIM>
__device__ __forceinline__ void gather_taps(const T* __restrict__ F, const AxisPair<T> (&ax)[3], T (&t)[2][2][2]) {
    const bool any_const = wave_any(ax[2].cst[0] | ax[2].cst[1] | ax[1].cst[0] | ax[1].cst[1] | (DIM == 3 ? (ax[0].cst[0] | ax[0].cst[1]) : false));
#pragma unroll
    for (int b0 = 0; b0 < (DIM == 3 ? 2 : 1); ++b0)
#pragma unroll
        for (int b1 = 0; b1 < 2; ++b1)
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) {
                const int o = (DIM == 3 ? ax[0].off[b0] : 0) + ax[1].off[b1] + ax[2].off[b2];
                T val;
                if (!any_const) val = F[o];   // wave-uniform
                else if (ax[2].cst[b2]) val = ax[2].cv[b2];
                else if (ax[1].cst[b1]) val = ax[1].cv[b1];
                else if (DIM == 3 && ax[0].cst[b0]) val = ax[0].cv[b0];
                else val = F[o];
                t[b0][b1][b2] = val;
            }
}
// multilinear interpolation of an ADVECTION lookup (the windows' arithmetic: lerp_cell)
template <typename T, int DIM>
__device__ __forceinline__ T gather_multilinear(const T* __restrict__ F, const AxisPair<T> (&ax)[3], const T (&fr)[3]) {
    T t[2][2][2];
    gather_taps<T, DIM>(F, ax, t);
    return lerp_cell<T, DIM>(t, fr);
}

// multilinear interpolation as PhiML's grid_sample writes it (phihip_grid_sample: lookups at caller-given coordinates, no window form exists):
// weights prod(where(bit, frac, 1 - frac)) summed in corner order (a0 = lowest bit).
template <typename T, int DIM>
__device__ __forceinline__ T gather_multilinear_weights(const T* __restrict__ F, const AxisPair<T> (&ax)[3], const T (&fr)[3]) {
    constexpr int A0 = 3 - DIM;
    const bool any_const = wave_any(ax[2].cst[0] | ax[2].cst[1] | ax[1].cst[0] | ax[1].cst[1] | (DIM == 3 ? (ax[0].cst[0] | ax[0].cst[1]) : false));
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
        if (!any_const) {   // wave-uniform
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
    T t[2][2][2];
    gather_taps<T, DIM>(F, ax, t);
    lo = hi = t[0][0][0];
#pragma unroll
    for (int k = 0; k < (DIM == 3 ? 2 : 1); ++k) {
        const T q[4] = {t[k][0][0], t[k][1][0], t[k][0][1], t[k][1][1]};      // (the order of advect_win.hip minmax_at; fmin / fmax like there)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lo = fmin(lo, q[i]);
            hi = fmax(hi, q[i]);
        }
    }
}

// the four values of component cb around the stored face `idx` of component CA: cells (m-1, m) along ca, physical faces (i, i+1) along cb,
// outside taps from the boundary rule (sample(velocity, field.geometry, at='face'), phi/field/_resample.py:158-161,279-287,341-364)
template <typename T, int DIM, int CA>
__device__ __forceinline__ void face_taps(const VelGrid& g, const CComp3a<T>& vel, int b, const int (&idx)[3], int cb, T (&v)[2][2]) {
    constexpr int A0 = 3 - DIM;
    constexpr int ca = CA;
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
    const bool any_const = wave_any(pa.cst[0] | pa.cst[1] | pb.cst[0] | pb.cst[1]);
#pragma unroll
    for (int ia = 0; ia < 2; ++ia)
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
            const bool ca_c = pa.cst[ia], cb_c = pb.cst[ib];
            if (any_const && (ca_c || cb_c)) {
                if (a_last) v[ia][ib] = ca_c ? pa.cv[ia] : pb.cv[ib];
                else v[ia][ib] = cb_c ? pb.cv[ib] : pa.cv[ia];
            } else {
                v[ia][ib] = C[rest + pa.off[ia] + pb.off[ib]];
            }
        }
}

// velocity at the stored face `idx` of component CA: own component + 4-point means of the others, the means as the reference's sample_subgrid writes them
// (lerps axis after axis with weights (0.5, 0.5)). The adjoint kernels use this form; the ADVECTION passes take face_disp below.
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
            T v[2][2];   // [ca offset][cb offset]
            face_taps<T, DIM, CA>(g, vel, b, idx, cb, v);
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

// FORWARD displacement dt u / dx (index units) of the stored face `idx` of component CA in the arithmetic of the LDS-staged kernels (advect_tile.hip,
// advect_win.hip): own component times shift, the others sum4_chain of their four values times (0.25 shift); shift[a] = (T)dt (T)(1 / dx[a]).
// The back-trace displacement is its negation (exact).
template <typename T, int DIM, int CA>
__device__ __forceinline__ void face_disp(const VelGrid& g, const CComp3a<T>& vel, int b, const int (&idx)[3], int f, T dt, T (&cf)[3]) {
    constexpr int A0 = 3 - DIM;
    constexpr int ca = CA;
    cf[0] = cf[1] = cf[2] = T(0);
#pragma unroll
    for (int cb = A0; cb < 3; ++cb) {
        const T shift = dt * (T)g.rdx[cb];
        if (cb == ca) {
            cf[cb] = vel.p[ca][(long long)b * g.ccells[ca] + f] * shift;
        } else {
            T v[2][2];
            face_taps<T, DIM, CA>(g, vel, b, idx, cb, v);
            cf[cb] = sum4_chain<T>(v) * (T(0.25) * shift);
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
        // NaN / infinite / absurd coordinates (finite_rk4 exists because velocities may be NaN) must not become wild indices: the
        // integer part is clamped to +-2^30 (NaN -> -2^30), the taps then resolve through the boundary rule and the result is NaN
        const T fc = fmin(fmax(fl, T(-1073741824.0)), T(1073741823.0));
        ax[a] = make_pair<T>((int)fc, n[a], stride[a], bc[a][0], bc[a][1], cv[a][0], cv[a][1]);
    }
    if (DIM == 2) { ax[0].off[0] = ax[0].off[1] = 0; ax[0].cst[0] = ax[0].cst[1] = false; ax[0].cv[0] = ax[0].cv[1] = T(0); }
}

// The same for a lookup given as (sample index, displacement in index units): coordinate = idx + disp, but integer part and fraction are
// formed from the DISPLACEMENT alone -- floor(disp), frac_part(disp) -- and the integer part is added to the index exactly. r4: the
// advection kernels used to round idx - dt u / dx to the element type first, which costs n eps / 2 of the lookup position on an axis of n
// samples (2.3e-5 cells at n = 384 in fp32: the parity tolerance had to grow with n); now the error is eps |disp| whatever the index, i.e.
// the kernels are MORE accurate than the NumPy path (absolute fp32 coordinates) they are checked against, and equal the fp64 evaluation of
// the same fp32 inputs to ~1e-7 of the field's variation per cell (tests/parity_cases.py check_advect_*: ground truth = fp64 oracle).
template <typename T, int DIM>
__device__ __forceinline__ void lookup_pairs_rel(const int (&idx)[3], const T (&disp)[3], const int (&n)[3], const int (&bc)[3][2], const T (&cv)[3][2],
                                                 AxisPair<T> (&ax)[3], T (&fr)[3]) {
    constexpr int A0 = 3 - DIM;
    const int stride[3] = {n[1] * n[2], n[2], 1};
    fr[0] = fr[1] = fr[2] = T(0);
#pragma unroll
    for (int a = A0; a < 3; ++a) {
        const T fl = floor(disp[a]);
        fr[a] = frac_part(disp[a]);      // r6: the SAME rounding as the LDS-staged kernels' v_fract (a displacement in [-eps/2, 0) gives the largest value below 1, not
                                         // 1.0), so a sample has the same bits whichever path computes it: tile, fix-up list, gather, narrow or wide reach
        // NaN / infinite / absurd displacements must not become wild indices: clamped to +-1e9 (NaN -> -1e9; idx + 1e9 < 2^31), the taps then
        // resolve through the boundary rule and the result is NaN
        const T fc = fmin(fmax(fl, T(-1.0e9)), T(1.0e9));
        ax[a] = make_pair<T>(idx[a] + (int)fc, n[a], stride[a], bc[a][0], bc[a][1], cv[a][0], cv[a][1]);
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

}  // namespace phihip
