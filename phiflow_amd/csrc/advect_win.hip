// advect_win.hip -- the advection passes that are NOT the velocity's semi-Lagrangian self-advection (advect_tile.hip), fed from LDS (r4):
//   WK_MC_STAG  correction pass of advect.mac_cormack(v, v, dt) on the staggered velocity, ALL components in one launch
//               (/root/reference phi/physics/advect.py:203-215; the limiter's cell frame: phi/field/_field.py:427-429)
//   WK_SL_CEN   advect.semi_lagrangian(s, v, dt) of a centred scalar (advect.py:156-179; velocity at the centres: _resample.py:145-157)
//   WK_MC_CEN   correction pass of advect.mac_cormack(s, v, dt) of a centred scalar
// Until round 3 these ran on the gather kernels of advect.hip: 17-27 scattered dword loads per sample, bound by the address units at
// 0.26-0.38 of the HBM rate (profiles/r03_kernel_roofline.json). Here a 256-thread workgroup owns a (T1 x T2) tile of the two fast axes,
// marches over a chunk of a0 planes and stages WINDOWS of every array a sample reads in LDS -- each window with the halo ITS taps need:
//   MC_STAG: the D velocity components with halo 1, but 2 along the component's own axis (the limiter looks up the field in the CELL
//            frame: its window is shifted by half a cell along that axis, taps reach -2 .. +2), and the D components of the forward pass
//            with halo 1: 71 KB (fp32, 3-D) -- DYNAMIC LDS beyond the 64 KB static limit, two workgroups per CU of the 160 KB;
//   SL_CEN:  the scalar with halo 1, velocity component c with halo 1 along c ONLY (u at a centre = mean of the cell's two c-faces);
//   MC_CEN:  scalar, forward pass (halo 1) and the velocity components as above: 39 KB.
// The boundary rule (wrap / clamp / constant; the last axis outside a constant side wins like PhiML's sequential padding) is applied while
// filling, the fill is cooperative and coalesced (row loads along the fast axis), a ring of 2 h0 + 2 planes per window keeps every plane
// read from HBM once per workgroup, one barrier per plane, requests two planes ahead (the discipline of advect_tile.hip: unconditional
// loads / stores so that s_waitcnt counts). A lookup that leaves its window (|displacement| >= 1 cell) flags the workgroup; the fix-up
// kernel (same grid, launched right behind) recomputes exactly those workgroups with the gather code of advect.hip -- the result does not
// depend on the path taken, and a CFL > 1 field is still correct.
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "advect_common.hpp"

namespace phihip {

enum WinKind { WK_MC_STAG = 0, WK_SL_CEN = 1, WK_MC_CEN = 2 };

// halo of window w along internal axis a. H (1 or 2, phihip_set_advect_halo): reach of the multilinear lookups of the centred kinds -- with 2
// a displacement below two cells stays in LDS (a smoke plume at dt = 1 moves faster than one cell per step in its core); the velocity
// windows of those kinds do not grow (the centre velocity is a mean of the cell's own faces), and the staggered correction pass has H = 1 only
// (six windows: 98 KB with H = 2, one workgroup per CU).
template <int KIND, int DIM, int H>
struct WinSpec;
template <int DIM, int H>
struct WinSpec<WK_MC_STAG, DIM, H> {                    // windows: velocity components A0 .. 2, then the forward-pass components
    static constexpr int NW = 2 * DIM;
    static constexpr int h(int w, int a) { return w < DIM ? ((3 - DIM + w) == a ? 2 : 1) : 1; }
};
template <int DIM, int H>
struct WinSpec<WK_SL_CEN, DIM, H> {                     // windows: the scalar, then the velocity components
    static constexpr int NW = 1 + DIM;
    static constexpr int h(int w, int a) { return w == 0 ? H : ((3 - DIM + w - 1) == a ? 1 : 0); }
};
template <int DIM, int H>
struct WinSpec<WK_MC_CEN, DIM, H> {                     // windows: the scalar, the forward pass, then the velocity components
    static constexpr int NW = 2 + DIM;
    static constexpr int h(int w, int a) { return w < 2 ? H : ((3 - DIM + w - 2) == a ? 1 : 0); }
};

// DMA (r6): the windows are filled by LDS-DMA (global_load_lds_dwordx4, advect_common.hpp lds_dma16) instead of through registers -- the lever that took the
// self-advection from 0.42 to 0.55 of peak in r5, carried over to these passes. A transfer moves 64 consecutive 16-byte chunks per wavefront instruction to
// CONSECUTIVE LDS addresses, so every window row is T2 / V + 2 whole chunks (the tile's columns and one chunk either side: the halo columns are its nearest
// elements) whatever the window's halo along the fast axis; the per-lane SOURCE resolves the row (wrap / clamp) and the chunk (wrap). Scope = regular grids:
// 3-D, no CLOSED side, the fast axis periodic with rows of whole vectors, 16-byte-aligned arrays.
template <typename T, int KIND, int DIM, int T1, int H = 1, bool DMA = false>
struct WinTile {
    using Spec = WinSpec<KIND, DIM, H>;
    static constexpr int NW = Spec::NW;
    static constexpr int V = 16 / (int)sizeof(T);
    static constexpr int T2 = sizeof(T) == 4 ? 64 : 32;   // tile columns = lanes along the fast axis (256 B rows)
    static constexpr int TY = kBlock / T2;
    static constexpr int S = T1 / TY;                     // tile positions per thread and plane
    static constexpr int h0(int w) { return DIM == 3 ? Spec::h(w, 0) : 0; }
    static constexpr int h1(int w) { return Spec::h(w, 1); }
    static constexpr int h2(int w) { return Spec::h(w, 2); }
    static constexpr int p1(int w) { return T1 + 2 * h1(w); }
    static constexpr int p2(int w) { return DMA ? T2 + 2 * V : T2 + 2 * h2(w); }
    static constexpr int c2(int w) { return DMA ? V : h2(w); }                    // window column of the tile's column 0
    static constexpr int nch(int w) { return p1(w) * (p2(w) / V); }               // DMA: 16-byte chunks per plane of window w, in (row, chunk) order
    static constexpr int ni(int w) { return (nch(w) + kWave - 1) / kWave; }       // DMA: wavefront instructions per plane of window w
    static constexpr int nitems() { int o = 0; for (int k = 0; k < NW; ++k) o += ni(k); return o; }
    static constexpr int NITEMS = nitems();                                       // DMA: (window, instruction) items per plane, dealt round-robin to the 4 wavefronts
    static constexpr int IPW = (NITEMS + kBlock / kWave - 1) / (kBlock / kWave);
    static constexpr int plane(int w) { return p1(w) * p2(w); }
    static constexpr int np(int w) { return DIM == 3 ? 2 * h0(w) + 2 : 1; }       // ring slots: planes p - h0 .. p + h0 in use, one being refilled
    // LDS layout: windows of EQUAL ring depth form a group whose slots are plane-major -- slot t of the group holds plane t of every member,
    // so the element offset of (window w, slot t) is gbase(w) + t * gstride(w) + gin(w) with gin a compile-time constant: one uniform plane
    // base per (group, relative plane) serves all its windows (per-window bases cost 20 scalar registers in the sample loop and spilled)
    static constexpr int gstride(int w) { int o = 0; for (int k = 0; k < NW; ++k) o += np(k) == np(w) ? plane(k) : 0; return o; }
    static constexpr int gin(int w) { int o = 0; for (int k = 0; k < w; ++k) o += np(k) == np(w) ? plane(k) : 0; return o; }
    static constexpr bool first_of_group(int w) { for (int k = 0; k < w; ++k) if (np(k) == np(w)) return false; return true; }
    static constexpr int gfirst(int w) { for (int k = 0; k < w; ++k) if (np(k) == np(w)) return k; return w; }
    static constexpr int gbase(int w) { int o = 0; for (int k = 0; k < gfirst(w); ++k) o += first_of_group(k) ? np(k) * gstride(k) : 0; return o; }
    static constexpr int total() { int o = 0; for (int k = 0; k < NW; ++k) o += np(k) * plane(k); return o; }
    static constexpr int kp(int w) { return (p1(w) + TY - 1) / TY; }              // fill passes of a thread per plane
    static constexpr int kpmax() { int m = 0; for (int w = 0; w < NW; ++w) m = kp(w) > m ? kp(w) : m; return m; }
    static constexpr int KP = kpmax();
    static constexpr int ntail(int w) { return DMA ? 0 : p1(w) * 2 * h2(w); }     // halo columns right of the T2 main columns: one element per thread
    static constexpr int tail_off(int w) { int o = 0; for (int k = 0; k < w; ++k) o += ntail(k); return o; }
    static constexpr int NTAIL = tail_off(NW);
    static constexpr int h0max() { int m = 0; for (int w = 0; w < NW; ++w) m = h0(w) > m ? h0(w) : m; return m; }
    static constexpr int H0MAX = h0max();
    static constexpr size_t BYTES = (size_t)total() * sizeof(T);
    static_assert(T1 % TY == 0, "tile rows must be a multiple of the thread rows");
    static_assert(NTAIL <= kBlock, "tail elements must fit one per thread");
    static_assert(BYTES <= 80 * 1024, "two workgroups per CU must fit the 160 KB of LDS");
    static_assert(!DMA || DIM == 3, "the LDS-DMA fill exists for 3-D grids");
};

// one staged array: where it lives, its stored extent, its padding rule (PHIHIP_BC_PERIODIC wrap / OPEN clamp / CLOSED constant)
template <typename T>
struct WinArray {
    const T* p;
    long long bstride;
    int n[3];
    int bc[3][2];
    T cv[3][2];
};

template <typename T>
struct WinParams {
    WinArray<T> arr[6];
    T* out[3];               // MC_STAG: the D components; centred kinds: out[0]
    long long ostride[3];    // elements per batch entry of out[k]
    int on[3][3];            // stored extent of out[k]
    int off[3];              // physical face number of stored index 0 per velocity component (runtime copy of OFFM)
    T shift[3];              // dt / dx: displacement in index units per unit velocity
    T ch;                    // 0.5 * correction strength (MacCormack kinds)
    int chunk, tiles1, tiles2, nblk, nmax0;
    FixList fix;             // (workgroup, plane) units whose lookups left a window: redone by the fix-up launch (advect_common.hpp)
    T* dump;
};

__device__ __forceinline__ int win_pad_index(int i, int n, int code_lo, int code_hi) {
    if (i < 0) {
        if (code_lo == PHIHIP_BC_PERIODIC) return wrap_index(i, n);
        return code_lo == PHIHIP_BC_CLOSED ? -1 : 0;
    }
    if (i >= n) {
        if (code_hi == PHIHIP_BC_PERIODIC) return wrap_index(i, n);
        return code_hi == PHIHIP_BC_CLOSED ? -2 : n - 1;
    }
    return i;
}
__device__ __forceinline__ int win_const_side(int i, int n, int code_lo, int code_hi) {
    return (i < 0 && code_lo == PHIHIP_BC_CLOSED) ? 1 : ((i >= n && code_hi == PHIHIP_BC_CLOSED) ? 2 : 0);
}
__device__ __forceinline__ void win_opaque(unsigned& v) {
#ifdef __HIP_DEVICE_COMPILE__
    asm volatile("" : "+v"(v));
#endif
}
__device__ __forceinline__ float win_clamp(float x, float lo, float hi) {
#ifdef __HIP_DEVICE_COMPILE__
    return __builtin_amdgcn_fmed3f(x, lo, hi);      // NaN -> lo, which the callers treat as "outside"
#else
    return x >= lo ? (x <= hi ? x : hi) : lo;
#endif
}
__device__ __forceinline__ double win_clamp(double x, double lo, double hi) { return x >= lo ? (x <= hi ? x : hi) : lo; }

// OFFM: bit a = the lower face of axis a is NOT stored (CLOSED lower side): the static offsets of the velocity means depend on it.
// CONSTS: some window may need constants patched in (a CLOSED side of the velocity or a constant extrapolation of the scalar).
template <typename T, int KIND, int DIM, int T1, int OFFM, bool CONSTS, int H, bool DMA = false>
__global__ __launch_bounds__(kBlock, 2) void advect_win_kernel(WinParams<T> P) {
    using C = WinTile<T, KIND, DIM, T1, H, DMA>;
    static_assert(!DMA || (!CONSTS && OFFM == 0), "LDS-DMA fill: regular grids (no CLOSED side)");
    constexpr int A0 = 3 - DIM;
    constexpr int NW = C::NW, T2 = C::T2, TY = C::TY, S = C::S, KP = C::KP;
    constexpr int OFF[3] = {(OFFM >> 0) & 1, (OFFM >> 1) & 1, (OFFM >> 2) & 1};
    PHIHIP_DYNAMIC_LDS(T, lds);
    __shared__ int slow_sh[2];          // "a lookup of plane p left its window", by the parity of p (read by thread 0 after the plane's barrier)

    const int tid = threadIdx.x, tx = tid % T2, ty = tid / T2;
    const int b = blockIdx.y;
    const int bid = xcd_order(blockIdx.x, P.nblk);   // neighbouring tiles share an XCD's L2
    const int t2 = bid % P.tiles2;
    const int t1 = (bid / P.tiles2) % P.tiles1;
    const int c0 = bid / (P.tiles2 * P.tiles1);
    const int lo1 = t1 * T1, lo2 = t2 * T2;
    const int pb = DIM == 3 ? c0 * P.chunk : 0;
    const int pe = DIM == 3 ? min(pb + P.chunk, P.nmax0) : 1;
    if (tid < 2) slow_sh[tid] = 0;      // (barriers of the ring warm-up / of the 2-D fill lie between this and the first plane of samples)
    if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) *P.fix.next = 0;    // the work list's other counter, for the launch after this one

    // does this workgroup's window reach beyond a constant side? (uniform: interior tiles and boxes without one skip every select)
    bool has_const = false;
    if (CONSTS) {
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const WinArray<T>& A = P.arr[w];
            has_const = has_const || (A.bc[1][0] == PHIHIP_BC_CLOSED && lo1 - C::h1(w) < 0) || (A.bc[1][1] == PHIHIP_BC_CLOSED && lo1 + T1 + C::h1(w) > A.n[1]) ||
                        (A.bc[2][0] == PHIHIP_BC_CLOSED && lo2 - C::h2(w) < 0) || (A.bc[2][1] == PHIHIP_BC_CLOSED && lo2 + T2 + C::h2(w) > A.n[2]);
            if (DIM == 3) has_const = has_const || (A.bc[0][0] == PHIHIP_BC_CLOSED && pb - C::h0(w) < 0) || (A.bc[0][1] == PHIHIP_BC_CLOSED && pe + C::h0(w) > A.n[0]);
        }
    }

    // ---- per-thread fill descriptors (plane-invariant): byte offset within a plane of element (row ty + kp TY, column tx) of window w ----
    unsigned eoff[NW][KP];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const WinArray<T>& A = P.arr[w];
        const int k = win_pad_index(lo2 - C::h2(w) + tx, A.n[2], A.bc[2][0], A.bc[2][1]);
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
            const int row = ty + kp * TY;
            const int j = win_pad_index(lo1 - C::h1(w) + (row < C::p1(w) ? row : 0), A.n[1], A.bc[1][0], A.bc[1][1]);
            eoff[w][kp] = (unsigned)((j < 0 ? 0 : j * A.n[2]) + (k < 0 ? 0 : k)) * (unsigned)sizeof(T);
        }
    }
    // tail element (halo columns T2 .. T2 + 2 h2 - 1 of every row of the windows that have them): window, row and column of THIS thread
    int tail_w = -1, tail_r = 0, tail_q = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w)
        if (C::ntail(w) > 0 && tid >= C::tail_off(w) && tid < C::tail_off(w) + C::ntail(w)) {
            tail_w = w;
            tail_r = (tid - C::tail_off(w)) / (2 * C::h2(w));
            tail_q = T2 + (tid - C::tail_off(w)) % (2 * C::h2(w));
        }
    unsigned tail_eoff = 0;
    int tail_lds = 0;          // LDS element offset of the tail element within slot 0 of its window
#pragma unroll
    for (int w = 0; w < NW; ++w)
        if (w == tail_w) {
            const WinArray<T>& A = P.arr[w];
            const int kk = win_pad_index(lo2 - C::h2(w) + tail_q, A.n[2], A.bc[2][0], A.bc[2][1]);
            const int j = win_pad_index(lo1 - C::h1(w) + tail_r, A.n[1], A.bc[1][0], A.bc[1][1]);
            tail_eoff = (unsigned)((j < 0 ? 0 : j * A.n[2]) + (kk < 0 ? 0 : kk)) * (unsigned)sizeof(T);
            tail_lds = C::gbase(w) + C::gin(w) + tail_r * C::p2(w) + tail_q;
        }

    // plane of window w that supplies staged plane i0: wrapped (one +-n suffices: every axis has >= 4 samples here) or clamped; beyond a
    // CONSTANT side any valid plane is read and patched. Branch-free scalar code.
    auto plane_src = [&](int w, int i0) -> long long {
        if (DIM == 2) return 0;
        const WinArray<T>& A = P.arr[w];
        const int n = A.n[0];
        int q = i0;
        if (A.bc[0][0] == PHIHIP_BC_PERIODIC) { q += q < 0 ? n : 0; q -= q >= n ? n : 0; }
        q = min(max(q, 0), n - 1);
        return (long long)q * ((long long)A.n[1] * A.n[2]);
    };
    auto slot_of = [&](int w, int i0) -> int {      // uniform
        if (DIM == 2) return 0;
        const int n = C::np(w);
        const int m = i0 % n;
        return m < 0 ? m + n : m;
    };
    // staged planes of window w for this chunk: pb - h0 .. pe - 1 + h0
    auto k_lo = [&](int w) { return DIM == 3 ? pb - C::h0(w) : 0; };
    auto k_hi = [&](int w) { return DIM == 3 ? pe - 1 + C::h0(w) : 0; };

    // Every global load of the plane loop is unconditional (hipcc then counts the outstanding operations: s_waitcnt vmcnt(N > 0)); in
    // half-step p window w requests plane p + h0(w) + 2, clamped into the range it stages.
    auto load_planes = [&](int p, T (&R)[NW][KP], T& tailv) {
        long long tail_src = 0;
        const T* tail_base = P.arr[0].p;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const WinArray<T>& A = P.arr[w];
            const int kl = min(max(p + C::h0(w) + 2, k_lo(w)), k_hi(w));
            const long long ps = plane_src(w, kl);
            const char* __restrict__ base = (const char*)(A.p + (long long)b * A.bstride + ps);
#pragma unroll
            for (int kp = 0; kp < C::kp(w); ++kp) {
                unsigned o = eoff[w][kp];
                win_opaque(o);
                R[w][kp] = *(const T*)(base + o);
            }
            if (w == tail_w) { tail_base = A.p + (long long)b * A.bstride; tail_src = ps; }
        }
        if (C::NTAIL > 0) {
            unsigned o = tail_eoff;
            win_opaque(o);
            tailv = *(const T*)((const char*)(tail_base + tail_src) + o);
        }
    };
    // Constants of CLOSED sides, patched in when a plane enters the ring (cold, uniform: only workgroups whose windows cross such a side).
    // The LAST axis outside a constant side decides (PhiML pads axis after axis): a2 over a1 over a0. Everything the patch needs is prepared
    // ONCE: the in-plane decision per fill element lives in registers (value + one bit), the per-window plane rule in a small LDS table --
    // read from the kernel arguments inside the plane loop, the boundary codes / constants / extents of six windows are ~100 live scalar
    // registers, i.e. several hundred SGPR spill instructions per plane also in the workgroups that never take this path.
    __shared__ T ctab[NW > 0 ? NW : 1][2];        // constant of the lower / upper a0 side of window w
    __shared__ int ztab[NW > 0 ? NW : 1][2];      // staged planes < ztab[w][0] / >= ztab[w][1] lie beyond a CLOSED a0 side
    T cval[NW][KP];
    T tail_cval = T(0);
    unsigned cbits = 0;                           // bit w * KP + kp: element (w, kp) of this thread is a constant; bit 31: its tail element
    if (CONSTS) {
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const WinArray<T>& A = P.arr[w];
            const int cside = win_const_side(lo2 - C::h2(w) + tx, A.n[2], A.bc[2][0], A.bc[2][1]);
#pragma unroll
            for (int kp = 0; kp < KP; ++kp) {
                const int j = win_const_side(lo1 - C::h1(w) + ty + kp * TY, A.n[1], A.bc[1][0], A.bc[1][1]);
                cval[w][kp] = cside ? (cside == 1 ? A.cv[2][0] : A.cv[2][1]) : (j == 1 ? A.cv[1][0] : A.cv[1][1]);
                if (cside || j) cbits |= 1u << (w * KP + kp);
            }
            if (w == tail_w) {
                const int rs = win_const_side(lo1 - C::h1(w) + tail_r, A.n[1], A.bc[1][0], A.bc[1][1]);
                const int cs = win_const_side(lo2 - C::h2(w) + tail_q, A.n[2], A.bc[2][0], A.bc[2][1]);
                tail_cval = cs ? (cs == 1 ? A.cv[2][0] : A.cv[2][1]) : (rs == 1 ? A.cv[1][0] : A.cv[1][1]);
                if (cs || rs) cbits |= 1u << 31;
            }
            if (tid == w) {
                ctab[w][0] = A.cv[0][0];
                ctab[w][1] = A.cv[0][1];
                ztab[w][0] = (DIM == 3 && A.bc[0][0] == PHIHIP_BC_CLOSED) ? 0 : -(1 << 30);
                ztab[w][1] = (DIM == 3 && A.bc[0][1] == PHIHIP_BC_CLOSED) ? A.n[0] : (1 << 30);
            }
        }
        __syncthreads();
    } else {
#pragma unroll
        for (int w = 0; w < NW; ++w)
#pragma unroll
            for (int kp = 0; kp < KP; ++kp) cval[w][kp] = T(0);
    }
    auto patch_planes = [&](int p, T (&R)[NW][KP], T& tailv) {
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int ks = p + C::h0(w) + 1;
            const bool below = ks < ztab[w][0], above = ks >= ztab[w][1];       // uniform (LDS broadcast reads)
            const T pv = below ? ctab[w][0] : ctab[w][1];
#pragma unroll
            for (int kp = 0; kp < C::kp(w); ++kp) {
                T v = R[w][kp];
                v = (below || above) ? pv : v;
                v = ((cbits >> (w * KP + kp)) & 1u) ? cval[w][kp] : v;
                R[w][kp] = v;
            }
            if (C::ntail(w) > 0 && w == tail_w) {
                tailv = (below || above) ? pv : tailv;
                tailv = (cbits >> 31) ? tail_cval : tailv;
            }
        }
    };
    // what half-step p - 1 requested (plane p + h0 + 1 of every window) enters the ring
    auto store_planes = [&](int p, const T (&R)[NW][KP], T tailv) {
        int tail_slot = -1;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int ks = p + C::h0(w) + 1;
            if (ks < k_lo(w) || ks > k_hi(w)) continue;             // uniform
            const int slot = slot_of(w, ks);
            T* L = lds + C::gbase(w) + slot * C::gstride(w) + C::gin(w);
#pragma unroll
            for (int kp = 0; kp < C::kp(w); ++kp)
                if (ty + kp * TY < C::p1(w)) L[(ty + kp * TY) * C::p2(w) + tx] = R[w][kp];
            if (w == tail_w) tail_slot = slot * C::gstride(w);
        }
        if (C::NTAIL > 0 && tail_slot >= 0) lds[tail_lds + tail_slot] = tailv;
    };

    // ---- one plane of samples ---------------------------------------------------------------------------------------------------
    // per-thread output bookkeeping (plane-invariant): in-plane element offset of position s = 0 per output (unsigned 32-bit + a uniform
    // 64-bit plane base: the address arithmetic of the stores stays scalar) and one bit per (position, output): the sample exists
    constexpr int NOUT = KIND == WK_MC_STAG ? 3 : 1;
    constexpr int O0 = KIND == WK_MC_STAG ? A0 : 0;
    unsigned obase[3] = {0, 0, 0};
    unsigned vbits = 0;
#pragma unroll
    for (int c = O0; c < NOUT; ++c) {
        obase[c] = (unsigned)((lo1 + ty) * P.on[c][2] + lo2 + tx);
#pragma unroll
        for (int k = 0; k < S; ++k)
            if (lo1 + ty + k * TY < P.on[c][1] && lo2 + tx < P.on[c][2]) vbits |= 1u << (k * 3 + c);
    }
    auto compute_plane = [&](int p) {
        bool slow_any = false;
        // element offset of plane p + d of window w (uniform): pbase[w][d + H0MAX]
        int pbase[NW][2 * C::H0MAX + 1];
#pragma unroll
        for (int w = 0; w < NW; ++w)
#pragma unroll
            for (int d = -C::H0MAX; d <= C::H0MAX; ++d)
                pbase[w][d + C::H0MAX] = (d >= -C::h0(w) && d <= C::h0(w)) ? C::gbase(w) + slot_of(w, p + d) * C::gstride(w) : 0;
        // multilinear lookup (or min / max over the taps) in window w: `rel` = lower tap relative to the sample per axis (integral values in
        // the element type, within [-h, h - 1]), `cen` = the sample's in-plane position in the window. The two tap planes are SELECTED among
        // the uniform plane bases (no per-lane multiplication / ring arithmetic).
        auto tap_base = [&](int w, int cen, const T (&rel)[3], int (&base)[2]) {
            const int inplane = cen + C::gin(w) + __mul24((int)rel[1], C::p2(w)) + (int)rel[2];
            if (DIM == 3) {
                int b0 = pbase[w][C::H0MAX], b1 = pbase[w][C::H0MAX + 1];      // rel = 0: planes p, p + 1
#pragma unroll
                for (int d = -C::h0(w); d < C::h0(w); ++d) {
                    if (d == 0) continue;
                    const bool hit = d < 0 ? rel[0] <= (T)d : rel[0] >= (T)d;      // (descending for d < 0: the last match wins)
                    if (d < 0) { b0 = rel[0] == (T)d ? pbase[w][d + C::H0MAX] : b0; b1 = rel[0] == (T)d ? pbase[w][d + 1 + C::H0MAX] : b1; }
                    else { b0 = hit ? pbase[w][d + C::H0MAX] : b0; b1 = hit ? pbase[w][d + 1 + C::H0MAX] : b1; }
                }
                base[0] = b0 + inplane;
                base[1] = b1 + inplane;
            } else {
                base[0] = base[1] = C::gbase(w) + inplane;
            }
        };
        auto lerp_taps = [&](int w, int cen, const T (&rel)[3], const T (&fr)[3]) -> T {
            int base[2];
            tap_base(w, cen, rel, base);
            T y[2];
#pragma unroll
            for (int k = 0; k < (DIM == 3 ? 2 : 1); ++k) {
                const int bk = base[k];
                const T a00 = lds[bk], a10 = lds[bk + C::p2(w)], a01 = lds[bk + 1], a11 = lds[bk + C::p2(w) + 1];
                const T x0 = fma(fr[2], a01 - a00, a00), x1 = fma(fr[2], a11 - a10, a10);
                y[k] = fma(fr[1], x1 - x0, x0);
            }
            return DIM == 3 ? fma(fr[0], y[1] - y[0], y[0]) : y[0];
        };
        auto minmax_taps = [&](int w, int cen, const T (&rel)[3], T& lo, T& hi) {
            int base[2];
            tap_base(w, cen, rel, base);
            lo = hi = lds[base[0]];
#pragma unroll
            for (int k = 0; k < (DIM == 3 ? 2 : 1); ++k) {
                const int bk = base[k];
                const T t[4] = {lds[bk], lds[bk + C::p2(w)], lds[bk + 1], lds[bk + C::p2(w) + 1]};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    lo = fmin(lo, t[q]);
                    hi = fmax(hi, t[q]);
                }
            }
        };
        // integer part / fraction of a lookup DISPLACEMENT (position relative to the sample, in index units: the integer part is exact whatever the
        // index -- advect_common.hpp lookup_pairs_rel); `dev` accumulates (rel - clamp(rel))^2: non-zero (or
        // NaN) when some tap leaves [lo_rel, hi_rel + 1] -- no lane masks in scalar registers, no compare per axis
        auto split = [&](T disp, int lo_rel, int hi_rel, T& fr, T& rel, T& dev) {
            const T r0 = floor(disp);
            fr = frac_part(disp);            // (the same rounding as split_sign and lookup_pairs_rel)
            rel = win_clamp(r0, (T)lo_rel, (T)hi_rel);
            const T d = r0 - rel;
            dev = fma(d, d, dev);
        };
        // r5 (instruction diet; ~4 cycles per wave64 VALU instruction of any class on gfx950, tools/micro/issue_rates.hip): a lookup whose taps may only
        // be the sample's own cell or the one below it (reach 1: the displacement lies in [-1, 1)) needs no floor / clamp / convert / multiply -- the
        // integer part is the displacement's SIGN: fraction = ONE v_fract, tap offset = one select on the sign, "left the window" = the running maximum
        // of the magnitudes compared with 1 once per sample (NaN included; exactly -1 goes to the fix-up pass, which computes the same value). The
        // offsets are inside the window whatever the displacement. GEN = the one axis that keeps the general form (the limiter's own axis: reach 2).
        auto split_sign = [&](T disp, T& fr, T& mx) {
            fr = frac_part(disp);
            mx = fmax(mx, fabs(disp));
        };
        auto tap_base_sign = [&](int w, int cen, const T (&disp)[3], const T (&rel)[3], int gen, int (&base)[2]) {
            int inplane = cen + C::gin(w);
            inplane += gen == 1 ? __mul24((int)rel[1], C::p2(w)) : (disp[1] < T(0) ? -C::p2(w) : 0);
            inplane += gen == 2 ? (int)rel[2] : (disp[2] < T(0) ? -1 : 0);
            if (DIM == 3) {
                int b0 = pbase[w][C::H0MAX], b1 = pbase[w][C::H0MAX + 1];      // rel = 0: planes p, p + 1
                if (gen == 0) {
#pragma unroll
                    for (int d = -C::h0(w); d < C::h0(w); ++d) {
                        if (d == 0) continue;
                        const bool hit = d < 0 ? rel[0] <= (T)d : rel[0] >= (T)d;
                        if (d < 0) { b0 = rel[0] == (T)d ? pbase[w][d + C::H0MAX] : b0; b1 = rel[0] == (T)d ? pbase[w][d + 1 + C::H0MAX] : b1; }
                        else { b0 = hit ? pbase[w][d + C::H0MAX] : b0; b1 = hit ? pbase[w][d + 1 + C::H0MAX] : b1; }
                    }
                } else {
                    const bool down = disp[0] < T(0);
                    b0 = down ? pbase[w][C::H0MAX > 0 ? C::H0MAX - 1 : 0] : b0;
                    b1 = down ? pbase[w][C::H0MAX] : b1;
                }
                base[0] = b0 + inplane;
                base[1] = b1 + inplane;
            } else {
                base[0] = base[1] = C::gbase(w) + inplane;
            }
        };
        auto lerp_at = [&](int w, const int (&base)[2], const T (&fr)[3]) -> T {
            T y[2];
#pragma unroll
            for (int k = 0; k < (DIM == 3 ? 2 : 1); ++k) {
                const int bk = base[k];
                const T a00 = lds[bk], a10 = lds[bk + C::p2(w)], a01 = lds[bk + 1], a11 = lds[bk + C::p2(w) + 1];
                const T x0 = fma(fr[2], a01 - a00, a00), x1 = fma(fr[2], a11 - a10, a10);
                y[k] = fma(fr[1], x1 - x0, x0);
            }
            return DIM == 3 ? fma(fr[0], y[1] - y[0], y[0]) : y[0];
        };
        auto minmax_at = [&](int w, const int (&base)[2], T& lo, T& hi) {
            lo = hi = lds[base[0]];
#pragma unroll
            for (int k = 0; k < (DIM == 3 ? 2 : 1); ++k) {
                const int bk = base[k];
                const T t[4] = {lds[bk], lds[bk + C::p2(w)], lds[bk + 1], lds[bk + C::p2(w) + 1]};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    lo = fmin(lo, t[q]);
                    hi = fmax(hi, t[q]);
                }
            }
        };
        // both tile positions of a thread in one straight-line body: at two workgroups per CU (LDS) the registers are there, and the second
        // position's LDS reads overlap the first one's arithmetic (same-box A/B, profiles/r04_time_frow_session_c.jsonl: 2-8 %)
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int r = ty + s * TY;
            int cen[NW];                    // the sample's in-plane position in every window
#pragma unroll
            for (int w = 0; w < NW; ++w) cen[w] = (r + C::h1(w)) * C::p2(w) + tx + C::c2(w);
            // static tap: window w at (plane offset d0, row offset d1, column offset d2) from the sample
            auto at = [&](int w, int d0, int d1, int d2) -> T { return lds[pbase[w][d0 + C::H0MAX] + cen[w] + (C::gin(w) + d1 * C::p2(w) + d2)]; };
            if (KIND == WK_MC_STAG) {
#pragma unroll
                for (int ca = A0; ca < 3; ++ca) {
                    const int wv = ca - A0, wf = DIM + ca - A0;
                    const T vc = at(wv, 0, 0, 0);
                    T cb[3] = {T(0), T(0), T(0)}, cf[3] = {T(0), T(0), T(0)};
                    cf[ca] = vc * P.shift[ca];
                    cb[ca] = -cf[ca];
#pragma unroll
                    for (int cbx = A0; cbx < 3; ++cbx) {
                        if (cbx == ca) continue;
                        // component cbx at this ca-face: cells (m - 1, m) along ca, faces (s, s + 1) along cbx (advect_common.hpp face_velocity)
                        T v4[2][2];
#pragma unroll
                        for (int ia = 0; ia < 2; ++ia)
#pragma unroll
                            for (int ib = 0; ib < 2; ++ib) {
                                int d[3] = {0, 0, 0};
                                d[ca] = OFF[ca] - 1 + ia;
                                d[cbx] = -OFF[cbx] + ib;
                                v4[ia][ib] = at(cbx - A0, d[0], d[1], d[2]);
                            }
                        const T sum = sum4_chain<T>(v4);
                        cf[cbx] = sum * (T(0.25) * P.shift[cbx]);
                        cb[cbx] = -cf[cbx];
                    }
                    T dev = T(0), mx = T(0);
                    T fr[3] = {T(0), T(0), T(0)}, rel[3] = {T(0), T(0), T(0)};
                    int tb[2];
#pragma unroll
                    for (int a = A0; a < 3; ++a) split_sign(cf[a], fr[a], mx);
                    tap_base_sign(wf, cen[wf], cf, rel, -1, tb);
                    const T bwd = lerp_at(wf, tb, fr);
                    const T nv = mc_correct(at(wf, 0, 0, 0), P.ch, vc, bwd);
                    // limiter: closest grid values of the backward lookup in the CELL frame (own axis: m - 1/2 instead of the stored index)
                    cb[ca] += (T)OFF[ca] - T(0.5);
                    {
                        T fdummy;
                        split(cb[ca], -2, 1, fdummy, rel[ca], dev);          // the own axis: reach 2 (general form); the others: the sign
#pragma unroll
                        for (int a = A0; a < 3; ++a)
                            if (a != ca) mx = fmax(mx, fabs(cb[a]));
                    }
                    tap_base_sign(wv, cen[wv], cb, rel, ca, tb);
                    T lo, hi;
                    minmax_at(wv, tb, lo, hi);
                    const T val = nv < lo ? lo : (nv > hi ? hi : nv);      // math.clip = minimum(maximum(x, lo), hi)
                    const bool valid = ((vbits >> (s * 3 + ca)) & 1u) && p < P.on[ca][0];
                    slow_any = slow_any || (valid && (!(dev == T(0)) || !(mx < T(1))));
                    T* const slot = P.out[ca] + (long long)b * P.ostride[ca] + (long long)p * ((long long)P.on[ca][1] * P.on[ca][2]) +
                                    (obase[ca] + (unsigned)(s * TY * P.on[ca][2]));
                    *(valid ? slot : P.dump) = val;
                }
            } else {
                constexpr int WV0 = KIND == WK_SL_CEN ? 1 : 2;      // first velocity window
                T cb[3] = {T(0), T(0), T(0)}, cf[3] = {T(0), T(0), T(0)};
#pragma unroll
                for (int c = A0; c < 3; ++c) {
                    // staggered velocity at the cell centre: mean of the cell's two c-faces (advect_common.hpp center_velocity)
                    int d[3] = {0, 0, 0};
                    d[c] = -OFF[c];
                    const T lo_f = at(WV0 + c - A0, d[0], d[1], d[2]);
                    d[c] = -OFF[c] + 1;
                    const T hi_f = at(WV0 + c - A0, d[0], d[1], d[2]);
                    const T u = hi_f * T(0.5) + lo_f * T(0.5);
                    const T sft = u * P.shift[c];
                    cb[c] = -sft;
                    cf[c] = sft;
                }
                T dev = T(0), mx = T(0);
                T fr[3] = {T(0), T(0), T(0)}, rel[3] = {T(0), T(0), T(0)};
                T val;
                if (H == 1) {          // reach 1: the sign form of every lookup
                    int tb[2];
                    if (KIND == WK_SL_CEN) {
#pragma unroll
                        for (int a = A0; a < 3; ++a) split_sign(cb[a], fr[a], mx);
                        tap_base_sign(0, cen[0], cb, rel, -1, tb);
                        val = lerp_at(0, tb, fr);
                    } else {
#pragma unroll
                        for (int a = A0; a < 3; ++a) split_sign(cf[a], fr[a], mx);      // (|cb| = |cf|: one maximum serves both lookups)
                        tap_base_sign(1, cen[1], cf, rel, -1, tb);
                        const T bwd = lerp_at(1, tb, fr);
                        const T nv = mc_correct(at(1, 0, 0, 0), P.ch, at(0, 0, 0, 0), bwd);
                        tap_base_sign(0, cen[0], cb, rel, -1, tb);
                        T lo, hi;
                        minmax_at(0, tb, lo, hi);
                        val = nv < lo ? lo : (nv > hi ? hi : nv);
                    }
                } else if (KIND == WK_SL_CEN) {
#pragma unroll
                    for (int a = A0; a < 3; ++a) split(cb[a], -H, H - 1, fr[a], rel[a], dev);
                    val = lerp_taps(0, cen[0], rel, fr);
                } else {
#pragma unroll
                    for (int a = A0; a < 3; ++a) split(cf[a], -H, H - 1, fr[a], rel[a], dev);
                    const T bwd = lerp_taps(1, cen[1], rel, fr);
                    const T nv = mc_correct(at(1, 0, 0, 0), P.ch, at(0, 0, 0, 0), bwd);
#pragma unroll
                    for (int a = A0; a < 3; ++a) split(cb[a], -H, H - 1, fr[a], rel[a], dev);
                    T lo, hi;
                    minmax_taps(0, cen[0], rel, lo, hi);
                    val = nv < lo ? lo : (nv > hi ? hi : nv);
                }
                const bool valid = ((vbits >> (s * 3)) & 1u) && p < P.on[0][0];
                slow_any = slow_any || (valid && (!(dev == T(0)) || !(mx < T(1))));
                T* const slot = P.out[0] + (long long)b * P.ostride[0] + (long long)p * ((long long)P.on[0][1] * P.on[0][2]) +
                                (obase[0] + (unsigned)(s * TY * P.on[0][2]));
                *(valid ? slot : P.dump) = val;
            }
        }
        if (slow_any) slow_sh[p & 1] = 1;
    };
    // after the barrier that ends plane p's half-step: one entry in the fix-up work list if any sample of the plane left a window
    auto report_plane = [&](int p) {
        if (tid == 0 && slow_sh[p & 1]) {
            slow_sh[p & 1] = 0;
            fix_append(P.fix, b * P.nblk + (int)blockIdx.x, p);
        }
    };

    // ---- pipeline: two planes per trip so that the two register sets keep static names ------------------------------------------
    T RA[NW][KP], RB[NW][KP];
    T tailA = T(0), tailB = T(0);
    auto half_step = [&](int p, bool compute, T (&Rld)[NW][KP], T& tail_ld, T (&Rst)[NW][KP], T& tail_st) {
        load_planes(p, Rld, tail_ld);
        if (compute) compute_plane(p);
        if (CONSTS && has_const) patch_planes(p, Rst, tail_st);
        store_planes(p, Rst, tail_st);
        __syncthreads();
        if (compute) report_plane(p);
    };
    if constexpr (DMA) {
        // ---- r6: the rings filled by LDS-DMA. Items = (window w, instruction k): 64 consecutive chunks of w's plane; item i belongs to wavefront i % 4. Per lane and
        // item the byte offset of its chunk within a plane is plane-invariant (row wrapped / clamped, chunk wrapped along the periodic fast axis).
        constexpr int NWAVE = kBlock / kWave;
        const int lane = tid & (kWave - 1);
#ifdef __HIP_DEVICE_COMPILE__
        const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
#else
        const int wave = tid / kWave;
#endif
        unsigned doff[C::IPW];
#pragma unroll
        for (int i = 0; i < C::IPW; ++i) doff[i] = 0;
        {
            int item = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const WinArray<T>& A = P.arr[w];
                const int G2 = C::p2(w) / C::V;
#pragma unroll
                for (int k = 0; k < C::ni(w); ++k, ++item) {
                    if ((item % NWAVE) != wave) continue;          // uniform
                    const int q = k * kWave + lane;
                    const int rr = q / G2, gg = q - rr * G2;
                    int j = lo1 - C::h1(w) + rr;
                    if (A.bc[1][0] == PHIHIP_BC_PERIODIC) j = wrap_index(j, A.n[1]);
                    j = min(max(j, 0), A.n[1] - 1);                                   // OPEN: zero-gradient padding = the edge row
                    const int k0 = wrap_index(lo2 - C::V + gg * C::V, A.n[2]);      // periodic fast axis, n2 a multiple of V: a chunk never straddles the seam
                    doff[item / NWAVE] = (unsigned)(j * A.n[2] + k0) * (unsigned)sizeof(T);
                }
            }
        }
        // request plane q + h0(w) + 1 of every window (half-step q): its slot held plane q - h0 - 1, whose last readers passed the barrier of half-step q - 1
        auto feed = [&](int q) {
            int item = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const WinArray<T>& A = P.arr[w];
                const int ks = q + C::h0(w) + 1;
                const bool in_range = ks >= k_lo(w) && ks <= k_hi(w);               // uniform
                const char* const src = (const char*)(A.p + (long long)b * A.bstride + plane_src(w, ks));
                T* const dst = lds + C::gbase(w) + slot_of(w, ks) * C::gstride(w) + C::gin(w);
#pragma unroll
                for (int k = 0; k < C::ni(w); ++k, ++item) {
                    if ((item % NWAVE) != wave || !in_range) continue;
                    if (k * kWave + lane < C::nch(w)) lds_dma16<T>(src + doff[item / NWAVE], dst + k * kWave * C::V, lane);
                }
            }
        };
        // every wavefront waits until at most `N` of its VMEM operations are in flight -- they retire in issue order, so everything older than the last N
        // (output stores of this half-step) has landed, the transfers included -- then the workgroup meets (nothing else orders a ds_read behind an LDS-DMA)
        auto landed_and_barrier = [&](auto n_tag) {
#ifdef __HIP_DEVICE_COMPILE__
            constexpr int N = decltype(n_tag)::value;
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory");
#else
            __syncthreads();
#endif
        };
        for (int q = pb - 2 * C::H0MAX - 1; q < pb; ++q) feed(q);                   // warm-up: planes pb - h0 .. pb + h0 of every window, all in flight at once
        landed_and_barrier(std::integral_constant<int, 0>{});
        for (int p = pb; p < pe; ++p) {
            feed(p);
            compute_plane(p);
            // (a half-step issues S x NOUT output stores per wavefront behind its transfers; half of that is assumed -- a wait that is too strong costs a few
            // cycles, one that is too weak reads a plane that has not landed)
            landed_and_barrier(std::integral_constant<int, (S * NOUT) / 2>{});
            report_plane(p);
        }
    } else if (DIM == 3) {
        // window w's first staged plane pb - h0 is requested in half-step pb - 2 h0 - 2: 2 H0MAX + 2 warm-up half-steps fill the rings
        int p = pb - 2 * C::H0MAX - 2;
        for (; p < pb; p += 2) {
            half_step(p, false, RA, tailA, RB, tailB);
            half_step(p + 1, false, RB, tailB, RA, tailA);
        }
        for (; p + 1 < pe; p += 2) {
            half_step(p, true, RA, tailA, RB, tailB);
            half_step(p + 1, true, RB, tailB, RA, tailA);
        }
        if (p < pe) half_step(p, true, RA, tailA, RB, tailB);
    } else {
        // 2-D: one plane. Request it (half-step -2), let it enter the windows (half-step -1), compute (half-step 0).
        load_planes(-2, RA, tailA);
        if (CONSTS && has_const) patch_planes(-1, RA, tailA);
        store_planes(-1, RA, tailA);
        __syncthreads();
        compute_plane(0);
        __syncthreads();
        report_plane(0);
    }
}

// ---- fix-up: the flagged workgroups' samples with the gather code of advect.hip ------------------------------------------------------
template <typename T, int KIND, int DIM, int T1>
__global__ __launch_bounds__(kBlock) void advect_win_fixup_kernel(VelGrid g, ScalarBc sb, CComp3a<T> field, const T* __restrict__ sfield, CComp3a<T> vel,
                                                                  CComp3a<T> fwd3, const T* __restrict__ fwd1, T* o0, T* o1, T* o2, T dt, T ch,
                                                                  int tiles1, int tiles2, int nblk, FixList fix) {
    using C = WinTile<T, KIND, DIM, T1>;
    constexpr int A0 = 3 - DIM;
    const int tid = threadIdx.x, tx = tid % C::T2, ty = tid / C::T2;
    T* const outp[3] = {o0, o1, o2};
    const int count = fix_count(fix);
    if (blockIdx.x == 0 && tid == 0) fix_publish(fix, count);
    for (int item = blockIdx.x; item < count; item += gridDim.x) {
        const FixItem e = fix.items[item];
        const int b = e.wg / nblk;
        const int bid = xcd_order(e.wg - b * nblk, nblk);
        const int t2 = bid % tiles2;
        const int t1 = (bid / tiles2) % tiles1;
        const int p = e.plane;
        for (int s = 0; s < C::S; ++s) {
            const int j1 = t1 * T1 + ty + s * C::TY, j2 = t2 * C::T2 + tx;
            const int idx[3] = {p, j1, j2};
            if (KIND == WK_MC_STAG) {
#pragma unroll
                for (int ca = A0; ca < 3; ++ca) {
                    if (p >= g.cn[ca][0] || j1 >= g.cn[ca][1] || j2 >= g.cn[ca][2]) continue;
                    const int n[3] = {g.cn[ca][0], g.cn[ca][1], g.cn[ca][2]};
                    const int f = (p * n[1] + j1) * n[2] + j2;
                    T cb_[3] = {T(0), T(0), T(0)}, cf_[3];
                    if (ca == 0) face_disp<T, DIM, 0>(g, vel, b, idx, f, dt, cf_);
                    else if (ca == 1) face_disp<T, DIM, 1>(g, vel, b, idx, f, dt, cf_);
                    else face_disp<T, DIM, 2>(g, vel, b, idx, f, dt, cf_);
#pragma unroll
                    for (int a = A0; a < 3; ++a) cb_[a] = -cf_[a];
                    int bc[3][2];
                    T cv[3][2];
                    comp_rule<T>(g, ca, bc, cv);
                    AxisPair<T> ax[3];
                    T fr[3];
                    const long long total = g.ccells[ca];
                    const T* __restrict__ F = field.p[ca] + (long long)b * total;
                    const T* __restrict__ W = fwd3.p[ca] + (long long)b * total;
                    lookup_pairs_rel<T, DIM>(idx, cf_, n, bc, cv, ax, fr);
                    const T bwd = gather_multilinear<T, DIM>(W, ax, fr);
                    const T nv = mc_correct(W[f], ch, F[f], bwd);
                    cb_[ca] += (T)g.off[ca] - T(0.5);
                    lookup_pairs_rel<T, DIM>(idx, cb_, n, bc, cv, ax, fr);
                    T lo, hi;
                    gather_minmax<T, DIM>(F, ax, lo, hi);
                    outp[ca][(long long)b * total + f] = nv < lo ? lo : (nv > hi ? hi : nv);
                }
            } else {
                if (p >= g.n[0] || j1 >= g.n[1] || j2 >= g.n[2]) continue;
                const int n[3] = {g.n[0], g.n[1], g.n[2]};
                const int f = (p * n[1] + j1) * n[2] + j2;
                T u[3];
                center_velocity<T, DIM>(g, vel, b, idx, u);
                T cb_[3] = {T(0), T(0), T(0)}, cf_[3] = {T(0), T(0), T(0)};
#pragma unroll
                for (int a = A0; a < 3; ++a) {
                    const T sft = u[a] * (dt * (T)g.rdx[a]);
                    cb_[a] = -sft;
                    cf_[a] = sft;
                }
                int bc[3][2];
                T cv[3][2];
                scalar_rule<T>(sb, bc, cv);
                AxisPair<T> ax[3];
                T fr[3];
                const T* __restrict__ F = sfield + (long long)b * g.cells;
                if (KIND == WK_SL_CEN) {
                    lookup_pairs_rel<T, DIM>(idx, cb_, n, bc, cv, ax, fr);
                    o0[(long long)b * g.cells + f] = gather_multilinear<T, DIM>(F, ax, fr);
                } else {
                    const T* __restrict__ W = fwd1 + (long long)b * g.cells;
                    lookup_pairs_rel<T, DIM>(idx, cf_, n, bc, cv, ax, fr);
                    const T bwd = gather_multilinear<T, DIM>(W, ax, fr);
                    const T nv = mc_correct(W[f], ch, F[f], bwd);
                    lookup_pairs_rel<T, DIM>(idx, cb_, n, bc, cv, ax, fr);
                    T lo, hi;
                    gather_minmax<T, DIM>(F, ax, lo, hi);
                    o0[(long long)b * g.cells + f] = nv < lo ? lo : (nv > hi ? hi : nv);
                }
            }
        }
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------------------
template <typename T>
static WinArray<T> component_array(const VelGrid& g, int c, const void* p) {
    WinArray<T> a;
    memset(&a, 0, sizeof(a));
    a.p = (const T*)p;
    a.bstride = g.ccells[c];
    for (int ax = 0; ax < 3; ++ax) {
        a.n[ax] = g.cn[c][ax];
        for (int s = 0; s < 2; ++s) {
            a.bc[ax][s] = ax < g.ax0 ? PHIHIP_BC_PERIODIC : g.bc[ax][s];
            a.cv[ax][s] = (T)g.bcv[ax][s][c];
        }
    }
    return a;
}
template <typename T>
static WinArray<T> scalar_array(const VelGrid& g, const ScalarBc& sb, const void* p) {
    WinArray<T> a;
    memset(&a, 0, sizeof(a));
    a.p = (const T*)p;
    a.bstride = g.cells;
    for (int ax = 0; ax < 3; ++ax) {
        a.n[ax] = g.n[ax];
        for (int s = 0; s < 2; ++s) {
            a.bc[ax][s] = ax < g.ax0 ? PHIHIP_BC_PERIODIC : sb.bc[ax][s];
            a.cv[ax][s] = (T)sb.val[ax][s];
        }
    }
    return a;
}

struct WinCall {
    const void* field[3];      // MC_STAG: the advected components (= the velocity)
    const void* sfield;        // centred kinds: the scalar
    const void* vel[3];
    const void* fwd[3];        // MC_STAG: forward-pass components; MC_CEN: fwd[0]
    void* out[3];
    const ScalarBc* sb;
    double dt, ch;
    int halo, kind;            // reach of the lookups of the centred kinds (1 / 2); AdvKind of the adaptive-reach bookkeeping
};

template <typename T, int KIND, int DIM, int T1, int OFFM, bool CONSTS, int H, bool DMA = false>
static int launch_win_inst(phihip_ctx* ctx, const GridView& v, const VelGrid& g, const WinCall& call, hipStream_t s) {
    using C = WinTile<T, KIND, DIM, T1, H, DMA>;
    constexpr int A0 = 3 - DIM;
    WinParams<T> P;
    memset(&P, 0, sizeof(P));
    int nmax[3] = {1, 1, 1};
    ScalarBc sb0;
    memset(&sb0, 0, sizeof(sb0));
    const ScalarBc& sb = call.sb ? *call.sb : sb0;
    if (KIND == WK_MC_STAG) {
        for (int c = A0; c < 3; ++c) {
            P.arr[c - A0] = component_array<T>(g, c, call.vel[c]);
            P.arr[DIM + c - A0] = component_array<T>(g, c, call.fwd[c]);
            P.out[c] = (T*)call.out[c];
            P.ostride[c] = g.ccells[c];
            for (int a = 0; a < 3; ++a) { P.on[c][a] = g.cn[c][a]; nmax[a] = g.cn[c][a] > nmax[a] ? g.cn[c][a] : nmax[a]; }
        }
    } else {
        P.arr[0] = scalar_array<T>(g, sb, call.sfield);
        int wv0 = 1;
        if (KIND == WK_MC_CEN) { P.arr[1] = scalar_array<T>(g, sb, call.fwd[0]); wv0 = 2; }
        for (int c = A0; c < 3; ++c) P.arr[wv0 + c - A0] = component_array<T>(g, c, call.vel[c]);
        P.out[0] = (T*)call.out[0];
        P.ostride[0] = g.cells;
        for (int a = 0; a < 3; ++a) { P.on[0][a] = g.n[a]; nmax[a] = g.n[a]; }
    }
    for (int a = 0; a < 3; ++a) {
        P.off[a] = g.off[a];
        P.shift[a] = (T)call.dt * (T)g.rdx[a];
    }
    P.ch = (T)call.ch;
    const int tiles1 = ceil_div(nmax[1], T1), tiles2 = ceil_div(nmax[2], C::T2);
    auto kernel = advect_win_kernel<T, KIND, DIM, T1, OFFM, CONSTS, H, DMA>;
    ctx->adv_last_dma = DMA ? 1 : 0;
    // LDS beyond the 64 KB a kernel gets by default: opt in once per instantiation and device
    static bool attr_set[16] = {false};
    bool& done = attr_set[ctx->device >= 0 && ctx->device < 16 ? ctx->device : 0];
    if (!done) {
        if (C::BYTES > 48 * 1024)
            PHIHIP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::BYTES));
        done = true;
    }
    int chunk = 1;
    if (DIM == 3) {
        // chunks of planes: every chunk stages 2 h0 + 1 extra planes of its windows; a launch that needs 1 < rounds < 2 of resident workgroups
        // costs two rounds. Score = (slot efficiency of the last round) x (useful / staged planes), as for the self-advection (advect_tile.hip)
        static int occ_dev[16] = {0};
        int& occ = occ_dev[ctx->device >= 0 && ctx->device < 16 ? ctx->device : 0];
        if (occ == 0) {
            int n = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, kBlock, C::BYTES) != hipSuccess || n < 1) n = 1;
            occ = n;
        }
        const double slots = (double)occ * ctx->num_cu;
        const long long tiles = (long long)tiles1 * tiles2 * v.batch;
        double best = -1.0;
        for (int c = 1; c <= nmax[0]; ++c) {
            const int ch = ceil_div(nmax[0], c);
            if (c > 1 && ch < 4) break;
            if (ceil_div(nmax[0], ch) != c) continue;
            const double rounds = (double)tiles * c / slots;
            const double eff = rounds / ceil(rounds - 1e-9);
            const double score = eff * ch / (ch + 2 * C::H0MAX + 1) * (rounds >= 2.0 ? 1.0 : (rounds >= 1.0 ? 0.97 : 0.9));
            if (score > best * 1.0001) { best = score; chunk = ch; }
        }
        if (ctx->adv_chunk > 0) chunk = ctx->adv_chunk < nmax[0] ? ctx->adv_chunk : nmax[0];
    }
    const int chunks0 = DIM == 3 ? ceil_div(nmax[0], chunk) : 1;
    const int nblk = tiles1 * tiles2 * chunks0;
    void* dump = nullptr;
    PHIHIP_TRY(prepare_fixlist(ctx, (long long)tiles1 * tiles2 * nmax[0] * v.batch, s, &P.fix, &dump, call.kind));
    P.chunk = chunk; P.tiles1 = tiles1; P.tiles2 = tiles2; P.nblk = nblk; P.nmax0 = nmax[0];
    P.dump = (T*)dump;
    hipLaunchKernelGGL(kernel, dim3(nblk, v.batch), dim3(kBlock), C::BYTES, s, P);
    CComp3a<T> ff{{(const T*)call.field[0], (const T*)call.field[1], (const T*)call.field[2]}};
    CComp3a<T> vv{{(const T*)call.vel[0], (const T*)call.vel[1], (const T*)call.vel[2]}};
    CComp3a<T> ww{{(const T*)call.fwd[0], (const T*)call.fwd[1], (const T*)call.fwd[2]}};
    const int fgrid = P.fix.cap < kFixupBlocks ? P.fix.cap : kFixupBlocks;
    hipLaunchKernelGGL((advect_win_fixup_kernel<T, KIND, DIM, T1>), dim3(fgrid), dim3(kBlock), 0, s, g, sb, ff, (const T*)call.sfield, vv, ww,
                       (const T*)call.fwd[0], (T*)call.out[0], (T*)call.out[1], (T*)call.out[2], (T)call.dt, (T)call.ch, tiles1, tiles2, nblk, P.fix);
    return PHIHIP_OK;
}

template <typename T, int KIND, int DIM, int T1, int OFFM, int H>
static int launch_win_off(phihip_ctx* ctx, const GridView& v, const VelGrid& g, const WinCall& call, hipStream_t s) {
    bool consts = false;
    for (int a = v.ax0; a < 3; ++a) {
        consts = consts || v.bc[a][0] == PHIHIP_BC_CLOSED || v.bc[a][1] == PHIHIP_BC_CLOSED;
        if (call.sb) consts = consts || call.sb->bc[a][0] == PHIHIP_BC_CLOSED || call.sb->bc[a][1] == PHIHIP_BC_CLOSED;
    }
    if constexpr (OFFM == 0) {
        if constexpr (DIM == 3 && KIND == WK_MC_STAG) {
            // r6: regular grids fill the windows by LDS-DMA (WinTile DMA): no CLOSED side, the fast axis periodic with rows of whole 16-byte vectors, aligned arrays
            // (the forward pass lives in the context's scratch: 256-byte aligned)
            constexpr int V = 16 / (int)sizeof(T);
            bool regular = !consts && ctx->adv_dma != 0 && !v.unaligned && v.bc[2][0] == PHIHIP_BC_PERIODIC;
            for (int c = 0; c < 3; ++c) regular = regular && g.cn[c][2] % V == 0 && ((size_t)call.vel[c] % 16 == 0) && ((size_t)call.fwd[c] % 16 == 0);
            if (regular) {
                // (experiment switch PHIHIP_WIN_DMA_ROWS=4: 4-row tiles -- 47 KB of LDS, three workgroups per CU instead of two at 100 VGPRs)
                static const int rows = [] { const char* e = getenv("PHIHIP_WIN_DMA_ROWS"); return e && e[0] == '4' ? 4 : 8; }();
                if constexpr (sizeof(T) == 4 && T1 == 8) {
                    if (rows == 4) return launch_win_inst<T, KIND, DIM, 4, OFFM, false, H, true>(ctx, v, g, call, s);
                }
                return launch_win_inst<T, KIND, DIM, T1, OFFM, false, H, true>(ctx, v, g, call, s);
            }
        }
        if (!consts) return launch_win_inst<T, KIND, DIM, T1, OFFM, false, H>(ctx, v, g, call, s);
    }
    return launch_win_inst<T, KIND, DIM, T1, OFFM, true, H>(ctx, v, g, call, s);
}

template <typename T, int KIND, int DIM, int T1, int H>
static int launch_win(phihip_ctx* ctx, const GridView& v, const VelGrid& g, const WinCall& call, hipStream_t s) {
    const int m = (g.off[0] & 1) | ((g.off[1] & 1) << 1) | ((g.off[2] & 1) << 2);     // (2-D: off[0] = 0)
    switch (m) {
        case 0: return launch_win_off<T, KIND, DIM, T1, 0, H>(ctx, v, g, call, s);
        case 2: return launch_win_off<T, KIND, DIM, T1, 2, H>(ctx, v, g, call, s);
        case 4: return launch_win_off<T, KIND, DIM, T1, 4, H>(ctx, v, g, call, s);
        case 6: return launch_win_off<T, KIND, DIM, T1, 6, H>(ctx, v, g, call, s);
        default: break;
    }
    if (DIM == 3) switch (m) {
        case 1: return launch_win_off<T, KIND, DIM, T1, (DIM == 3 ? 1 : 0), H>(ctx, v, g, call, s);
        case 3: return launch_win_off<T, KIND, DIM, T1, (DIM == 3 ? 3 : 0), H>(ctx, v, g, call, s);
        case 5: return launch_win_off<T, KIND, DIM, T1, (DIM == 3 ? 5 : 0), H>(ctx, v, g, call, s);
        case 7: return launch_win_off<T, KIND, DIM, T1, (DIM == 3 ? 7 : 0), H>(ctx, v, g, call, s);
        default: break;
    }
    set_error("advect: unexpected face-offset pattern %d", m);
    return PHIHIP_ERR_BAD_ARG;
}

template <int KIND>
static int run_win(phihip_ctx* ctx, const GridView& v, const WinCall& call, hipStream_t s) {
    const VelGrid g = make_velgrid(v);
    // 2-D grids keep the gather kernels: a 2-D workgroup has ONE plane, i.e. fill -> barrier -> compute without any overlap and 512 samples
    // per fill; measured on the same box (profiles/r04_time_frow_session_c.jsonl) 8 x 512^2: semi_lagrangian(s, v) 12.9 us gather / 24.0 us
    // windows, mac_cormack(s, v) 28.3 / 54.4, mac_cormack(v, v) 55.0 / 54.6; 2048^2: 22.5 / 40.5, 49.5 / 91.3, 94.4 / 93.1
    if (v.rank != 3 && !ctx->adv_win_2d) return PHIHIP_ERR_UNSUPPORTED;
    for (int c = v.ax0; c < 3; ++c)
        for (int a = v.ax0; a < 3; ++a)
            if (v.cn[c][a] < 4 || v.n[a] < 4) return PHIHIP_ERR_UNSUPPORTED;   // a window wider than the axis: the caller keeps the gather kernels
    LaunchScope ls(ctx, PHIHIP_K_ADVECT, s);
    const bool wide = KIND != WK_MC_STAG && call.halo >= 2;           // lookups of the centred kinds reach two cells
    if (v.dtype == PHIHIP_F64) {
        if (v.rank == 3) { if (wide) PHIHIP_TRY((launch_win<double, KIND, 3, 8, (KIND != WK_MC_STAG ? 2 : 1)>(ctx, v, g, call, s))); else PHIHIP_TRY((launch_win<double, KIND, 3, 8, 1>(ctx, v, g, call, s))); }
        else PHIHIP_TRY((launch_win<double, KIND, 2, 8, 1>(ctx, v, g, call, s)));
    } else {
        if (v.rank == 3) { if (wide) PHIHIP_TRY((launch_win<float, KIND, 3, 8, (KIND != WK_MC_STAG ? 2 : 1)>(ctx, v, g, call, s))); else PHIHIP_TRY((launch_win<float, KIND, 3, 8, 1>(ctx, v, g, call, s))); }
        else PHIHIP_TRY((launch_win<float, KIND, 2, 8, 1>(ctx, v, g, call, s)));
    }
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

// correction pass of mac_cormack(v, v, dt): fwd = the semi-Lagrangian result (advect_tile.hip), out = the corrected, limited velocity
int run_mc_correct_self_tiled(phihip_ctx* ctx, const GridView& v, const void* const vel[3], const void* const fwd[3], void* const out[3], double dt,
                              double ch, int kind, hipStream_t s) {
    WinCall c;
    memset(&c, 0, sizeof(c));
    for (int k = 0; k < 3; ++k) { c.field[k] = vel[k]; c.vel[k] = vel[k]; c.fwd[k] = fwd[k]; c.out[k] = out[k]; }
    c.dt = dt; c.ch = ch; c.halo = 1; c.kind = kind;
    return run_win<WK_MC_STAG>(ctx, v, c, s);
}

int run_advect_centered_tiled(phihip_ctx* ctx, const GridView& v, const void* sfield, const ScalarBc& sb, const void* const vel[3], void* out, double dt,
                              int halo, int kind, hipStream_t s) {
    WinCall c;
    memset(&c, 0, sizeof(c));
    c.sfield = sfield; c.sb = &sb; c.out[0] = out;
    for (int k = 0; k < 3; ++k) c.vel[k] = vel[k];
    c.dt = dt; c.halo = halo; c.kind = kind;
    return run_win<WK_SL_CEN>(ctx, v, c, s);
}

int run_mc_correct_centered_tiled(phihip_ctx* ctx, const GridView& v, const void* sfield, const ScalarBc& sb, const void* const vel[3], const void* fwd,
                                  void* out, double dt, double ch, int halo, int kind, hipStream_t s) {
    WinCall c;
    memset(&c, 0, sizeof(c));
    c.sfield = sfield; c.sb = &sb; c.out[0] = out; c.fwd[0] = fwd;
    for (int k = 0; k < 3; ++k) c.vel[k] = vel[k];
    c.dt = dt; c.ch = ch; c.halo = halo; c.kind = kind;
    return run_win<WK_MC_CEN>(ctx, v, c, s);
}

}  // namespace phihip
