// advect_tile.hip -- semi-Lagrangian SELF-advection of the staggered velocity (advect.semi_lagrangian(v, v, dt),
// /root/reference phi/physics/advect.py:156-179 with the euler back-trace :20-24) as ONE launch for all D components, fed from LDS.
//
// A 256-thread workgroup owns a (T1 x T2) tile of stored indices of the two fast axes and marches over a chunk of a0 planes. A ring
// of 2H+2 planes of EVERY velocity component -- tile + a halo of H samples, boundary rule (wrap / clamp / constant, the last axis
// outside a constant side wins like PhiML's sequential padding) already applied while filling -- lives in LDS:
//   * the fill is cooperative and coalesced: per plane each thread issues ~(T1+2H)/TY + 1 loads per component along the fast axis
//     (the gather kernels of advect.hip issue 17 scattered loads per SAMPLE: they are bound by the address unit, not by HBM);
//   * the 4-point means of the other components at a face (phi/field/_resample.py:279-287,341-364) and the 2^D taps of the
//     multilinear lookup (phi/field/_resample.py:257-259) are LDS reads at uniform / compile-time offsets from one address;
//   * a lookup that leaves the staged window (|displacement| >= H cells) cannot be served from LDS: the workgroup raises a flag and
//     `advect_self_fixup_kernel` (same grid, launched right behind; workgroups whose flag is clear exit at once) recomputes the samples
//     of exactly those workgroups with the boundary-resolved global gather of advect_common.hpp -- same arithmetic, so the result does
//     not depend on the path, and the LDS kernel carries no gather code (it was two thirds of its instructions and most of its
//     scalar-register pressure);
//   * plane p+H+2 is requested before plane p is computed and enters the ring after plane p+1 (HBM latency exceeds one plane of
//     arithmetic; two register sets alternate); one barrier per plane.
// Algorithmic traffic 2 D words per cell; the kernel reads each velocity sample once per workgroup (+ halo) and writes each once.
#include <math.h>

#include <type_traits>

#include "advect_common.hpp"

#ifndef PHIHIP_DMA_FENCE
#define PHIHIP_DMA_FENCE 0      // r5 same-box A/B (profiles/r05_time_advect_dma_fence.jsonl): with 48-62 VGPRs and the occupancy set by LDS the scheduler may overlap the samples' LDS reads: -5 %
#endif
#ifndef PHIHIP_TILE_FENCE
#define PHIHIP_TILE_FENCE 1
#endif
#ifndef PHIHIP_DMA_UNROLL_S
#define PHIHIP_DMA_UNROLL_S 0
#endif


namespace phihip {

template <typename T, int DIM, int H, int T1>
struct AdvTile {
    static constexpr int T2 = sizeof(T) == 4 ? 64 : 32;   // tile columns = lanes along the fast axis (256 B rows)
    static constexpr int TY = kBlock / T2;                // thread rows
    static constexpr int S = T1 / TY;                     // tile positions per thread and plane
    static constexpr int P1 = T1 + 2 * H, P2 = T2 + 2 * H;
    static constexpr int PLANE = P1 * P2;
    static constexpr int NC = DIM;                        // staged components
    static constexpr int NP = DIM == 3 ? 2 * H + 2 : 1;   // ring slots (planes p-H .. p+H in use, one being refilled)
    static constexpr int KP = (P1 + TY - 1) / TY;         // fill passes of a thread per component and plane
    static constexpr int NTAIL = NC * P1 * 2 * H;         // halo columns right of the T2 main columns: one element per thread
    static_assert(T1 % TY == 0, "tile rows must be a multiple of the thread rows");
    static_assert(NTAIL <= kBlock, "tail elements must fit one per thread");
    static_assert((size_t)NC * NP * PLANE * sizeof(T) <= 65536, "static LDS limit");
};

// what the kernel needs of the staggered layout, in the element type (VelGrid carries doubles and more: converting / spilling its ~90
// scalar registers inside the plane loop cost more than the arithmetic)
template <typename T>
struct TileGrid {
    int cn[3][3];        // [component][axis] stored samples
    int off[3];          // physical face number of stored index 0
    int bc[3][2];
    T bcv[3][2][3];      // [axis][side][component] constant of CLOSED sides
    T shift[3];          // dt / dx: back-trace displacement in index units per unit velocity
    long long ccells[3];
};

template <typename T>
static TileGrid<T> make_tilegrid(const VelGrid& g, double dt) {
    TileGrid<T> t;
    memset(&t, 0, sizeof(t));
    for (int a = 0; a < 3; ++a) {
        t.off[a] = g.off[a];
        t.shift[a] = (T)dt * (T)g.rdx[a];
        t.ccells[a] = g.ccells[a];
        for (int c = 0; c < 3; ++c) t.cn[a][c] = g.cn[a][c];
        for (int side = 0; side < 2; ++side) {
            t.bc[a][side] = g.bc[a][side];
            for (int c = 0; c < 3; ++c) t.bcv[a][side][c] = (T)g.bcv[a][side][c];
        }
    }
    return t;
}

// stored index i of an axis with n entries under the velocity padding rule: the index to read, or -1 / -2 when the lower / upper
// CONSTANT side supplies the value (PhiML pads with the constant; clamp = BOUNDARY; wrap = PERIODIC)
__device__ __forceinline__ int pad_index(int i, int n, int code_lo, int code_hi) {
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

// 1 / 2 when index i of an axis with n entries lies beyond the lower / upper CONSTANT (closed) side, else 0 (no wrap arithmetic: the value
// itself was loaded from a valid address already, this only decides whether the constant replaces it)
__device__ __forceinline__ int const_side(int i, int n, int code_lo, int code_hi) {
    return (i < 0 && code_lo == PHIHIP_BC_CLOSED) ? 1 : ((i >= n && code_hi == PHIHIP_BC_CLOSED) ? 2 : 0);
}

// keeps the instruction scheduler from hoisting every sample's LDS reads to the top of the plane (that costs > 200 registers)
__device__ __forceinline__ void sched_fence() {
#ifdef __HIP_DEVICE_COMPILE__
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// hides a value from the optimiser (no instruction): what is computed from it cannot be hoisted out of the loop and kept in a register
__device__ __forceinline__ void opaque_int(int& v) {
#ifdef __HIP_DEVICE_COMPILE__
    asm volatile("" : "+v"(v));
#endif
}

__device__ __forceinline__ void opaque_uint(unsigned& v) {
#ifdef __HIP_DEVICE_COMPILE__
    asm volatile("" : "+v"(v));
#endif
}

// min(max(x, lo), hi) as ONE v_med3 (NaN -> lo, which the callers treat as "outside")
__device__ __forceinline__ float clamp_real(float x, float lo, float hi) {
#ifdef __HIP_DEVICE_COMPILE__
    return __builtin_amdgcn_fmed3f(x, lo, hi);
#else
    return x >= lo ? (x <= hi ? x : hi) : lo;
#endif
}
__device__ __forceinline__ double clamp_real(double x, double lo, double hi) { return x >= lo ? (x <= hi ? x : hi) : lo; }

// CONSTS: the grid has a CLOSED (constant) side somewhere, i.e. a staged window may need wall values patched in. Without one -- periodic /
// open boxes, the benchmark configuration -- the whole patch path is compiled out: its plane-invariant per-thread predicates were hoisted
// out of the plane loop as ~100 lane masks in SGPR pairs, 325 of them spilled to VGPR lanes and reloaded every plane (r3: 7631 -> 3210
// instructions, 148 -> 88 VGPRs, 3 -> 5 waves per SIMD for the fp32 halo-1 kernel).
template <typename T, int DIM, int H, int T1, int OFFM, bool CONSTS>
__global__ __launch_bounds__(kBlock, (H == 1 && DIM == 3 && T1 == 8) ? (sizeof(T) == 4 ? 4 : 2) : 2) void advect_self_tile_kernel(TileGrid<T> g, CComp3a<T> vel, T* __restrict__ o0, T* __restrict__ o1,
                                                                  T* __restrict__ o2, int chunk, int tiles1, int tiles2, int nblk, int nmax0,
                                                                  FixList fix, T* __restrict__ dump) {
    using C = AdvTile<T, DIM, H, T1>;
    constexpr int A0 = 3 - DIM;
    constexpr int T2 = C::T2, TY = C::TY, S = C::S, P1 = C::P1, P2 = C::P2, PLANE = C::PLANE, NC = C::NC, NP = C::NP, KP = C::KP;
    // physical face number of stored index 0 per axis (1 below a CLOSED side): compile-time, so that every tap of the 4-point means is
    // an immediate LDS offset from three per-position addresses (27 scalar offsets per plane otherwise -- they spilled)
    constexpr int OFF[3] = {(OFFM >> 0) & 1, (OFFM >> 1) & 1, (OFFM >> 2) & 1};
    __shared__ T lds[NC * NP * PLANE];
    __shared__ int slow_sh[2];          // "a lookup of plane p left the window", by the parity of p (read by thread 0 after the plane's barrier)

    const int tid = threadIdx.x, tx = tid % T2, ty = tid / T2;
    const int b = blockIdx.y;
    int bid = blockIdx.x;
    bid = xcd_order(bid, nblk);   // neighbouring tiles share an XCD's L2
    const int t2 = bid % tiles2;
    const int t1 = (bid / tiles2) % tiles1;
    const int c0 = bid / (tiles2 * tiles1);
    const int lo1 = t1 * T1, lo2 = t2 * T2;                            // tile origin (stored indices, the same for every component)
    const int pb = DIM == 3 ? c0 * chunk : 0;
    const int pe = DIM == 3 ? min(pb + chunk, nmax0) : 1;
    T* const outp[3] = {o0, o1, o2};
    if (tid < 2) slow_sh[tid] = 0;      // (a barrier of the ring warm-up / of the 2-D fill lies between this and the first plane of samples)
    if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) *fix.next = 0;      // the work list's other counter, for the launch after this one
    // per-thread output bookkeeping: element offset of position s = 0 in plane 0 per component, plane strides, and one bit per
    // (position, component): the sample exists in that component's array
    unsigned obase[3] = {0, 0, 0};      // in-plane element offset (unsigned 32-bit + uniform 64-bit base: address arithmetic stays scalar)
    long long pstride[3] = {0, 0, 0};
    unsigned vbits = 0;
#pragma unroll
    for (int c = A0; c < 3; ++c) {
        obase[c] = (unsigned)((lo1 + ty) * g.cn[c][2] + lo2 + tx);
        pstride[c] = (long long)g.cn[c][1] * g.cn[c][2];
#pragma unroll
        for (int k = 0; k < S; ++k)
            if (lo1 + ty + k * TY < g.cn[c][1] && lo2 + tx < g.cn[c][2]) vbits |= 1u << (k * 3 + c);
    }

    // Does this workgroup's window reach beyond a CLOSED side (constant padding)? Uniform, so interior tiles (and periodic / open
    // domains altogether) run the fill without a single select.
    bool has_const = false;
#pragma unroll
    for (int c = A0; CONSTS && c < 3; ++c) {
        has_const = has_const || (g.bc[1][0] == PHIHIP_BC_CLOSED && lo1 - H < 0) || (g.bc[1][1] == PHIHIP_BC_CLOSED && lo1 + T1 + H > g.cn[c][1]) ||
                    (g.bc[2][0] == PHIHIP_BC_CLOSED && lo2 - H < 0) || (g.bc[2][1] == PHIHIP_BC_CLOSED && lo2 + T2 + H > g.cn[c][2]);
        if (DIM == 3) has_const = has_const || (g.bc[0][0] == PHIHIP_BC_CLOSED && pb - H < 0) || (g.bc[0][1] == PHIHIP_BC_CLOSED && pe + H > g.cn[c][0]);
    }

    // ---- per-thread fill descriptors (plane-invariant) -------------------------------------------------------------------------
    // element kp of component c: row ty + kp TY, column tx of the window. eoff = in-plane element offset after wrap / clamp;
    // ccode: 0 = stored sample, 1 / 2 = the lower / upper CONSTANT side of a2 supplies the value (rows: recomputed in the cold path)
    unsigned eoff[3][KP];
    const bool last_ok = ty + (KP - 1) * TY < P1;        // only the last pass can run past the window's rows
#pragma unroll
    for (int c = A0; c < 3; ++c) {
        const int k = pad_index(lo2 - H + tx, g.cn[c][2], g.bc[2][0], g.bc[2][1]);
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
            const int j = pad_index(lo1 - H + ty + kp * TY, g.cn[c][1], g.bc[1][0], g.bc[1][1]);
            eoff[c][kp] = (unsigned)((j < 0 ? 0 : j * g.cn[c][2]) + (k < 0 ? 0 : k)) * (unsigned)sizeof(T);   // BYTES (planes are < 4 GiB)
        }
        if (!last_ok) eoff[c][KP - 1] = eoff[c][0];   // (row past the window: the load still runs -- see load_plane -- on a valid address)
    }
    // tail element (halo columns T2 .. T2+2H-1 of every row): component, row and column of THIS thread
    const bool has_tail = tid < C::NTAIL;
    const int tail_ci = has_tail ? tid / (P1 * 2 * H) : 0;
    const int tail_c = A0 + tail_ci;
    const int tail_r = has_tail ? (tid % (P1 * 2 * H)) / (2 * H) : 0;
    const int tail_q = T2 + tid % (2 * H);
    unsigned tail_off = 0;
    int tail_n0 = 1, tail_n1 = 1, tail_n2 = 1;
    const T* tail_base = vel.p[2];
#pragma unroll
    for (int c = A0; c < 3; ++c)     // (selects over the static component index: dynamic indexing of kernel arguments goes through scratch)
        if (c == tail_c) { tail_n0 = g.cn[c][0]; tail_n1 = g.cn[c][1]; tail_n2 = g.cn[c][2]; tail_base = vel.p[c] + (long long)b * g.ccells[c]; }
    if (has_tail) {
        const int kk = pad_index(lo2 - H + tail_q, tail_n2, g.bc[2][0], g.bc[2][1]);
        const int j = pad_index(lo1 - H + tail_r, tail_n1, g.bc[1][0], g.bc[1][1]);
        tail_off = (unsigned)((j < 0 ? 0 : j * tail_n2) + (kk < 0 ? 0 : kk)) * (unsigned)sizeof(T);
    }
    (void)tail_n0;

    // request plane i0 of every component into registers: plain loads; the constant sides are patched in afterwards (cold, uniform)
    // plane of component c that supplies staged plane i0: wrapped (every axis has >= 4 samples here, so one +-n suffices) or clamped;
    // beyond a CONSTANT side any valid plane is read and patched below. Branch-free scalar code: it runs every half-step.
    auto plane_src = [&](int c, int i0) -> long long {
        if (DIM == 2) return 0;
        const int n = g.cn[c][0];
        int w = i0;
        if (g.bc[0][0] == PHIHIP_BC_PERIODIC) { w += w < 0 ? n : 0; w -= w >= n ? n : 0; }
        w = min(max(w, 0), n - 1);
        return (long long)w * pstride[c];
    };
    // Every global load (and, in compute_plane, store) of the plane loop is unconditional: hipcc can then count the outstanding operations
    // and waits with s_waitcnt vmcnt(N > 0) for the register set that is due, while the younger requests stay in flight (with loads
    // under `if` it drains the queue, i.e. it also waits for the output stores it has just issued -- stencil_march.hpp has the story).
    auto load_plane = [&](int i0, T (&R)[3][KP], T& tailv) {
        long long psrc[3] = {0, 0, 0};
#pragma unroll
        for (int c = A0; c < 3; ++c) {
            psrc[c] = plane_src(c, i0);
            // uniform base + 32-bit byte offset per lane = ONE instruction (global_load ... saddr). The offset is hidden from the optimiser at
            // each use: hoisted out of the plane loop, zext(offset) becomes a 64-bit register pair and the load a 64-bit vector address add
            const char* __restrict__ base = (const char*)(vel.p[c] + (long long)b * g.ccells[c] + psrc[c]);
#pragma unroll
            for (int kp = 0; kp < KP; ++kp) {
                unsigned o = eoff[c][kp];
                opaque_uint(o);
                R[c][kp] = *(const T*)(base + o);
            }
        }
        {
            unsigned o = tail_off;
            opaque_uint(o);
            tailv = *(const T*)((const char*)(tail_base + (tail_c == 0 ? psrc[0] : (tail_c == 1 ? psrc[1] : psrc[2]))) + o);
        }
    };
    // constant sides are patched in when the plane enters the ring (cold, uniform: only workgroups whose window crosses a CLOSED side).
    // Every per-thread predicate is recomputed here from the (optimiser-opaque) thread coordinates: kept across planes they are lane
    // masks in SGPR pairs -- ~100 of them, spilled and reloaded every plane, also by the workgroups that never take this path.
    auto patch_plane = [&](int i0, T (&R)[3][KP], T& tailv) {
        int tyo = ty, txo = tx, tido = tid;
        opaque_int(tyo); opaque_int(txo); opaque_int(tido);
        // PhiML pads axis after axis: the LAST axis outside a constant side decides (a2 over a1 over a0)
#pragma unroll
        for (int c = A0; c < 3; ++c) {
            const int k = DIM == 3 ? const_side(i0, g.cn[c][0], g.bc[0][0], g.bc[0][1]) : 0;       // uniform
            // (values first, selects after: a select between two kernel-argument LOADS becomes a per-lane address + flat load)
            const T k00 = g.bcv[0][0][c], k01 = g.bcv[0][1][c], k10 = g.bcv[1][0][c], k11 = g.bcv[1][1][c], k20 = g.bcv[2][0][c], k21 = g.bcv[2][1][c];
            const T pv = k == 1 ? k00 : k01;
            const int cside = const_side(lo2 - H + txo, g.cn[c][2], g.bc[2][0], g.bc[2][1]);
            const T cv = cside == 1 ? k20 : k21;
#pragma unroll
            for (int kp = 0; kp < KP; ++kp) {
                T v = R[c][kp];
                const int j = const_side(lo1 - H + tyo + kp * TY, g.cn[c][1], g.bc[1][0], g.bc[1][1]);
                const T rv = j == 1 ? k10 : k11;
                v = k ? pv : v;
                v = j ? rv : v;
                v = cside ? cv : v;
                R[c][kp] = v;
            }
            if (tido < C::NTAIL && tido / (P1 * 2 * H) == c - A0) {
                const int tr = (tido % (P1 * 2 * H)) / (2 * H), tq = T2 + tido % (2 * H);
                const int rs = const_side(lo1 - H + tr, g.cn[c][1], g.bc[1][0], g.bc[1][1]), cs = const_side(lo2 - H + tq, g.cn[c][2], g.bc[2][0], g.bc[2][1]);
                const T rv = rs == 1 ? k10 : k11, cv2 = cs == 1 ? k20 : k21;
                tailv = k ? pv : tailv;
                tailv = rs ? rv : tailv;
                tailv = cs ? cv2 : tailv;
            }
        }
    };
    auto store_plane = [&](int slot, const T (&R)[3][KP], T tailv) {
#pragma unroll
        for (int c = A0; c < 3; ++c) {
            T* L = lds + ((c - A0) * NP + slot) * PLANE;
#pragma unroll
            for (int kp = 0; kp < KP; ++kp)
                if (kp < KP - 1 || last_ok) L[(ty + kp * TY) * P2 + tx] = R[c][kp];
        }
        // The tail element's ring position is recomputed from the thread index every time: kept live it was spilled to scratch, and a scratch
        // reload counts as a VMEM load -- the s_waitcnt vmcnt(0) behind it drained every request in flight, once per plane.
        int t = tid;
        opaque_int(t);
        if (t < C::NTAIL) lds[((t / (P1 * 2 * H)) * NP + slot) * PLANE + ((t % (P1 * 2 * H)) / (2 * H)) * P2 + T2 + t % (2 * H)] = tailv;
    };
    auto slot_of = [&](int i0) -> int {   // uniform
        if (DIM == 2) return 0;
        if ((NP & (NP - 1)) == 0) return i0 & (NP - 1);
        const int m = i0 % NP;
        return m < 0 ? m + NP : m;
    };


    // one plane of samples
    // one plane of samples. Per sample: displacement from the own component and the 4-point sums of the others (x 1/4 folded into the
    // scale), coordinate = index - displacement rounded like the reference rounds it (absolute index space), taps relative to the
    // sample's own LDS position, blend as nested lerps along a2, a1, a0.
    // FULL: every sample of the tile exists in every component's array (interior tiles; a uniform test per workgroup picks the instantiation): no
    // validity bits, no dump slot, the store is base (uniform) + 32-bit offset
    bool tile_full = pe <= g.cn[2][0] && lo1 + T1 <= g.cn[2][1] && lo2 + T2 <= g.cn[2][2];
#pragma unroll
    for (int c = A0; c < 2; ++c) tile_full = tile_full && pe <= g.cn[c][0] && lo1 + T1 <= g.cn[c][1] && lo2 + T2 <= g.cn[c][2];
    auto compute_plane = [&](int p, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int so_m = slot_of(p - 1) * PLANE, so_0 = slot_of(p) * PLANE, so_p = slot_of(p + 1) * PLANE;   // uniform element offsets
        bool slow_any = false;
#pragma unroll 1
        for (int s = 0; s < S; ++s) {
            const int center = (ty + s * TY + H) * P2 + tx + H;
            const int cen[3] = {center + so_m, center + so_0, center + so_p};     // this position in the planes p-1, p, p+1
#pragma unroll
            for (int ca = A0; ca < 3; ++ca) {
                // component x at (plane offset d0 in {-1,0,1}, row offset d1, column offset d2) from the sample: immediate offsets
                auto at = [&](int x, int d0, int d1, int d2) -> T { return lds[cen[d0 + 1] + ((x - A0) * NP * PLANE + d1 * P2 + d2)]; };
                T coord[3] = {T(0), T(0), T(0)};      // lookup position RELATIVE to the sample, in index units (r4: integer part and fraction are
                                                      // formed from the displacement alone, advect_common.hpp lookup_pairs_rel)
                coord[ca] = at(ca, 0, 0, 0) * -g.shift[ca];
#pragma unroll
                for (int cb = A0; cb < 3; ++cb) {
                    if (cb == ca) continue;
                    // component cb at this ca-face: cells (m-1, m) along ca, faces (s, s+1) along cb, in cb's stored indices:
                    // offsets (off[ca] - 1 + ia) along ca and (-off[cb] + ib) along cb (advect_common.hpp face_velocity)
                    T v4[2][2];
#pragma unroll
                    for (int ia = 0; ia < 2; ++ia)
#pragma unroll
                        for (int ib = 0; ib < 2; ++ib) {
                            int d[3] = {0, 0, 0};
                            d[ca] = OFF[ca] - 1 + ia;
                            d[cb] = -OFF[cb] + ib;
                            v4[ia][ib] = at(cb, d[0], d[1], d[2]);
                        }
                    // (a chain, not a tree: the tree is packed into v_pk_add_f32 + three register moves -- 5 instructions for 3 additions)
                    coord[cb] = (((v4[0][0] + v4[0][1]) + v4[1][0]) + v4[1][1]) * (T(-0.25) * g.shift[cb]);
                }
                // taps relative to the sample: rel = floor(displacement) in [-H, H-1] or the lookup leaves the window
                T fr[3] = {T(0), T(0), T(0)};
                int di[3] = {0, 0, 0};
                bool slow = false;
                int base0, base1;
                if (H == 1) {
                    // r5 (instruction diet, measured ~4 cycles per wave64 VALU instruction of any class on this part -- tools/micro/issue_rates.hip): inside the
                    // window the integer part of a displacement is -1 or 0, i.e. its SIGN. No floor / clamp / convert / multiply: the fraction is ONE
                    // v_fract, a tap offset one v_cndmask on the sign, "left the window" one v_max3 of the magnitudes and one compare (also true for NaN;
                    // a displacement of exactly -1 cell is sent to the fix-up pass, which computes the same value). The offsets are in the window
                    // whatever the displacement, so nothing has to be clamped: 27 -> 15 instructions per sample.
                    const T m01 = DIM == 3 ? fmax(fabs(coord[0]), fabs(coord[1])) : fabs(coord[1]);
                    slow = !(fmax(m01, fabs(coord[2])) < T(1));
#pragma unroll
                    for (int a = A0; a < 3; ++a) fr[a] = frac_part(coord[a]);
                    const int off = (ca - A0) * NP * PLANE + center + (coord[1] < T(0) ? -P2 : 0) + (coord[2] < T(0) ? -1 : 0);
                    base0 = base1 = off;
                    if (DIM == 3) {                                // lower tap plane p-1 or p, upper p or p+1
                        const bool down = coord[0] < T(0);
                        base1 += down ? so_0 : so_p;
                        base0 += down ? so_m : so_0;
                    }
                } else {
#pragma unroll
                    for (int a = A0; a < 3; ++a) {
                        const T fl = floor(coord[a]);
                        fr[a] = frac_part(coord[a]);                   // (the reach-1 form's rounding: every path gives a sample the same bits)
                        const T rel = fl;
                        const T relc = clamp_real(rel, T(-H), T(H - 1));
                        slow = slow || !(rel == relc);                 // also true for NaN
                        di[a] = (int)relc;
                    }
                    base0 = (ca - A0) * NP * PLANE + center + __mul24(di[1], P2) + di[2];
                    base1 = base0;
                }
                const bool valid = FULL || (((vbits >> (s * 3 + ca)) & 1u) && p < g.cn[ca][0]);
                slow_any = slow_any || (valid && slow);          // -> this plane of the tile is redone by advect_self_fixup_kernel
                if (DIM == 3) {
                    if (H == 1) {
                    } else {
                        int s0 = slot_of(p) + di[0];
                        s0 += s0 < 0 ? NP : 0; s0 -= s0 >= NP ? NP : 0;
                        int s1 = s0 + 1; s1 -= s1 >= NP ? NP : 0;
                        base1 += s1 * PLANE;
                        base0 += s0 * PLANE;
                    }
                }
                T y[2];
#pragma unroll
                for (int k = 0; k < (DIM == 3 ? 2 : 1); ++k) {
                    const int bk = k ? base1 : base0;
                    const T a00 = lds[bk], a10 = lds[bk + P2], a01 = lds[bk + 1], a11 = lds[bk + P2 + 1];   // (row pairs: packed lerp along a2)
                    const T x0 = fma(fr[2], a01 - a00, a00), x1 = fma(fr[2], a11 - a10, a10);
                    y[k] = fma(fr[1], x1 - x0, x0);
                }
                const T val = DIM == 3 ? fma(fr[0], y[1] - y[0], y[0]) : y[0];
                if (FULL) {
                    T* const plane = outp[ca] + (long long)b * g.ccells[ca] + (long long)p * pstride[ca];      // uniform
                    plane[obase[ca] + (unsigned)(s * TY * g.cn[ca][2])] = val;
                } else {
                    T* const slot = outp[ca] + (long long)b * g.ccells[ca] + (long long)p * pstride[ca] + (obase[ca] + (unsigned)(s * TY * g.cn[ca][2]));
                    *(valid ? slot : dump) = val;     // unconditional store (see load_plane)
                }
#if PHIHIP_TILE_FENCE == 1
                sched_fence();     // one sample's LDS reads in flight at a time (fence per position or none: +10 % time, profiles/r02_ab_advect*.jsonl)
#endif
            }
#if PHIHIP_TILE_FENCE == 2
            sched_fence();
#endif
        }
        if (slow_any) slow_sh[p & 1] = 1;
    };
    // after the barrier that ends plane p's half-step: one entry in the fix-up work list if any sample of the plane left the window
    auto report_plane = [&](int p) {
        if (tid == 0 && slow_sh[p & 1]) {
            slow_sh[p & 1] = 0;
            fix_append(fix, b * nblk + (int)blockIdx.x, p);
        }
    };

    // Pipeline, two planes per trip so that the two register sets keep static names. In half-step p: plane p+H+2 is requested, plane p
    // is computed (once the ring holds p-H .. p+H), plane p+H+1 -- requested one half-step earlier -- enters the ring; one barrier.
    // The trips before pb only fill the ring (their loads are back to back, so the warm-up costs about one memory round trip).
    T RA[3][KP], RB[3][KP];
    T tailA = T(0), tailB = T(0);
    const int k_lo = DIM == 3 ? pb - H : 0, k_hi = DIM == 3 ? pe - 1 + H : 0;     // planes this workgroup stages
    auto half_step = [&](int p, auto mode_tag, T (&Rld)[3][KP], T& tail_ld, T (&Rst)[3][KP], T& tail_st) {
        constexpr int MODE_ = decltype(mode_tag)::value;   // 0: the general half-step (2-D), 1: ring warm-up (no samples yet), 2: steady state
        const int kl = p + H + 2, ks = p + H + 1;
        if (MODE_ == 0) {
            if (kl >= k_lo && kl <= k_hi) load_plane(kl, Rld, tail_ld);
            if (p >= pb && p < pe) compute_plane(p, std::false_type{});
        } else {
            load_plane(min(max(kl, k_lo), k_hi), Rld, tail_ld);   // (past the last staged plane: that plane again, nobody stores it)
            if (MODE_ == 2) {
                if (tile_full) compute_plane(p, std::true_type{}); else compute_plane(p, std::false_type{});
            }
        }
        if (ks >= k_lo && ks <= k_hi) {
            if (CONSTS && has_const) patch_plane(ks, Rst, tail_st);
            store_plane(slot_of(ks), Rst, tail_st);
        }
        __syncthreads();
        if (MODE_ == 2 || (MODE_ == 0 && p >= pb && p < pe)) report_plane(p);
    };
    const int p_first = k_lo - H - 2;     // the half-step that requests the first staged plane
    if (DIM == 3) {
        // pb - p_first = 2H + 2 half-steps fill the ring, then every half-step computes a plane: both loops are straight-line code
        int p = p_first;
        for (; p < pb; p += 2) {
            half_step(p, std::integral_constant<int, 1>{}, RA, tailA, RB, tailB);
            half_step(p + 1, std::integral_constant<int, 1>{}, RB, tailB, RA, tailA);
        }
        for (; p + 1 < pe; p += 2) {
            half_step(p, std::integral_constant<int, 2>{}, RA, tailA, RB, tailB);
            half_step(p + 1, std::integral_constant<int, 2>{}, RB, tailB, RA, tailA);
        }
        if (p < pe) half_step(p, std::integral_constant<int, 2>{}, RA, tailA, RB, tailB);
    } else {
        for (int p = p_first; p < pe; p += 2) {
            half_step(p, std::integral_constant<int, 0>{}, RA, tailA, RB, tailB);
            half_step(p + 1, std::integral_constant<int, 0>{}, RB, tailB, RA, tailA);
        }
    }
}

// Redo of the (tile, plane) units that met lookups outside their LDS window (work list filled by advect_self_tile_kernel): every sample of
// the plane with the gather code of the per-component kernels (advect.hip), reading the untouched input velocity. Fixed grid; the
// workgroups stride over the list.
template <typename T, int DIM, int T1>
__global__ __launch_bounds__(kBlock) void advect_self_fixup_kernel(VelGrid g, CComp3a<T> vel, T* __restrict__ o0, T* __restrict__ o1, T* __restrict__ o2,
                                                                   T dt, int tiles1, int tiles2, int nblk, FixList fix) {
    using C = AdvTile<T, DIM, 1, T1>;
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
#pragma unroll
            for (int ca = A0; ca < 3; ++ca) {
                if (p >= g.cn[ca][0] || j1 >= g.cn[ca][1] || j2 >= g.cn[ca][2]) continue;
                const int idx[3] = {p, j1, j2};
                const int f = (p * g.cn[ca][1] + j1) * g.cn[ca][2] + j2;
                T cf[3];      // (the tile kernels' arithmetic: advect_common.hpp "ONE arithmetic per advection sample")
                if (ca == 0) face_disp<T, DIM, 0>(g, vel, b, idx, f, dt, cf);
                else if (ca == 1) face_disp<T, DIM, 1>(g, vel, b, idx, f, dt, cf);
                else face_disp<T, DIM, 2>(g, vel, b, idx, f, dt, cf);
                T disp[3] = {T(0), T(0), T(0)};
#pragma unroll
                for (int a = A0; a < 3; ++a) disp[a] = -cf[a];
                const int n[3] = {g.cn[ca][0], g.cn[ca][1], g.cn[ca][2]};
                int bc[3][2];
                T cv[3][2];
                comp_rule<T>(g, ca, bc, cv);
                AxisPair<T> ax[3];
                T fr[3];
                lookup_pairs_rel<T, DIM>(idx, disp, n, bc, cv, ax, fr);
                outp[ca][(long long)b * g.ccells[ca] + f] = gather_multilinear<T, DIM>(vel.p[ca] + (long long)b * g.ccells[ca], ax, fr);
            }
        }
    }
}

// =====================================================================================================================================
// r5: the same pass with the ring filled by LDS-DMA (`global_load_lds_dwordx4`: HBM / L2 -> LDS without a VGPR round trip) -- the lever the
// round-3 and round-4 verdicts asked for. Scope = grids whose staged rows are REGULAR: 3-D, reach 1, no CLOSED (constant) side, the fast axis
// periodic with rows of whole 16-byte vectors -- the benchmark configuration. (A closed box has a component with N - 1 faces per row: its
// windows need per-element padding, which only the register-staged kernel above can do.)
//   * window of a (component, plane) = P1 rows x G2 = T2 / V + 2 chunks of 16 bytes: the tile's columns and ONE chunk either side (the halo column is
//     its nearest element), contiguous in LDS in (row, chunk) order -- the destination of LDS-DMA is wave-uniform base + 16 lane, so a
//     wavefront instruction moves 64 consecutive chunks = 1 KiB; the per-lane SOURCE resolves the row (wrap / clamp) and the chunk (wrap);
//   * wavefront w (of 4) feeds component w: NI = ceil(P1 G2 / 64) instructions per plane (3 for the fp32 tile instead of 10 loads + 10
//     ds_write per THREAD and ~70 VALU instructions per wavefront and plane of address arithmetic and staging);
//   * ring of 4 planes: plane p + 2 is requested at the start of half-step p into the slot plane p - 2 left, and has to have landed at the END of
//     the half-step (one plane of arithmetic, ~2 500 cycles, against ~1 000 of memory latency): `s_waitcnt vmcnt(#stores of this half-step)` on
//     the feeding wavefronts -- VMEM operations retire in issue order, so that many may stay in flight --, then the workgroup barrier that makes
//     the landed bytes visible to the other wavefronts (MI355X_MICROARCH.md: nothing else orders a ds_read behind an LDS-DMA). The transfers
//     are inline assembly: the compiler would drain vmcnt before every LDS read that may alias an LDS-DMA destination it knows of.
// Same samples, same arithmetic as the kernel above (compute: the H = 1 sign form); fix-up list and launch geometry are shared.
// =====================================================================================================================================
template <typename T, int T1>
struct AdvDma {
    static constexpr int V = 16 / (int)sizeof(T);
    static constexpr int T2 = sizeof(T) == 4 ? 64 : 32;
    static constexpr int TY = kBlock / T2, S = T1 / TY;
    static constexpr int P1 = T1 + 2, G2 = T2 / V + 2, PITCH = G2 * V;
    static constexpr int NCH = P1 * G2, NI = (NCH + kWave - 1) / kWave;
    static constexpr int PLANE = NCH * V;                 // elements; the last instruction of a plane masks its idle lanes
    static constexpr int NP = 4;
    static_assert((size_t)3 * NP * PLANE * sizeof(T) <= 65536, "static LDS limit");
};

// (lds_dma16: advect_common.hpp)
// GEN (r5, second step): grids with CLOSED / OPEN sides and rows that are not whole vectors (a closed box stores N - 1 faces of the component along its own
// axis). The ring is still filled by LDS-DMA; what a 16-byte chunk cannot express is settled three ways:
//   * a chunk (or row, or plane) that lies entirely beyond a CLOSED side is transferred from a small table in global memory that holds every wall constant
//     replicated to 16 bytes (`kconst`, written when the constants change) -- PhiML pads axis after axis, so a2 wins over a1 over a0;
//   * rows / planes beyond an OPEN side are the clamped row / plane (zero-gradient padding = the edge sample): an address, nothing else;
//   * the few ELEMENTS left over -- the halo column beyond an open side (a copy of the edge element), the elements of a chunk that straddles a row end
//     (N - 1 is not a multiple of 4) beyond that end, and the chunk that straddles the end of a plane's LAST row (its natural read would leave the plane: it is
//     not transferred at all) -- are PATCHED by up to one thread each after the plane has landed: constant, copy of an LDS element of the same row, or one
//     scalar global load. Workgroups with patches (tiles at a non-periodic fast-axis end) pay a second barrier per plane; the others run the regular loop.
// OFFM: face-offset mask (1 below a CLOSED side), as in the register-staged kernel.
template <typename T, int T1, int OFFM, bool GEN>
__global__ __launch_bounds__(kBlock, sizeof(T) == 4 ? 4 : 2) void advect_self_dma_kernel(TileGrid<T> g, CComp3a<T> vel, T* __restrict__ o0, T* __restrict__ o1, T* __restrict__ o2,
                                                                                         int chunk, int tiles1, int tiles2, int nblk, int nmax0, FixList fix, T* __restrict__ dump,
                                                                                         const T* __restrict__ kconst) {
    using C = AdvDma<T, T1>;
    constexpr int T2 = C::T2, TY = C::TY, S = C::S, V = C::V, PITCH = C::PITCH, PLANE = C::PLANE, NP = C::NP, NI = C::NI, G2 = C::G2, P1 = C::P1;
    constexpr int OFF[3] = {(OFFM >> 0) & 1, (OFFM >> 1) & 1, (OFFM >> 2) & 1};
    static_assert(GEN || OFFM == 0, "regular grids store every lower face");
    __shared__ __attribute__((aligned(16))) T lds[3 * NP * PLANE];
    __shared__ int slow_sh[2];
    __shared__ int patch_sh;            // GEN: some thread of this workgroup has an element to patch

    const int tid = threadIdx.x, tx = tid % T2, ty = tid / T2, lane = tid & (kWave - 1);
#ifdef __HIP_DEVICE_COMPILE__
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
#else
    const int wave = tid / kWave;
#endif
    const int b = blockIdx.y;
    const int bid = xcd_order((int)blockIdx.x, nblk);
    const int t2 = bid % tiles2;
    const int t1 = (bid / tiles2) % tiles1;
    const int c0 = bid / (tiles2 * tiles1);
    const int lo1 = t1 * T1, lo2 = t2 * T2;
    const int pb = c0 * chunk, pe = min(pb + chunk, nmax0);
    T* const outp[3] = {o0, o1, o2};
    if (tid < 2) slow_sh[tid] = 0;
    if (GEN && tid == 2) patch_sh = 0;
    if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) *fix.next = 0;
    unsigned obase[3];
    long long pstride[3];
    unsigned vbits = 0;
    bool tile_full = true;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        obase[c] = (unsigned)((lo1 + ty) * g.cn[c][2] + lo2 + tx);
        pstride[c] = (long long)g.cn[c][1] * g.cn[c][2];
#pragma unroll
        for (int k = 0; k < S; ++k)
            if (lo1 + ty + k * TY < g.cn[c][1] && lo2 + tx < g.cn[c][2]) vbits |= 1u << (k * 3 + c);
        tile_full = tile_full && pe <= g.cn[c][0] && lo1 + T1 <= g.cn[c][1] && lo2 + T2 <= g.cn[c][2];
    }

    // ---- feed: wavefront w < 3 owns component w. Plane-invariant byte offset of the chunk each lane moves, per instruction -----------------
    const int fc = wave < 3 ? wave : 0;
    const int fn0 = fc == 0 ? g.cn[0][0] : (fc == 1 ? g.cn[1][0] : g.cn[2][0]);
    const int fn1 = fc == 0 ? g.cn[0][1] : (fc == 1 ? g.cn[1][1] : g.cn[2][1]);
    const int fn2 = fc == 0 ? g.cn[0][2] : (fc == 1 ? g.cn[1][2] : g.cn[2][2]);
    const T* const fbase = (fc == 0 ? vel.p[0] : (fc == 1 ? vel.p[1] : vel.p[2])) + (long long)b * (fc == 0 ? g.ccells[0] : (fc == 1 ? g.ccells[1] : g.ccells[2]));
    const long long fstride = (long long)fn1 * fn2;
    // a row index / chunk start of the window under the component's padding rule: the stored index to read and, beyond a CLOSED side, which constant replaces
    // it (kind = axis * 2 + side, -1 = none)
    auto resolve_row = [&](int j, int n1, int& kind) -> int {
        kind = -1;
        if (j < 0) {
            if (g.bc[1][0] == PHIHIP_BC_PERIODIC) return wrap_index(j, n1);
            if (g.bc[1][0] == PHIHIP_BC_CLOSED) kind = 2;
            return 0;
        }
        if (j >= n1) {
            if (g.bc[1][1] == PHIHIP_BC_PERIODIC) return wrap_index(j, n1);
            if (g.bc[1][1] == PHIHIP_BC_CLOSED) kind = 3;
            return n1 - 1;
        }
        return j;
    };
    unsigned doff[NI];
    int ckind[GEN ? NI : 1];            // GEN: -1 = transfer from the array, 0 .. 5 = from the constant table (axis * 2 + side), 7 = not transferred (patched)
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int q = k * kWave + lane;
        const int rr = q / G2, gg = q - rr * G2;
        if (!GEN) {
            int j = lo1 - 1 + rr;
            if (g.bc[1][0] == PHIHIP_BC_PERIODIC) j = wrap_index(j, fn1);
            j = min(max(j, 0), fn1 - 1);                           // OPEN: zero-gradient padding = the edge row
            const int k0 = wrap_index(lo2 - V + gg * V, fn2);      // the fast axis is periodic and fn2 a multiple of V: a chunk never straddles the seam
            doff[k] = (unsigned)(j * fn2 + k0) * (unsigned)sizeof(T);
        } else {
            int rkind, kind = -1;
            const int j = resolve_row(lo1 - 1 + rr, fn1, rkind);
            int k0 = lo2 - V + gg * V;
            if (k0 + V <= 0) {                                     // entirely below the row
                if (g.bc[2][0] == PHIHIP_BC_PERIODIC) k0 = wrap_index(k0, fn2);
                else if (g.bc[2][0] == PHIHIP_BC_CLOSED) { kind = 4; k0 = 0; }
                else k0 = 0;                                       // OPEN: any valid chunk -- the halo element is patched (copy of the edge element)
            } else if (k0 >= fn2) {                                // entirely beyond the row
                if (g.bc[2][1] == PHIHIP_BC_PERIODIC) k0 = wrap_index(k0, fn2);
                else if (g.bc[2][1] == PHIHIP_BC_CLOSED) { kind = 5; k0 = 0; }
                else k0 = fn2 - V;
            } else if (k0 + V > fn2) {                             // straddles the row's end (rows of N - 1 / N + 1 faces): the elements beyond it are patched;
                if (j == fn1 - 1) kind = 7;                        // in a plane's LAST row the read would leave the plane (the array, at its end): patched whole
            }
            if (kind < 0 && rkind >= 0) kind = rkind;              // (a2 wins over a1)
            ckind[k] = kind;
            doff[k] = (unsigned)(j * fn2 + k0) * (unsigned)sizeof(T);
        }
    }
    // plane i0 of the wavefront's component: element offset of the plane to read and, beyond a CLOSED a0 side, the constant kind (0 / 1)
    auto plane_src = [&](int i0, int& pkind) -> long long {
        int w = i0;
        pkind = -1;
        if (g.bc[0][0] == PHIHIP_BC_PERIODIC) { w += w < 0 ? fn0 : 0; w -= w >= fn0 ? fn0 : 0; }
        if (GEN && w < 0 && g.bc[0][0] == PHIHIP_BC_CLOSED) pkind = 0;
        if (GEN && w >= fn0 && g.bc[0][1] == PHIHIP_BC_CLOSED) pkind = 1;
        w = min(max(w, 0), fn0 - 1);
        return (long long)w * fstride;
    };
    auto feed = [&](int i0) {      // request plane i0 of the wavefront's component into its ring slot
        if (wave < 3) {
            int pkind;
            const char* const src = reinterpret_cast<const char*>(fbase + plane_src(i0, pkind));
            T* const dst = lds + (fc * NP + (i0 & (NP - 1))) * PLANE;
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                if (!GEN) {
                    if (k * kWave + lane < C::NCH) lds_dma16<T>(src + doff[k], dst + k * kWave * V, lane);
                } else {
                    const int kind = ckind[k] >= 0 ? ckind[k] : pkind;      // chunk / row constants win over the plane's
                    const char* const from = kind >= 0 ? reinterpret_cast<const char*>(kconst + (kind * 3 + fc) * V) : src + doff[k];
                    if (k * kWave + lane < C::NCH && ckind[k] != 7) lds_dma16<T>(from, dst + k * kWave * V, lane);
                }
            }
        }
    };
    // ---- GEN: the patch element of this thread (plane-invariant): thread e of component c, window row rr -> one column --------------------------------
    //   slot 0: the halo column below the row (OPEN: copy of column 0; CLOSED: constant -- redundant with the constant chunk, harmless)
    //   slot 1, 2: the first / second column beyond the row's end (OPEN: copy of the last column; CLOSED: constant)
    //   slot 3 ..: the in-row elements of a straddling chunk that is not transferred (last row of a plane): one scalar global load each
    constexpr int PE = V + 2;
    static_assert(3 * P1 * PE <= kBlock, "one patch element per thread");
    int pmode = 0;                      // 0 none, 1 constant, 2 copy of an LDS element of the same (component, slot, row), 3 global element
    int pdst = 0, psrc = 0, pcomp = 0;
    T pval = T(0);
    long long pgoff = 0, ppstr = 0;
    int prow_kind = -1, pn0 = 1;
    const T* pbase = vel.p[2];
    if (GEN && tid < 3 * P1 * PE) {
        const int c = tid / (P1 * PE), rr = (tid / PE) % P1, e = tid % PE;
        const int n1 = c == 0 ? g.cn[0][1] : (c == 1 ? g.cn[1][1] : g.cn[2][1]), n2 = c == 0 ? g.cn[0][2] : (c == 1 ? g.cn[1][2] : g.cn[2][2]);
        int rkind;
        const int j = resolve_row(lo1 - 1 + rr, n1, rkind);
        const int w0 = lo2 - V;                                 // stored column of window column 0
        const int ks = (n2 / V) * V;                            // start of the chunk that straddles the row's end (if n2 % V != 0)
        int col = 0;
        pcomp = c;
        prow_kind = rkind;
        if (e == 0) {
            col = lo2 - 1;
            if (col < 0 && g.bc[2][0] != PHIHIP_BC_PERIODIC) {
                pmode = g.bc[2][0] == PHIHIP_BC_CLOSED ? 1 : 2;
                pval = c == 0 ? g.bcv[2][0][0] : (c == 1 ? g.bcv[2][0][1] : g.bcv[2][0][2]);      // (selects: a per-lane index into kernel arguments goes through scratch)
                psrc = rr * PITCH + (0 - w0);
            }
        } else if (e <= 2) {
            col = n2 + (e - 1);
            if (g.bc[2][1] != PHIHIP_BC_PERIODIC && col >= w0 && col <= lo2 + T2 && n2 - 1 >= w0) {
                pmode = g.bc[2][1] == PHIHIP_BC_CLOSED ? 1 : 2;
                pval = c == 0 ? g.bcv[2][1][0] : (c == 1 ? g.bcv[2][1][1] : g.bcv[2][1][2]);
                psrc = rr * PITCH + (n2 - 1 - w0);
                if (pmode == 2 && n2 % V != 0 && j == n1 - 1) {      // the edge element is itself patched in this phase (slot 3 ..): read what it reads
                    pmode = 3;
                    pgoff = (long long)j * n2 + (n2 - 1);
                }
            }
        } else {
            col = ks + (e - 3);
            if (n2 % V != 0 && col < n2 && j == n1 - 1 && ks >= w0 && ks < w0 + G2 * V) {
                pmode = 3;
                pgoff = (long long)j * n2 + col;
            }
        }
        pdst = rr * PITCH + (col - w0);
        // (the component's array, extent and plane stride of this thread's element: resolved HERE -- inside the plane loop the compiler turns the select
        // chains into a table in scratch, and a scratch load is a VMEM operation between the counted waits)
        pn0 = c == 0 ? g.cn[0][0] : (c == 1 ? g.cn[1][0] : g.cn[2][0]);
        ppstr = (long long)n1 * n2;
        pbase = (c == 0 ? vel.p[0] : (c == 1 ? vel.p[1] : vel.p[2])) + (long long)b * (c == 0 ? g.ccells[0] : (c == 1 ? g.ccells[1] : g.ccells[2]));
        if (pmode) patch_sh = 1;
    }
    auto patch = [&](int i0) {          // after plane i0 has landed (and a barrier): the elements no chunk can express
        if (pmode) {
            const int n0 = pn0;
            const long long pstr = ppstr;
            int w = i0, pkind = -1;
            if (g.bc[0][0] == PHIHIP_BC_PERIODIC) { w += w < 0 ? n0 : 0; w -= w >= n0 ? n0 : 0; }
            if (w < 0 && g.bc[0][0] == PHIHIP_BC_CLOSED) pkind = 0;
            if (w >= n0 && g.bc[0][1] == PHIHIP_BC_CLOSED) pkind = 1;
            w = min(max(w, 0), n0 - 1);
            T* const P = lds + (pcomp * NP + (i0 & (NP - 1))) * PLANE;
            T val;
            if (pmode == 1) val = pval;
            else if (pmode == 2) val = P[psrc];
            else {
                const int kind = prow_kind >= 0 ? prow_kind : pkind;
                val = kind >= 0 ? kconst[(kind * 3 + pcomp) * V] : pbase[(long long)w * pstr + pgoff];
            }
            P[pdst] = val;
        }
    };
    // the feeding wavefronts wait until at most `stores` of their VMEM operations are in flight (= everything older than this half-step's output
    // stores has retired, the transfers included), then the workgroup meets
    auto landed_and_barrier = [&](auto stores_tag) {
#ifdef __HIP_DEVICE_COMPILE__
        constexpr int N = decltype(stores_tag)::value;
        if (wave < 3) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory");
#else
        __syncthreads();
#endif
    };

    auto compute_plane = [&](int p, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int so_m = ((p - 1) & (NP - 1)) * PLANE, so_0 = (p & (NP - 1)) * PLANE, so_p = ((p + 1) & (NP - 1)) * PLANE;
        bool slow_any = false;
#if PHIHIP_DMA_UNROLL_S
#pragma unroll
#else
#pragma unroll 1
#endif
        for (int s = 0; s < S; ++s) {
            const int center = (ty + s * TY + 1) * PITCH + tx + V;
            const int cen[3] = {center + so_m, center + so_0, center + so_p};
#pragma unroll
            for (int ca = 0; ca < 3; ++ca) {
                auto at = [&](int x, int d0, int d1, int d2) -> T { return lds[cen[d0 + 1] + (x * NP * PLANE + d1 * PITCH + d2)]; };
                T coord[3] = {T(0), T(0), T(0)};
                coord[ca] = at(ca, 0, 0, 0) * -g.shift[ca];
#pragma unroll
                for (int cb = 0; cb < 3; ++cb) {
                    if (cb == ca) continue;
                    // component cb at this ca-face: cells (m - 1, m) along ca, faces (s, s + 1) along cb, in cb's stored indices (regular grids: off = 0)
                    T v4[2][2];
#pragma unroll
                    for (int ia = 0; ia < 2; ++ia)
#pragma unroll
                        for (int ib = 0; ib < 2; ++ib) {
                            int d[3] = {0, 0, 0};
                            d[ca] = OFF[ca] - 1 + ia;
                            d[cb] = -OFF[cb] + ib;
                            v4[ia][ib] = at(cb, d[0], d[1], d[2]);
                        }
                    coord[cb] = (((v4[0][0] + v4[0][1]) + v4[1][0]) + v4[1][1]) * (T(-0.25) * g.shift[cb]);
                }
                const bool slow = !(fmax(fmax(fabs(coord[0]), fabs(coord[1])), fabs(coord[2])) < T(1));
                T fr[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) fr[a] = frac_part(coord[a]);
                const int off = ca * NP * PLANE + center + (coord[1] < T(0) ? -PITCH : 0) + (coord[2] < T(0) ? -1 : 0);
                const bool down = coord[0] < T(0);
                const int base1 = off + (down ? so_0 : so_p), base0 = off + (down ? so_m : so_0);
                const bool valid = FULL || (((vbits >> (s * 3 + ca)) & 1u) && p < g.cn[ca][0]);
                slow_any = slow_any || (valid && slow);
                T y[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int bk = k ? base1 : base0;
                    const T a00 = lds[bk], a10 = lds[bk + PITCH], a01 = lds[bk + 1], a11 = lds[bk + PITCH + 1];
                    const T x0 = fma(fr[2], a01 - a00, a00), x1 = fma(fr[2], a11 - a10, a10);
                    y[k] = fma(fr[1], x1 - x0, x0);
                }
                const T val = fma(fr[0], y[1] - y[0], y[0]);
                if (FULL) {
                    T* const plane = outp[ca] + (long long)b * g.ccells[ca] + (long long)p * pstride[ca];
                    plane[obase[ca] + (unsigned)(s * TY * g.cn[ca][2])] = val;
                } else {
                    T* const slot = outp[ca] + (long long)b * g.ccells[ca] + (long long)p * pstride[ca] + (obase[ca] + (unsigned)(s * TY * g.cn[ca][2]));
                    *(valid ? slot : dump) = val;     // unconditional: the feeding wavefronts count their stores
                }
#if PHIHIP_DMA_FENCE == 1
                sched_fence();      // one sample's LDS reads in flight at a time
#endif
            }
#if PHIHIP_DMA_FENCE == 2
            sched_fence();          // one position's (three samples') LDS reads in flight at a time
#endif
        }
        if (slow_any) slow_sh[p & 1] = 1;
    };
    auto report_plane = [&](int p) {
        if (tid == 0 && slow_sh[p & 1]) {
            slow_sh[p & 1] = 0;
            fix_append(fix, b * nblk + (int)blockIdx.x, p);
        }
    };

    // ring warm-up: planes pb - 1, pb, pb + 1 (back to back: one memory round trip), then one plane per half-step
    feed(pb - 1);
    feed(pb);
    feed(pb + 1);
    landed_and_barrier(std::integral_constant<int, 0>{});
    auto wg_barrier = [&]() {
#ifdef __HIP_DEVICE_COMPILE__
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory");
#else
        __syncthreads();
#endif
    };
    const bool patching = GEN && patch_sh != 0;       // (uniform: written before the barrier above)
    if (patching) {
        patch(pb - 1); patch(pb); patch(pb + 1);
        wg_barrier();
    }
    for (int p = pb; p < pe; ++p) {
        feed(p + 2);       // (beyond the chunk's last plane + 1 nobody reads it: requested all the same, the counts stay uniform)
        if (tile_full) compute_plane(p, std::true_type{}); else compute_plane(p, std::false_type{});
        landed_and_barrier(std::integral_constant<int, 3 * S>{});
        if (patching) {        // plane p + 2 has landed and is visible: its patch elements, then the workgroup meets again before anybody reads the plane
            patch(p + 2);
            wg_barrier();
        }
        report_plane(p);
    }
}

// the wall constants of a grid, every one replicated to 16 bytes: kconst[(axis * 2 + side) * 3 + component][0 .. V)
template <typename T>
__global__ void advect_consts_kernel(TileGrid<T> g, T* __restrict__ kconst) {
    constexpr int V = 16 / (int)sizeof(T);
    const int t = threadIdx.x;
    if (t < 18 * V) {
        const int e = t / V, c = e % 3, as = e / 3;
        kconst[t] = g.bcv[as >> 1][as & 1][c];
    }
}

template <typename T, int DIM, int H, int T1, int OFFM, bool CONSTS>
static int launch_tile_consts(phihip_ctx* ctx, const GridView& v, const VelGrid& vg, const void* const vel[3], void* const out[3], double dt, int kind, hipStream_t s) {
    const TileGrid<T> g = make_tilegrid<T>(vg, dt);
    using C = AdvTile<T, DIM, H, T1>;
    int nmax[3] = {1, 1, 1};
    for (int a = 0; a < 3; ++a)
        for (int c = v.ax0; c < 3; ++c) nmax[a] = v.cn[c][a] > nmax[a] ? v.cn[c][a] : nmax[a];
    const int tiles1 = ceil_div(nmax[1], T1), tiles2 = ceil_div(nmax[2], C::T2);
    int chunk = 1, chunks0 = 1;
    if (DIM == 3) {
        // Chunks of planes per workgroup: every chunk stages 2H+1 extra planes, and a launch that needs 1 < rounds < 2 of resident
        // workgroups costs two rounds (at 256^3 1536 workgroups on 1024 slots ran at 55 % VALU utilisation). Score every chunk count by
        // (slot efficiency of the last round) x (useful planes / staged planes) and prefer >= 2 rounds at equal score.
        static int occ_dev[16] = {0};       // per device: a process may drive several (the query runs on the current one = ctx->device)
        int& occ = occ_dev[ctx->device >= 0 && ctx->device < 16 ? ctx->device : 0];
        if (occ == 0) {
            int n = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, advect_self_tile_kernel<T, DIM, H, T1, OFFM, CONSTS>, kBlock, 0) != hipSuccess || n < 1) n = 1;
            occ = n;
        }
        const double slots = (double)occ * ctx->num_cu;
        const long long tiles = (long long)tiles1 * tiles2 * v.batch;
        double best = -1.0;
        for (int c = 1; c <= nmax[0]; ++c) {
            const int ch = ceil_div(nmax[0], c);
            if (c > 1 && ch < 4) break;
            if (ceil_div(nmax[0], ch) != c) continue;            // same decomposition as a smaller c
            const double rounds = (double)tiles * c / slots;
            const double eff = rounds / ceil(rounds - 1e-9);
            const double score = eff * ch / (ch + 2 * H + 1) * (rounds >= 2.0 ? 1.0 : (rounds >= 1.0 ? 0.97 : 0.9));
            if (score > best * 1.0001) { best = score; chunk = ch; }
        }
    }
    CComp3a<T> vv{{(const T*)vel[0], (const T*)vel[1], (const T*)vel[2]}};
    int nblk = 0;
    // the tile kernel + its fix-up launch (fixed grid striding over the work list)
    // r5: the ring is filled by LDS-DMA -- `dma` = 1: regular grids (no CLOSED side, periodic fast axis with rows of whole 16-byte vectors, 16-byte-aligned
    // arrays), 2: the GEN instantiation (closed / open sides, rows of N - 1 / N + 1 faces: constants from a table, patch elements)
    int dma = 0;
    constexpr int VV = 16 / (int)sizeof(T);
    if constexpr (DIM == 3 && H == 1 && (sizeof(T) == 4 ? T1 == 8 : T1 == 16)) {
        bool ok = ctx->adv_dma != 0, regular = !CONSTS && OFFM == 0 && v.bc[2][0] == PHIHIP_BC_PERIODIC;
        bool any_open = false;
        for (int a = 0; a < 3; ++a) any_open = any_open || v.bc[a][0] == PHIHIP_BC_OPEN || v.bc[a][1] == PHIHIP_BC_OPEN;
        for (int c = 0; c < 3; ++c) {
            ok = ok && v.cn[c][2] >= 2 * VV && v.cn[c][1] >= 4 && v.cn[c][0] >= 4;
            if (v.bc[2][0] == PHIHIP_BC_PERIODIC) ok = ok && v.cn[c][2] % VV == 0;            // (a wrapped chunk must not straddle the seam)
            regular = regular && v.cn[c][2] % VV == 0 && (reinterpret_cast<uintptr_t>(vel[c]) & 15u) == 0;
        }
        (void)any_open;                                                                        // (open a0 / a1 sides are addresses only: still regular)
        dma = ok ? (regular ? 1 : 2) : 0;
    }
    const T* kconst = nullptr;
    if (dma == 2) {
        // the wall constants, replicated to 16 bytes each, in a table the transfers can read (rewritten when they -- or the element type -- change)
        PHIHIP_TRY(ensure_buffer(ctx->ws_adv_const, 18 * 16));
        double key[19];
        for (int a = 0; a < 3; ++a) for (int sd = 0; sd < 2; ++sd) for (int c = 0; c < 3; ++c) key[(a * 2 + sd) * 3 + c] = (double)g.bcv[a][sd][c];
        key[18] = (double)sizeof(T);
        if (!ctx->adv_const_valid || memcmp(key, ctx->adv_const_key, sizeof(key)) != 0 || stream_is_capturing(s)) {
            hipLaunchKernelGGL((advect_consts_kernel<T>), dim3(1), dim3(128), 0, s, g, (T*)ctx->ws_adv_const.ptr);
            memcpy(ctx->adv_const_key, key, sizeof(key));
            ctx->adv_const_valid = !stream_is_capturing(s);      // (a captured launch rewrites the table in its own graph; eager launches after it write it again)
        }
        kconst = (const T*)ctx->ws_adv_const.ptr;
    }
    auto launch = [&](int ch) -> int {
        FixList fix;
        void* dump = nullptr;
        PHIHIP_TRY(prepare_fixlist(ctx, (long long)tiles1 * tiles2 * nmax[0] * v.batch, s, &fix, &dump, kind));
        chunks0 = DIM == 3 ? ceil_div(nmax[0], ch) : 1;
        nblk = tiles1 * tiles2 * chunks0;
        if constexpr (DIM == 3 && H == 1 && (sizeof(T) == 4 ? T1 == 8 : T1 == 16)) {
            if constexpr (!CONSTS && OFFM == 0) {
                if (dma == 1)
                    hipLaunchKernelGGL((advect_self_dma_kernel<T, T1, 0, false>), dim3(nblk, v.batch), dim3(kBlock), 0, s, g, vv, (T*)out[0], (T*)out[1], (T*)out[2], ch,
                                       tiles1, tiles2, nblk, nmax[0], fix, (T*)dump, kconst);
            }
            if (dma == 2)
                hipLaunchKernelGGL((advect_self_dma_kernel<T, T1, OFFM, true>), dim3(nblk, v.batch), dim3(kBlock), 0, s, g, vv, (T*)out[0], (T*)out[1], (T*)out[2], ch,
                                   tiles1, tiles2, nblk, nmax[0], fix, (T*)dump, kconst);
        }
        if (!dma)
        hipLaunchKernelGGL((advect_self_tile_kernel<T, DIM, H, T1, OFFM, CONSTS>), dim3(nblk, v.batch), dim3(kBlock), 0, s, g, vv, (T*)out[0], (T*)out[1],
                           (T*)out[2], ch, tiles1, tiles2, nblk, nmax[0], fix, (T*)dump);
        const int fgrid = fix.cap < kFixupBlocks ? fix.cap : kFixupBlocks;
        hipLaunchKernelGGL((advect_self_fixup_kernel<T, DIM, T1>), dim3(fgrid), dim3(kBlock), 0, s, vg, vv, (T*)out[0], (T*)out[1], (T*)out[2], (T)dt,
                           tiles1, tiles2, nblk, fix);
        return PHIHIP_OK;
    };
    if (DIM == 3 && ctx->adv_chunk > 0) {
        chunk = ctx->adv_chunk < nmax[0] ? ctx->adv_chunk : nmax[0];
    } else if (DIM == 3 && H == 1 && ctx->autotune && (long long)nmax[0] * nmax[1] * nmax[2] * v.batch >= (1 << 21) && !stream_is_capturing(s)) {
        // (reach 1 only: the wide kernel is what the adaptive policy switches to in the middle of a run -- timing six chunk lengths there cost the step in
        // which the switch fell ~4.5 ms at 256^3, 20 steps' worth of what the better chunk length could save; it keeps the planner's length)
        // First call on this grid: the planner's chunk length against a few others, timed on the call's own operands (every length
        // writes the same values, so the output is simply overwritten; ~2 ms once per grid). The planner ranks slot efficiency x halo
        // overhead and misses e.g. the ring warm-up per chunk: 384^3 fp64 runs 6 % faster with 16 planes than with its 64.
        const std::array<long long, 6> key = {(long long)sizeof(T) + (dma ? 100 : 0), DIM, H, nmax[0], (long long)tiles1 * tiles2, v.batch};
        const auto it = ctx->adv_tuned.find(key);
        if (it != ctx->adv_tuned.end()) {
            chunk = it->second;
        } else {
            SlowTrace tr("advect_self_tiled: first-call chunk timing");
            hipEvent_t e0, e1;
            PHIHIP_CHECK_HIP(hipEventCreate(&e0));
            PHIHIP_CHECK_HIP(hipEventCreate(&e1));
            const int cand[6] = {chunk, 16, 24, 32, 48, 64};
            float best_ms = 1e30f;
            int best = chunk;
            for (int k = 0; k < 6; ++k) {
                const int ch = cand[k] < nmax[0] ? cand[k] : nmax[0];
                if (k > 0 && ch == chunk) continue;
                float ms_min = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {        // (the first repetition also warms the instruction cache)
                    PHIHIP_CHECK_HIP(hipEventRecord(e0, s));
                    PHIHIP_TRY(launch(ch));
                    PHIHIP_CHECK_HIP(hipEventRecord(e1, s));
                    PHIHIP_CHECK_HIP(hipEventSynchronize(e1));
                    float ms = 0;
                    PHIHIP_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
                    ms_min = ms < ms_min ? ms : ms_min;
                }
                if (ms_min < best_ms * (k == 0 ? 1.0f : 0.98f)) { best_ms = ms_min; best = ch; }   // the planner's length stays unless > 2 % slower
            }
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            chunk = best;
            ctx->adv_tuned[key] = chunk;
        }
    }
    PHIHIP_TRY(launch(chunk));
    ctx->adv_last_chunk = DIM == 3 ? chunk : 0;
    ctx->adv_last_dma = dma;
    return PHIHIP_OK;
}

#ifdef PHIHIP_ONLY_LEAN
// development aid (tools/kernel_resources.py, ISA audits): `hipcc -DPHIHIP_ONLY_LEAN=1 --cuda-device-only -S` compiles the benchmark configuration's
// instantiation alone (3 s instead of 30)
int run_advect_self_tiled(phihip_ctx* ctx, const GridView& v, const void* const vel[3], void* const out[3], double dt, int halo, int kind, hipStream_t s) {
    const VelGrid g = make_velgrid(v);
#if PHIHIP_ONLY_LEAN == 2
    return launch_tile_consts<double, 3, 1, 16, 7, true>(ctx, v, g, vel, out, dt, kind, s);
#elif PHIHIP_ONLY_LEAN == 3
    return launch_tile_consts<float, 3, 1, 8, 7, true>(ctx, v, g, vel, out, dt, kind, s);
#else
    return launch_tile_consts<float, 3, 1, 8, 0, false>(ctx, v, g, vel, out, dt, kind, s);
#endif
}
}  // namespace phihip
#else
// a stored lower face (OFFM bit clear) on every axis does not exclude a CLOSED upper side (mixed boxes): the periodic / open code without
// the wall-value patch path is a separate instantiation of OFFM = 0 only
template <typename T, int DIM, int H, int T1, int OFFM>
static int launch_tile_off(phihip_ctx* ctx, const GridView& v, const VelGrid& vg, const void* const vel[3], void* const out[3], double dt, int kind, hipStream_t s) {
    bool closed = false;
    for (int a = v.ax0; a < 3; ++a) closed = closed || v.bc[a][0] == PHIHIP_BC_CLOSED || v.bc[a][1] == PHIHIP_BC_CLOSED;
    if constexpr (OFFM == 0) {
        if (!closed) return launch_tile_consts<T, DIM, H, T1, OFFM, false>(ctx, v, vg, vel, out, dt, kind, s);
    }
    return launch_tile_consts<T, DIM, H, T1, OFFM, true>(ctx, v, vg, vel, out, dt, kind, s);
}

template <typename T, int DIM, int H, int T1>
static int launch_tile(phihip_ctx* ctx, const GridView& v, const VelGrid& vg, const void* const vel[3], void* const out[3], double dt, int kind, hipStream_t s) {
    const int m = (vg.off[0] & 1) | ((vg.off[1] & 1) << 1) | ((vg.off[2] & 1) << 2);     // (2-D: off[0] = 0)
    switch (m) {
        case 0: return launch_tile_off<T, DIM, H, T1, 0>(ctx, v, vg, vel, out, dt, kind, s);
        case 2: return launch_tile_off<T, DIM, H, T1, 2>(ctx, v, vg, vel, out, dt, kind, s);
        case 4: return launch_tile_off<T, DIM, H, T1, 4>(ctx, v, vg, vel, out, dt, kind, s);
        case 6: return launch_tile_off<T, DIM, H, T1, 6>(ctx, v, vg, vel, out, dt, kind, s);
        default: break;
    }
    if (DIM == 3) switch (m) {
        case 1: return launch_tile_off<T, DIM, H, T1, (DIM == 3 ? 1 : 0)>(ctx, v, vg, vel, out, dt, kind, s);
        case 3: return launch_tile_off<T, DIM, H, T1, (DIM == 3 ? 3 : 0)>(ctx, v, vg, vel, out, dt, kind, s);
        case 5: return launch_tile_off<T, DIM, H, T1, (DIM == 3 ? 5 : 0)>(ctx, v, vg, vel, out, dt, kind, s);
        case 7: return launch_tile_off<T, DIM, H, T1, (DIM == 3 ? 7 : 0)>(ctx, v, vg, vel, out, dt, kind, s);
        default: break;
    }
    set_error("advect: unexpected face-offset pattern %d", m);
    return PHIHIP_ERR_BAD_ARG;
}

// halo: 1 or 2 samples (taps reach |displacement| < halo cells without leaving LDS)
int run_advect_self_tiled(phihip_ctx* ctx, const GridView& v, const void* const vel[3], void* const out[3], double dt, int halo, int kind, hipStream_t s) {
    const VelGrid g = make_velgrid(v);
    for (int c = v.ax0; c < 3; ++c)
        for (int a = v.ax0; a < 3; ++a)
            if (v.cn[c][a] < 4) return PHIHIP_ERR_UNSUPPORTED;   // a ring / window wider than the axis: not worth tiling (caller: gather kernels)
    LaunchScope ls(ctx, PHIHIP_K_ADVECT, s);
    const bool f64 = v.dtype == PHIHIP_F64;
    // fp64, 3-D: halo 1 with the 16-row tile (2 workgroups per CU, half the halo rows per sample) -- same-box A/B after the r3 instruction diet:
    // 384^3 closed 1.14 -> 0.94 ms, 256^3 periodic 0.232 -> 0.227 ms; fp32 keeps the 8-row tile (256^3: 0.119 vs 0.146 ms)
    // (profiles/r03_time_advect.jsonl). halo == 3 selects the 16-row tile for fp32 as well (A/B measurements).
    if (v.rank == 3 && (halo == 3 || (halo == 1 && f64))) {
        if (f64) PHIHIP_TRY((launch_tile<double, 3, 1, 16>(ctx, v, g, vel, out, dt, kind, s))); else PHIHIP_TRY((launch_tile<float, 3, 1, 16>(ctx, v, g, vel, out, dt, kind, s)));
    } else if (v.rank == 3) {
        if (halo >= 2) { if (f64) PHIHIP_TRY((launch_tile<double, 3, 2, 8>(ctx, v, g, vel, out, dt, kind, s))); else PHIHIP_TRY((launch_tile<float, 3, 2, 8>(ctx, v, g, vel, out, dt, kind, s))); }
        else PHIHIP_TRY((launch_tile<float, 3, 1, 8>(ctx, v, g, vel, out, dt, kind, s)));      // (fp64 halo 1 took the 16-row tile above)
    } else {
        if (halo >= 2) { if (f64) PHIHIP_TRY((launch_tile<double, 2, 2, 8>(ctx, v, g, vel, out, dt, kind, s))); else PHIHIP_TRY((launch_tile<float, 2, 2, 8>(ctx, v, g, vel, out, dt, kind, s))); }
        else { if (f64) PHIHIP_TRY((launch_tile<double, 2, 1, 8>(ctx, v, g, vel, out, dt, kind, s))); else PHIHIP_TRY((launch_tile<float, 2, 1, 8>(ctx, v, g, vel, out, dt, kind, s))); }
    }
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

}  // namespace phihip
#endif
