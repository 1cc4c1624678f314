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
//   * a lookup that leaves the staged window (|displacement| >= H cells, NaN) takes the global gather of advect_common.hpp behind a
//     wave-uniform branch -- same arithmetic, so the result does not depend on the path;
//   * planes p+H+1 are requested before plane p is computed and written to the ring after it: one barrier per plane.
// Algorithmic traffic 2 D words per cell; the kernel reads each velocity sample once per workgroup (+ halo) and writes each once.
#include "advect_common.hpp"

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

__device__ __forceinline__ int clamp_int(int x, int lo, int hi) { return min(max(x, lo), hi); }

template <typename T, int DIM, int H, int T1>
__global__ __launch_bounds__(kBlock) void advect_self_tile_kernel(TileGrid<T> g, CComp3a<T> vel, T* __restrict__ o0, T* __restrict__ o1,
                                                                  T* __restrict__ o2, int chunk, int tiles1, int tiles2, int nblk, int nmax0) {
    using C = AdvTile<T, DIM, H, T1>;
    constexpr int A0 = 3 - DIM;
    constexpr int T2 = C::T2, TY = C::TY, S = C::S, P1 = C::P1, P2 = C::P2, PLANE = C::PLANE, NC = C::NC, NP = C::NP, KP = C::KP;
    __shared__ T lds[NC * NP * PLANE];

    const int tid = threadIdx.x, tx = tid % T2, ty = tid / T2;
    const int b = blockIdx.y;
    int bid = blockIdx.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);   // XCD-aware order: neighbouring tiles share an XCD's L2
    const int t2 = bid % tiles2;
    const int t1 = (bid / tiles2) % tiles1;
    const int c0 = bid / (tiles2 * tiles1);
    const int lo1 = t1 * T1, lo2 = t2 * T2;                            // tile origin (stored indices, the same for every component)
    const int pb = DIM == 3 ? c0 * chunk : 0;
    const int pe = DIM == 3 ? min(pb + chunk, nmax0) : 1;
    T* const outp[3] = {o0, o1, o2};

    // Does this workgroup's window reach beyond a CLOSED side (constant padding)? Uniform, so interior tiles (and periodic / open
    // domains altogether) run the fill without a single select.
    bool has_const = false;
#pragma unroll
    for (int c = A0; c < 3; ++c) {
        has_const = has_const || (g.bc[1][0] == PHIHIP_BC_CLOSED && lo1 - H < 0) || (g.bc[1][1] == PHIHIP_BC_CLOSED && lo1 + T1 + H > g.cn[c][1]) ||
                    (g.bc[2][0] == PHIHIP_BC_CLOSED && lo2 - H < 0) || (g.bc[2][1] == PHIHIP_BC_CLOSED && lo2 + T2 + H > g.cn[c][2]);
        if (DIM == 3) has_const = has_const || (g.bc[0][0] == PHIHIP_BC_CLOSED && pb - H < 0) || (g.bc[0][1] == PHIHIP_BC_CLOSED && pe + H > g.cn[c][0]);
    }

    // ---- per-thread fill descriptors (plane-invariant) -------------------------------------------------------------------------
    // element kp of component c: row ty + kp TY, column tx of the window. eoff = in-plane element offset after wrap / clamp;
    // rcode / ccode: 0 = stored sample, 1 / 2 = the lower / upper CONSTANT side of a1 resp. a2 supplies the value
    int eoff[3][KP];
    int rcode[3][KP];
    int ccode[3];
    const bool last_ok = ty + (KP - 1) * TY < P1;        // only the last pass can run past the window's rows
#pragma unroll
    for (int c = A0; c < 3; ++c) {
        const int k = pad_index(lo2 - H + tx, g.cn[c][2], g.bc[2][0], g.bc[2][1]);
        ccode[c] = k < 0 ? -k : 0;
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
            const int j = pad_index(lo1 - H + ty + kp * TY, g.cn[c][1], g.bc[1][0], g.bc[1][1]);
            rcode[c][kp] = j < 0 ? -j : 0;
            eoff[c][kp] = (j < 0 ? 0 : j * g.cn[c][2]) + (k < 0 ? 0 : k);
        }
    }
    // tail element (halo columns T2 .. T2+2H-1 of every row): component, row and column of THIS thread
    const bool has_tail = tid < C::NTAIL;
    const int tail_ci = has_tail ? tid / (P1 * 2 * H) : 0;
    const int tail_c = A0 + tail_ci;
    const int tail_r = has_tail ? (tid % (P1 * 2 * H)) / (2 * H) : 0;
    const int tail_q = T2 + tid % (2 * H);
    int tail_off = 0, tail_rcode = 0, tail_ccode = 0, tail_n0 = 1, tail_n1 = 1, tail_n2 = 1;
    const T* tail_base = vel.p[2];
#pragma unroll
    for (int c = A0; c < 3; ++c)     // (selects over the static component index: dynamic indexing of kernel arguments goes through scratch)
        if (c == tail_c) { tail_n0 = g.cn[c][0]; tail_n1 = g.cn[c][1]; tail_n2 = g.cn[c][2]; tail_base = vel.p[c] + (long long)b * g.ccells[c]; }
    if (has_tail) {
        const int kk = pad_index(lo2 - H + tail_q, tail_n2, g.bc[2][0], g.bc[2][1]);
        const int j = pad_index(lo1 - H + tail_r, tail_n1, g.bc[1][0], g.bc[1][1]);
        tail_ccode = kk < 0 ? -kk : 0;
        tail_rcode = j < 0 ? -j : 0;
        tail_off = (j < 0 ? 0 : j * tail_n2) + (kk < 0 ? 0 : kk);
    }
    const long long tail_pstride = (long long)tail_n1 * tail_n2;

    // request plane i0 of every component into registers: plain loads; the constant sides are patched in afterwards (cold, uniform)
    auto load_plane = [&](int i0, T (&R)[3][KP], T& tailv) {
#pragma unroll
        for (int c = A0; c < 3; ++c) {
            const int k = DIM == 3 ? pad_index(i0, g.cn[c][0], g.bc[0][0], g.bc[0][1]) : 0;
            const T* __restrict__ base = vel.p[c] + (long long)b * g.ccells[c] + (k < 0 ? 0 : (long long)k * g.cn[c][1] * g.cn[c][2]);
#pragma unroll
            for (int kp = 0; kp < KP; ++kp)
                if (kp < KP - 1 || last_ok) R[c][kp] = base[eoff[c][kp]];
        }
        if (has_tail) {
            const int k = DIM == 3 ? pad_index(i0, tail_n0, g.bc[0][0], g.bc[0][1]) : 0;
            tailv = tail_base[(k < 0 ? 0 : (long long)k * tail_pstride) + tail_off];
        }
        if (has_const) {   // PhiML pads axis after axis: the LAST axis outside a constant side decides (a2 over a1 over a0)
#pragma unroll
            for (int c = A0; c < 3; ++c) {
                const int k = DIM == 3 ? pad_index(i0, g.cn[c][0], g.bc[0][0], g.bc[0][1]) : 0;
                // (values first, selects after: a select between two kernel-argument LOADS becomes a per-lane address + flat load)
                const T k00 = g.bcv[0][0][c], k01 = g.bcv[0][1][c], k10 = g.bcv[1][0][c], k11 = g.bcv[1][1][c], k20 = g.bcv[2][0][c], k21 = g.bcv[2][1][c];
                const T pv = k == -1 ? k00 : k01;
                const T cv = ccode[c] == 1 ? k20 : k21;
#pragma unroll
                for (int kp = 0; kp < KP; ++kp) {
                    T v = R[c][kp];
                    const T rv = rcode[c][kp] == 1 ? k10 : k11;
                    v = k < 0 ? pv : v;
                    v = rcode[c][kp] ? rv : v;
                    v = ccode[c] ? cv : v;
                    R[c][kp] = v;
                }
                if (has_tail && c == tail_c) {
                    const T rv = tail_rcode == 1 ? k10 : k11, cv2 = tail_ccode == 1 ? k20 : k21;
                    tailv = k < 0 ? pv : tailv;
                    tailv = tail_rcode ? rv : tailv;
                    tailv = tail_ccode ? cv2 : tailv;
                }
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
        if (has_tail) lds[(tail_ci * NP + slot) * PLANE + tail_r * P2 + tail_q] = tailv;
    };
    auto slot_of = [&](int i0) -> int {   // uniform
        if (DIM == 2) return 0;
        if ((NP & (NP - 1)) == 0) return i0 & (NP - 1);
        const int m = i0 % NP;
        return m < 0 ? m + NP : m;
    };

    // ---- prologue: planes pb-H .. pb+H ------------------------------------------------------------------------------------------
    T R[3][KP];
    T tailv = T(0);
    if (DIM == 3) {
        for (int i0 = pb - H; i0 <= pb + H; ++i0) {
            load_plane(i0, R, tailv);
            store_plane(slot_of(i0), R, tailv);
        }
    } else {
        load_plane(0, R, tailv);
        store_plane(0, R, tailv);
    }
    __syncthreads();

    int bc[3][2];
#pragma unroll
    for (int a = 0; a < 3; ++a) { bc[a][0] = g.bc[a][0]; bc[a][1] = g.bc[a][1]; }

    for (int p = pb; p < pe; ++p) {
        const bool more = DIM == 3 && p + 1 < pe;
        if (more) load_plane(p + H + 1, R, tailv);
        const int sl_m = slot_of(p - 1), sl_0 = slot_of(p), sl_p = slot_of(p + 1);

#pragma unroll 1
        for (int s = 0; s < S; ++s) {   // (not unrolled: the components of one position already give D independent samples)
            const int j1 = lo1 + ty + s * TY, j2 = lo2 + tx;
            const int center = (ty + s * TY + H) * P2 + tx + H;
#pragma unroll
            for (int ca = A0; ca < 3; ++ca) {
                const bool valid = p < g.cn[ca][0] && j1 < g.cn[ca][1] && j2 < g.cn[ca][2];   // every LDS read below is in bounds regardless
                const int idx[3] = {p, j1, j2};
                // LDS element of component x at (plane offset d0 in {-1,0,1}, row offset d1, column offset d2) from this sample
                auto at = [&](int x, int d0, int d1, int d2) -> T {
                    const int sl = d0 < 0 ? sl_m : (d0 > 0 ? sl_p : sl_0);
                    return lds[((x - A0) * NP + sl) * PLANE + center + d1 * P2 + d2];
                };
                T u[3] = {T(0), T(0), T(0)};
                u[ca] = at(ca, 0, 0, 0);
#pragma unroll
                for (int cb = A0; cb < 3; ++cb) {
                    if (cb == ca) continue;
                    // component cb at this ca-face: cells (m-1, m) along ca, faces (s, s+1) along cb, in cb's stored indices:
                    // offsets (off[ca] - 1 + ia) along ca and (-off[cb] + ib) along cb (advect_common.hpp face_velocity)
                    T v[2][2];
#pragma unroll
                    for (int ia = 0; ia < 2; ++ia)
#pragma unroll
                        for (int ib = 0; ib < 2; ++ib) {
                            int d[3] = {0, 0, 0};
                            d[ca] = g.off[ca] - 1 + ia;
                            d[cb] = -g.off[cb] + ib;
                            v[ia][ib] = at(cb, d[0], d[1], d[2]);
                        }
                    if (ca < cb) {
                        const T a0 = v[1][0] * T(0.5) + v[0][0] * T(0.5), a1 = v[1][1] * T(0.5) + v[0][1] * T(0.5);
                        u[cb] = a1 * T(0.5) + a0 * T(0.5);
                    } else {
                        const T a0 = v[0][1] * T(0.5) + v[0][0] * T(0.5), a1 = v[1][1] * T(0.5) + v[1][0] * T(0.5);
                        u[cb] = a1 * T(0.5) + a0 * T(0.5);
                    }
                }
                // back-trace in the index space of component ca's own array
                T coord[3] = {T(0), T(0), T(0)}, fl[3] = {T(0), T(0), T(0)}, fr[3] = {T(0), T(0), T(0)};
#pragma unroll
                for (int a = A0; a < 3; ++a) {
                    coord[a] = (T)idx[a] - u[a] * g.shift[a];
                    fl[a] = floor(coord[a]);
                    fr[a] = coord[a] - fl[a];
                }
                // the 2^D taps i_lo, i_lo + 1 must lie in the staged window (NaN compares false -> global path)
                bool inwin = fl[1] >= (T)(lo1 - H) && fl[1] <= (T)(lo1 + T1 + H - 2) && fl[2] >= (T)(lo2 - H) && fl[2] <= (T)(lo2 + T2 + H - 2);
                if (DIM == 3) inwin = inwin && fl[0] >= (T)(p - H) && fl[0] <= (T)(p + H - 1);
                // LDS lookup for every lane (addresses clamped into the window; lanes outside it are overwritten below)
                const int i1 = clamp_int((int)fl[1] - (lo1 - H), 0, P1 - 2), i2 = clamp_int((int)fl[2] - (lo2 - H), 0, P2 - 2);
                int base0 = (ca - A0) * NP * PLANE + i1 * P2 + i2, base1 = base0;
                if (DIM == 3) {
                    const int dz = clamp_int((int)fl[0] - p, -H, H - 1);
                    int s0, s1;
                    if ((NP & (NP - 1)) == 0) { s0 = (p + dz) & (NP - 1); s1 = (p + dz + 1) & (NP - 1); }
                    else {
                        s0 = sl_0 + dz; s0 += s0 < 0 ? NP : 0; s0 -= s0 >= NP ? NP : 0;
                        s1 = s0 + 1; s1 -= s1 >= NP ? NP : 0;
                    }
                    base1 = base0 + s1 * PLANE;
                    base0 += s0 * PLANE;
                }
                T val = T(0);
                // same corner order and weight products as gather_multilinear (a0 = lowest bit)
#pragma unroll
                for (int corner = 0; corner < (1 << DIM); ++corner) {
                    const int b0 = DIM == 3 ? (corner & 1) : 0;
                    const int b1 = DIM == 3 ? ((corner >> 1) & 1) : (corner & 1);
                    const int b2 = DIM == 3 ? ((corner >> 2) & 1) : ((corner >> 1) & 1);
                    T w = T(1);
                    if (DIM == 3) w *= b0 ? fr[0] : (T(1) - fr[0]);
                    w *= b1 ? fr[1] : (T(1) - fr[1]);
                    w *= b2 ? fr[2] : (T(1) - fr[2]);
                    val += lds[(b0 ? base1 : base0) + b1 * P2 + b2] * w;
                }
                const bool slow = valid && !inwin;
                if (wave_any(slow)) {
                    if (slow) {
                        const int n[3] = {g.cn[ca][0], g.cn[ca][1], g.cn[ca][2]};
                        const T cv[3][2] = {{g.bcv[0][0][ca], g.bcv[0][1][ca]}, {g.bcv[1][0][ca], g.bcv[1][1][ca]}, {g.bcv[2][0][ca], g.bcv[2][1][ca]}};
                        AxisPair<T> ax[3];
                        T fr2[3];
                        lookup_pairs<T, DIM>(coord, n, bc, cv, ax, fr2);
                        val = gather_multilinear<T, DIM>(vel.p[ca] + (long long)b * g.ccells[ca], ax, fr2);
                    }
                }
                if (valid) outp[ca][(long long)b * g.ccells[ca] + ((long long)p * g.cn[ca][1] + j1) * g.cn[ca][2] + j2] = val;
            }
        }
        if (more) store_plane(slot_of(p + H + 1), R, tailv);
        __syncthreads();
    }
}

template <typename T, int DIM, int H, int T1>
static int launch_tile(const GridView& v, const VelGrid& vg, const void* const vel[3], void* const out[3], double dt, hipStream_t s) {
    const TileGrid<T> g = make_tilegrid<T>(vg, dt);
    using C = AdvTile<T, DIM, H, T1>;
    int nmax[3] = {1, 1, 1};
    for (int a = 0; a < 3; ++a)
        for (int c = v.ax0; c < 3; ++c) nmax[a] = v.cn[c][a] > nmax[a] ? v.cn[c][a] : nmax[a];
    const int tiles1 = ceil_div(nmax[1], T1), tiles2 = ceil_div(nmax[2], C::T2);
    int chunk = 1, chunks0 = 1;
    if (DIM == 3) {
        // ~1536 workgroups (3 rounds of 2 per CU) unless that makes the chunks shorter than 8 planes (2H+1 prologue planes per chunk)
        const long long tiles = (long long)tiles1 * tiles2 * v.batch;
        const int want = (int)((1536 + tiles - 1) / tiles);
        chunk = ceil_div(nmax[0], want < 1 ? 1 : want);
        chunk = chunk < 8 ? 8 : chunk;
        chunk = chunk > nmax[0] ? nmax[0] : chunk;
        chunks0 = ceil_div(nmax[0], chunk);
    }
    const int nblk = tiles1 * tiles2 * chunks0;
    CComp3a<T> vv{{(const T*)vel[0], (const T*)vel[1], (const T*)vel[2]}};
    hipLaunchKernelGGL((advect_self_tile_kernel<T, DIM, H, T1>), dim3(nblk, v.batch), dim3(kBlock), 0, s, g, vv, (T*)out[0], (T*)out[1], (T*)out[2],
                       chunk, tiles1, tiles2, nblk, nmax[0]);
    return PHIHIP_OK;
}

// halo: 1 or 2 samples (taps reach |displacement| < halo cells without leaving LDS)
int run_advect_self_tiled(phihip_ctx* ctx, const GridView& v, const void* const vel[3], void* const out[3], double dt, int halo, hipStream_t s) {
    const VelGrid g = make_velgrid(v);
    LaunchScope ls(ctx, PHIHIP_K_ADVECT, s);
    const bool f64 = v.dtype == PHIHIP_F64;
    if (v.rank == 3) {
        if (halo >= 2) { if (f64) launch_tile<double, 3, 2, 8>(v, g, vel, out, dt, s); else launch_tile<float, 3, 2, 8>(v, g, vel, out, dt, s); }
        else { if (f64) launch_tile<double, 3, 1, 16>(v, g, vel, out, dt, s); else launch_tile<float, 3, 1, 16>(v, g, vel, out, dt, s); }
    } else {
        if (halo >= 2) { if (f64) launch_tile<double, 2, 2, 8>(v, g, vel, out, dt, s); else launch_tile<float, 2, 2, 8>(v, g, vel, out, dt, s); }
        else { if (f64) launch_tile<double, 2, 1, 8>(v, g, vel, out, dt, s); else launch_tile<float, 2, 1, 8>(v, g, vel, out, dt, s); }
    }
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

}  // namespace phihip
