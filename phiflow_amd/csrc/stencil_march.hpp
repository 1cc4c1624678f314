// stencil_march.hpp -- the 5/7-point pressure operator (fluid.masked_laplace, /root/reference phi/physics/fluid.py:165-202)
// as a matrix-free plane-marching kernel for gfx950, shared by every phase of the CG loop.
//
// One 256-thread workgroup owns a (T1 x T2) tile of the two fast axes and marches over a chunk of planes of the slow
// axis a0. Per plane each thread owns R consecutive rows x one 16-byte vector of the fast axis:
//   * a0 neighbours live in registers (previous / current / next plane are rotated, every plane is read from HBM once),
//   * a1 / a2 neighbours of the current plane come from an LDS copy of the tile (+ a one-cell halo ring that designated
//     threads fetch with the boundary rule applied: wrap / clamp / zero), double buffered => one barrier per plane,
//   * every request is issued one whole trip ahead (source plane i+2, halo items and own-cell operands of plane i+1 while plane i is
//     computed) and every load / store of the loop is unconditional, so that the compiler's s_waitcnt counts instead of draining.
// The "source" S whose Laplacian is taken and the epilogue differ per MODE:
//   APPLY   S = p                  out = A S
//   RESID   S = x                  r = y - A S                     sum r^2, sum y^2
//   MATVEC  S = r + beta * d_old   d_new = S                       sum S * (A S)        (q = A d is never stored)
//   UPDATE  S = d                  x += alpha S ; r -= alpha A S   sum r_new^2          (q recomputed from d)
// so one CG iteration moves 3 + 5 = 8 words per cell through HBM instead of the textbook 10-11.
// Variants of the same source (template flags, r4): UNAL -- rows that are not whole vectors / unaligned buffers (element-aligned global vectors,
// the last vector of a row overlaps its neighbour); ROWT -- tiles of WHOLE rows with a run-time lane count per row and no halo columns.
// UPDATE_R / UPDATE_X2 halve the traffic of `x`: the solution does not enter the recurrence, so every other iteration skips it
// (UPDATE_R: r -= alpha A S only, 3 words) and the next one adds both steps at once -- the previous search direction is recovered
// from operands the kernel reads anyway, d_k = (d_{k+1} - r_{k+1}) / beta_{k+1}:
//   UPDATE_X2   x += (alpha_k / beta_{k+1}) (S - r) + alpha_{k+1} S ; r -= alpha_{k+1} A S        => 7 words per iteration on average.
// CG1 is the whole iteration of the SINGLE-REDUCTION form of CG (Chronopoulos & Gear 1989) in ONE launch, for grids whose iteration is
// bound by the two kernel boundaries rather than by traffic (batched 2-D, small 3-D): with w = A r and s = A p carried as vectors,
//   S = r - alpha (w + beta s)   [= r_new, also on the halo]      p = r + beta p ; s = w + beta s ; x += alpha p ; r = S ; w = A S
//   sum S^2 (= gamma'), sum (A S) S (= delta'), sum S s (= mu'), sum (A S) p (= nu'), sum p s (= sigma)
//   ->   beta' = gamma' / gamma ,  alpha' = gamma' / (p'.A p') with p'.A p' = delta' + beta' (mu' + nu') + beta'^2 sigma
// (r3) The textbook closure p'.A p' = delta' - beta' gamma' / alpha rests on the orthogonality relations of exact CG; in fp32 they erode and
// the attainable residual stalled 1-2 digits above the two-launch form (closed 512^2: 9e-4 against 3e-5). The five-sum form is an IDENTITY
// for the vectors the kernel actually holds -- (r' + beta' p).(w' + beta' s) expanded -- so it is as accurate as computing p'.s' directly, which
// is what the two-launch form does, at the same single reduction point (tools/cg1_accuracy.py: same iteration counts, same floor).
// Same iterates as PhiML's cg, one global reduction point per iteration, 10 words per cell instead of 7.
// MATVEC_AD / UPDATE_AD are the same passes for PhiML's 'CG-adaptive' (SURVEY Appendix B.2): they additionally reduce
// sum d * r resp. sum r_new * (A d), from which alpha = (d.r)/(d.q) and d = r - ((r.q)/(d.q)) d are formed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace phihip {

constexpr int kBlock = 256;
constexpr int kWave = 64;

// XCD-aware workgroup order: the dispatcher deals workgroups round-robin to the 8 XCDs (workgroup b runs on XCD b % 8), each with its own
// L2. Map the hardware index to a logical one so that every XCD works on ONE contiguous range of logical indices -- neighbouring tiles
// (which share halo rows / planes) then share an L2. Bijection for any count: XCD x owns n/8 (+1 if x < n % 8) consecutive indices.
__device__ __forceinline__ int xcd_order(int b, int n) {
    const int q = n >> 3, r = n & 7, x = b & 7;
    return x * q + (x < r ? x : r) + (b >> 3);
}
// the same ranges walked from their far ends: the k-th workgroup an XCD receives takes the LAST-but-k tile of that XCD's range (sawtooth, see march_kernel)
__device__ __forceinline__ int xcd_order_rev(int b, int n) {
    const int q = n >> 3, r = n & 7, x = b & 7;
    const int len = q + (x < r ? 1 : 0);
    return x * q + (x < r ? x : r) + (len - 1 - (b >> 3));
}


enum NeighbourRule { NB_WRAP = 0, NB_CLAMP = 1, NB_ZERO = 2, NB_HALO = 3 };   // NB_HALO (axis a0 only): the plane comes from a
                                                                            // neighbour slab's halo buffer (MarchArgs::a_lo ...)
enum MarchMode { MODE_APPLY = 0, MODE_RESID = 1, MODE_MATVEC = 2, MODE_UPDATE = 3, MODE_MATVEC_AD = 4, MODE_UPDATE_AD = 5, MODE_UPDATE_R = 6, MODE_UPDATE_X2 = 7,
                 MODE_RESID_BAL = 8,     // RESID that also balances y: y -= shift * active, written back (once per projection)
                 MODE_APPLY_DOT = 9,     // w = A r with sum r^2 and sum (A r) r: start / refresh of the single-reduction CG
                 MODE_CG1 = 10 };        // one WHOLE iteration of the single-reduction (Chronopoulos-Gear) CG, see below

// Per batch entry CG control block (device memory). There is no separate "scalar" kernel between the phases of an
// iteration: every workgroup of the NEXT kernel re-reduces the previous kernel's per-workgroup partial sums in a fixed order
// (deterministic, no atomics / fences -- the kernel boundary orders the data) and advances the control block in registers;
// workgroup 0 stores it to the other of two slots for the kernel after that.
struct CgState {
    double alpha, beta;
    double rsq, rsq0, rhs_sq, tol_sq, dq;
    double alpha_prev;       // alpha of the previous iteration (UPDATE_X2 applies it together with the current one)
    double sigma;            // single-reduction CG: p.Ap of the step just taken (diagnostic; the five-sum closure does not chain it)
    int32_t cont, iterations, converged, diverged;
    int32_t pending;         // 1: x still lacks alpha * d of the last iteration (UPDATE_R ran); flushed by the paired update or at the end
    int32_t pend_buf;        // which of the two d buffers holds that direction
};

struct CgParams {
    double rtol, atol;
    int32_t max_iter, pad;
};

enum CgPrologue {
    PRO_NONE = 0,       // no control block involved (APPLY, initial residual)
    PRO_CONT = 1,       // read the continue flag only (true-residual refresh)
    PRO_FIRST = 2,      // build the control block from the initial residual's sums (rr, yy); beta = 0
    PRO_BETA = 3,       // rsq_new from the UPDATE / refresh partials: beta, convergence flags
    PRO_ALPHA = 4,      // dq from the MATVEC partials: alpha, iteration count
    PRO_BETA_AD = 5,    // 'CG-adaptive': (rsq_new, r_new.q) -> beta = -(r.q)/(d.q), convergence flags
    PRO_ALPHA_AD = 6,   // 'CG-adaptive': (d.q, d.r) -> alpha = (d.r)/(d.q), iteration count
    PRO_CG1 = 7         // single-reduction CG: (gamma', delta') of the previous launch -> convergence flags, then beta, alpha, count
};

// one 64-bit store that the HOST may read at any time (pinned, device-mapped memory)
__device__ __forceinline__ void publish_flag(unsigned long long* p, unsigned long long v) {
#ifdef __HIP_DEVICE_COMPILE__
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#else
    *reinterpret_cast<volatile unsigned long long*>(p) = v;
#endif
}

__device__ __forceinline__ bool cg_finite(double v) { return (v == v) && v <= 1.7e308 && v >= -1.7e308; }

// PhiML's cg loop body bookkeeping, split at its two reductions
__device__ __forceinline__ CgState cg_advance(int kind, CgState s, double sum1, double sum2, const CgParams& prm, double sum3 = 0, double sum4 = 0,
                                             double sum5 = 0) {
    if (kind == PRO_FIRST) {
        s.alpha = 0; s.beta = 0; s.dq = 0; s.alpha_prev = 0;
        s.pending = 0; s.pend_buf = 0;
        s.rsq = sum1; s.rsq0 = sum1; s.rhs_sq = sum2;
        const double t1 = prm.rtol * prm.rtol * sum2, t2 = prm.atol * prm.atol;
        s.tol_sq = t1 > t2 ? t1 : t2;
        s.iterations = 0;
        s.diverged = cg_finite(sum1) ? 0 : 1;
        s.converged = sum1 <= s.tol_sq ? 1 : 0;
        s.cont = (!s.converged && !s.diverged && prm.max_iter > 0) ? 1 : 0;
    } else if (kind == PRO_BETA || kind == PRO_BETA_AD) {
        if (s.cont) {
            if (kind == PRO_BETA) s.beta = s.rsq != 0 ? sum1 / s.rsq : 0;   // divide_no_nan
            else s.beta = s.dq != 0 ? -sum2 / s.dq : 0;
            s.rsq = sum1;
            s.diverged = (!cg_finite(sum1) || (s.rsq0 > 0 && sum1 / s.rsq0 > 100 && s.iterations >= 8)) ? 1 : 0;
            s.converged = sum1 <= s.tol_sq ? 1 : 0;
            s.cont = (!s.converged && !s.diverged && s.iterations < prm.max_iter) ? 1 : 0;
        }
    } else if (kind == PRO_CG1) {
        if (s.cont) {
            // the residual the previous launch (or the start / refresh) produced: PhiML's checks first ...
            const double g_old = s.rsq;
            if (s.iterations > 0 || s.dq != 0) {   // (not the very first launch: PRO_FIRST already judged r0)
                s.rsq = sum1;
                s.diverged = (!cg_finite(sum1) || (s.rsq0 > 0 && sum1 / s.rsq0 > 100 && s.iterations >= 8)) ? 1 : 0;
                s.converged = sum1 <= s.tol_sq ? 1 : 0;
                s.cont = (!s.converged && !s.diverged && s.iterations < prm.max_iter) ? 1 : 0;
            }
            if (s.cont) {   // ... then this launch's step
                const bool first = s.iterations == 0 && s.dq == 0;
                s.beta = first ? 0 : (g_old != 0 ? sum1 / g_old : 0);
                // p'.A p' = (r' + beta p).(w' + beta s) = delta' + beta (mu' + nu') + beta^2 sigma: sums over the vectors as they are (see the header)
                const double denom = first ? sum2 : sum2 + s.beta * (sum3 + sum4) + s.beta * s.beta * sum5;
                s.sigma = denom;
                s.alpha_prev = s.alpha;
                s.alpha = denom != 0 ? sum1 / denom : 0;
                s.dq = denom != 0 ? denom : 1;        // d.Ad of this step (kept non-zero: marks "not the first launch")
                s.iterations += 1;
            }
        }
    } else if (kind == PRO_ALPHA || kind == PRO_ALPHA_AD) {
        if (s.cont) {
            s.iterations += 1;
            s.dq = sum1;
            s.alpha_prev = s.alpha;
            s.alpha = sum1 != 0 ? (kind == PRO_ALPHA ? s.rsq : sum2) / sum1 : 0;
        }
    }
    return s;
}

struct MarchGrid {
    int n0, n1, n2;        // cells per internal axis (n0 == 1 for 2-D grids)
    int nb[3][2];          // NeighbourRule per internal axis / side for the pressure
    long long cells;       // n0 * n1 * n2
    int tiles1, tiles2, chunks0, chunk, nblk;
    int flags_per_batch;   // 1: flags array has a batch dimension, 0: shared by all batch entries
    int bidir;             // 1: launch the BIDIR instantiation: odd chunks march down (short chunks share their boundary planes in time)
    int tpr_rt;            // ROWT instantiation: lanes per row (n2 / V), else 0
};

template <typename T>
struct MarchArgs {
    const T* a;            // APPLY p | RESID x | MATVEC r     | UPDATE d
    const T* b;            //         | RESID y | MATVEC d_old |
    T* o1;                 // APPLY out | RESID r | MATVEC d_new | UPDATE x (in/out)
    T* o2;                 //                                     | UPDATE r (in/out)
    const uint8_t* flags;  // per-cell stencil flags or nullptr
    const CgState* st_in;  // [batch] control block written by the previous kernel
    CgState* st_out;       // [batch] slot for the next kernel (written by workgroup 0)
    const double* pin1;    // [batch][nblk] partial sums to reduce in the prologue
    const double* pin2;    //               (PRO_FIRST: sum y^2, PRO_ALPHA_AD: sum d r, PRO_BETA_AD: sum r q)
    double* part1;         // [batch][nblk] partial sums produced by this kernel
    double* part2;         // [batch][nblk]   (RESID: sum y^2, MATVEC_AD: sum d r, UPDATE_AD: sum r q)
    // single-reduction CG (CG1, APPLY_DOT): three more sums per launch -- mu = r'.s, nu = (A r').p, sigma = p.s -- and the previous launch's
    const double* pin3; const double* pin4; const double* pin5;
    double* part3; double* part4; double* part5;
    CgParams prm;
    int prologue;          // CgPrologue
    int nblk_in;           // workgroups per batch entry of the kernel that produced pin1 / pin2
    int pend_buf;          // UPDATE_R: index of the d buffer this launch reads (recorded with the pending flag)
    unsigned long long* host_flags;   // MATVEC: host-mapped [batch] array that receives (seq << 32 | continue flag), or nullptr
    unsigned int seq;
    T w0, w1, w2;          // scale / dx^2 per internal axis (scale = 1: the pressure operator)
    T ident;               // the kernels apply ident * S + sum_a w_a (d^2 S)_a: 0 for the pressure, 1 for implicit diffusion (I - k dt L)
    // slab decomposition along a0: one plane [batch][n1][n2] of the source array(s) below plane 0 / above plane
    // n0 - 1, received from the neighbouring rank; read where g.nb[0][side] == NB_HALO
    const T* a_lo; const T* a_hi;
    const T* b_lo; const T* b_hi;
    // CG1 only: third stencil source (s = A p of the previous step) and the remaining outputs. a = r, b = w, c = s (inputs of this
    // launch); o1 = r_new, o2 = w_new, o3 = s_new (the OTHER set of the three ping-pong pairs: neighbours still read the inputs),
    // o4 = p, o5 = x (own cells only: in place)
    const T* c;
    T* o3; T* o4; T* o5;
    // RESID_BAL only: fluid._balance_divergence folded into the initial residual -- y is read as y - shift[b] * active and written
    // back balanced to `yout` (the refreshes and the caller see the balanced right-hand side); saves the separate read + write pass
    const double* shift;
    T* yout;
    // 16 bytes that lanes outside the grid store into (every store of the plane loop is unconditional, see march_kernel); set by launch_march
    T* dump;
};

template <typename T, int V>
struct alignas(sizeof(T) * V) Vec {
    T v[V];
};

template <typename T, int V>
__device__ __forceinline__ Vec<T, V> vec_zero() {
    Vec<T, V> r;
#pragma unroll
    for (int i = 0; i < V; ++i) r.v[i] = T(0);
    return r;
}

template <typename T, int V>
__device__ __forceinline__ Vec<T, V> vec_load(const T* p) {
    return *reinterpret_cast<const Vec<T, V>*>(p);
}

template <typename T, int V>
__device__ __forceinline__ void vec_store(T* p, const Vec<T, V>& x) {
    *reinterpret_cast<Vec<T, V>*>(p) = x;
}

// ---- rows that are not whole 16-byte vectors / buffers that are not 16-byte aligned (UNAL instantiation of march_kernel, r4) -----------------
// Global accesses take the vector at ELEMENT alignment (gfx950 serves dwordx4 at any 4-byte address), and the thread that would own the
// partial vector at the end of a row owns the row's LAST V cells instead: its vector overlaps its left neighbour's by V - (n2 mod V) cells,
// both compute the same bits for those cells from the same operands and store them twice, only the sums count them once. No element is
// rotated or masked in registers (a first version that loaded the partial vector early and rotated it with selects was turned into a
// dynamically indexed array in SCRATCH by the compiler: 80-270 bytes per lane), no byte outside the arrays is touched.
template <typename T, int V>
struct __attribute__((packed, aligned(sizeof(T)))) VecP {
    T v[V];
};
template <typename T, int V, bool UNAL>
__device__ __forceinline__ Vec<T, V> vec_load_g(const T* p) {
    if (!UNAL) return *reinterpret_cast<const Vec<T, V>*>(p);
    const VecP<T, V> t = *reinterpret_cast<const VecP<T, V>*>(p);
    Vec<T, V> r;
#pragma unroll
    for (int i = 0; i < V; ++i) r.v[i] = t.v[i];
    return r;
}
template <typename T, int V, bool UNAL>
__device__ __forceinline__ void vec_store_g(T* p, const Vec<T, V>& x) {
    if (!UNAL) { *reinterpret_cast<Vec<T, V>*>(p) = x; return; }
    VecP<T, V> t;
#pragma unroll
    for (int i = 0; i < V; ++i) t.v[i] = x.v[i];
    *reinterpret_cast<VecP<T, V>*>(p) = t;
}
// LDS copy of the tile: vectors sit on 16-byte boundaries except the overlapping last vector of a ragged row (`odd`: element by element)
template <typename T, int V>
__device__ __forceinline__ Vec<T, V> lds_load(const T* p, bool odd) {
    if (!odd) return *reinterpret_cast<const Vec<T, V>*>(p);
    Vec<T, V> r;
#pragma unroll
    for (int i = 0; i < V; ++i) r.v[i] = p[i];
    return r;
}
template <typename T, int V>
__device__ __forceinline__ void lds_store(T* p, const Vec<T, V>& x, bool odd) {
    if (!odd) { *reinterpret_cast<Vec<T, V>*>(p) = x; return; }
#pragma unroll
    for (int i = 0; i < V; ++i) p[i] = x.v[i];
}

// index of the neighbour one step outside [0, n): wrap / clamp / zero ghost
__device__ __forceinline__ int nb_index(int i, int n, int rule_lo, int rule_hi, bool& zero) {
    if (i < 0) {
        if (rule_lo == NB_WRAP) return i + n;
        if (rule_lo == NB_ZERO) zero = true;
        return 0;
    }
    if (i >= n) {
        if (rule_hi == NB_WRAP) return i - n;
        if (rule_hi == NB_ZERO) zero = true;
        return n - 1;
    }
    return i;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

// sum over the 256 threads of a block; result valid in thread 0. `red` = kBlock / kWave doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < kBlock / kWave; ++w) s += red[w];
    }
    return s;
}

// N sums at once over the 256 threads of a block (one pair of barriers instead of N): results valid in thread 0. `red` = N * kBlock / kWave doubles.
template <int N>
__device__ __forceinline__ void block_sum_n(double (&v)[N], double* red) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = wave_sum(v[k]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) red[k * (kBlock / kWave) + wave] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            double s = 0;
#pragma unroll
            for (int w = 0; w < kBlock / kWave; ++w) s += red[k * (kBlock / kWave) + w];
            v[k] = s;
        }
    }
}

// fixed-order sum of n partials by the whole block; valid in thread 0
__device__ __forceinline__ double reduce_partials(const double* part, int n, double* red) {
    double s = 0;
    for (int i = threadIdx.x; i < n; i += kBlock) s += part[i];
    return block_sum(s, red);
}

// Prologue shared by every kernel of the CG loop: returns the advanced control block to all threads of the workgroup.
__device__ __forceinline__ CgState cg_prologue(int kind, const CgState* st_in, CgState* st_out, const double* pin1, const double* pin2,
                                              int nblk, const CgParams& prm, int b, bool writer, double* red, CgState* sh, int pending = -1,
                                              int pend_buf = 0, const double* pin3 = nullptr, const double* pin4 = nullptr, const double* pin5 = nullptr) {
    // the control block is fetched BEFORE the reductions so that its memory round trip overlaps theirs
    CgState s = CgState();
    if (threadIdx.x == 0 && kind != PRO_FIRST) s = st_in[b];
    double s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0;
    if (kind == PRO_CG1 && pin3) {      // the five sums of the single-reduction CG in ONE pass (that kernel is latency-bound: barriers count)
        double t[5] = {0, 0, 0, 0, 0};
        for (int i = threadIdx.x; i < nblk; i += kBlock) {
            const long long o = (long long)b * nblk + i;
            t[0] += pin1[o]; t[1] += pin2[o]; t[2] += pin3[o]; t[3] += pin4[o]; t[4] += pin5[o];
        }
        block_sum_n<5>(t, red);
        s1 = t[0]; s2 = t[1]; s3 = t[2]; s4 = t[3]; s5 = t[4];
    } else {
        if (kind >= PRO_FIRST) s1 = reduce_partials(pin1 + (long long)b * nblk, nblk, red);
        if (kind == PRO_FIRST || kind >= PRO_BETA_AD) s2 = reduce_partials(pin2 + (long long)b * nblk, nblk, red);   // (incl. PRO_CG1 without the extra sums)
    }
    if (threadIdx.x == 0) {
        s = cg_advance(kind, s, s1, s2, prm, s3, s4, s5);
        if (pending >= 0 && s.cont) {   // an UPDATE phase of a running entry: does x lag one step behind afterwards?
            s.pending = pending;
            s.pend_buf = pend_buf;
        }
        *sh = s;
        if (writer && kind >= PRO_FIRST) st_out[b] = s;
    }
    __syncthreads();
    return *sh;
}

// waves per SIMD the register allocator has to leave room for: the one-row tiles of the 3-word phases sit at 78-82 VGPRs -- 80 is the
// step between 5 and 6 resident waves
// (fp32 without cell flags: 78-82 registers, no spill at 80; the flag / fp64 variants would spill 2-26 dwords and keep their allocation)
template <typename T, int R, int MODE, bool FLAGS>
constexpr int march_min_waves() {
    return (sizeof(T) == 4 && !FLAGS && R == 1 && (MODE == MODE_APPLY || MODE == MODE_RESID || MODE == MODE_MATVEC || MODE == MODE_UPDATE_R)) ? 6 : 1;
}

// ROWT (r4), the ROW tile: a tile spans WHOLE rows whose vector count is not a power of two -- 72 ... 128 lanes per row for 288 ... 512-cell
// fp32 rows, TPR is then the upper bound (128) and MarchGrid::tpr_rt the count in use; the workgroup holds floor(256 / tpr_rt) thread
// rows, the remaining threads idle. No halo COLUMNS exist: the neighbours beyond a row's ends are read from the row's own LDS copy with
// the boundary rule (wrap / clamp / zero). Why: the power-of-two tiles cannot span a 288- ... 448-cell row, and their halo columns are what
// the mid-size dip is made of (1.49 fabric read requests per needed one at 320^3 against 1.13 at 512^3).
#ifndef PHIHIP_SAWTOOTH
#define PHIHIP_SAWTOOTH 1
#endif
// march_body: the kernel's body as a device function (r6), so that TWO entry points share it -- march_kernel (one lattice per launch: every CG phase) and
// march_apply_multi_kernel (MODE_APPLY on up to three lattices in ONE launch: the components of diffuse.explicit, phi/physics/diffuse.py:13-60 is one call).
template <typename T, int V, int R, int TPR, int MODE, bool FLAGS, bool DIM3, bool BIDIR = false, bool UNAL = false, bool ROWT = false>
__device__ __forceinline__ void march_body(const MarchGrid& g, const MarchArgs<T>& p) {
    constexpr int TRc = ROWT ? kBlock / (TPR / 2 + 1) : kBlock / TPR;   // thread rows (ROWT: at most -- rows of more than TPR / 2 lanes)
    constexpr int T1c = TRc * R;       // tile rows (axis a1)
    constexpr int T2c = TPR * V;       // tile columns (axis a2)
    constexpr int LS = T2c + 2 * V;    // LDS row stride; interior starts at column V so vector accesses stay aligned
    constexpr int LROWS = T1c + 2;
    // WIDE (r6): a row tile whose rows take MORE than half of the workgroup's lanes (129 ... 256 vectors: 384-cell fp64 rows = 192 lanes) -- one thread row, and each
    // lane fetches BOTH halo-row vectors of its column (the narrower forms give every halo vector a thread of its own: 2 tpr <= 256)
    constexpr bool WIDE = ROWT && TPR > 128;
    static_assert(ROWT || 2 * TPR + 2 * T1c <= kBlock, "halo items must fit one per thread");
    static_assert(!(ROWT && (UNAL || BIDIR)), "the row tile exists for aligned rows and one marching direction");
    const int tpr = ROWT ? g.tpr_rt : TPR;                      // lanes per row
    const int T1 = ROWT ? (kBlock / tpr) * R : T1c;
    const int T2 = ROWT ? g.n2 : T2c;
    using VT = Vec<T, V>;
    using VF = Vec<uint8_t, V>;
    constexpr bool IS_CG1 = MODE == MODE_CG1;
    constexpr bool IS_MV = MODE == MODE_MATVEC || MODE == MODE_MATVEC_AD || IS_CG1;   // the stencil source is a combination: A + beta B [+ gam C]
    constexpr bool IS_AP = MODE == MODE_APPLY || MODE == MODE_APPLY_DOT;
    constexpr bool IS_UP = MODE == MODE_UPDATE || MODE == MODE_UPDATE_AD || MODE == MODE_UPDATE_R || MODE == MODE_UPDATE_X2;
    constexpr bool HAS_X = IS_UP && MODE != MODE_UPDATE_R;   // UPDATE_R leaves x alone
    constexpr bool AD = MODE == MODE_MATVEC_AD || MODE == MODE_UPDATE_AD;

    __shared__ __attribute__((aligned(16))) T lds[2][LROWS * LS];
    __shared__ double red[5 * (kBlock / kWave)];
    __shared__ CgState sh_state;

    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    constexpr bool IS_RES = MODE == MODE_RESID || MODE == MODE_RESID_BAL;
    const T yshift = MODE == MODE_RESID_BAL ? (T)p.shift[b] : T(0);
    T alpha = T(0), beta = T(0), gam = T(0), beta_cg = T(0);   // CG1: S = A + beta B + gam C with beta = -alpha, gam = -alpha beta_cg
    // per-thread partial sums of the dot products: one row of a plane is summed in T, rows and planes are added up in double (a thread of
    // a 512^3 launch adds ~1000 rows; in fp32 that alone cost the eigenfunction test five extra iterations)
    double acc1 = 0.0, acc2 = 0.0;
    double acc3 = 0.0, acc4 = 0.0, acc5 = 0.0;   // single-reduction CG only: r'.s, (A r').p, p.s

    // XCD-aware block order: blocks b and b+8 share an XCD (and its L2); make consecutive tiles neighbours there.
    int bid = blockIdx.x;
    // Sawtooth (r5): the UPDATE kernels walk the grid BACK to front -- the planes of a chunk in descending order and, where a launch has more workgroups
    // than the chip holds at once, every XCD its range of tiles from the far end (the SAME range: a tile stays on the XCD whose L2 the MATVEC before left
    // its last planes in; mirroring the whole block order moved the tiles to other XCDs and took the gain away at 256^3). A CG iteration alternates MATVEC
    // and UPDATE launches that share two of their three vectors; what one launch touched last is what is still in the L2 / the 256 MiB Infinity Cache when
    // the next one starts, so the next one starts there (tools/micro/mall_sawtooth.hip: a plain triad over rotating vectors gains 3 % at 256^3 / 512^3
    // and 8 ... 15 % at 320^3 ... 384^3).
    constexpr bool DOWN = PHIHIP_SAWTOOTH && IS_UP && DIM3;
    bid = DOWN ? xcd_order_rev(bid, g.nblk) : xcd_order(bid, g.nblk);
    const int t2 = bid % g.tiles2;
    const int t1 = (bid / g.tiles2) % g.tiles1;
    const int c0 = bid / (g.tiles2 * g.tiles1);

    const int tx = tid % tpr, ty = tid / tpr;
    const bool lane_on = !ROWT || ty < kBlock / tpr;             // ROWT: the threads behind the last whole thread row idle
    const int j2_grid = t2 * T2 + tx * V;
    // UNAL: the thread whose vector would cross the end of the row owns the row's last V cells instead; `tsh` of them belong to its neighbour too
    const int tsh = (UNAL && j2_grid < g.n2 && j2_grid + V > g.n2) ? j2_grid + V - g.n2 : 0;
    const bool odd = UNAL && tsh != 0;
    const int j2 = j2_grid - tsh;
    const int j1b = t1 * T1 + ty * R;
    const int i_begin = c0 * g.chunk;
    const int i_end = min(i_begin + g.chunk, g.n0);
    // Odd chunks march DOWN: two neighbouring chunks of a tile then start at their common boundary, so the two planes they both
    // need (each other's first plane as halo) are requested at the same time and are served once from HBM (L2 / Infinity Cache
    // for the second requester) instead of once at the start of one chunk and again at the end of the other.
    const int step = (DIM3 && ((BIDIR && (c0 & 1)) != DOWN)) ? -1 : 1;   // compile-time unless the BIDIR instantiation (MATVEC, short chunks)
    const int i_first = step > 0 ? i_begin : i_end - 1;
    const int count = i_end - i_begin;
    const long long base = (long long)b * g.cells;
    const long long fbase = g.flags_per_batch ? base : 0;
    const int n1 = g.n1, n2 = g.n2;

    // Every global load / store of the plane loop is UNCONDITIONAL (lanes outside the grid read element 0 of the plane and store into a
    // dump slot; lanes without a halo role read element 0 as well): hipcc can then count the outstanding operations and waits with
    // s_waitcnt vmcnt(N > 0) for exactly the plane it needs, while the requests for later planes stay in flight across the barrier. With
    // loads under `if (ok)` / `if (halo role)` the compiler has to assume the branch was skipped and drains the queue (vmcnt(0)) -- the
    // round-1 loop paid two to three serialised memory round trips per plane that way (halo rows, halo columns, own cells).
    bool ok[R];
    int own_off[R];        // in-plane element offset of the thread's vector in its row rr
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
        ok[rr] = lane_on && (j2 < n2) && (j1b + rr < n1);
        own_off[rr] = ok[rr] ? (j1b + rr) * n2 + j2 : 0;
    }
    const long long plane = (long long)n1 * n2;

    // raw operands of a source plane (own cells): the plane selection incl. the slab halos and the ghost-plane rule is uniform; `zero` =
    // the plane is a zero ghost (NB_ZERO): the loads still run (plane 0), `combine` discards them
    auto load_raw = [&](int i, VT (&A)[R], VT (&B)[R], VT (&Cc)[R], bool& zero) {
        zero = false;
        const T* pa = p.a + base;
        const T* pb = IS_MV ? p.b + base : nullptr;
        const T* pc = IS_CG1 ? p.c + base : nullptr;
        if (DIM3) {
            if (i < 0 && g.nb[0][0] == NB_HALO) {
                pa = p.a_lo + (long long)b * plane;
                if (IS_MV) pb = p.b_lo + (long long)b * plane;
            } else if (i >= g.n0 && g.nb[0][1] == NB_HALO) {
                pa = p.a_hi + (long long)b * plane;
                if (IS_MV) pb = p.b_hi + (long long)b * plane;
            } else {
                const long long po = (long long)nb_index(i, g.n0, g.nb[0][0], g.nb[0][1], zero) * plane;
                pa += po;
                if (IS_MV) pb += po;
                if (IS_CG1) pc += po;
            }
        }
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            A[rr] = vec_load_g<T, V, UNAL>(pa + own_off[rr]);
            if (IS_MV) B[rr] = vec_load_g<T, V, UNAL>(pb + own_off[rr]);
            if (IS_CG1) Cc[rr] = vec_load_g<T, V, UNAL>(pc + own_off[rr]);
        }
    };
    VT Ra_p[R], Rb_p[R], Rc_p[R], Ra_c[R], Rb_c[R], Rc_c[R];
    bool zero_p = false, zero_c = false;
#pragma unroll
    for (int rr = 0; rr < R; ++rr) Ra_p[rr] = Rb_p[rr] = Rc_p[rr] = Rb_c[rr] = Rc_c[rr] = vec_zero<T, V>();
    // The first two planes are requested BEFORE the prologue's partial-sum reduction so that their HBM latency overlaps it.
    if (DIM3) load_raw(i_first - step, Ra_p, Rb_p, Rc_p, zero_p);   // the plane behind the marching direction
    load_raw(i_first, Ra_c, Rb_c, Rc_c, zero_c);

    if (p.prologue != PRO_NONE) {
        const CgState S = cg_prologue(p.prologue, p.st_in, p.st_out, p.pin1, p.pin2, p.nblk_in, p.prm, b, blockIdx.x == 0, red, &sh_state,
                                      IS_UP ? (MODE == MODE_UPDATE_R ? 1 : 0) : -1, p.pend_buf, IS_CG1 ? p.pin3 : nullptr, p.pin4, p.pin5);
        if (IS_MV && p.host_flags && blockIdx.x == 0 && tid == 0)   // the host stops enqueueing once every entry reports 0
            publish_flag(p.host_flags + b, ((unsigned long long)p.seq << 32) | (unsigned long long)(S.cont != 0));
        if (S.cont == 0) return;   // frozen batch entry: x, r, d stay as they are
        alpha = (T)S.alpha;
        beta = (T)S.beta;
        if (MODE == MODE_UPDATE_X2) beta = (T)(S.alpha_prev / S.beta);   // coefficient of (S - r) = beta_{k+1} d_k; beta > 0 while running
        if (IS_CG1) { beta_cg = (T)S.beta; beta = (T)(-S.alpha); gam = (T)(-S.alpha * S.beta); }
    }
    // `own`: the plane belongs to this workgroup's chunk (MATVEC_AD sums d_new * r over exactly those)
    auto combine = [&](const VT (&A)[R], const VT (&B)[R], const VT (&Cc)[R], VT (&S)[R], bool own, bool zero) {
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            S[rr] = A[rr];
            if (IS_MV) {
#pragma unroll
                for (int v = 0; v < V; ++v) S[rr].v[v] = fma(beta, B[rr].v[v], A[rr].v[v]);
                if (IS_CG1) {
#pragma unroll
                    for (int v = 0; v < V; ++v) S[rr].v[v] = fma(gam, Cc[rr].v[v], S[rr].v[v]);
                }
                if (AD && own && ok[rr]) {
                    T sa = T(0);
#pragma unroll
                    for (int v = 0; v < V; ++v) sa += (UNAL && v < tsh) ? T(0) : S[rr].v[v] * A[rr].v[v];      // (overlap cells: the neighbour counts them)
                    acc2 += (double)sa;
                }
            }
            if (zero) S[rr] = vec_zero<T, V>();
        }
    };

    // ---- halo roles (fixed per thread) ------------------------------------------------------------------------------
    // vector items: rows just below / above the tile; scalar items: columns just left / right of the tile
    const bool hv_role = WIDE ? tid < tpr : tid < 2 * tpr;
    const int hv_side = WIDE ? 0 : tid / tpr, hv_col = tid % tpr;
    const int hs_idx = tid - 2 * tpr;
    const bool hs_role = !ROWT && hs_idx >= 0 && hs_idx < 2 * T1;      // (the row tile has no halo columns)
    const int hs_side = hs_idx / T1, hs_row = hs_idx % T1;
    const int rows_here = min(T1, n1 - t1 * T1);   // valid rows of this tile
    const int cols_here = min(T2, n2 - t2 * T2);   // valid columns of this tile
    bool hv_ok = false, hv_zero = false, hs_ok = false, hs_zero = false;
    int hv_lrow = 0, hs_lcol = 0;
    int h_o = 0, hs_e = 0;    // ONE vector load per thread and source plane serves both kinds of item: offset of the 16-byte vector within a
                              // plane (0 = a harmless address for lanes that have nothing to fetch); scalar items pick element hs_e of it
    bool hv_odd = false;          // UNAL: the halo-row vector at the end of a ragged row is the row's last V cells, like the own cells
    if (hv_role) {
        int jh2 = t2 * T2 + hv_col * V;
        if (UNAL && jh2 < n2 && jh2 + V > n2) { jh2 = n2 - V; hv_odd = true; }
        const int jh1 = hv_side == 0 ? t1 * T1 - 1 : t1 * T1 + rows_here;
        hv_lrow = hv_side == 0 ? 0 : rows_here + 1;
        const int jt1 = nb_index(jh1, n1, g.nb[1][0], g.nb[1][1], hv_zero);
        hv_ok = jh2 < n2;
        if (hv_ok && !hv_zero) h_o = jt1 * n2 + jh2;
    }
    // WIDE: the same lane's second halo vector -- the row just ABOVE the tile (side 1) at its column
    int h_o2 = 0, hv_lrow2 = 0;
    bool hv_zero2 = false;
    if (WIDE && hv_role) {
        const int jh1 = t1 * T1 + rows_here;
        hv_lrow2 = rows_here + 1;
        const int jt1 = nb_index(jh1, n1, g.nb[1][0], g.nb[1][1], hv_zero2);
        if (hv_ok && !hv_zero2) h_o2 = jt1 * n2 + hv_col * V;
    }
    if (hs_role) {
        const int jh1 = t1 * T1 + hs_row;
        const int jh2 = hs_side == 0 ? t2 * T2 - 1 : t2 * T2 + cols_here;
        hs_lcol = hs_side == 0 ? V - 1 : V + cols_here;
        const int jt2 = nb_index(jh2, n2, g.nb[2][0], g.nb[2][1], hs_zero);
        hs_ok = jh1 < n1;
        if (hs_ok && !hs_zero) {
            int vs = (jt2 / V) * V;              // rows start on vector boundaries (n2 % V == 0 on the vector path) ...
            if (UNAL && vs + V > n2) vs = n2 - V;      // ... or the vector that ENDS with the row holds the element (UNAL; n2 >= V)
            h_o = jh1 * n2 + vs;
            hs_e = jt2 - vs;
        }
    }
    struct HaloRaw {
        VT va, vb, vc;
        VT va2, vb2, vc2;      // WIDE only
    };
    auto load_halo = [&](int i, HaloRaw& H) {
        const long long poff = base + (long long)i * plane;
        H.va = vec_load_g<T, V, UNAL>(p.a + poff + h_o);
        if (IS_MV) H.vb = vec_load_g<T, V, UNAL>(p.b + poff + h_o);
        if (IS_CG1) H.vc = vec_load_g<T, V, UNAL>(p.c + poff + h_o);
        if (WIDE) {
            H.va2 = vec_load_g<T, V, false>(p.a + poff + h_o2);
            if (IS_MV) H.vb2 = vec_load_g<T, V, false>(p.b + poff + h_o2);
            if (IS_CG1) H.vc2 = vec_load_g<T, V, false>(p.c + poff + h_o2);
        }
    };
    auto combine_halo2 = [&](const HaloRaw& H, VT& hv2) {      // WIDE: the second halo vector, combined like the first
        hv2 = H.va2;
        if (IS_MV) {
#pragma unroll
            for (int v = 0; v < V; ++v) hv2.v[v] = fma(beta, H.vb2.v[v], H.va2.v[v]);
        }
        if (IS_CG1) {
#pragma unroll
            for (int v = 0; v < V; ++v) hv2.v[v] = fma(gam, H.vc2.v[v], hv2.v[v]);
        }
        if (hv_zero2) hv2 = vec_zero<T, V>();
    };
    auto combine_halo = [&](const HaloRaw& H, VT& hv, T& hs) {
        hv = H.va;
        if (IS_MV) {
#pragma unroll
            for (int v = 0; v < V; ++v) hv.v[v] = fma(beta, H.vb.v[v], H.va.v[v]);
        }
        if (IS_CG1) {
#pragma unroll
            for (int v = 0; v < V; ++v) hv.v[v] = fma(gam, H.vc.v[v], hv.v[v]);
        }
        hs = hv.v[0];
#pragma unroll
        for (int v = 1; v < V; ++v) hs = hs_e == v ? hv.v[v] : hs;
        if (hv_zero) hv = vec_zero<T, V>();
        if (hs_zero) hs = T(0);
    };

    // ---- per-plane extra operands (own cells only) --------------------------------------------------------------------
    struct Extra {
        VT e1[R];   // RESID: y      UPDATE: x      CG1: p
        VT e2[R];   //               UPDATE: r      CG1: x
        VT e3[R], e4[R], e5[R];   // CG1: r, w, s of the own cells (the stencil source only keeps their combination)
        VF fl[R];   // stencil flags
    };
    auto load_extra = [&](int i, Extra& E) {
        const long long poff = base + (long long)i * plane;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const long long off = poff + own_off[rr];
            if (IS_RES) E.e1[rr] = vec_load_g<T, V, UNAL>(p.b + off);
            if (IS_UP) {
                if (HAS_X) E.e1[rr] = vec_load_g<T, V, UNAL>(p.o1 + off);
                E.e2[rr] = vec_load_g<T, V, UNAL>(p.o2 + off);
            }
            if (MODE == MODE_APPLY_DOT && p.c) {      // refresh of the single-reduction CG: mu, nu, sigma against the standing p and s
                E.e1[rr] = vec_load_g<T, V, UNAL>(p.o4 + off);
                E.e5[rr] = vec_load_g<T, V, UNAL>(p.c + off);
            }
            if (IS_CG1) {
                E.e1[rr] = vec_load_g<T, V, UNAL>(p.o4 + off);
                E.e2[rr] = vec_load_g<T, V, UNAL>(p.o5 + off);
                E.e3[rr] = vec_load_g<T, V, UNAL>(p.a + off);
                E.e4[rr] = vec_load_g<T, V, UNAL>(p.b + off);
                E.e5[rr] = vec_load_g<T, V, UNAL>(p.c + off);
            }
            if (FLAGS) E.fl[rr] = vec_load_g<uint8_t, V, UNAL>(p.flags + (fbase - base) + off);   // (UNAL: the V flag bytes of a ragged row start at any byte)
        }
    };
    // destination of a store: the cell's slot, or the dump slot for lanes outside the grid
    auto dst = [&](T* arr, long long off, int rr) -> T* { return ok[rr] ? arr + off : p.dump; };
    auto put = [&](T* arr, long long off, int rr, const VT& val) { vec_store_g<T, V, UNAL>(dst(arr, off, rr), val); };

    // ---- prologue ---------------------------------------------------------------------------------------------------
    VT Sp[R], Sc[R], Sn[R];
    VT Rn_a[R], Rn_b[R], Rn_c[R];
    bool zero_n = false;
    HaloRaw Hn;
    Extra En;
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
        Sp[rr] = vec_zero<T, V>();
        Sn[rr] = vec_zero<T, V>();
        Rn_a[rr] = Rn_b[rr] = Rn_c[rr] = vec_zero<T, V>();
    }
    if (DIM3) combine(Ra_p, Rb_p, Rc_p, Sp, false, zero_p);
    combine(Ra_c, Rb_c, Rc_c, Sc, true, zero_c);
    if (DIM3) load_raw(i_first + step, Rn_a, Rn_b, Rn_c, zero_n);
    load_halo(i_first, Hn);
    load_extra(i_first, En);

    int buf = 0;
    // ROWT: LDS columns of the left / right neighbour of this thread's vector (inside the row: the adjacent cells; at its ends: wrap / clamp / zero)
    int rt_lf_col[2] = {0, 0};
    bool rt_lf_zero[2] = {false, false};
    if (ROWT) {
        const int jl = nb_index(j2 - 1, n2, g.nb[2][0], g.nb[2][1], rt_lf_zero[0]);
        const int jr = nb_index(j2 + V, n2, g.nb[2][0], g.nb[2][1], rt_lf_zero[1]);
        rt_lf_col[0] = V + jl;
        rt_lf_col[1] = V + jr;
    }
    const int lrow0 = ty * R + 1;            // LDS row of this thread's first own row
    const int lcol = V + tx * V - tsh;       // LDS column of this thread's vector

    // Sp = the plane behind, Sn = the plane ahead in marching direction (the a0 stencil is symmetric in them)
    const unsigned bit_behind = step > 0 ? 1u : 2u, bit_ahead = step > 0 ? 2u : 1u;
    int i = i_first;
    // One plane. What the previous trip requested is consumed at the top (it has had a whole trip to arrive), then the requests for the
    // trips to come are issued -- source plane i + 2, halo and own-cell operands of plane i + 1 -- and stay in flight across the barrier.
    auto one_plane = [&](auto has_next_tag) {
        constexpr bool has_next = decltype(has_next_tag)::value;
        if (DIM3) combine(Rn_a, Rn_b, Rn_c, Sn, has_next, zero_n);
        VT hv_c;
        T hs_c;
        combine_halo(Hn, hv_c, hs_c);
        VT hv_c2;
        if (WIDE) combine_halo2(Hn, hv_c2);
        const Extra Ec = En;
        if (has_next) {
            if (DIM3) load_raw(i + 2 * step, Rn_a, Rn_b, Rn_c, zero_n);
            load_halo(i + step, Hn);
            load_extra(i + step, En);
        }
        T* L = lds[buf];
#pragma unroll
        for (int rr = 0; rr < R; ++rr)
            if (ok[rr]) lds_store<T, V>(L + (lrow0 + rr) * LS + lcol, Sc[rr], odd);
        if (hv_ok) lds_store<T, V>(L + hv_lrow * LS + V + hv_col * V - (hv_odd ? t2 * T2 + hv_col * V + V - n2 : 0), hv_c, hv_odd);
        if (WIDE && hv_ok) lds_store<T, V>(L + hv_lrow2 * LS + V + hv_col * V, hv_c2, false);
        if (hs_ok) L[(hs_row + 1) * LS + hs_lcol] = hs_c;
        __syncthreads();

        const long long poff = base + (long long)i * plane;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int lr = lrow0 + rr;
            VT up, dn;
            if (rr > 0) up = Sc[rr - 1]; else up = lds_load<T, V>(L + (lr - 1) * LS + lcol, odd);
            bool dn_reg = false;
            if (rr < R - 1) dn_reg = (j1b + rr + 1 < n1);
            if (dn_reg) dn = Sc[rr < R - 1 ? rr + 1 : rr]; else dn = lds_load<T, V>(L + (lr + 1) * LS + lcol, odd);
            T lf, rt;
            if (!ROWT) {
                lf = L[lr * LS + lcol - 1];
                rt = L[lr * LS + lcol + V];
            } else {      // the cells beyond the ends of the row: its own LDS copy under the boundary rule
                lf = rt_lf_zero[0] ? T(0) : L[lr * LS + rt_lf_col[0]];
                rt = rt_lf_zero[1] ? T(0) : L[lr * LS + rt_lf_col[1]];
            }
            VT q;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const T c = Sc[rr].v[v];
                const T lo2 = v > 0 ? Sc[rr].v[v > 0 ? v - 1 : 0] : lf;
                const T hi2 = v < V - 1 ? Sc[rr].v[v < V - 1 ? v + 1 : v] : rt;
                T r;
                if (FLAGS) {
                    const unsigned f = Ec.fl[rr].v[v];
                    r = T(0);
                    if (DIM3) {
                        if (f & bit_behind) r += (Sp[rr].v[v] - c) * p.w0;
                        if (f & bit_ahead) r += (Sn[rr].v[v] - c) * p.w0;
                    }
                    if (f & 4u) r += (up.v[v] - c) * p.w1;
                    if (f & 8u) r += (dn.v[v] - c) * p.w1;
                    if (f & 16u) r += (lo2 - c) * p.w2;
                    if (f & 32u) r += (hi2 - c) * p.w2;
                    r = fma(p.ident, c, r);
                    if (!(f & 64u)) r = c;   // inactive cell: identity row (fluid.py:202)
                } else {
                    // flux form like the reference (differences of neighbours first, then the difference of the two face
                    // fluxes): the rounding error scales with |grad p| instead of |p| -- with (lo + hi - 2c) CG stagnates
                    // an order of magnitude above the reference's residual floor in fp32.
                    const T t2 = ((hi2 - c) - (c - lo2)) * p.w2;
                    const T t1 = ((dn.v[v] - c) - (c - up.v[v])) * p.w1;
                    if (DIM3) r = ((Sn[rr].v[v] - c) - (c - Sp[rr].v[v])) * p.w0 + t1 + t2;
                    else r = t1 + t2;
                    r = fma(p.ident, c, r);
                }
                q.v[v] = r;
            }
            // UNAL: the cells this vector shares with its left neighbour count once -- `cnt` is 0 for them (the stores keep the full values)
            T cnt[V];
#pragma unroll
            for (int v = 0; v < V; ++v) cnt[v] = (UNAL && v < tsh) ? T(0) : T(1);
            auto once = [&](int v, T x) -> T { return UNAL ? cnt[v] * x : x; };
            const long long off = poff + own_off[rr];
            T s1 = T(0), s2 = T(0);   // this row's contributions to the two dot products
            if (IS_AP) {
                put(p.o1, off, rr, q);
                if (MODE == MODE_APPLY_DOT) {
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        s1 += once(v, Sc[rr].v[v] * Sc[rr].v[v]);
                        s2 += once(v, q.v[v] * Sc[rr].v[v]);
                    }
                    if (p.c && ok[rr]) {
                        T t3 = T(0), t4 = T(0), t5 = T(0);
#pragma unroll
                        for (int v = 0; v < V; ++v) {
                            t3 += once(v, Sc[rr].v[v] * Ec.e5[rr].v[v]);          // r . s
                            t4 += once(v, q.v[v] * Ec.e1[rr].v[v]);               // (A r) . p
                            t5 += once(v, Ec.e1[rr].v[v] * Ec.e5[rr].v[v]);       // p . s
                        }
                        acc3 += (double)t3; acc4 += (double)t4; acc5 += (double)t5;
                    }
                }
            } else if (IS_CG1) {
                VT pn, sn, xn;
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    pn.v[v] = fma(beta_cg, Ec.e1[rr].v[v], Ec.e3[rr].v[v]);       // p = r + beta p
                    sn.v[v] = fma(beta_cg, Ec.e5[rr].v[v], Ec.e4[rr].v[v]);       // s = w + beta s  (= A p)
                    xn.v[v] = fma(-beta, pn.v[v], Ec.e2[rr].v[v]);                // x += alpha p   (beta holds -alpha here)
                    s1 += once(v, Sc[rr].v[v] * Sc[rr].v[v]);                  // gamma' = |r_new|^2
                    s2 += once(v, q.v[v] * Sc[rr].v[v]);                       // delta' = (A r_new) . r_new
                }
                if (ok[rr]) {
                    T t3 = T(0), t4 = T(0), t5 = T(0);
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        t3 += once(v, Sc[rr].v[v] * sn.v[v]);                  // mu' = r_new . s
                        t4 += once(v, q.v[v] * pn.v[v]);                       // nu' = (A r_new) . p
                        t5 += once(v, pn.v[v] * sn.v[v]);                      // sigma = p . s
                    }
                    acc3 += (double)t3; acc4 += (double)t4; acc5 += (double)t5;
                }
                put(p.o4, off, rr, pn);
                put(p.o3, off, rr, sn);
                put(p.o5, off, rr, xn);
                put(p.o1, off, rr, Sc[rr]);
                put(p.o2, off, rr, q);
            } else if (IS_RES) {
                VT r, yb;
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    T y = Ec.e1[rr].v[v];
                    if (MODE == MODE_RESID_BAL) {   // div -= active * mean(div) / mean(active)  (fluid.py:205-209)
                        if (FLAGS) y -= (Ec.fl[rr].v[v] & 64u) ? yshift : T(0);
                        else y -= yshift;
                        yb.v[v] = y;
                    }
                    r.v[v] = y - q.v[v];
                    s1 += once(v, r.v[v] * r.v[v]);
                    s2 += once(v, y * y);
                }
                if (MODE == MODE_RESID_BAL) put(p.yout, off, rr, yb);
                put(p.o1, off, rr, r);
            } else if (IS_MV) {
#pragma unroll
                for (int v = 0; v < V; ++v) s1 += once(v, Sc[rr].v[v] * q.v[v]);
                put(p.o1, off, rr, Sc[rr]);
            } else {
                VT xn, rn;
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    if (MODE == MODE_UPDATE_X2) xn.v[v] = Ec.e1[rr].v[v] + beta * (Sc[rr].v[v] - Ec.e2[rr].v[v]) + alpha * Sc[rr].v[v];
                    else if (HAS_X) xn.v[v] = Ec.e1[rr].v[v] + alpha * Sc[rr].v[v];
                    rn.v[v] = Ec.e2[rr].v[v] - alpha * q.v[v];
                    s1 += once(v, rn.v[v] * rn.v[v]);
                    if (AD) s2 += once(v, rn.v[v] * q.v[v]);
                }
                if (HAS_X) put(p.o1, off, rr, xn);
                put(p.o2, off, rr, rn);
            }
            if (ok[rr]) {   // lanes outside the grid computed on garbage (possibly NaN): keep them out of the sums
                acc1 += (double)s1;
                acc2 += (double)s2;
            }
        }
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            Sp[rr] = Sc[rr];
            Sc[rr] = Sn[rr];
        }
        buf ^= 1;
        i += step;
    };
    for (int k = 0; k + 1 < count; ++k) one_plane(std::true_type{});
    if (count > 0) one_plane(std::false_type{});

    if (MODE != MODE_APPLY) {
        const double s1 = block_sum(acc1, red);
        if (tid == 0) p.part1[(long long)b * g.nblk + blockIdx.x] = s1;
        if (IS_RES || AD || IS_CG1 || MODE == MODE_APPLY_DOT) {
            const double s2 = block_sum(acc2, red);
            if (tid == 0) p.part2[(long long)b * g.nblk + blockIdx.x] = s2;
        }
        if ((IS_CG1 || MODE == MODE_APPLY_DOT) && p.part3) {
            double t[3] = {acc3, acc4, acc5};
            block_sum_n<3>(t, red);
            if (tid == 0) {
                p.part3[(long long)b * g.nblk + blockIdx.x] = t[0];
                p.part4[(long long)b * g.nblk + blockIdx.x] = t[1];
                p.part5[(long long)b * g.nblk + blockIdx.x] = t[2];
            }
        }
    }
}

template <typename T, int V, int R, int TPR, int MODE, bool FLAGS, bool DIM3, bool BIDIR = false, bool UNAL = false, bool ROWT = false>
__global__ __launch_bounds__(kBlock, (UNAL ? 1 : march_min_waves<T, R, MODE, FLAGS>())) void march_kernel(MarchGrid g, MarchArgs<T> p) {
    march_body<T, V, R, TPR, MODE, FLAGS, DIM3, BIDIR, UNAL, ROWT>(g, p);
}

// MODE_APPLY (out = (ident + sum_a w_a d^2_a) in) on up to three lattices of one tile configuration in ONE launch: blockIdx.z selects the lattice -- its shape,
// neighbour rules, decomposition (MarchGrid) and its arrays; weights, ident and the dump slot are common. The launch has max(nblk) workgroups per lattice and
// batch entry, the surplus of a smaller lattice (a closed box: N - 1 faces along the component's axis) leaves at once.
template <typename T>
struct MarchMulti {
    MarchGrid g[3];
    const T* in[3];
    T* out[3];
};
template <typename T, int V, int R, int TPR, bool DIM3, bool UNAL = false, bool ROWT = false>
__global__ __launch_bounds__(kBlock, (UNAL ? 1 : march_min_waves<T, R, MODE_APPLY, false>())) void march_apply_multi_kernel(MarchMulti<T> m, MarchArgs<T> p) {
    const int l = blockIdx.z;              // uniform: the lattice's descriptor comes from the kernel arguments with scalar loads
    const MarchGrid g = m.g[l];
    if ((int)blockIdx.x >= g.nblk) return;
    MarchArgs<T> q = p;
    q.a = m.in[l];
    q.o1 = m.out[l];
    march_body<T, V, R, TPR, MODE_APPLY, false, DIM3, false, UNAL, ROWT>(g, q);
}

}  // namespace phihip
