// cg_small.hip -- the whole CG solve of a SMALL grid in one kernel: one 512- or 1024-thread workgroup per batch entry, the
// search direction d and q = A d live in LDS (sized by the variant: 2 * threads * cells-per-thread words, <= 64 KB), x / r / d
// of the thread's own cells in registers, alpha / beta / the convergence logic in the workgroup -- no kernel boundary and no
// host polling inside the loop.
//
// Why: the marching kernels need two dependent launches per iteration, i.e. >= 9.5 us per iteration however small the grid
// (tools/host_bound_check.py); PhiFlow's typical learning workloads are large batches of small simulations (64^2 ... 128^2),
// and BASELINE configs[0] is a 128^2 plume. Here a 128^2 entry costs the arithmetic of one CU per iteration and every batch
// entry gets its own CU. Same algorithm and control flow as cg.hip / stencil_march.hpp (PhiML cg).
#include "common.hpp"
#include "march_dispatch.hpp"

namespace phihip {

// The per-cell loops are fully unrolled (register arrays need constant indices). Making the packed neighbour code opaque to the
// optimiser keeps its decode inside the loop; hoisted, the decoded offsets cost 6 registers per cell and the kernel spills.
#ifdef __HIP_DEVICE_COMPILE__
#define PHIHIP_OPAQUE(v) asm volatile("" : "+v"(v))
#else
#define PHIHIP_OPAQUE(v) do { } while (0)
#endif

constexpr int kSmallMaxThreads = 1024;

template <typename T>
struct SmallArgs {
    const T* rhs;
    T* x;
    const uint8_t* flags;
    CgState* st_out;
    CgParams prm;
    int refresh_every;
    int dim3;              // 1: rank-3 grid (a 3-D grid may have ONE plane, and its a0 boundary rule still applies)
    int adaptive;          // 1: PhiML 'CG-adaptive' (alpha = d.r / d.q, beta = -(r'.q) / d.q)
    T w0, w1, w2;
};

template <int NT>
__device__ __forceinline__ double small_block_sum(double v, double* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0;
#pragma unroll
    for (int w = 0; w < NT / kWave; ++w) s += red[w];   // every thread adds the same values in the same order
    return s;
}

// 2-bit neighbour rule per axis side, packed per cell: 0 interior, 1 wrap, 2 clamp (self), 3 zero ghost
__device__ __forceinline__ unsigned side_code(int i, int n, int rule, bool lower) {
    if (lower ? i > 0 : i < n - 1) return 0u;
    return rule == NB_WRAP ? 1u : (rule == NB_CLAMP ? 2u : 3u);
}

template <typename T>
__device__ __forceinline__ T small_nb(const T* L, int c, T self, unsigned code, int stride, int wrap_shift, bool lower) {
    if (code == 0u) return L[lower ? c - stride : c + stride];
    if (code == 1u) return L[lower ? c - stride + wrap_shift : c + stride - wrap_shift];
    return code == 2u ? self : T(0);
}

template <typename T, int NT, int CPT, bool FLAGS>
__global__ __launch_bounds__(NT) void cg_small_kernel(MarchGrid g, SmallArgs<T> p) {
    constexpr int kSmallThreads = NT;
    __shared__ __attribute__((aligned(16))) T L[NT * CPT];   // the vector whose Laplacian is taken (x, then d)
    __shared__ __attribute__((aligned(16))) T Q[NT * CPT];   // q = A d of the thread's own cells (spares CPT registers)
    __shared__ double red[NT / kWave];

    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int cells = (int)g.cells;
    const int n1 = g.n1, n2 = g.n2;
    const int s0 = n1 * n2, s1 = n2;
    const long long base = (long long)b * cells;
    const bool dim3_ = p.dim3 != 0;

    T x[CPT], r[CPT], d[CPT];
    unsigned code[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int c = k * kSmallThreads + tid;
        x[k] = r[k] = d[k] = T(0);
        code[k] = 0u;
        if (c < cells) {
            const int i2 = c % n2, t = c / n2, i1 = t % n1, i0 = t / n1;
            unsigned cd = side_code(i2, n2, g.nb[2][0], true) | (side_code(i2, n2, g.nb[2][1], false) << 2) |
                          (side_code(i1, n1, g.nb[1][0], true) << 4) | (side_code(i1, n1, g.nb[1][1], false) << 6);
            if (dim3_) cd |= (side_code(i0, g.n0, g.nb[0][0], true) << 8) | (side_code(i0, g.n0, g.nb[0][1], false) << 10);
            if (FLAGS) cd |= (unsigned)p.flags[(g.flags_per_batch ? base : 0) + c] << 16;
            code[k] = cd;
            x[k] = p.x[base + c];
        }
    }

    // A applied to the vector in L at the thread's k-th cell (same operation order as march_kernel)
    auto apply = [&](int k, T self) -> T {
        const int c = k * kSmallThreads + tid;
        unsigned cd = code[k];
        PHIHIP_OPAQUE(cd);   // keeps the neighbour-offset decode inside the loop (hoisted it costs 6 registers per cell)
        const T lo2 = small_nb<T>(L, c, self, cd & 3u, 1, n2, true), hi2 = small_nb<T>(L, c, self, (cd >> 2) & 3u, 1, n2, false);
        const T up = small_nb<T>(L, c, self, (cd >> 4) & 3u, s1, n1 * s1, true), dn = small_nb<T>(L, c, self, (cd >> 6) & 3u, s1, n1 * s1, false);
        T lo0 = T(0), hi0 = T(0);
        if (dim3_) {
            lo0 = small_nb<T>(L, c, self, (cd >> 8) & 3u, s0, g.n0 * s0, true);
            hi0 = small_nb<T>(L, c, self, (cd >> 10) & 3u, s0, g.n0 * s0, false);
        }
        if (FLAGS) {
            const unsigned f = cd >> 16;
            T q = T(0);
            if (dim3_) {
                if (f & 1u) q += (lo0 - self) * p.w0;
                if (f & 2u) q += (hi0 - self) * p.w0;
            }
            if (f & 4u) q += (up - self) * p.w1;
            if (f & 8u) q += (dn - self) * p.w1;
            if (f & 16u) q += (lo2 - self) * p.w2;
            if (f & 32u) q += (hi2 - self) * p.w2;
            if (!(f & 64u)) q = self;   // inactive cell: identity row (fluid.py:202)
            return q;
        }
        const T t2 = ((hi2 - self) - (self - lo2)) * p.w2;
        const T t1 = ((dn - self) - (self - up)) * p.w1;
        return dim3_ ? ((hi0 - self) - (self - lo0)) * p.w0 + t1 + t2 : t1 + t2;
    };
    auto publish = [&](const T (&v)[CPT]) {   // own cells of a register vector -> L (callers sync around it)
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int c = k * kSmallThreads + tid;
            if (c < cells) L[c] = v[k];
        }
    };
    // r = y - A x from the current x; returns (sum r^2, sum y^2)
    auto residual = [&](double& rr, double& yy) {
        __syncthreads();
        publish(x);
        __syncthreads();
        T a1 = T(0), a2 = T(0);
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int c = k * kSmallThreads + tid;
            if (c < cells) {
                const T y = p.rhs[base + c];
                r[k] = y - apply(k, x[k]);
                a1 += r[k] * r[k];
                a2 += y * y;
            }
        }
        rr = small_block_sum<NT>((double)a1, red);
        yy = small_block_sum<NT>((double)a2, red);
    };

    double rr, yy;
    residual(rr, yy);
    CgState S = cg_advance(PRO_FIRST, CgState(), rr, yy, p.prm);
    for (int it = 1; it <= p.prm.max_iter && S.cont; ++it) {
        // ---- MATVEC: d = r + beta d ; dq = d . A d ----
        const T beta = (T)S.beta;
#pragma unroll
        for (int k = 0; k < CPT; ++k) d[k] = fma(beta, d[k], r[k]);
        __syncthreads();
        publish(d);
        __syncthreads();
        T acc = T(0), acc_dr = T(0);
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int c = k * kSmallThreads + tid;
            if (c < cells) {
                const T q = apply(k, d[k]);
                Q[c] = q;
                acc += d[k] * q;
                acc_dr += d[k] * r[k];
            }
        }
        const double dq = small_block_sum<NT>((double)acc, red);
        if (p.adaptive) S = cg_advance(PRO_ALPHA_AD, S, dq, small_block_sum<NT>((double)acc_dr, red), p.prm);
        else S = cg_advance(PRO_ALPHA, S, dq, 0.0, p.prm);
        const T alpha = (T)S.alpha;
        // ---- UPDATE: x += alpha d ; r -= alpha A d (or the true residual every refresh_every-th iteration) ----
#pragma unroll
        for (int k = 0; k < CPT; ++k) x[k] = x[k] + alpha * d[k];
        T a2 = T(0);   // sum r_new . q ('CG-adaptive')
        if (p.refresh_every > 0 && it % p.refresh_every == 0) {
            double dummy;
            residual(rr, dummy);
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                const int c = k * kSmallThreads + tid;
                if (c < cells) a2 += r[k] * Q[c];
            }
        } else {
            T a1 = T(0);
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                const int c = k * kSmallThreads + tid;
                if (c < cells) {
                    const T q = Q[c];
                    r[k] = r[k] - alpha * q;
                    a1 += r[k] * r[k];
                    a2 += r[k] * q;
                }
            }
            rr = small_block_sum<NT>((double)a1, red);
        }
        if (p.adaptive) S = cg_advance(PRO_BETA_AD, S, rr, small_block_sum<NT>((double)a2, red), p.prm);
        else S = cg_advance(PRO_BETA, S, rr, 0.0, p.prm);
    }
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int c = k * kSmallThreads + tid;
        if (c < cells) p.x[base + c] = x[k];
    }
    if (tid == 0) p.st_out[b] = S;
}

template <typename T, int NT, int CPT>
static void launch_small(const MarchGrid& g, const SmallArgs<T>& a, int batch, bool flags, hipStream_t s) {
    if (flags)
        hipLaunchKernelGGL((cg_small_kernel<T, NT, CPT, true>), dim3(batch), dim3(NT), 0, s, g, a);
    else
        hipLaunchKernelGGL((cg_small_kernel<T, NT, CPT, false>), dim3(batch), dim3(NT), 0, s, g, a);
}

// cells per batch entry up to which the single-workgroup solver is used (register + LDS budget of one CU)
// Measured on MI355X (tools/sweep_cg2d.py): 64^2 x 256 entries 4.2 us / iteration against 14.7 us with the marching kernels,
// 90^2 x 128 5.3 against 31.6 us; at 128^2 (16384 cells on one CU: 11.4 us) the marching kernels win (9.2 us), so the
// limit is 8192 cells.
long long small_cg_limit(const phihip_ctx* ctx, const GridView& v) {
    const long long cap = v.dtype == PHIHIP_F64 ? 8192 : 16384;   // LDS: 2 * cells * sizeof(T) <= 128 KB
    if (ctx->small_cg_cells > 0) return ctx->small_cg_cells < cap ? ctx->small_cg_cells : cap;
    // 8193 ... 16384 cells (fp32, 1024 threads x 16 cells): one CU per entry needs ~11-12 us per iteration, the marching kernels 9.2 us
    // for ONE 128^2 entry but they grow with the batch: 25^3 x 64 entries 12.9 against 35.8 us, 100^2 x 32 8.8 / 12.1, 128^2 x 32 11.0 / 11.6
    return v.batch >= 8 ? cap : 8192;
}

template <typename T>
static int cg_small_t(phihip_ctx* ctx, const GridView& v, const uint8_t* flags, int mask_batch, const void* rhs, void* x,
                      const phihip_solve* solve, CgState* st_out, hipStream_t s) {
    MarchConfig c;
    MarchGrid g;
    PHIHIP_TRY(plan_march(ctx, v, mask_batch, flags != nullptr, FAM_APPLY, &c, &g));   // only the grid description is used
    SmallArgs<T> a;
    memset(&a, 0, sizeof(a));
    a.rhs = (const T*)rhs;
    a.x = (T*)x;
    a.flags = flags;
    a.st_out = st_out;
    a.prm.rtol = solve->rel_tol; a.prm.atol = solve->abs_tol; a.prm.max_iter = solve->max_iterations; a.prm.pad = 0;
    a.refresh_every = solve->refresh_every;
    a.dim3 = v.rank == 3 ? 1 : 0;
    a.adaptive = solve->method == PHIHIP_METHOD_CG_ADAPTIVE ? 1 : 0;
    a.w0 = (T)(1.0 / (v.dx[0] * v.dx[0])); a.w1 = (T)(1.0 / (v.dx[1] * v.dx[1])); a.w2 = (T)(1.0 / (v.dx[2] * v.dx[2]));
    LaunchScope ls(ctx, PHIHIP_K_CG_UPDATE, s);
    const bool fl = flags != nullptr;
    // (threads, cells per thread) variants; LDS = 2 * threads * cells per thread * sizeof(T), so small grids leave room for
    // several workgroups (batch entries) per CU. Measured (tools/sweep_cg2d.py, us per iteration, 512- vs 1024-thread form):
    // 32^2 x 512: 3.96 / 5.6, 45^2 x 256: 2.91 / 3.19, 64^2 x 256: 4.16 / 4.03, 90^2 x 128: 6.6 / 5.3.
    if (v.cells <= 1024) launch_small<T, 512, 2>(g, a, v.batch, fl, s);
    else if (v.cells <= 2048) launch_small<T, 512, 4>(g, a, v.batch, fl, s);
    else if (v.cells <= 4096) launch_small<T, 512, 8>(g, a, v.batch, fl, s);
    else if (sizeof(T) == 8) launch_small<T, 512, 16>(g, a, v.batch, fl, s);   // fp64: 1024 x 8 would exceed 128 VGPRs by far
    else if (v.cells <= 8192) launch_small<T, 1024, 8>(g, a, v.batch, fl, s);
    else if constexpr (sizeof(T) == 4) launch_small<T, 1024, 16>(g, a, v.batch, fl, s);   // 128 KB of LDS: fp32 only (small_cg_limit)
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

int run_cg_small(phihip_ctx* ctx, const GridView& v, const uint8_t* flags, int mask_batch, const void* rhs, void* x,
                 const phihip_solve* solve, void* st_out, hipStream_t s) {
    return v.dtype == PHIHIP_F64 ? cg_small_t<double>(ctx, v, flags, mask_batch, rhs, x, solve, (CgState*)st_out, s)
                                 : cg_small_t<float>(ctx, v, flags, mask_batch, rhs, x, solve, (CgState*)st_out, s);
}

}  // namespace phihip
