// cg.hip -- matrix-free conjugate gradients on the masked 5/7-point Laplacian.
// Replaces math.solve_linear(masked_laplace, div, Solve('CG', ...)) (/root/reference phi/physics/fluid.py:156) whose
// PhiML implementation assembles a sparse matrix per call and runs {SpMV, 2 dots, 3 AXPY} as separate array passes.
// Here one iteration = TWO launches: MATVEC (d = r + beta d ; dq = d.Ad) and UPDATE (x += a d ; r -= a Ad ; rsq); each
// reduces its predecessor's partial sums in its prologue (stencil_march.hpp), so alpha / beta / the per-batch continue
// flags never leave the device and no scalar kernel sits between the phases. The host only enqueues launches.
#include <math.h>
#include <stdlib.h>

#include "common.hpp"
#include "march_dispatch.hpp"

namespace phihip {

// ---------------------------------------------------------------------------------------------------------------------
// planning
// ---------------------------------------------------------------------------------------------------------------------
// Launch plan of one kernel family. Measured on MI355X (profiles/r01_sweep_family*.jsonl): a grid that needs 1 < rounds < 2
// of resident workgroups costs almost two full rounds, chunks longer than 64 planes starve the chip, short chunks re-read two
// halo planes of the stencil source per chunk, and the families prefer different tiles -- MATVEC (2 loads / cell, 76-138
// VGPRs) wants >= 3 medium workgroups per CU, UPDATE (3 loads, 2 stores, up to 224 VGPRs) the largest tile. So every (tile,
// chunk) candidate gets   score = slot efficiency / relative traffic
//     slot efficiency = rounds / ceil(rounds)            (rounds = workgroups * batch / (occupancy(kernel) * CUs) > 1)
//                     = min(1, workgroups / min(slots, 4 [MATVEC, UPDATE_R] or 2 [UPDATE, residual; 1.5 in fp64] per CU))   (one round)
//     relative traffic = 1 + (2 / chunk) * (source words / all words)
// and the best score (x a small per-family tile preference) wins; for small grids the serial march of a chunk (latency) replaces the
// traffic term (see best_chunk).
// coefficients of the operator the marching kernels apply: ident * I + scale * L (GridView::op_*; default 0 * I + 1 * L = the pressure operator)
template <typename T>
static inline void set_operator(MarchArgs<T>& a, const GridView& v) {
    const double sc = v.op_custom ? v.op_scale : 1.0;
    a.w0 = (T)(sc / (v.dx[0] * v.dx[0])); a.w1 = (T)(sc / (v.dx[1] * v.dx[1])); a.w2 = (T)(sc / (v.dx[2] * v.dx[2]));
    a.ident = (T)(v.op_custom ? v.op_ident : 0.0);
}

static PlanKey plan_key(const GridView& v, int mask_batch, bool flags, int family) {
    return PlanKey{v.dtype, v.rank, v.n[0], v.n[1], v.n[2], v.batch, flags ? 1 : 0, mask_batch > 1 ? 1 : 0, family, v.unaligned ? 0 : 1};
}

int plan_march(const phihip_ctx* ctx, const GridView& v, int mask_batch, bool flags, int family, MarchConfig* c, MarchGrid* g, int force_id) {
    const int esize = v.dtype == PHIHIP_F64 ? 8 : 4;
    const int vmax = 16 / esize;
    memset(g, 0, sizeof(*g));
    g->n0 = v.n[0]; g->n1 = v.n[1]; g->n2 = v.n[2];
    g->cells = v.cells;
    for (int ax = 0; ax < 3; ++ax)
        for (int s = 0; s < 2; ++s) {
            const int code = v.bc[ax][s];
            g->nb[ax][s] = v.op_custom ? v.op_rule[ax][s] : (code == PHIHIP_BC_PERIODIC ? NB_WRAP : (code == PHIHIP_BC_CLOSED ? NB_CLAMP : NB_ZERO));
        }
    g->flags_per_batch = mask_batch > 1 ? 1 : 0;
    c->batch = v.batch;
    c->vec = march_vector_width(v.n[2], esize, v.unaligned);
    const Tuning& t = ctx->tuning[family];
    const int mode = family_mode(family);
    const double src_share = family == FAM_MATVEC ? 2.0 / 3.0 : (family == FAM_UPDATE ? 0.2 : (family == FAM_CG1 ? 0.3 : 1.0 / 3.0));   // UPDATE_R: d is 1 of 3 words
    // the r-only update (2 loads + 1 store per cell) prefers the tiles of MATVEC (family sweep: 256^3 (1,64) x 16, 512^3 (2,32) x 64)
    // APPLY / RESID (never autotuned) take MATVEC's tiles while a vector fits the Infinity Cache regime (<= 72 MB: 256^3 fp32), the large
    // tiles beyond: same-box A/B 256^3 APPLY 27.5 -> 22 us, RESID 46 -> 40 us, but 512^3 APPLY 195 -> 224 us with MATVEC's (1,64) tile
    const bool mv_like = family == FAM_MATVEC || family == FAM_UPDATE_R || (family == FAM_APPLY && (double)v.cells * v.batch * esize <= 72e6);

    auto tile_of = [&](int id, int* t1, int* t2) { march_tile_extent(c->vec, id, v.n[2], t1, t2); };
    auto tiles_of = [&](int id) -> long long {
        int t1, t2;
        tile_of(id, &t1, &t2);
        return (long long)ceil_div(v.n[1], t1) * ceil_div(v.n[2], t2);
    };
    // best chunk of a tile configuration and its score
    auto best_chunk = [&](int cand, double* score_out) -> int {
        const double slots = (double)march_occupancy_any(v, cand, c->vec, mode, flags) * ctx->num_cu;
        const double tiles = (double)tiles_of(cand) * v.batch;
        const double words = family == FAM_UPDATE ? 5.0 : (family == FAM_CG1 ? 10.0 : 3.0);
        // every workgroup of the NEXT kernel re-reads all partial sums of its batch entry: nblk^2 doubles per entry, served by L2 (~3x
        // HBM speed). Negligible in 3-D (<= 0.5 % at 512^3), decisive for large 2-D grids, which have one workgroup per tile:
        // 2048^2 with 4096 small tiles 43.9 us per iteration, with 1024 (4,64) tiles 26.1 us (tools/sweep_cg2d.py)
        auto partials_share = [&](double nblk_entry) { return nblk_entry * nblk_entry * 8.0 / 3.0 / ((double)v.cells * words * esize); };
        if (v.rank != 3) {
            const double rounds = tiles / slots;
            const double per_cu = mv_like ? 4.0 : (esize == 8 ? 1.5 : 2.0);
            const double wanted = slots < per_cu * ctx->num_cu ? slots : per_cu * ctx->num_cu;
            const double eff = rounds > 1.0 ? rounds / ceil(rounds - 1e-9) : (tiles < wanted ? tiles / wanted : 1.0);
            *score_out = eff / (1.0 + partials_share((double)tiles_of(cand)));
            return 1;
        }
        // time model of one launch (us): the workgroups of a round march `chunk` planes one after the other (~0.75 us per plane after
        // ~3 us of prologue + first loads: 64^3 (1,16) 7.9 / 8.5 / 10.1 / 13.2 us at chunk 1 / 2 / 4 / 8 incl. the launch gap), and the
        // launch cannot beat its HBM traffic at ~5.5 TB/s. Large grids are traffic-bound (long chunks: fewer re-read halo planes), small
        // ones latency-bound (short chunks, many workgroups: 64^3 iteration 20.7 -> 11.1 us, 128^3 24.2 -> 18.1 us).
        static const int kChunks[10] = {64, 48, 32, 24, 16, 12, 8, 4, 2, 1};
        int best = 0;
        double best_score = -1.0;
        const double per_cu = mv_like ? 4.0 : (esize == 8 ? 1.5 : 2.0);
        const double wanted = slots < per_cu * ctx->num_cu ? slots : per_cu * ctx->num_cu;
        const double us_traffic = (double)v.cells * v.batch * words * esize / 5.5e6;
        // candidates: the fixed list, then the chunk lengths that put m = 1 .. occupancy workgroups on EVERY CU
        const int occ_i = (int)(slots / ctx->num_cu + 0.5);
        for (int k = 0; k < 10 + occ_i; ++k) {
            int ch;
            if (k < 10) ch = kChunks[k] < v.n[0] ? kChunks[k] : v.n[0];
            else {
                long long chunks0 = (long long)((double)(k - 9) * ctx->num_cu / tiles);
                if (chunks0 < 1) continue;
                if (chunks0 > v.n[0]) chunks0 = v.n[0];
                ch = ceil_div(v.n[0], (int)chunks0);
                if (ch > 128) continue;
            }
            const double blocks = tiles * ceil_div(v.n[0], ch);
            const double rounds = blocks / slots;
            // (a round filled to >= 90 % counts as full: 320^3 runs 8 % faster with 1000 workgroups of 32 planes than with 1400 of 24)
            double eff = rounds > 1.0 ? rounds / ceil(rounds - 1e-9) : (blocks < 0.9 * wanted ? blocks / (0.9 * wanted) : 1.0);
            // within a round the CUs should carry the same number of workgroups: 384^3 (1,16) 1152 workgroups (4.5 per CU) 146 us,
            // 1008 (3.9) and 1440 (5.6) 137 us (profiles/r02_midsize_chunks.jsonl)
            if (rounds <= 1.0 && blocks >= ctx->num_cu) eff *= blocks / (ceil(blocks / ctx->num_cu - 1e-9) * ctx->num_cu);
            const double planes = ((family == FAM_MATVEC || family == FAM_UPDATE_R) && ch <= 16 ? 1.0 : 2.0) / ch;   // bidirectional marching shares one of the two
            const double us_bw = us_traffic * (1.0 + planes * src_share + partials_share(blocks / v.batch)) / eff;
            const double us_lat = ceil(rounds - 1e-9) * (3.0 + 0.75 * (ch + 1));
            const double score = us_traffic / (us_bw > us_lat ? us_bw : us_lat);   // = eff / relative traffic when traffic-bound
            if (score > best_score * 1.0001) { best_score = score; best = ch; }
        }
        *score_out = best_score;
        return best;
    };

    int id = -1, chunk = 1;
    double score = 0;
    const auto tuned = (t.rows > 0 || t.chunk > 0 || v.halo[0] || v.halo[1]) ? ctx->tuned.end() : ctx->tuned.find(plan_key(v, mask_batch, flags, family));
    if (force_id >= 0 && !march_one_tile(c->vec) && march_tile_available(c->vec, esize, force_id, v.n[2])) {      // the tile of a sibling lattice (one launch for both)
        id = force_id;
        chunk = best_chunk(id, &score);
    } else if (tuned != ctx->tuned.end()) {   // measured on this device (autotune_cg)
        id = tuned->second.id;
        chunk = tuned->second.chunk;
    } else if (march_one_tile(c->vec)) {
        id = 5;   // (1, 64): the only scalar / UNAL instantiation
        chunk = best_chunk(id, &score);
    } else if (t.rows > 0 && t.tpr > 0) {
        for (int k = 0; k < kNumTileConfigs; ++k)
            if (kTileShapes[k].rows == t.rows && kTileShapes[k].tpr == t.tpr) id = k;
        if (id < 0) {
            set_error("tuning: no tile config with rows=%d threads_per_row=%d", t.rows, t.tpr);
            return PHIHIP_ERR_BAD_ARG;
        }
        if (!march_tile_available(c->vec, esize, id, v.n[2])) id = 5;          // (a pinned tile that this vector width does not have: the full-row tile)
        chunk = best_chunk(id, &score);
    } else {
        // candidate order + a small preference factor per family (from the family sweeps): UPDATE / residual favour large tiles
        // at equal chunk length, MATVEC medium tiles and the full-row tile (1, 64)
        static const int pref_mv[kNumTileConfigs] = {2, 5, 6, 7, 1, 0, 3, 4, 8, 9, 10, 11, 12}, pref_up[kNumTileConfigs] = {4, 3, 6, 2, 7, 1, 0, 5, 10, 9, 8, 12, 11};
        static const double bonus_mv[kNumTileConfigs] = {1.0, 1.0, 0.99, 0.98, 0.98, 0.97, 0.93, 0.93, 0.9, 0.9, 0.9, 0.85, 0.85},
                            bonus_up[kNumTileConfigs] = {1.0, 1.0, 0.99, 0.98, 0.98, 0.98, 0.97, 0.97, 0.9, 0.9, 0.9, 0.85, 0.85};
        // (the row tile is last and discounted in the ANALYTIC plan: it is the first-call autotune that decides for it, on the device)
        const int* pref = mv_like ? pref_mv : pref_up;
        const double* bonus = mv_like ? bonus_mv : bonus_up;
        double best_score = -1.0;
        for (int k = 0; k < kNumTileConfigs; ++k) {
            const int cand = pref[k];
            if (!march_tile_available(c->vec, esize, cand, v.n[2])) continue;
            int t1, t2;
            tile_of(cand, &t1, &t2);
            const double waste = (double)tiles_of(cand) * t1 * t2 / ((double)v.n[1] * v.n[2]);
            double sc;
            const int ch = best_chunk(cand, &sc);
            sc = sc / waste * bonus[k];
            if (sc > best_score) { best_score = sc; id = cand; chunk = ch; }
        }
    }
    if (v.rank == 3 && t.chunk > 0) chunk = t.chunk < v.n[0] ? t.chunk : v.n[0];
    tile_of(id, &c->t1, &c->t2);
    // every workgroup of the next kernel re-reduces all partial sums of its batch entry: keep that list short
    while (v.rank == 3 && t.chunk == 0 && tuned == ctx->tuned.end() && chunk < v.n[0] && tiles_of(id) * ceil_div(v.n[0], chunk) > 8192) chunk = chunk * 2 < v.n[0] ? chunk * 2 : v.n[0];
    c->id = id;
    c->chunk = chunk;
    g->tiles1 = ceil_div(v.n[1], c->t1);
    g->tiles2 = ceil_div(v.n[2], c->t2);
    g->chunks0 = ceil_div(v.n[0], chunk);
    g->chunk = chunk;
    g->nblk = g->tiles1 * g->tiles2 * g->chunks0;
    // short chunks re-read a large share of source planes (2 / chunk): let neighbouring chunks meet at their common boundary
    // (256^3 MATVEC with chunk 8: -7 %; at chunk 64 the shared planes are 3 % of the traffic and the reversal only costs)
    g->bidir = ((family == FAM_MATVEC || family == FAM_UPDATE_R) && v.rank == 3 && chunk <= 16 && g->chunks0 > 1) ? 1 : 0;
    g->tpr_rt = 0;
    if (id >= kRowTile && !march_one_tile(c->vec)) {      // a row tile: lanes per row at run time, one marching direction
        g->tpr_rt = v.n[2] / c->vec;
        g->bidir = 0;
    }
    return PHIHIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// control-block kernels outside the iteration: finalize / peek (one block per batch entry) and the refresh AXPY
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void cg_state_kernel(int kind, const CgState* st_in, CgState* st_out, const double* pin1,
                                                          const double* pin2, int nblk, CgParams prm) {
    __shared__ double red[kBlock / kWave];
    __shared__ CgState sh;
    cg_prologue(kind, st_in, st_out, pin1, pin2, nblk, prm, blockIdx.x, true, red, &sh);
}

// x += alpha * d for the true-residual refresh iterations (PhiML recomputes r = y - A x every 50th iteration)
// `r_pending` != nullptr: x also lacks the previous step (UPDATE_R ran last): add both like UPDATE_X2, r = the residual before this step
template <typename T>
__global__ __launch_bounds__(kBlock) void cg_axpy_x(T* x, const T* d, const T* r_pending, int kind, const CgState* st_in, CgState* st_out,
                                                    const double* part_dq, const double* part_dr, int nblk, CgParams prm, long long cells) {
    __shared__ double red[kBlock / kWave];
    __shared__ CgState sh;
    const int b = blockIdx.y;
    const CgState S = cg_prologue(kind, st_in, st_out, part_dq, part_dr, nblk, prm, b, blockIdx.x == 0, red, &sh, 0, 0);
    if (S.cont == 0) return;
    const T alpha = (T)S.alpha;
    const long long base = (long long)b * cells;
    if (r_pending) {
        const T c1 = (T)(S.alpha_prev / S.beta);
        for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < cells; i += (long long)gridDim.x * kBlock) {
            const T dv = d[base + i];
            x[base + i] = x[base + i] + c1 * (dv - r_pending[base + i]) + alpha * dv;
        }
        return;
    }
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < cells; i += (long long)gridDim.x * kBlock)
        x[base + i] = fma(alpha, d[base + i], x[base + i]);
}

// after the loop: batch entries that stopped right after an UPDATE_R step still lack alpha * d of that step
template <typename T>
__global__ __launch_bounds__(kBlock) void cg_flush_x(T* x, const T* d0, const T* d1, const CgState* st, long long cells) {
    const int b = blockIdx.y;
    const CgState S = st[b];
    if (!S.pending) return;
    const T alpha = (T)S.alpha;
    const T* d = (S.pend_buf ? d1 : d0) + (long long)b * cells;
    const long long base = (long long)b * cells;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < cells; i += (long long)gridDim.x * kBlock)
        x[base + i] = fma(alpha, d[i], x[base + i]);
}

// per-workgroup partial sums of a . b ('CG-adaptive' refresh iterations: r_new . A d with both vectors stored)
template <typename T>
__global__ __launch_bounds__(kBlock) void dot_partials_kernel(const T* a, const T* b, long long cells, double* part) {
    __shared__ double red[kBlock / kWave];
    const long long base = (long long)blockIdx.y * cells;
    T acc = T(0);
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < cells; i += (long long)gridDim.x * kBlock) acc += a[base + i] * b[base + i];
    const double s = block_sum((double)acc, red);
    if (threadIdx.x == 0) part[(long long)blockIdx.y * gridDim.x + blockIdx.x] = s;
}

__global__ void cg_export_residuals(const CgState* st, int batch, double* out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < batch) {
        out[2 * b] = st[b].rsq;
        out[2 * b + 1] = st[b].rhs_sq;
    }
}

// max over the batch entries of ||r|| / ||rhs||: one wavefront, fixed order
__global__ void cg_export_relative(const CgState* st, int batch, double* out) {
    double m = 0.0;
    for (int b = threadIdx.x; b < batch; b += kWave) {
        const double rel = st[b].rhs_sq > 0 ? sqrt(st[b].rsq / st[b].rhs_sq) : 0.0;
        m = rel > m ? rel : m;       // (NaN residuals compare false and are reported by the solve info, not here)
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        const double o = __shfl_down(m, off, kWave);
        m = o > m ? o : m;
    }
    if (threadIdx.x == 0) out[0] = m;
}

int run_export_relative_residual(phihip_ctx* ctx, int batch, double* out, hipStream_t s) {
    if (!ctx->last_state || ctx->last_state_batch < batch) {
        set_error("solve_relative_residual: no solve with batch >= %d has run on this context", batch);
        return PHIHIP_ERR_BAD_ARG;
    }
    hipLaunchKernelGGL(cg_export_relative, dim3(1), dim3(kWave), 0, s, (const CgState*)ctx->last_state, batch, out);
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

int run_export_residuals(phihip_ctx* ctx, int batch, double* out, hipStream_t s) {
    if (!ctx->last_state || ctx->last_state_batch < batch) {
        set_error("solve_residuals: no solve with batch >= %d has run on this context", batch);
        return PHIHIP_ERR_BAD_ARG;
    }
    hipLaunchKernelGGL(cg_export_residuals, dim3(ceil_div(batch, 64)), dim3(64), 0, s, (const CgState*)ctx->last_state, batch, out);
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// drivers
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
static int laplace_apply_t(phihip_ctx* ctx, const GridView& v, const uint8_t* flags, int mask_batch, const void* p, void* out,
                           hipStream_t s) {
    MarchConfig c;
    MarchGrid g;
    PHIHIP_TRY(plan_march(ctx, v, mask_batch, flags != nullptr, FAM_APPLY, &c, &g));
    MarchArgs<T> a;
    memset(&a, 0, sizeof(a));
    a.a = (const T*)p;
    a.o1 = (T*)out;
    a.flags = flags;
    set_operator(a, v);
    a.prologue = PRO_NONE;
    LaunchScope ls(ctx, PHIHIP_K_OTHER, s);
    PHIHIP_TRY(launch_march_any<T>(v, c, MODE_APPLY, flags != nullptr, g, a, s));
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

// (I + k dt L) on the `count` <= 3 lattices of a staggered field (diffuse.explicit: ONE reference call, phi/physics/diffuse.py:13-60): lattices whose plans share
// a tile configuration go into one launch of march_apply_multi_kernel -- all D components of a periodic box, the D - 1 components with whole rows of a closed
// one -- the rest one launch each. r5: always one launch per component (3 x 35 us on HBM-resident data at 256^3 + two launch gaps).
template <typename T>
static int laplace_apply_multi_t(phihip_ctx* ctx, const GridView* w, int count, const void* const* in, void* const* out, hipStream_t s) {
    MarchConfig c[3];
    MarchGrid g[3];
    MarchArgs<T> a[3];
    for (int l = 0; l < count; ++l) {
        PHIHIP_TRY(plan_march(ctx, w[l], 1, false, FAM_APPLY, &c[l], &g[l], l > 0 ? c[0].id : -1));
        memset(&a[l], 0, sizeof(a[l]));
        a[l].a = (const T*)in[l];
        a[l].o1 = (T*)out[l];
        set_operator(a[l], w[l]);
        a[l].prologue = PRO_NONE;
    }
    LaunchScope ls(ctx, PHIHIP_K_OTHER, s);
    bool done[3] = {false, false, false};
    for (int l = 0; l < count; ++l) {
        if (done[l]) continue;
        int members[3], nm = 0;
        for (int k = l; k < count; ++k)
            if (!done[k] && c[k].id == c[l].id && c[k].vec == c[l].vec && c[k].batch == c[l].batch && a[k].w0 == a[l].w0 && a[k].w1 == a[l].w1 &&
                a[k].w2 == a[l].w2 && a[k].ident == a[l].ident) members[nm++] = k;
        if (nm == 1) {
            PHIHIP_TRY(launch_march_any<T>(w[l], c[l], MODE_APPLY, false, g[l], a[l], s));
            done[l] = true;
            continue;
        }
        MarchGrid gm[3];
        MarchArgs<T> am[3];
        for (int k = 0; k < nm; ++k) { gm[k] = g[members[k]]; am[k] = a[members[k]]; done[members[k]] = true; }
        PHIHIP_TRY(launch_march_multi_any<T>(w[l], c[l], nm, gm, am, s));
    }
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

int run_laplace_apply_multi(phihip_ctx* ctx, const GridView* w, int count, const void* const* in, void* const* out, hipStream_t s) {
    return w[0].dtype == PHIHIP_F64 ? laplace_apply_multi_t<double>(ctx, w, count, in, out, s) : laplace_apply_multi_t<float>(ctx, w, count, in, out, s);
}

int run_laplace_apply(phihip_ctx* ctx, const GridView& v, const uint8_t* flags, int mask_batch, const void* p, void* out,
                      hipStream_t s) {
    return v.dtype == PHIHIP_F64 ? laplace_apply_t<double>(ctx, v, flags, mask_batch, p, out, s)
                                 : laplace_apply_t<float>(ctx, v, flags, mask_batch, p, out, s);
}

long long small_cg_limit(const phihip_ctx* ctx, const GridView& v);
int run_cg_small(phihip_ctx*, const GridView&, const uint8_t* flags, int mask_batch, const void* rhs, void* x, const phihip_solve*, void* st_out,
                 hipStream_t);

// small grids: the whole solve in ONE kernel, one workgroup per batch entry (cg_small.hip)
static int cg_small_path(phihip_ctx* ctx, const GridView& v, const uint8_t* flags, int mask_batch, const void* rhs, void* x,
                         const phihip_solve* solve, phihip_solve_info* info, hipStream_t s) {
    PHIHIP_TRY(ensure_buffer(ctx->ws_state, (size_t)3 * v.batch * sizeof(CgState)));
    CgState* st = (CgState*)ctx->ws_state.ptr;
    PHIHIP_TRY(run_cg_small(ctx, v, flags, mask_batch, rhs, x, solve, st, s));
    ctx->last_state = st;
    ctx->last_state_batch = v.batch;
    if (info) {
        if (ctx->host_state_bytes < (size_t)v.batch * sizeof(CgState)) {
            if (ctx->host_state) (void)hipHostFree(ctx->host_state);
            ctx->host_state = nullptr;
            ctx->host_state_bytes = 0;
            PHIHIP_CHECK_HIP(hipHostMalloc(&ctx->host_state, (size_t)v.batch * sizeof(CgState), hipHostMallocDefault));
            ctx->host_state_bytes = (size_t)v.batch * sizeof(CgState);
        }
        CgState* hst = (CgState*)ctx->host_state;
        PHIHIP_CHECK_HIP(hipMemcpyAsync(hst, st, (size_t)v.batch * sizeof(CgState), hipMemcpyDeviceToHost, s));
        PHIHIP_CHECK_HIP(hipStreamSynchronize(s));
        for (int b = 0; b < v.batch; ++b) {
            info[b].residual_sq = hst[b].rsq;
            info[b].rhs_sq = hst[b].rhs_sq;
            info[b].iterations = hst[b].iterations;
            info[b].converged = hst[b].converged;
            info[b].diverged = hst[b].diverged;
            info[b].reserved = 0;
        }
    }
    return PHIHIP_OK;
}

// 2-D fp32 grids whose iteration is bound by kernel boundaries: the whole solve in ONE launch of resident workgroups (cg_resident.hip)
bool cg_resident_applicable(const phihip_ctx* ctx, const GridView& v, const uint8_t* flags, const phihip_solve* solve);
bool cg_resident_batch_pays(const phihip_ctx* ctx, const GridView& v, const uint8_t* flags, const phihip_solve* solve);
int run_cg_resident(phihip_ctx*, const GridView&, const uint8_t* flags, int mask_batch, const void* rhs, void* x, const phihip_solve*, void* st_out, const double* shift,
                    hipStream_t);

static int cg_resident_path(phihip_ctx* ctx, const GridView& v, const uint8_t* flags, int mask_batch, const void* rhs, void* x, const phihip_solve* solve,
                            phihip_solve_info* info, const double* shift, hipStream_t s) {
    // A caller that passes no `info` is never synchronised with, so an aborted launch (not resident as a whole) cannot fail ITS call: the kernel
    // raises a word in host-mapped memory and the next resident solve of the context reports it (include/phihip.h).
    if (ctx->adv_host && ctx->adv_host[15]) {
        ctx->adv_host[15] = 0;
        set_error("cg (resident): an EARLIER resident solve of this context gave up (~1 s wait: the launch was not resident as a whole) -- its pressure is invalid");
        return PHIHIP_ERR_HIP;
    }
    PHIHIP_TRY(ensure_buffer(ctx->ws_state, (size_t)4 * v.batch * sizeof(CgState)));
    CgState* st = (CgState*)ctx->ws_state.ptr;
    PHIHIP_TRY(run_cg_resident(ctx, v, flags, mask_batch, rhs, x, solve, st, shift, s));
    ctx->last_state = st;
    ctx->last_state_batch = v.batch;
    if (info) {
        if (ctx->host_state_bytes < (size_t)v.batch * sizeof(CgState)) {
            if (ctx->host_state) (void)hipHostFree(ctx->host_state);
            ctx->host_state = nullptr;
            ctx->host_state_bytes = 0;
            PHIHIP_CHECK_HIP(hipHostMalloc(&ctx->host_state, (size_t)v.batch * sizeof(CgState), hipHostMallocDefault));
            ctx->host_state_bytes = (size_t)v.batch * sizeof(CgState);
        }
        CgState* hst = (CgState*)ctx->host_state;
        PHIHIP_CHECK_HIP(hipMemcpyAsync(hst, st, (size_t)v.batch * sizeof(CgState), hipMemcpyDeviceToHost, s));
        PHIHIP_CHECK_HIP(hipStreamSynchronize(s));
        for (int b = 0; b < v.batch; ++b) {
            if (hst[b].iterations < 0) {
                if (ctx->adv_host) ctx->adv_host[15] = 0;      // (reported here)
                set_error("cg (resident): a workgroup waited at a barrier for ~1 s -- the launch was not resident as a whole (is another stream using the device?)");
                return PHIHIP_ERR_HIP;
            }
            info[b].residual_sq = hst[b].rsq;
            info[b].rhs_sq = hst[b].rhs_sq;
            info[b].iterations = hst[b].iterations;
            info[b].converged = hst[b].converged;
            info[b].diverged = hst[b].diverged;
            info[b].reserved = 0;
        }
    }
    return PHIHIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// first-call autotune: time the (tile, chunk) candidates of MATVEC / UPDATE_X2 / UPDATE_R for this grid on the context's workspace
// (r, d0, d1 -- the solve that follows initialises them) and cache the fastest. The analytic plan of plan_march misses by up to
// 25 % between its fitted sizes (320^3 ... 448^3: profiles/r01_size_scan.jsonl); a measurement does not.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
static int autotune_cg(phihip_ctx* ctx, const GridView& v, const uint8_t* flags, int mask_batch, const T* rhs, T* r, T* d0, T* d1, T* xsol, double* part, hipStream_t s) {
    const bool has_flags = flags != nullptr;
    const int esize = (int)sizeof(T);
    const int vec = march_vector_width(v.n[2], esize, v.unaligned);
    static const int kChunks[12] = {128, 96, 64, 48, 32, 24, 16, 12, 8, 4, 2, 1};
    struct Cand { int id, chunk; float us; };
    const size_t vec_bytes = (size_t)v.batch * v.cells * sizeof(T);
    // r6: the candidates move the caller's right-hand side, not zeros (alpha = beta = 0 keep r = d = rhs and x as it is). PHIHIP_AUTOTUNE_DATA=0: zeros as until r5
    static const bool kRealData = [] { const char* e = getenv("PHIHIP_AUTOTUNE_DATA"); return !(e && e[0] == '0'); }();
    if (kRealData && rhs) {
        PHIHIP_CHECK_HIP(hipMemcpyAsync(r, rhs, vec_bytes, hipMemcpyDeviceToDevice, s));
        PHIHIP_CHECK_HIP(hipMemcpyAsync(d0, rhs, vec_bytes, hipMemcpyDeviceToDevice, s));
        PHIHIP_CHECK_HIP(hipMemcpyAsync(d1, rhs, vec_bytes, hipMemcpyDeviceToDevice, s));
    } else {
        PHIHIP_CHECK_HIP(hipMemsetAsync(r, 0, vec_bytes, s));
        PHIHIP_CHECK_HIP(hipMemsetAsync(d0, 0, vec_bytes, s));
        PHIHIP_CHECK_HIP(hipMemsetAsync(d1, 0, vec_bytes, s));
    }
    hipEvent_t e0, e1;
    PHIHIP_CHECK_HIP(hipEventCreate(&e0));
    PHIHIP_CHECK_HIP(hipEventCreate(&e1));
    int status = PHIHIP_OK;
    if (getenv("PHIHIP_AUTOTUNE_LOG")) fprintf(stderr, "[phihip autotune] grid %d x %d x %d batch %d flags %d mask_batch %d\n", v.n[0], v.n[1], v.n[2], v.batch, (int)has_flags, mask_batch);
    struct Pick { int id, chunk; float us, us_model; };
    Pick model_pick[FAM_COUNT], tuned_pick[FAM_COUNT];
    std::vector<Pick> challengers[FAM_COUNT];
    int fresh = 0;
    for (int family = FAM_MATVEC; family <= FAM_UPDATE_R && status == PHIHIP_OK; ++family) {
        const PlanKey key = plan_key(v, mask_batch, has_flags, family);
        if (ctx->tuned.count(key)) continue;
        const int mode = family == FAM_MATVEC ? MODE_MATVEC : (family == FAM_UPDATE ? MODE_UPDATE_X2 : MODE_UPDATE_R);
        MarchConfig c_model;
        MarchGrid g_model;
        status = plan_march(ctx, v, mask_batch, has_flags, family, &c_model, &g_model);
        if (status != PHIHIP_OK) break;
        const long long maxblk = g_model.nblk > 8192 ? g_model.nblk : 8192;
        if (ensure_buffer(ctx->ws_part, 5 * (size_t)v.batch * maxblk * sizeof(double)) != PHIHIP_OK) { status = PHIHIP_ERR_ALLOC; break; }
        part = (double*)ctx->ws_part.ptr;
        std::vector<Cand> cands;
        cands.push_back({c_model.id, c_model.chunk, 0.f});                      // the model's choice goes first (ties keep it)
        for (int id = 0; id < kNumTileConfigs; ++id) {
            if (!march_tile_available(vec, esize, id, v.n[2])) continue;
            if (v.rank != 3) {
                int e1, e2;
                march_tile_extent(vec, id, v.n[2], &e1, &e2);
                const long long blocks = (long long)ceil_div(v.n[1], e1) * ceil_div(v.n[2], e2);
                if (id != c_model.id && blocks <= maxblk) cands.push_back({id, 1, 0.f});
                continue;
            }
            for (int k = 0; k < 12; ++k) {
                const int ch = kChunks[k] < v.n[0] ? kChunks[k] : v.n[0];
                if (id == c_model.id && ch == c_model.chunk) continue;
                bool dup = false;
                for (const Cand& o : cands) dup = dup || (o.id == id && o.chunk == ch);
                if (dup) continue;
                int e1, e2;
                march_tile_extent(vec, id, v.n[2], &e1, &e2);
                const long long blocks = (long long)ceil_div(v.n[1], e1) * ceil_div(v.n[2], e2) * ceil_div(v.n[0], ch);
                // starved chip (but a little under one workgroup per CU is a candidate: 384^3 fp64 UPDATE_X2 runs fastest with 216 (4,64) workgroups of
                // 128 planes, profiles/r03_sweep_config5.jsonl) / partial-sum lists too long
                if (blocks * v.batch < ctx->num_cu * 3 / 4 || blocks > 8192) continue;
                cands.push_back({id, ch, 0.f});
            }
            // chunk lengths that fill the chip EVENLY: m workgroups on every CU (m = 1 .. resident workgroups, and two such rounds).
            // 384^3 MATVEC (1,16): 1152 workgroups of 48 planes (4.5 per CU) 146 us, 1008 of 55 planes (3.9) or 1440 of 39 (5.6) 137 us
            // (profiles/r02_midsize_chunks.jsonl) -- the fixed list above has no such member for most sizes
            {
                int e1, e2;
                march_tile_extent(vec, id, v.n[2], &e1, &e2);
                const long long tiles = (long long)ceil_div(v.n[1], e1) * ceil_div(v.n[2], e2) * v.batch;
                const int occ = march_occupancy_any(v, id, vec, mode, has_flags);
                for (int m = 1; m <= occ + 1; ++m) {
                    const long long target = (long long)(m <= occ ? m : 2 * occ) * ctx->num_cu;
                    long long chunks0 = target / tiles;
                    if (chunks0 < 1) continue;
                    if (chunks0 > v.n[0]) chunks0 = v.n[0];
                    const int ch = ceil_div(v.n[0], (int)chunks0);
                    const long long blocks = tiles / v.batch * ceil_div(v.n[0], ch);
                    if (blocks * v.batch < ctx->num_cu || blocks > 8192) continue;
                    bool dup = false;
                    for (const Cand& o : cands) dup = dup || (o.id == id && o.chunk == ch);
                    if (!dup) cands.push_back({id, ch, 0.f});
                }
            }
        }
        const Tuning saved = ctx->tuning[family];
        auto run = [&](Cand& cd, int reps) -> int {
            ctx->tuning[family].rows = march_one_tile(vec) ? 1 : kTileShapes[cd.id].rows;
            ctx->tuning[family].tpr = march_one_tile(vec) ? 64 : kTileShapes[cd.id].tpr;
            ctx->tuning[family].chunk = v.rank == 3 ? cd.chunk : 0;
            MarchConfig c;
            MarchGrid g;
            PHIHIP_TRY(plan_march(ctx, v, mask_batch, has_flags, family, &c, &g));
            MarchArgs<T> a;
            memset(&a, 0, sizeof(a));
            a.flags = flags;
            set_operator(a, v);
            a.prologue = PRO_NONE;                       // alpha = beta = 0: every vector stays zero
            a.part1 = part; a.part2 = part + (size_t)v.batch * maxblk;
            if (family == FAM_MATVEC) { a.a = r; a.b = d0; a.o1 = d1; }
            else { a.a = d1; a.o1 = family == FAM_UPDATE ? xsol : d0; a.o2 = r; }      // (UPDATE_X2 updates the caller's x: alpha = beta = 0 leave it as it is)
            float best = 1e30f;
            for (int k = 0; k < reps; ++k) {
                PHIHIP_CHECK_HIP(hipEventRecord(e0, s));
                PHIHIP_TRY(launch_march_any<T>(v, c, mode, has_flags, g, a, s));
                PHIHIP_CHECK_HIP(hipEventRecord(e1, s));
                PHIHIP_CHECK_HIP(hipEventSynchronize(e1));
                float ms = 0;
                PHIHIP_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            cd.us = best * 1e3f;
            return PHIHIP_OK;
        };
        Cand warm = cands[0];
        status = run(warm, 2);                                                   // clocks up, code loaded
        // Time budget of the first pass: max(60 ms, 40 launches of the model's plan) per family -- everything at <= 512^3, the model's tile
        // with its other chunk lengths first and then whatever fits at 1024^3 (2.5 ms per launch), where trying all ~100 would cost seconds.
        // Untimed candidates keep us = 1e30 and cannot win.
        {
            std::vector<Cand> ordered;
            ordered.push_back(cands[0]);
            for (size_t k = 1; k < cands.size(); ++k) if (cands[k].id == cands[0].id) ordered.push_back(cands[k]);
            for (size_t k = 1; k < cands.size(); ++k) if (cands[k].id != cands[0].id) ordered.push_back(cands[k]);
            cands.swap(ordered);
        }
        double spent_us = 0, budget_us = 6e4;
        for (size_t k = 0; k < cands.size() && status == PHIHIP_OK; ++k) {
            if (k > 0 && spent_us > budget_us) { cands[k].us = 1e30f; continue; }
            status = run(cands[k], 2);
            if (k == 0 && 40.0 * cands[0].us > budget_us) budget_us = 40.0 * cands[0].us;
            spent_us += 2.0 * cands[k].us;
        }
        // confirm the leaders with more repetitions (single launches of ~30 us carry a few % of timer noise)
        std::vector<size_t> order(cands.size());
        for (size_t k = 0; k < order.size(); ++k) order[k] = k;
        for (size_t a_ = 0; a_ + 1 < order.size(); ++a_)
            for (size_t b_ = a_ + 1; b_ < order.size(); ++b_)
                if (cands[order[b_]].us < cands[order[a_]].us) { const size_t t_ = order[a_]; order[a_] = order[b_]; order[b_] = t_; }
        const float us_model_first = cands[0].us;
        for (size_t k = 0; k < order.size() && k < 3 && status == PHIHIP_OK; ++k) status = run(cands[order[k]], 5);
        if (status == PHIHIP_OK) status = run(cands[0], 5);
        ctx->tuning[family] = saved;
        if (status != PHIHIP_OK) break;
        size_t win = 0;
        for (size_t k = 1; k < cands.size(); ++k)
            if (cands[k].us < cands[win].us * 0.98f) win = k;                   // the model's plan stays unless something is > 2 % faster
        const float us_model = cands[0].us < us_model_first ? cands[0].us : us_model_first;
        model_pick[family] = Pick{cands[0].id, cands[0].chunk, us_model, us_model};
        tuned_pick[family] = Pick{cands[win].id, cands[win].chunk, cands[win].us, us_model};
        for (size_t k = 0; k < order.size() && challengers[family].size() < 3; ++k)      // the three fastest in isolation (after the re-timing)
            if (order[k] != 0 && cands[order[k]].us < us_model * 1.02f) challengers[family].push_back(Pick{cands[order[k]].id, cands[order[k]].chunk, cands[order[k]].us, us_model});
        ++fresh;
    }
    // A candidate timed on its own re-reads operands its previous launch left in the Infinity Cache; inside the iteration the other phase
    // has replaced them (192^3: a plan that won in isolation lost 15 % in the loop). So every challenger is confirmed in the loop it will
    // run in -- MATVEC, UPDATE_R, MATVEC, UPDATE_X2 on the (zeroed) workspace -- against the model's plans, one family at a time.
    if (status == PHIHIP_OK && fresh == 3) {
        Tuning saved[FAM_COUNT];
        for (int f = 0; f < FAM_COUNT; ++f) saved[f] = ctx->tuning[f];
        auto loop_us = [&](const Pick (&pk)[FAM_COUNT], float* us) -> int {
            MarchConfig c[FAM_COUNT];
            MarchGrid g[FAM_COUNT];
            for (int f = FAM_MATVEC; f <= FAM_UPDATE_R; ++f) {
                ctx->tuning[f].rows = march_one_tile(vec) ? 1 : kTileShapes[pk[f].id].rows;
                ctx->tuning[f].tpr = march_one_tile(vec) ? 64 : kTileShapes[pk[f].id].tpr;
                ctx->tuning[f].chunk = v.rank == 3 ? pk[f].chunk : 0;
                PHIHIP_TRY(plan_march(ctx, v, mask_batch, has_flags, f, &c[f], &g[f]));
            }
            long long maxblk = 8192;
            for (int f = FAM_MATVEC; f <= FAM_UPDATE_R; ++f) maxblk = g[f].nblk > maxblk ? g[f].nblk : maxblk;
            if (ensure_buffer(ctx->ws_part, 5 * (size_t)v.batch * maxblk * sizeof(double)) != PHIHIP_OK) return PHIHIP_ERR_ALLOC;
            double* pp = (double*)ctx->ws_part.ptr;
            MarchArgs<T> a;
            memset(&a, 0, sizeof(a));
            a.flags = flags;
            set_operator(a, v);
            a.prologue = PRO_NONE;
            a.part1 = pp; a.part2 = pp + (size_t)v.batch * maxblk;
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                PHIHIP_CHECK_HIP(hipEventRecord(e0, s));
                for (int it = 0; it < 2; ++it) {
                    T* dn = it ? d0 : d1;
                    T* dold = it ? d1 : d0;
                    a.a = r; a.b = dold; a.o1 = dn; a.o2 = nullptr;
                    PHIHIP_TRY(launch_march_any<T>(v, c[FAM_MATVEC], MODE_MATVEC, has_flags, g[FAM_MATVEC], a, s));
                    // r6: UPDATE_X2 runs on the caller's solution vector (alpha = beta = 0: x + 0 = x), a FOURTH array like in the solve. Until r5 the idle direction
                    // buffer stood in for x: three arrays instead of four changes what the Infinity Cache holds at 288^3 ... 448^3 (an array is 100-360 MB there),
                    // and the loop ranked UPDATE_X2 plans differently from the solve (384^3: the confirmed plan ran 0.30 ms per iteration where another ran 0.27,
                    // profiles/r06_autotune_stability.txt)
                    a.a = dn; a.b = nullptr; a.o1 = it == 0 ? dold : xsol; a.o2 = r;
                    if (it == 0) PHIHIP_TRY(launch_march_any<T>(v, c[FAM_UPDATE_R], MODE_UPDATE_R, has_flags, g[FAM_UPDATE_R], a, s));
                    else PHIHIP_TRY(launch_march_any<T>(v, c[FAM_UPDATE], MODE_UPDATE_X2, has_flags, g[FAM_UPDATE], a, s));
                }
                PHIHIP_CHECK_HIP(hipEventRecord(e1, s));
                PHIHIP_CHECK_HIP(hipEventSynchronize(e1));
                float ms = 0;
                PHIHIP_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            *us = best * 1e3f;
            return PHIHIP_OK;
        };
        float us_base = 0, us_tmp = 0;
        status = loop_us(model_pick, &us_tmp);                                     // (first pass warms the clocks)
        if (status == PHIHIP_OK) status = loop_us(model_pick, &us_base);
        Pick final_pick[FAM_COUNT];
        for (int f = 0; f < FAM_COUNT; ++f) final_pick[f] = model_pick[f];
        // a challenger replaces the model's plan only if it shortens the loop by >= kAccept TWICE, the second time against a fresh timing of
        // the model's loop (box noise is ~1-2 %; a wrong replacement costs more than a missed one gains). r6: 1.5 % -> 3 % (PHIHIP_AUTOTUNE_ACCEPT, percent):
        // six fresh contexts at 384^3 picked UPDATE_X2 (4,32) x 28 four times on a 1.5-2 % margin and ran 0.28-0.30 ms per iteration with it against
        // 0.27 with the model's (2,32) x 55 (profiles/r06_autotune_stability.txt)
        static const bool kLog = getenv("PHIHIP_AUTOTUNE_LOG") != nullptr;      // the loop timings behind every decision, on stderr
        static const float kAccept = [] { const char* e = getenv("PHIHIP_AUTOTUNE_ACCEPT"); const double p = e ? atof(e) : 3.0; return (float)(1.0 - (p > 0 && p < 50 ? p : 3.0) / 100.0); }();
        for (int f = FAM_MATVEC; f <= FAM_UPDATE_R && status == PHIHIP_OK; ++f) {
            float us_best = us_base * kAccept;
            for (const Pick& ch : challengers[f]) {
                Pick trial[FAM_COUNT];
                for (int k = 0; k < FAM_COUNT; ++k) trial[k] = model_pick[k];
                trial[f] = ch;
                float us_trial = 0, us_again = 0;
                status = loop_us(trial, &us_trial);
                if (status != PHIHIP_OK) break;
                if (us_trial >= us_best) continue;
                status = loop_us(model_pick, &us_tmp);
                if (status == PHIHIP_OK) status = loop_us(trial, &us_again);
                if (status != PHIHIP_OK) break;
                us_base = us_tmp < us_base ? us_tmp : us_base;
                const float us_t = us_again > us_trial ? us_again : us_trial;     // the slower of the two timings has to win as well
                if (kLog) fprintf(stderr, "[phihip autotune] n=%d family %d challenger (%d,%d) x %d: loop %.1f / %.1f us, model's loop %.1f us (fresh %.1f)\n", v.n[0], f,
                                  kTileShapes[ch.id].rows, kTileShapes[ch.id].tpr, ch.chunk, us_trial, us_again, us_base, us_tmp);
                if (us_t < us_base * kAccept && us_t < us_best) { us_best = us_t; final_pick[f] = ch; }
            }
        }
        for (int f = 0; f < FAM_COUNT; ++f) ctx->tuning[f] = saved[f];
        if (status == PHIHIP_OK)
            for (int f = FAM_MATVEC; f <= FAM_UPDATE_R; ++f) {
                TunedPlan tp;
                tp.id = final_pick[f].id; tp.chunk = final_pick[f].chunk; tp.us = final_pick[f].us; tp.us_model = final_pick[f].us_model;
                ctx->tuned[plan_key(v, mask_batch, has_flags, f)] = tp;
            }
    } else if (status == PHIHIP_OK) {
        for (int f = FAM_MATVEC; f <= FAM_UPDATE_R; ++f) {
            const PlanKey key = plan_key(v, mask_batch, has_flags, f);
            if (ctx->tuned.count(key)) continue;
            TunedPlan tp;
            tp.id = tuned_pick[f].id; tp.chunk = tuned_pick[f].chunk; tp.us = tuned_pick[f].us; tp.us_model = tuned_pick[f].us_model;
            ctx->tuned[key] = tp;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return status;
}

// ---------------------------------------------------------------------------------------------------------------------
// Workspace placement (r6, last session). What a CG iteration costs depends on WHICH allocations hold r, d0, d1: six contexts alive at once, identical launch
// plans, the caller's x / rhs shared -- 512^3: 0.666 ... 0.731 ms per iteration, 384^3: 0.284 ... 0.314, 256^3: 0.0738 ... 0.0755, every context within 0.1 % of
// itself from round to round (tools/micro/ws_placement_probe.py, profiles/r06_ws_placement_probe.jsonl). A buffer streamed ALONE runs at the same rate as any
// other (8 x 512 MiB: 6.09-6.11 TB/s); two reads + one write over three of them differ by 3.5 % with the triple (tools/micro/buffer_bandwidth_probe.py): it is the
// relative position of the streams in the physical address space (channel / bank conflicts), which a library cannot see -- but it can hold several candidate
// allocations, time the loop the solve runs (MATVEC, UPDATE_R, MATVEC, UPDATE_X2 with the tuned plans, on the caller's x) on each and keep the fastest. Runs once
// per growth of the workspace, behind the first-call autotune (not under capture; with the plans the solve runs: tuned, pinned or analytic). The arithmetic does not know where a vector
// lives: results are bit-identical. PHIHIP_WS_CANDIDATES / phihip_workspace_placement = number of (r, d0, d1) triples to choose from (default 12; <= 1: the first
// allocation is kept).
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
static int place_workspace(phihip_ctx* ctx, const GridView& v, const uint8_t* flags, int mask_batch, const T* rhs, T* xsol, hipStream_t s) {
    const int kCandidates = ctx->ws_candidates;
    static const bool kLog = getenv("PHIHIP_AUTOTUNE_LOG") != nullptr;
    const bool has_flags = flags != nullptr;
    const size_t vec_bytes = (size_t)v.batch * v.cells * sizeof(T);
    // vectors the Infinity Cache regime holds (<= 72 MB: 256^3 fp32, the bound plan_march uses) cost the same wherever they live (256^3: 14 candidates within 1 %)
    if (kCandidates <= 1 || vec_bytes <= ctx->ws_place_min_bytes) return PHIHIP_OK;
    MarchConfig c[FAM_COUNT];
    MarchGrid g[FAM_COUNT];
    long long maxblk = 8192;
    for (int f = FAM_MATVEC; f <= FAM_UPDATE_R; ++f) {
        PHIHIP_TRY(plan_march(ctx, v, mask_batch, has_flags, f, &c[f], &g[f]));
        maxblk = g[f].nblk > maxblk ? g[f].nblk : maxblk;
    }
    PHIHIP_TRY(ensure_buffer(ctx->ws_part, 5 * (size_t)v.batch * maxblk * sizeof(double)));
    double* pp = (double*)ctx->ws_part.ptr;
    struct Triple { DeviceBuffer b[3]; float us; };
    std::vector<Triple> cand(1);
    cand[0].b[0] = ctx->ws_r; cand[0].b[1] = ctx->ws_d0; cand[0].b[2] = ctx->ws_d1;
    for (int k = 1; k < kCandidates; ++k) {
        size_t free_b = 0, total_b = 0;
        // (never more than half of what is free, and never more than 32 GiB of candidates at once: 1024^3 fp32 chooses between three triples, 512^3 between all)
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b / 2 < 3 * vec_bytes || (size_t)(k + 1) * 3 * vec_bytes > ((size_t)32 << 30) + 3 * vec_bytes) break;
        Triple t;
        bool ok = true;
        for (int i = 0; i < 3 && ok; ++i) {
            const hipError_t e = hipMalloc(&t.b[i].ptr, vec_bytes);
            ok = e == hipSuccess;
            if (ok) t.b[i].bytes = vec_bytes; else t.b[i].ptr = nullptr;
        }
        if (!ok) {
            (void)hipGetLastError();
            for (int i = 0; i < 3; ++i) if (t.b[i].ptr) (void)hipFree(t.b[i].ptr);
            break;
        }
        cand.push_back(t);
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int status = PHIHIP_OK;
    if (cand.size() > 1 && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) status = PHIHIP_ERR_HIP;
    MarchArgs<T> a;
    memset(&a, 0, sizeof(a));
    a.flags = flags;
    set_operator(a, v);
    a.prologue = PRO_NONE;
    a.part1 = pp; a.part2 = pp + (size_t)v.batch * maxblk;
    auto loops = [&](const Triple& t, int n) -> int {      // alpha = beta = 0: r, d and x keep their values
        T* r = (T*)t.b[0].ptr;
        for (int k = 0; k < n; ++k)
            for (int it = 0; it < 2; ++it) {
                T* dn = (T*)(it ? t.b[1].ptr : t.b[2].ptr);
                T* dold = (T*)(it ? t.b[2].ptr : t.b[1].ptr);
                a.a = r; a.b = dold; a.o1 = dn; a.o2 = nullptr;
                PHIHIP_TRY(launch_march_any<T>(v, c[FAM_MATVEC], MODE_MATVEC, has_flags, g[FAM_MATVEC], a, s));
                a.a = dn; a.b = nullptr; a.o1 = it == 0 ? dold : xsol; a.o2 = r;
                if (it == 0) PHIHIP_TRY(launch_march_any<T>(v, c[FAM_UPDATE_R], MODE_UPDATE_R, has_flags, g[FAM_UPDATE_R], a, s));
                else PHIHIP_TRY(launch_march_any<T>(v, c[FAM_UPDATE], MODE_UPDATE_X2, has_flags, g[FAM_UPDATE], a, s));
            }
        return PHIHIP_OK;
    };
    int reps = 2;
    for (size_t k = 0; k < cand.size() && status == PHIHIP_OK && cand.size() > 1; ++k) {
        Triple& t = cand[k];
        for (int i = 0; i < 3 && status == PHIHIP_OK; ++i) {
            const hipError_t e = rhs ? hipMemcpyAsync(t.b[i].ptr, rhs, vec_bytes, hipMemcpyDeviceToDevice, s) : hipMemsetAsync(t.b[i].ptr, 0, vec_bytes, s);
            if (e != hipSuccess) status = PHIHIP_ERR_HIP;
        }
        t.us = 1e30f;
        for (int round = 0; round < 3 && status == PHIHIP_OK; ++round) {      // round 0 warms the candidate and (on the first) sizes the repetition count: >= 2 ms per timing
            if (hipEventRecord(e0, s) != hipSuccess) { status = PHIHIP_ERR_HIP; break; }
            status = loops(t, round == 0 ? 1 : reps);
            if (status != PHIHIP_OK) break;
            if (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) { status = PHIHIP_ERR_HIP; break; }
            float ms = 0;
            if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { status = PHIHIP_ERR_HIP; break; }
            if (round == 0) {
                if (k == 0) { const int n = ms > 0 ? (int)(2.0f / ms) + 1 : 2; reps = n < 2 ? 2 : (n > 24 ? 24 : n); }
                continue;
            }
            const float us = ms * 1e3f / (2 * reps);
            t.us = us < t.us ? us : t.us;
        }
    }
    size_t win = 0;
    if (status == PHIHIP_OK)
        for (size_t k = 1; k < cand.size(); ++k)
            if (cand[k].us < cand[win].us * (win == 0 ? 0.995f : 1.0f)) win = k;      // (the first allocation stays unless another is > 0.5 % faster)
    if (kLog && cand.size() > 1) {
        fprintf(stderr, "[phihip placement] grid %d x %d x %d batch %d: us per iteration", v.n[0], v.n[1], v.n[2], v.batch);
        for (size_t k = 0; k < cand.size(); ++k) fprintf(stderr, " %.2f%s", cand[k].us, k == win ? "*" : "");
        fprintf(stderr, "\n");
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (cand.size() > 1 && hipStreamSynchronize(s) != hipSuccess && status == PHIHIP_OK) status = PHIHIP_ERR_HIP;      // nothing is freed under a running launch
    ctx->ws_r = cand[win].b[0]; ctx->ws_d0 = cand[win].b[1]; ctx->ws_d1 = cand[win].b[2];
    for (size_t k = 0; k < cand.size(); ++k)
        if (k != win)
            for (int i = 0; i < 3; ++i) (void)hipFree(cand[k].b[i].ptr);
    ctx->ws_place_count = (int)cand.size();
    ctx->ws_place_best_us = cand[win].us;
    ctx->ws_place_first_us = cand[0].us;
    return status;
}

// ---------------------------------------------------------------------------------------------------------------------
// Single-reduction CG (Chronopoulos & Gear 1989; stencil_march.hpp MODE_CG1): ONE launch per iteration. Used where the two launches of
// the form above are bound by their boundaries (dependent launch ~2.7 us + prologue chain, profiles/r02_xcd_barrier_microbench.txt), not
// by traffic. Vectors: r, w = A r, s = A p in ping-pong pairs (a launch's neighbours still read the inputs), p and x in place.
//   start:  r = y - A x (RESID[_BAL]) ; state from (|r|^2, |y|^2) ; w = A r with gamma = |r|^2, delta = (A r).r (APPLY_DOT)
//   k-th launch: prologue (gamma', delta' of the previous launch) -> PhiML's convergence / divergence tests, beta, alpha, count; body above
//   refresh (every 50th like PhiML): r = y - A x ; w = A r, gamma, delta recomputed -- s = A p continues by its recurrence
// ---------------------------------------------------------------------------------------------------------------------
static long long cg1_threshold(const phihip_ctx* ctx, const GridView& v) {
    if (ctx->cg1_cells > 0) return ctx->cg1_cells;
    // cells x batch up to which ONE launch with 10 words per cell beats TWO with 7 (tools/sweep_cg1.py, profiles/r03_cg1_sweep.jsonl: 1.10-1.31x
    // faster at 0.26-1.05 M cells, 0.67-0.81x at 2.1 M)
    return v.dtype == PHIHIP_F64 ? 600000 : 1200000;
}

template <typename T>
static int cg1_t(phihip_ctx* ctx, const GridView& v, const uint8_t* flags, int mask_batch, const void* rhs, void* x, const phihip_solve* solve,
                 phihip_solve_info* info, const double* shift, hipStream_t s) {
    MarchConfig c, c1;
    MarchGrid g, g1;
    PHIHIP_TRY(plan_march(ctx, v, mask_batch, flags != nullptr, FAM_APPLY, &c, &g));
    PHIHIP_TRY(plan_march(ctx, v, mask_batch, flags != nullptr, FAM_CG1, &c1, &g1));
    const bool has_flags = flags != nullptr;
    const size_t vec_bytes = (((size_t)v.batch * v.cells * sizeof(T)) + 255) / 256 * 256;
    const int nblk_max = g.nblk > g1.nblk ? g.nblk : g1.nblk;
    const size_t part_n = (size_t)v.batch * nblk_max;
    PHIHIP_TRY(ensure_buffer(ctx->ws_r, vec_bytes));
    PHIHIP_TRY(ensure_buffer(ctx->ws_d0, vec_bytes));
    PHIHIP_TRY(ensure_buffer(ctx->ws_d1, vec_bytes));
    PHIHIP_TRY(ensure_buffer(ctx->ws_cg1, 4 * vec_bytes));
    PHIHIP_TRY(ensure_buffer(ctx->ws_part, 12 * part_n * sizeof(double)));
    PHIHIP_TRY(ensure_buffer(ctx->ws_state, (size_t)4 * v.batch * sizeof(CgState)));
    if (ctx->host_state_bytes < (size_t)2 * v.batch * sizeof(CgState)) {
        if (ctx->host_state) (void)hipHostFree(ctx->host_state);
        ctx->host_state = nullptr;
        ctx->host_state_bytes = 0;
        PHIHIP_CHECK_HIP(hipHostMalloc(&ctx->host_state, (size_t)2 * v.batch * sizeof(CgState), hipHostMallocDefault));
        ctx->host_state_bytes = (size_t)2 * v.batch * sizeof(CgState);
    }
    if (!ctx->poll_ev[0]) {
        PHIHIP_CHECK_HIP(hipEventCreate(&ctx->poll_ev[0]));
        PHIHIP_CHECK_HIP(hipEventCreate(&ctx->poll_ev[1]));
    }
    if (ctx->host_flags_count < (size_t)v.batch) {
        if (ctx->host_flags) (void)hipHostFree(ctx->host_flags);
        ctx->host_flags = nullptr;
        ctx->host_flags_count = 0;
        PHIHIP_CHECK_HIP(hipHostMalloc((void**)&ctx->host_flags, (size_t)v.batch * sizeof(unsigned long long), hipHostMallocMapped));
        memset(ctx->host_flags, 0, (size_t)v.batch * sizeof(unsigned long long));
        PHIHIP_CHECK_HIP(hipHostGetDevicePointer((void**)&ctx->host_flags_dev, ctx->host_flags, 0));
        ctx->host_flags_count = (size_t)v.batch;
    }
    const unsigned int seq = ++ctx->solve_seq;
    char* extra = (char*)ctx->ws_cg1.ptr;
    T* rv[2] = {(T*)ctx->ws_r.ptr, (T*)extra};
    T* wv[2] = {(T*)ctx->ws_d0.ptr, (T*)(extra + vec_bytes)};
    T* sv[2] = {(T*)ctx->ws_d1.ptr, (T*)(extra + 2 * vec_bytes)};
    T* pv = (T*)(extra + 3 * vec_bytes);
    double* part = (double*)ctx->ws_part.ptr;
    double* part_g[2] = {part, part + part_n};            // gamma / delta of the launch before, ping-pong (a launch reads one pair, writes the other)
    double* part_d[2] = {part + 2 * part_n, part + 3 * part_n};
    double* part_rr = part + 4 * part_n;
    double* part_yy = part + 5 * part_n;
    double* part_mu[2] = {part + 6 * part_n, part + 7 * part_n};       // r'.s, (A r').p, p.s of the launch before (five-sum closure of alpha)
    double* part_nu[2] = {part + 8 * part_n, part + 9 * part_n};
    double* part_sg[2] = {part + 10 * part_n, part + 11 * part_n};
    CgState* st[2] = {(CgState*)ctx->ws_state.ptr, (CgState*)ctx->ws_state.ptr + v.batch};
    int cur = 0;
    CgParams prm;
    prm.rtol = solve->rel_tol; prm.atol = solve->abs_tol; prm.max_iter = solve->max_iterations; prm.pad = 0;
    MarchArgs<T> base;
    memset(&base, 0, sizeof(base));
    base.flags = flags;
    base.prm = prm;
    set_operator(base, v);

    int vc = 0;     // which half of the (r, w, s) pairs holds the current vectors
    int pc = 0;     // which (gamma, delta) partial pair the next CG1 prologue reads
    int nblk_in = g.nblk;
    PHIHIP_CHECK_HIP(hipMemsetAsync(pv, 0, vec_bytes, s));          // p_{-1} = s_{-1} = 0 (beta_0 = 0 alone would keep NaN garbage alive)
    PHIHIP_CHECK_HIP(hipMemsetAsync(sv[0], 0, vec_bytes, s));
    auto residual_and_w = [&](int prologue, bool balance_now) -> int {
        {   // r = y - A x
            MarchArgs<T> a = base;
            a.a = (const T*)x; a.b = (const T*)rhs; a.o1 = rv[vc];
            a.part1 = part_rr; a.part2 = part_yy;
            a.prologue = prologue;
            a.st_in = st[cur];
            a.shift = balance_now ? shift : nullptr;
            a.yout = balance_now ? (T*)const_cast<void*>(rhs) : nullptr;
            LaunchScope ls(ctx, PHIHIP_K_CG_RESIDUAL, s);
            PHIHIP_TRY(launch_march_any<T>(v, c, balance_now ? MODE_RESID_BAL : MODE_RESID, has_flags, g, a, s));
        }
        if (prologue == PRO_NONE) {   // the control block of this solve: tolerances, |r0|^2, converged at once?
            LaunchScope ls(ctx, PHIHIP_K_CG_SCALAR, s);
            hipLaunchKernelGGL(cg_state_kernel, dim3(v.batch), dim3(kBlock), 0, s, (int)PRO_FIRST, (const CgState*)st[cur], st[cur ^ 1], (const double*)part_rr,
                               (const double*)part_yy, g.nblk, prm);
            cur ^= 1;
        }
        {   // w = A r ; gamma = |r|^2 ; delta = (A r) . r ; and against the standing p, s (zero at the start): mu = r.s, nu = (A r).p, sigma = p.s
            MarchArgs<T> a = base;
            a.a = rv[vc]; a.o1 = wv[vc];
            a.c = sv[vc]; a.o4 = pv;
            a.part1 = part_g[pc]; a.part2 = part_d[pc];
            a.part3 = part_mu[pc]; a.part4 = part_nu[pc]; a.part5 = part_sg[pc];
            a.prologue = PRO_CONT;
            a.st_in = st[cur];
            LaunchScope ls(ctx, PHIHIP_K_OTHER, s);
            PHIHIP_TRY(launch_march_any<T>(v, c, MODE_APPLY_DOT, has_flags, g, a, s));
        }
        nblk_in = g.nblk;
        return PHIHIP_OK;
    };
    PHIHIP_TRY(residual_and_w(PRO_NONE, shift != nullptr));
    int checks = 0;
    for (int k = 1; k <= solve->max_iterations; ++k) {
        {
            MarchArgs<T> a = base;
            a.a = rv[vc]; a.b = wv[vc]; a.c = sv[vc];
            a.o1 = rv[vc ^ 1]; a.o2 = wv[vc ^ 1]; a.o3 = sv[vc ^ 1]; a.o4 = pv; a.o5 = (T*)x;
            a.pin1 = part_g[pc]; a.pin2 = part_d[pc]; a.nblk_in = nblk_in;
            a.pin3 = part_mu[pc]; a.pin4 = part_nu[pc]; a.pin5 = part_sg[pc];
            a.part1 = part_g[pc ^ 1]; a.part2 = part_d[pc ^ 1];
            a.part3 = part_mu[pc ^ 1]; a.part4 = part_nu[pc ^ 1]; a.part5 = part_sg[pc ^ 1];
            a.prologue = PRO_CG1;
            a.st_in = st[cur]; a.st_out = st[cur ^ 1];
            if (solve->check_every > 0) { a.host_flags = ctx->host_flags_dev; a.seq = seq; }
            LaunchScope ls(ctx, PHIHIP_K_CG_UPDATE, s);
            PHIHIP_TRY(launch_march_any<T>(v, c1, MODE_CG1, has_flags, g1, a, s));
            cur ^= 1; vc ^= 1; pc ^= 1;
            nblk_in = g1.nblk;
        }
        if (solve->refresh_every > 0 && k % solve->refresh_every == 0) {
            // true residual like PhiML every 50th iteration: r = y - A x and w, gamma, delta from it REPLACE the sums of launch k in the
            // pair the next prologue reads -- beta = gamma_true / gamma_k exactly as PhiML forms it (rsq / rsq_old); entries that were
            // frozen before launch k skip both passes (PRO_CONT)
            PHIHIP_TRY(residual_and_w(PRO_CONT, false));
        }
        if (solve->check_every > 0 && k < solve->max_iterations) {
            bool any = false;
            for (int b = 0; b < v.batch && !any; ++b) {
                const unsigned long long f = *(volatile unsigned long long*)(ctx->host_flags + b);
                any = (unsigned int)(f >> 32) != seq || (f & 1ull);
            }
            if (!any) break;
            if (k % solve->check_every == 0) {
                const int slot = checks & 1;
                PHIHIP_CHECK_HIP(hipEventRecord(ctx->poll_ev[slot], s));
                if (checks > 0) PHIHIP_CHECK_HIP(hipEventSynchronize(ctx->poll_ev[slot ^ 1]));
                ++checks;
            }
        }
    }
    {   // fold the last (gamma, delta) into the control block: converged / diverged / residual of the final iterate
        LaunchScope ls(ctx, PHIHIP_K_CG_SCALAR, s);
        hipLaunchKernelGGL(cg_state_kernel, dim3(v.batch), dim3(kBlock), 0, s, (int)PRO_BETA, (const CgState*)st[cur], st[cur ^ 1], (const double*)part_g[pc],
                           (const double*)part_d[pc], nblk_in, prm);
        cur ^= 1;
    }
    ctx->last_state = st[cur];
    ctx->last_state_batch = v.batch;
    PHIHIP_CHECK_HIP(hipGetLastError());
    if (info) {
        CgState* hst = (CgState*)ctx->host_state;
        PHIHIP_CHECK_HIP(hipMemcpyAsync(hst, st[cur], (size_t)v.batch * sizeof(CgState), hipMemcpyDeviceToHost, s));
        PHIHIP_CHECK_HIP(hipStreamSynchronize(s));
        for (int b = 0; b < v.batch; ++b) {
            info[b].residual_sq = hst[b].rsq;
            info[b].rhs_sq = hst[b].rhs_sq;
            info[b].iterations = hst[b].iterations;
            info[b].converged = hst[b].converged;
            info[b].diverged = hst[b].diverged;
            info[b].reserved = 0;
        }
    }
    return PHIHIP_OK;
}

// shift != nullptr: rhs is the UNBALANCED divergence and shift[b] its mean over the active cells (device doubles): the initial residual
// kernel subtracts it on the fly and writes the balanced right-hand side back into `rhs` (marching path only)
template <typename T>
static int cg_t(phihip_ctx* ctx, const GridView& v, const uint8_t* flags, int mask_batch, const void* rhs, void* x,
                const phihip_solve* solve, phihip_solve_info* info, const double* shift, hipStream_t s) {
    if (!v.op_custom && ctx->small_cg && v.cells <= small_cg_limit(ctx, v)) {      // (the single-kernel solver knows the pressure operator only)
        if (shift) { set_error("cg: the single-kernel solver takes a balanced right-hand side"); return PHIHIP_ERR_BAD_ARG; }
        return cg_small_path(ctx, v, flags, mask_batch, rhs, x, solve, info, s);
    }
    // 2-D fp32 grids up to the resident solver's cell limit: ONE launch for the whole solve (cg_resident.hip), r6 by default and under capture too (the solve
    // number of its tags lives on the device: a replay gets a fresh one). A launch the runtime refuses (cooperative launch too large) takes the forms below.
    if (!v.op_custom && ctx->resident_cg > 0 && ctx->small_cg && !std::is_same<T, double>::value && cg_resident_applicable(ctx, v, flags, solve) &&
        (ctx->resident_cg == 2 || ((v.batch >= 2 || v.n[2] <= 256) && (long long)v.cells * v.batch <= ctx->resident_cg_cells && cg_resident_batch_pays(ctx, v, flags, solve)))) {
        // (mode 1: batches, and single entries with rows of <= 256 cells = one vector per thread -- 128^2 ... 256^2: 7.1-7.2 -> 6.2-6.6 us per iteration, a tolerance solve
        // -15 ... -20 % without the host's polling; a single entry with two vectors per thread LOSES: 320^2 ... 512^2 7.3-7.9 -> 8.0-8.3 us, profiles/r06_sweep_resident_single.txt)
        const int st = cg_resident_path(ctx, v, flags, mask_batch, rhs, x, solve, info, shift, s);
        if (st != PHIHIP_ERR_UNSUPPORTED) return st;
    }
    // (phihip_set_small_grid_solver(ctx, 0) = "the two-launch marching kernels at every size": it also switches the automatic choice off)
    if (solve->method == PHIHIP_METHOD_CG && !v.halo[0] && !v.halo[1] &&
        (ctx->cg1_mode == 2 || (ctx->cg1_mode == 1 && ctx->small_cg && (long long)v.cells * v.batch <= cg1_threshold(ctx, v))))
        return cg1_t<T>(ctx, v, flags, mask_batch, rhs, x, solve, info, shift, s);
    const size_t vec_bytes = (size_t)v.batch * v.cells * sizeof(T);
    const bool ws_grew = ctx->ws_r.bytes < vec_bytes || ctx->ws_d0.bytes < vec_bytes || ctx->ws_d1.bytes < vec_bytes;
    if (ctx->autotune && !v.halo[0] && !v.halo[1] && ctx->tuning[FAM_MATVEC].rows == 0 && ctx->tuning[FAM_MATVEC].chunk == 0 &&
        !ctx->tuned.count(plan_key(v, mask_batch, flags != nullptr, FAM_UPDATE_R)) && !stream_is_capturing(s)) {
        const size_t vb = (size_t)v.batch * v.cells * sizeof(T);
        PHIHIP_TRY(ensure_buffer(ctx->ws_r, vb));
        PHIHIP_TRY(ensure_buffer(ctx->ws_d0, vb));
        PHIHIP_TRY(ensure_buffer(ctx->ws_d1, vb));
        PHIHIP_TRY(ensure_buffer(ctx->ws_part, 5 * (size_t)v.batch * 8192 * sizeof(double)));
        PHIHIP_TRY(autotune_cg<T>(ctx, v, flags, mask_batch, (const T*)rhs, (T*)ctx->ws_r.ptr, (T*)ctx->ws_d0.ptr, (T*)ctx->ws_d1.ptr, (T*)x, (double*)ctx->ws_part.ptr, s));
    }
    MarchConfig c, c_mv, c_up, c_ur;   // residual / MATVEC / UPDATE / UPDATE_R may run different tile shapes
    MarchGrid g, g_mv, g_up, g_ur;
    PHIHIP_TRY(plan_march(ctx, v, mask_batch, flags != nullptr, FAM_APPLY, &c, &g));
    PHIHIP_TRY(plan_march(ctx, v, mask_batch, flags != nullptr, FAM_MATVEC, &c_mv, &g_mv));
    PHIHIP_TRY(plan_march(ctx, v, mask_batch, flags != nullptr, FAM_UPDATE, &c_up, &g_up));
    PHIHIP_TRY(plan_march(ctx, v, mask_batch, flags != nullptr, FAM_UPDATE_R, &c_ur, &g_ur));
    int nblk_max = g.nblk > g_mv.nblk ? (g.nblk > g_up.nblk ? g.nblk : g_up.nblk) : (g_mv.nblk > g_up.nblk ? g_mv.nblk : g_up.nblk);
    nblk_max = nblk_max > g_ur.nblk ? nblk_max : g_ur.nblk;
    const size_t part_n = (size_t)v.batch * nblk_max;
    PHIHIP_TRY(ensure_buffer(ctx->ws_r, vec_bytes));
    PHIHIP_TRY(ensure_buffer(ctx->ws_d0, vec_bytes));
    PHIHIP_TRY(ensure_buffer(ctx->ws_d1, vec_bytes));
    // a workspace that has just been (re)allocated for this grid: choose between candidate allocations with the launch plans this solve runs -- tuned by the
    // autotune above, pinned by the caller or analytic (see place_workspace; its own switch: phihip_workspace_placement, not the autotune's)
    if (ws_grew && !v.halo[0] && !v.halo[1] && !stream_is_capturing(s) && ctx->ws_r.bytes == vec_bytes && ctx->ws_d0.bytes == vec_bytes && ctx->ws_d1.bytes == vec_bytes)
        PHIHIP_TRY(place_workspace<T>(ctx, v, flags, mask_batch, (const T*)rhs, (T*)x, s));
    PHIHIP_TRY(ensure_buffer(ctx->ws_part, 5 * part_n * sizeof(double)));
    PHIHIP_TRY(ensure_buffer(ctx->ws_state, (size_t)4 * v.batch * sizeof(CgState)));
    if (ctx->host_state_bytes < (size_t)2 * v.batch * sizeof(CgState)) {
        if (ctx->host_state) (void)hipHostFree(ctx->host_state);
        ctx->host_state = nullptr;
        ctx->host_state_bytes = 0;
        PHIHIP_CHECK_HIP(hipHostMalloc(&ctx->host_state, (size_t)2 * v.batch * sizeof(CgState), hipHostMallocDefault));
        ctx->host_state_bytes = (size_t)2 * v.batch * sizeof(CgState);
    }
    if (!ctx->poll_ev[0]) {
        PHIHIP_CHECK_HIP(hipEventCreate(&ctx->poll_ev[0]));
        PHIHIP_CHECK_HIP(hipEventCreate(&ctx->poll_ev[1]));
    }
    if (ctx->host_flags_count < (size_t)v.batch) {
        if (ctx->host_flags) (void)hipHostFree(ctx->host_flags);
        ctx->host_flags = nullptr;
        ctx->host_flags_count = 0;
        PHIHIP_CHECK_HIP(hipHostMalloc((void**)&ctx->host_flags, (size_t)v.batch * sizeof(unsigned long long), hipHostMallocMapped));
        memset(ctx->host_flags, 0, (size_t)v.batch * sizeof(unsigned long long));
        PHIHIP_CHECK_HIP(hipHostGetDevicePointer((void**)&ctx->host_flags_dev, ctx->host_flags, 0));
        ctx->host_flags_count = (size_t)v.batch;
    }
    const unsigned int seq = ++ctx->solve_seq;   // 0 never matches: flags of earlier solves read as "still running"
    int checks = 0;
    T* r = (T*)ctx->ws_r.ptr;
    T* d[2] = {(T*)ctx->ws_d0.ptr, (T*)ctx->ws_d1.ptr};
    double* part_rr = (double*)ctx->ws_part.ptr;
    double* part_dq = part_rr + part_n;
    double* part_yy = part_dq + part_n;
    double* part_rq = part_yy + part_n;   // 'CG-adaptive' only: sum r_new . A d (UPDATE) and sum d . r (MATVEC)
    double* part_dr = part_rq + part_n;
    const bool ad = solve->method == PHIHIP_METHOD_CG_ADAPTIVE;
    const int mode_mv = ad ? MODE_MATVEC_AD : MODE_MATVEC, mode_up = ad ? MODE_UPDATE_AD : MODE_UPDATE;
    const int pro_alpha = ad ? PRO_ALPHA_AD : PRO_ALPHA, pro_beta = ad ? PRO_BETA_AD : PRO_BETA;
    CgState* st[2] = {(CgState*)ctx->ws_state.ptr, (CgState*)ctx->ws_state.ptr + v.batch};
    int cur = 0;   // slot holding the most recent control block
    const bool has_flags = flags != nullptr;
    CgParams prm;
    prm.rtol = solve->rel_tol; prm.atol = solve->abs_tol; prm.max_iter = solve->max_iterations; prm.pad = 0;

    MarchArgs<T> base;
    memset(&base, 0, sizeof(base));
    base.flags = flags;
    base.prm = prm;
    set_operator(base, v);

    // ---- r0 = y - A x0 ; d0 = r0 (the first MATVEC runs with beta = 0 and reads r in place of d_old: no zeroed buffer needed) ----
    {
        MarchArgs<T> a = base;
        a.a = (const T*)x; a.b = (const T*)rhs; a.o1 = r;
        a.part1 = part_rr; a.part2 = part_yy;
        a.prologue = PRO_NONE;
        a.shift = shift;
        a.yout = shift ? (T*)const_cast<void*>(rhs) : nullptr;
        LaunchScope ls(ctx, PHIHIP_K_CG_RESIDUAL, s);
        PHIHIP_TRY(launch_march_any<T>(v, c, shift ? MODE_RESID_BAL : MODE_RESID, has_flags, g, a, s));
    }
    bool first = true;
    // x does not enter the recurrence: update it every other iteration only (UPDATE_R / UPDATE_X2, stencil_march.hpp)
    const bool defer = ctx->defer_x && !ad;
    bool pending = false;
    int nblk_rr = g.nblk;   // workgroup count of the kernel that last wrote part_rr (RESID or UPDATE)
    const int axpy_blocks = (int)((v.cells + kBlock - 1) / kBlock < 2048 ? (v.cells + kBlock - 1) / kBlock : 2048);
    CgState* hst = (CgState*)ctx->host_state;
    for (int k = 1; k <= solve->max_iterations; ++k) {
        T* d_old = d[(k - 1) & 1];
        T* d_new = d[k & 1];
        {
            MarchArgs<T> a = base;
            a.a = r; a.b = first ? r : d_old; a.o1 = d_new; a.part1 = part_dq; a.part2 = part_dr;
            if (solve->check_every > 0) { a.host_flags = ctx->host_flags_dev; a.seq = seq; }
            a.prologue = first ? PRO_FIRST : pro_beta;
            a.st_in = st[cur]; a.st_out = st[cur ^ 1]; a.pin1 = part_rr; a.pin2 = (first || !ad) ? part_yy : part_rq; a.nblk_in = nblk_rr;
            LaunchScope ls(ctx, PHIHIP_K_CG_MATVEC_DOT, s);
            PHIHIP_TRY(launch_march_any<T>(v, c_mv, mode_mv, has_flags, g_mv, a, s));
            cur ^= 1;
            first = false;
        }
        if (solve->refresh_every > 0 && k % solve->refresh_every == 0) {
            {
                LaunchScope ls(ctx, PHIHIP_K_OTHER, s);
                hipLaunchKernelGGL(cg_axpy_x<T>, dim3(axpy_blocks, v.batch), dim3(kBlock), 0, s, (T*)x, (const T*)d_new,
                                   (const T*)(pending ? r : nullptr), pro_alpha, (const CgState*)st[cur], st[cur ^ 1], (const double*)part_dq,
                                   (const double*)part_dr, g_mv.nblk, prm, v.cells);
                cur ^= 1;
                pending = false;
            }
            if (ad) {   // q = A d is not kept by the fused kernels: store it once (into the free d buffer) for sum r_new . q below
                MarchArgs<T> a = base;
                a.a = d_new; a.o1 = d_old;
                a.prologue = PRO_NONE;
                LaunchScope ls(ctx, PHIHIP_K_OTHER, s);
                PHIHIP_TRY(launch_march_any<T>(v, c, MODE_APPLY, has_flags, g, a, s));
            }
            MarchArgs<T> a = base;
            a.a = (const T*)x; a.b = (const T*)rhs; a.o1 = r;
            a.part1 = part_rr; a.part2 = part_yy;   // sum y^2 is not needed again: part_yy is scratch from now on
            a.prologue = PRO_CONT;
            a.st_in = st[cur];
            LaunchScope ls(ctx, PHIHIP_K_CG_RESIDUAL, s);
            PHIHIP_TRY(launch_march_any<T>(v, c, MODE_RESID, has_flags, g, a, s));
            nblk_rr = g.nblk;
            if (ad) {
                LaunchScope ls2(ctx, PHIHIP_K_OTHER, s);
                hipLaunchKernelGGL(dot_partials_kernel<T>, dim3(g.nblk, v.batch), dim3(kBlock), 0, s, (const T*)r, (const T*)d_old, v.cells, part_rq);
            }
        } else {
            MarchArgs<T> a = base;
            a.a = d_new; a.o1 = (T*)x; a.o2 = r; a.part1 = part_rr; a.part2 = part_rq;
            a.prologue = pro_alpha;
            a.st_in = st[cur]; a.st_out = st[cur ^ 1]; a.pin1 = part_dq; a.pin2 = part_dr; a.nblk_in = g_mv.nblk;
            a.pend_buf = k & 1;
            const int mode = !defer ? mode_up : (pending ? MODE_UPDATE_X2 : MODE_UPDATE_R);
            if (defer) pending = !pending;
            const bool r_only = mode == MODE_UPDATE_R;
            LaunchScope ls(ctx, r_only ? PHIHIP_K_CG_UPDATE_R : PHIHIP_K_CG_UPDATE, s);
            PHIHIP_TRY(launch_march_any<T>(v, r_only ? c_ur : c_up, mode, has_flags, r_only ? g_ur : g_up, a, s));
            cur ^= 1;
            nblk_rr = r_only ? g_ur.nblk : g_up.nblk;
        }
        if (solve->check_every > 0 && k < solve->max_iterations) {
            // Tolerance mode. Every MATVEC prologue publishes its continue decision into host-mapped memory (publish_flag), so the host
            // looks before each enqueue -- no peek kernel, no copy -- and stops as soon as every entry reports "done" for THIS solve
            // (sequence number). Flags only go 1 -> 0 and frozen entries make every kernel return at once, so the launches that
            // were enqueued ahead are harmless; an event every `check_every` iterations bounds that run-ahead to two intervals.
            bool any = false;
            for (int b = 0; b < v.batch && !any; ++b) {
                const unsigned long long f = *(volatile unsigned long long*)(ctx->host_flags + b);
                any = (unsigned int)(f >> 32) != seq || (f & 1ull);
            }
            if (!any) break;
            if (k % solve->check_every == 0) {
                const int slot = checks & 1;
                PHIHIP_CHECK_HIP(hipEventRecord(ctx->poll_ev[slot], s));
                if (checks > 0) PHIHIP_CHECK_HIP(hipEventSynchronize(ctx->poll_ev[slot ^ 1]));
                ++checks;
            }
        }
    }
    {   // fold the last reduction into the control block (or build it when no iteration ran)
        LaunchScope ls(ctx, PHIHIP_K_CG_SCALAR, s);
        hipLaunchKernelGGL(cg_state_kernel, dim3(v.batch), dim3(kBlock), 0, s, (int)(first ? PRO_FIRST : pro_beta), (const CgState*)st[cur],
                           st[cur ^ 1], (const double*)part_rr, (const double*)((first || !ad) ? part_yy : part_rq), nblk_rr, prm);
        cur ^= 1;
    }
    if (defer) {
        LaunchScope ls(ctx, PHIHIP_K_OTHER, s);
        hipLaunchKernelGGL(cg_flush_x<T>, dim3(axpy_blocks, v.batch), dim3(kBlock), 0, s, (T*)x, (const T*)d[0], (const T*)d[1],
                           (const CgState*)st[cur], v.cells);
    }
    ctx->last_state = st[cur];
    ctx->last_state_batch = v.batch;
    PHIHIP_CHECK_HIP(hipGetLastError());
    if (info) {
        PHIHIP_CHECK_HIP(hipMemcpyAsync(hst, st[cur], (size_t)v.batch * sizeof(CgState), hipMemcpyDeviceToHost, s));
        PHIHIP_CHECK_HIP(hipStreamSynchronize(s));
        for (int b = 0; b < v.batch; ++b) {
            info[b].residual_sq = hst[b].rsq;
            info[b].rhs_sq = hst[b].rhs_sq;
            info[b].iterations = hst[b].iterations;
            info[b].converged = hst[b].converged;
            info[b].diverged = hst[b].diverged;
            info[b].reserved = 0;
        }
    }
    return PHIHIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Slab-decomposed CG: ONE simulation split along a0 over several ranks. Each rank runs the same kernels on its
// slab; planes across a slab boundary come from halo buffers (NB_HALO) that the host layer fills by exchanging boundary planes
// with the neighbouring ranks, and every dot product is completed by an all-reduce between the phases: the kernels write
// per-workgroup partials, `sum_partials_kernel` folds them into one double per batch entry (the all-reduce operand), and the
// next kernel's prologue reads the GLOBAL sum as a single "partial" (nblk_in = 1). The control block logic is unchanged.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void sum_partials_kernel(const double* part, int nblk, double* out) {
    __shared__ double red[kBlock / kWave];
    const double s = reduce_partials(part + (long long)blockIdx.x * nblk, nblk, red);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

struct SlabSetup {
    MarchConfig c;
    MarchGrid g;
    double* part1;
    double* part2;
    CgState* st[2];
};

static int slab_setup(phihip_ctx* ctx, const GridView& v, int mask_batch, bool flags, int family, SlabSetup* out) {
    PHIHIP_TRY(plan_march(ctx, v, mask_batch, flags, family, &out->c, &out->g));
    for (int side = 0; side < 2; ++side)
        if (v.halo[side]) out->g.nb[0][side] = NB_HALO;
    const size_t part_n = (size_t)v.batch * out->g.nblk;
    PHIHIP_TRY(ensure_buffer(ctx->ws_part, 3 * ((part_n + 8191) / 8192 * 8192 + 8192) * sizeof(double)));
    PHIHIP_TRY(ensure_buffer(ctx->ws_state, (size_t)3 * v.batch * sizeof(CgState)));
    out->part1 = (double*)ctx->ws_part.ptr;
    out->part2 = out->part1 + part_n;
    out->st[0] = (CgState*)ctx->ws_state.ptr;
    out->st[1] = out->st[0] + v.batch;
    return PHIHIP_OK;
}

template <typename T>
static MarchArgs<T> slab_args(const GridView& v, const uint8_t* flags, const phihip_solve* solve) {
    MarchArgs<T> a;
    memset(&a, 0, sizeof(a));
    a.flags = flags;
    if (solve) { a.prm.rtol = solve->rel_tol; a.prm.atol = solve->abs_tol; a.prm.max_iter = solve->max_iterations; }
    set_operator(a, v);
    return a;
}

// r = rhs - A x ; sums[0..batch) = local sum r^2, sums[batch..2 batch) = local sum rhs^2.  keep_going != 0: skip frozen entries
// (true-residual refresh inside the loop; the control block of the running solve decides)
template <typename T>
static int slab_residual_t(phihip_ctx* ctx, const GridView& v, const uint8_t* flags, int mask_batch, const void* x, const void* x_lo,
                           const void* x_hi, const void* rhs, void* r, double* sums, int keep_going, hipStream_t s) {
    SlabSetup su;
    PHIHIP_TRY(slab_setup(ctx, v, mask_batch, flags != nullptr, FAM_APPLY, &su));
    MarchArgs<T> a = slab_args<T>(v, flags, nullptr);
    a.a = (const T*)x; a.b = (const T*)rhs; a.o1 = (T*)r;
    a.a_lo = (const T*)x_lo; a.a_hi = (const T*)x_hi;
    a.part1 = su.part1; a.part2 = su.part2;
    a.prologue = keep_going ? PRO_CONT : PRO_NONE;
    a.st_in = su.st[ctx->slab_cur];
    {
        LaunchScope ls(ctx, PHIHIP_K_CG_RESIDUAL, s);
        PHIHIP_TRY(launch_march_any<T>(v, su.c, MODE_RESID, flags != nullptr, su.g, a, s));
    }
    hipLaunchKernelGGL(sum_partials_kernel, dim3(v.batch), dim3(kBlock), 0, s, (const double*)su.part1, su.g.nblk, sums);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(v.batch), dim3(kBlock), 0, s, (const double*)su.part2, su.g.nblk, sums + v.batch);
    if (!keep_going) ctx->slab_cur = 0;
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

// d_new = r + beta d_old (beta from the GLOBAL sums_in) ; sum_out = local d_new . A d_new
template <typename T>
static int slab_matvec_t(phihip_ctx* ctx, const GridView& v, const uint8_t* flags, int mask_batch, int first, const double* sums_in,
                         const void* r, const void* r_lo, const void* r_hi, const void* d_old, const void* d_lo, const void* d_hi,
                         void* d_new, double* sum_out, const phihip_solve* solve, hipStream_t s) {
    SlabSetup su;
    PHIHIP_TRY(slab_setup(ctx, v, mask_batch, flags != nullptr, FAM_MATVEC, &su));
    MarchArgs<T> a = slab_args<T>(v, flags, solve);
    a.a = (const T*)r; a.b = (const T*)d_old; a.o1 = (T*)d_new;
    a.a_lo = (const T*)r_lo; a.a_hi = (const T*)r_hi; a.b_lo = (const T*)d_lo; a.b_hi = (const T*)d_hi;
    a.part1 = su.part1;
    a.prologue = first ? PRO_FIRST : PRO_BETA;
    a.pin1 = sums_in; a.pin2 = sums_in + v.batch; a.nblk_in = 1;
    a.st_in = su.st[ctx->slab_cur]; a.st_out = su.st[ctx->slab_cur ^ 1];
    {
        LaunchScope ls(ctx, PHIHIP_K_CG_MATVEC_DOT, s);
        PHIHIP_TRY(launch_march_any<T>(v, su.c, MODE_MATVEC, flags != nullptr, su.g, a, s));
    }
    ctx->slab_cur ^= 1;
    hipLaunchKernelGGL(sum_partials_kernel, dim3(v.batch), dim3(kBlock), 0, s, (const double*)su.part1, su.g.nblk, sum_out);
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

// x += alpha d ; r -= alpha A d (alpha from the GLOBAL sum_in = d . A d) ; sum_out = local sum r^2.   x_only: the true-residual
// refresh iteration (PhiML every 50th): only x is advanced, the caller recomputes r = rhs - A x afterwards
template <typename T>
static int slab_update_t(phihip_ctx* ctx, const GridView& v, const uint8_t* flags, int mask_batch, const double* sum_in, const void* d,
                         const void* d_lo, const void* d_hi, void* x, void* r, double* sum_out, int x_only, const phihip_solve* solve,
                         hipStream_t s) {
    SlabSetup su;
    PHIHIP_TRY(slab_setup(ctx, v, mask_batch, flags != nullptr, FAM_UPDATE, &su));
    CgParams prm;
    prm.rtol = solve->rel_tol; prm.atol = solve->abs_tol; prm.max_iter = solve->max_iterations; prm.pad = 0;
    if (x_only) {
        const int axpy_blocks = (int)((v.cells + kBlock - 1) / kBlock < 2048 ? (v.cells + kBlock - 1) / kBlock : 2048);
        LaunchScope ls(ctx, PHIHIP_K_OTHER, s);
        hipLaunchKernelGGL(cg_axpy_x<T>, dim3(axpy_blocks, v.batch), dim3(kBlock), 0, s, (T*)x, (const T*)d, (const T*)nullptr, (int)PRO_ALPHA,
                           (const CgState*)su.st[ctx->slab_cur], su.st[ctx->slab_cur ^ 1], sum_in, (const double*)nullptr, 1, prm, v.cells);
        ctx->slab_cur ^= 1;
        PHIHIP_CHECK_HIP(hipGetLastError());
        return PHIHIP_OK;
    }
    MarchArgs<T> a = slab_args<T>(v, flags, solve);
    a.a = (const T*)d; a.o1 = (T*)x; a.o2 = (T*)r;
    a.a_lo = (const T*)d_lo; a.a_hi = (const T*)d_hi;
    a.part1 = su.part1;
    a.prologue = PRO_ALPHA;
    a.pin1 = sum_in; a.nblk_in = 1;
    a.st_in = su.st[ctx->slab_cur]; a.st_out = su.st[ctx->slab_cur ^ 1];
    {
        LaunchScope ls(ctx, PHIHIP_K_CG_UPDATE, s);
        PHIHIP_TRY(launch_march_any<T>(v, su.c, MODE_UPDATE, flags != nullptr, su.g, a, s));
    }
    ctx->slab_cur ^= 1;
    hipLaunchKernelGGL(sum_partials_kernel, dim3(v.batch), dim3(kBlock), 0, s, (const double*)su.part1, su.g.nblk, sum_out);
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

int run_slab_residual(phihip_ctx* ctx, const GridView& v, const uint8_t* flags, int mask_batch, const void* x, const void* x_lo, const void* x_hi,
                      const void* rhs, void* r, double* sums, int keep_going, hipStream_t s) {
    return v.dtype == PHIHIP_F64 ? slab_residual_t<double>(ctx, v, flags, mask_batch, x, x_lo, x_hi, rhs, r, sums, keep_going, s)
                                 : slab_residual_t<float>(ctx, v, flags, mask_batch, x, x_lo, x_hi, rhs, r, sums, keep_going, s);
}

int run_slab_matvec(phihip_ctx* ctx, const GridView& v, const uint8_t* flags, int mask_batch, int first, const double* sums_in, const void* r,
                    const void* r_lo, const void* r_hi, const void* d_old, const void* d_lo, const void* d_hi, void* d_new, double* sum_out,
                    const phihip_solve* solve, hipStream_t s) {
    return v.dtype == PHIHIP_F64
               ? slab_matvec_t<double>(ctx, v, flags, mask_batch, first, sums_in, r, r_lo, r_hi, d_old, d_lo, d_hi, d_new, sum_out, solve, s)
               : slab_matvec_t<float>(ctx, v, flags, mask_batch, first, sums_in, r, r_lo, r_hi, d_old, d_lo, d_hi, d_new, sum_out, solve, s);
}

int run_slab_update(phihip_ctx* ctx, const GridView& v, const uint8_t* flags, int mask_batch, const double* sum_in, const void* d, const void* d_lo,
                    const void* d_hi, void* x, void* r, double* sum_out, int x_only, const phihip_solve* solve, hipStream_t s) {
    return v.dtype == PHIHIP_F64 ? slab_update_t<double>(ctx, v, flags, mask_batch, sum_in, d, d_lo, d_hi, x, r, sum_out, x_only, solve, s)
                                 : slab_update_t<float>(ctx, v, flags, mask_batch, sum_in, d, d_lo, d_hi, x, r, sum_out, x_only, solve, s);
}

// folds the last GLOBAL residual sum into the control block and reports it (synchronises the stream)
int run_slab_finish(phihip_ctx* ctx, const GridView& v, int first, const double* sums_in, const phihip_solve* solve, phihip_solve_info* info,
                    int peek, hipStream_t s) {
    PHIHIP_TRY(ensure_buffer(ctx->ws_state, (size_t)3 * v.batch * sizeof(CgState)));
    CgState* st[3] = {(CgState*)ctx->ws_state.ptr, (CgState*)ctx->ws_state.ptr + v.batch, (CgState*)ctx->ws_state.ptr + 2 * v.batch};
    CgParams prm;
    prm.rtol = solve->rel_tol; prm.atol = solve->abs_tol; prm.max_iter = solve->max_iterations; prm.pad = 0;
    // peek: the decision the next MATVEC prologue will take, written to a third slot so that the chain does not advance
    const int dst = peek ? 2 : (ctx->slab_cur ^ 1);
    hipLaunchKernelGGL(cg_state_kernel, dim3(v.batch), dim3(kBlock), 0, s, (int)(first ? PRO_FIRST : PRO_BETA), (const CgState*)st[ctx->slab_cur],
                       st[dst], sums_in, sums_in + v.batch, 1, prm);
    if (!peek) {
        ctx->slab_cur ^= 1;
        ctx->last_state = st[ctx->slab_cur];
        ctx->last_state_batch = v.batch;
    }
    if (ctx->host_state_bytes < (size_t)v.batch * sizeof(CgState)) {
        if (ctx->host_state) (void)hipHostFree(ctx->host_state);
        ctx->host_state = nullptr;
        ctx->host_state_bytes = 0;
        PHIHIP_CHECK_HIP(hipHostMalloc(&ctx->host_state, (size_t)v.batch * sizeof(CgState), hipHostMallocDefault));
        ctx->host_state_bytes = (size_t)v.batch * sizeof(CgState);
    }
    CgState* hst = (CgState*)ctx->host_state;
    PHIHIP_CHECK_HIP(hipMemcpyAsync(hst, st[peek ? 2 : ctx->slab_cur], (size_t)v.batch * sizeof(CgState), hipMemcpyDeviceToHost, s));
    PHIHIP_CHECK_HIP(hipStreamSynchronize(s));
    if (info)
        for (int b = 0; b < v.batch; ++b) {
            info[b].residual_sq = hst[b].rsq;
            info[b].rhs_sq = hst[b].rhs_sq;
            info[b].iterations = hst[b].iterations;
            info[b].converged = hst[b].converged;
            info[b].diverged = hst[b].diverged;
            info[b].reserved = hst[b].cont;   // 1: the entry would keep iterating (used by the host loop to stop early)
        }
    return PHIHIP_OK;
}

int run_cg(phihip_ctx* ctx, const GridView& v, const uint8_t* flags, int mask_batch, const void* rhs, void* x,
           const phihip_solve* solve, phihip_solve_info* info, hipStream_t s) {
    return v.dtype == PHIHIP_F64 ? cg_t<double>(ctx, v, flags, mask_batch, rhs, x, solve, info, nullptr, s)
                                 : cg_t<float>(ctx, v, flags, mask_batch, rhs, x, solve, info, nullptr, s);
}

bool cg_uses_marching(const phihip_ctx* ctx, const GridView& v) { return v.op_custom || !(ctx->small_cg && v.cells <= small_cg_limit(ctx, v)); }

int run_cg_balancing(phihip_ctx* ctx, const GridView& v, const uint8_t* flags, int mask_batch, void* rhs, void* x, const phihip_solve* solve,
                     phihip_solve_info* info, const double* shift, hipStream_t s) {
    return v.dtype == PHIHIP_F64 ? cg_t<double>(ctx, v, flags, mask_batch, rhs, x, solve, info, shift, s)
                                 : cg_t<float>(ctx, v, flags, mask_batch, rhs, x, solve, info, shift, s);
}

}  // namespace phihip
