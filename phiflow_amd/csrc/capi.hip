// capi.hip -- extern "C" surface of libphihip.so (declared in include/phihip.h) + context / workspace / profiling plumbing.
#include <dlfcn.h>
#include <stdarg.h>
#include <stdlib.h>

#include <mutex>

#include "common.hpp"
#include "march_dispatch.hpp"

namespace phihip {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int make_view(const phihip_grid* grid, GridView* out) {
    PHIHIP_REQUIRE(grid != nullptr, "grid is NULL");
    PHIHIP_REQUIRE(grid->rank == 2 || grid->rank == 3, "grid.rank must be 2 or 3 (got %d)", grid->rank);
    PHIHIP_REQUIRE(grid->dtype == PHIHIP_F32 || grid->dtype == PHIHIP_F64, "grid.dtype must be PHIHIP_F32 or PHIHIP_F64");
    PHIHIP_REQUIRE(grid->batch >= 1, "grid.batch must be >= 1");
    PHIHIP_REQUIRE(grid->batch <= 65535, "grid.batch must be <= 65535 (the batch index is the y dimension of the launch grids; split larger batches)");
    GridView v;
    memset(&v, 0, sizeof(v));
    v.rank = grid->rank;
    v.ax0 = 3 - grid->rank;
    v.dtype = grid->dtype;
    v.batch = grid->batch;
    for (int ax = 0; ax < 3; ++ax) {
        v.n[ax] = 1;
        v.bc[ax][0] = v.bc[ax][1] = PHIHIP_BC_PERIODIC;
        v.lower[ax] = 0;
        v.dx[ax] = 1;
    }
    for (int d = 0; d < grid->rank; ++d) {
        const int ax = d + v.ax0;
        PHIHIP_REQUIRE(grid->res[d] >= 1, "grid.res[%d] must be >= 1", d);
        PHIHIP_REQUIRE(grid->upper[d] > grid->lower[d], "grid bounds must have positive size along axis %d", d);
        v.n[ax] = grid->res[d];
        v.lower[ax] = grid->lower[d];
        v.dx[ax] = (grid->upper[d] - grid->lower[d]) / grid->res[d];
        for (int s = 0; s < 2; ++s) {
            const int code = grid->bc[d][s];
            PHIHIP_REQUIRE(code >= PHIHIP_BC_PERIODIC && code <= PHIHIP_BC_OPEN, "grid.bc[%d][%d] invalid", d, s);
            v.bc[ax][s] = code;
            for (int c = 0; c < grid->rank; ++c) v.bcv[ax][s][c + v.ax0] = grid->bc_val[d][s][c];
        }
        PHIHIP_REQUIRE((v.bc[ax][0] == PHIHIP_BC_PERIODIC) == (v.bc[ax][1] == PHIHIP_BC_PERIODIC),
                       "axis %d: PERIODIC must be set on both sides", d);
    }
    v.cells = (long long)v.n[0] * v.n[1] * v.n[2];
    for (int ca = 0; ca < 3; ++ca) {
        const bool lo_valid = v.bc[ca][0] != PHIHIP_BC_CLOSED;
        const bool hi_valid = v.bc[ca][1] == PHIHIP_BC_OPEN;
        v.off[ca] = lo_valid ? 0 : 1;
        for (int ax = 0; ax < 3; ++ax) v.cn[ca][ax] = v.n[ax];
        if (ca >= v.ax0) v.cn[ca][ca] = v.n[ca] + (int)lo_valid + (int)hi_valid - 1;
        v.ccells[ca] = (long long)v.cn[ca][0] * v.cn[ca][1] * v.cn[ca][2];
        if (ca >= v.ax0) PHIHIP_REQUIRE(v.cn[ca][ca] >= 1, "axis %d has no stored faces (resolution too small)", ca - v.ax0);
    }
    *out = v;
    return PHIHIP_OK;
}

VelGrid make_velgrid(const GridView& v) {
    VelGrid g;
    memset(&g, 0, sizeof(g));
    for (int a = 0; a < 3; ++a) {
        g.n[a] = v.n[a];
        g.off[a] = v.off[a];
        g.dx[a] = v.dx[a];
        g.rdx[a] = 1.0 / v.dx[a];
        g.ccells[a] = v.ccells[a];
        for (int s = 0; s < 2; ++s) {
            g.bc[a][s] = v.bc[a][s];
            for (int c = 0; c < 3; ++c) g.bcv[a][s][c] = v.bcv[a][s][c];
        }
        for (int c = 0; c < 3; ++c) g.cn[a][c] = v.cn[a][c];
    }
    g.ax0 = v.ax0;
    g.cells = v.cells;
    return g;
}

ScalarBc make_scalar_bc(const GridView& v, const int32_t s_bc[3][2], const double s_val[3][2]) {
    ScalarBc sb;
    memset(&sb, 0, sizeof(sb));
    for (int d = 0; d < v.rank; ++d)
        for (int side = 0; side < 2; ++side) {
            sb.bc[d + v.ax0][side] = s_bc[d][side];
            sb.val[d + v.ax0][side] = s_val ? s_val[d][side] : 0.0;
        }
    return sb;
}

static int ensure_adv_host(phihip_ctx* ctx) {
    if (ctx->adv_host) return PHIHIP_OK;
    void* h = nullptr;
    PHIHIP_CHECK_HIP(hipHostMalloc(&h, phihip_ctx::kAdvHostInts * sizeof(int), hipHostMallocMapped));
    memset(h, 0, phihip_ctx::kAdvHostInts * sizeof(int));
    void* d = nullptr;
    PHIHIP_CHECK_HIP(hipHostGetDevicePointer(&d, h, 0));
    ctx->adv_host = (int*)h;
    ctx->adv_host_dev = (int*)d;
    return PHIHIP_OK;
}

// Fallback fractions (share of the (tile, plane) units whose lookups left the window) at which the reach changes. r5, measured with the smoke256 workload at
// three stages of its plume with the reach FIXED (profiles/r05_smoke256_reach.jsonl; narrow + fix-up list / wide / gather): self-advection 0.24 / 0.27 / 0.46 ms at
// 5.6 % of the units, 0.32 / 0.31 / 0.45 at 15 %, 0.39 / 0.34 / 0.47 at 19 % -- the cross-over to the wide window is near 12 %, not at the 2 % of round 4 (the
// work list made the fix-up cheap: a flagged unit costs one gather workgroup, a wide window costs every workgroup a third of its occupancy); the centred kinds
// (MacCormack smoke: 0.29 / 0.34, 0.34 / 0.39, 0.36 / 0.40) are faster narrow at every fraction seen.
constexpr double kAdvWideAtSelf = 0.12, kAdvWideAtCentred = 0.30, kAdvGatherAtNarrow = 0.15, kAdvGatherAtWide = 0.25;

int adv_choose(phihip_ctx* ctx, int kind, bool has_wide, long long grid_fp, hipStream_t s) {
    phihip_ctx::AdvKindState& K = ctx->adv_policy[kind];
    const bool capturing = stream_is_capturing(s);
    // the policy of THIS grid (SlabFluid's whole-slab and window passes, several simulations on one context: each keeps its own fallback history);
    // an unknown grid takes a free entry or the least recently used one and starts from the narrow reach
    K.clock += 1;
    int slot = -1, lru = 0;
    for (int i = 0; i < phihip_ctx::kAdvGrids; ++i) {
        if (K.e[i].fp == grid_fp) { slot = i; break; }
        if (K.e[i].used < K.e[lru].used) lru = i;
    }
    if (slot < 0) {
        slot = lru;
        K.e[slot] = phihip_ctx::AdvPolicy{K.e[slot].ev, grid_fp};
    }
    K.cur = slot;
    phihip_ctx::AdvPolicy& P = K.e[slot];
    P.used = K.clock;
    // ONE observation at a time, resolved at a FIXED distance: the pass that recorded the event published its count into its own slot; exactly kAdvMaxLag
    // passes later the host waits for that event -- it is then several steps old: the wait is free unless the host runs that far ahead of the device, and
    // then the device still has that many passes queued -- and reads that slot. Nothing here depends on WHEN the host looks (r5, first version: hipEventQuery
    // and "the newest completed pass's count" -- non-blocking, but the reach of a pass, and with it the last bits of its result, depended on host timing;
    // before that, r4: a wait in every pass, which serialised host and device).
    constexpr int kAdvMaxLag = 2;      // (the decision of pass k + 3 uses pass k: a host that only enqueues stays at most three passes ahead of the device)
    if (!capturing) {
        // an observation whose slot passes of OTHER grids have reused since (the ring has kAdvSlots entries per kind) says nothing any more: dropped, by the count
        // of passes alone -- never by timing
        if (P.pending && K.seq + 1 - P.obs_seq >= (unsigned)phihip_ctx::kAdvSlots) P.pending = false;
        K.seq += 1;                                                        // this pass
        if (ctx->adv_host) ctx->adv_host[phihip_ctx::kAdvSlotBase + kind * phihip_ctx::kAdvSlots + (K.seq % phihip_ctx::kAdvSlots)] = 0;   // (its slot's last user was resolved or dropped)
    }
    bool resolved = false;
    if (P.pending && !capturing && P.age >= kAdvMaxLag) {
        SlowTrace tr("adv_choose: hipEventSynchronize on the observed pass");
        PHIHIP_CHECK_HIP(hipEventSynchronize(P.ev));
        resolved = true;
    }
    if (resolved) {
        P.pending = false;
        const int word = ctx->adv_host[phihip_ctx::kAdvSlotBase + kind * phihip_ctx::kAdvSlots + (P.obs_seq % phihip_ctx::kAdvSlots)];
        const int seen = (word >> 28) & 3;          // the reach the pass ran with (0: it published nothing)
        const double frac = P.units > 0 ? (double)(word & ((1 << 28) - 1)) / (double)P.units : 0.0;
        const double wide_at = kind == AK_SL_SELF ? kAdvWideAtSelf : kAdvWideAtCentred;
        if (seen == 1) P.mode = frac > (has_wide ? wide_at : kAdvGatherAtNarrow) ? (has_wide ? 2 : 0) : 1;
        else if (seen == 2) P.mode = frac > kAdvGatherAtWide ? 0 : 2;
    }
    P.calls += 1;
    if (!capturing && P.calls % 64 == 0) {                  // is the cheaper form good enough again?
        if (P.mode == 2) return 1;
        if (P.mode == 0) return has_wide ? 2 : 1;
    }
    return P.mode;
}

int ensure_adv_host_public(phihip_ctx* ctx) { return ensure_adv_host(ctx); }

int adv_record(phihip_ctx* ctx, int kind, int reach, hipStream_t s) {
    if (stream_is_capturing(s)) return PHIHIP_OK;           // (an event record would become a node of the graph; captured passes keep their reach)
    phihip_ctx::AdvKindState& K = ctx->adv_policy[kind];
    phihip_ctx::AdvPolicy& P = K.e[K.cur];
    // one observation at a time: while the event of an earlier pass is unresolved (a host that runs steps ahead of the device) it is NOT re-recorded --
    // re-recording every pass kept the event forever in the future and the policy never adapted in an enqueue-only loop (found with the smoke256
    // workload: the switch to the wide reach came 50 steps late). The published count is the newest completed pass's: same reach, fresher data.
    if (P.pending) { P.age += 1; return PHIHIP_OK; }
    if (reach == 0) return PHIHIP_OK;                       // a gather pass publishes nothing: no observation (the probe every 64 calls is the next one)
    P.age = 0;
    P.obs_seq = K.seq;
    if (!P.ev) PHIHIP_CHECK_HIP(hipEventCreateWithFlags(&P.ev, hipEventDisableTiming));
    PHIHIP_CHECK_HIP(hipEventRecord(P.ev, s));
    P.pending = true;
    P.last = reach;
    P.units = ctx->adv_last_nblk;
    return PHIHIP_OK;
}

int prepare_fixlist(phihip_ctx* ctx, long long units, hipStream_t s, FixList* list, void** dump, int kind) {
    if (units >= (1LL << 28)) {
        set_error("advect: more than 2^28 (tile, plane) units in one launch");
        return PHIHIP_ERR_UNSUPPORTED;
    }
    const size_t bytes = 128 + (size_t)units * sizeof(FixItem);
    // (a reallocation is recognised by the SIZE: an allocator may well hand the freed address out again, with its own bookkeeping in the
    // first bytes -- glibc does under the emulation -- and a stale count would send the fix-up launch over `cap` garbage entries)
    const size_t had = ctx->ws_adv_flags.ptr ? ctx->ws_adv_flags.bytes : 0;
    SlowTrace tr("prepare_fixlist: ensure_buffer");
    PHIHIP_TRY(ensure_buffer(ctx->ws_adv_flags, bytes));
    tr.done();
    if (ctx->ws_adv_flags.bytes != had || !ctx->adv_ctl_clear) {        // a fresh buffer: the control block starts at zero (the fix-up launch keeps it so)
        PHIHIP_CHECK_HIP(hipMemsetAsync(ctx->ws_adv_flags.ptr, 0, 64, s));
        ctx->adv_ctl_clear = true;
    }
    if (stream_is_capturing(s)) {
        // a captured launch is replayed with the kernel arguments of the capture: the alternation below would leave a stale count when a
        // graph holds an odd number of such launches. Captured launches get their own counter, cleared by a memset node in front of them.
        list->count = (int*)ctx->ws_adv_flags.ptr + 3;
        list->next = (int*)ctx->ws_adv_flags.ptr + 4;
        PHIHIP_CHECK_HIP(hipMemsetAsync(list->count, 0, sizeof(int), s));
        ctx->adv_seq_captured = true;
    } else {
        ctx->adv_seq += 1;
        ctx->adv_seq_captured = false;
        list->count = (int*)ctx->ws_adv_flags.ptr + (ctx->adv_seq & 1u);
        list->next = (int*)ctx->ws_adv_flags.ptr + ((ctx->adv_seq + 1u) & 1u);
    }
    list->items = (FixItem*)((char*)ctx->ws_adv_flags.ptr + 128);
    list->publish = nullptr;
    if (kind != AK_NONE && ctx->adv_halo < 0) {
        PHIHIP_TRY(ensure_adv_host(ctx));
        // every eager pass has its own slot (adv_choose reads the one of the pass it observes); captured passes, whose reach is fixed, share one per kind
        list->publish = ctx->adv_host_dev + (stream_is_capturing(s) ? phihip_ctx::kAdvCaptureBase + kind
                                                                    : phihip_ctx::kAdvSlotBase + kind * phihip_ctx::kAdvSlots + (int)(ctx->adv_policy[kind].seq % phihip_ctx::kAdvSlots));
    }
    list->cap = (int)units;
    list->reach_tag = (ctx->adv_reach_now & 3) << 28;       // (units < 2^28, checked above)
    *dump = (char*)ctx->ws_adv_flags.ptr + 64;
    ctx->adv_last_nblk = (int)units;
    return PHIHIP_OK;
}

int ensure_buffer(DeviceBuffer& buf, size_t bytes) {
    if (buf.bytes >= bytes && buf.ptr) return PHIHIP_OK;
    if (buf.ptr) {
        PHIHIP_CHECK_HIP(hipFree(buf.ptr));
        buf.ptr = nullptr;
        buf.bytes = 0;
    }
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(&buf.ptr, bytes);
    if (e != hipSuccess) {
        buf.ptr = nullptr;
        set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        return PHIHIP_ERR_ALLOC;
    }
    buf.bytes = bytes;
    return PHIHIP_OK;
}

int profile_begin(phihip_ctx* ctx, int kid, hipStream_t s, int* slot) {
    if (ctx->ev_used == ctx->ev_pool.size()) {
        phihip_ctx::EventPair p;
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return PHIHIP_ERR_HIP;
        ctx->ev_pool.push_back(p);
    }
    *slot = (int)ctx->ev_used++;
    ctx->ev_pool[*slot].kid = kid;
    (void)hipEventRecord(ctx->ev_pool[*slot].a, s);
    return PHIHIP_OK;
}

int profile_end(phihip_ctx* ctx, int slot, hipStream_t s) {
    (void)hipEventRecord(ctx->ev_pool[slot].b, s);
    return PHIHIP_OK;
}

int profile_collect(phihip_ctx* ctx) {
    for (size_t i = 0; i < ctx->ev_used; ++i) {
        auto& p = ctx->ev_pool[i];
        PHIHIP_CHECK_HIP(hipEventSynchronize(p.b));
        float ms = 0;
        PHIHIP_CHECK_HIP(hipEventElapsedTime(&ms, p.a, p.b));
        ctx->prof_launches[p.kid] += 1;
        ctx->prof_ms[p.kid] += ms;
    }
    ctx->ev_used = 0;
    return PHIHIP_OK;
}

static void remap3(const GridView& v, const void* const in[3], const void* out[3]) {
    out[0] = out[1] = out[2] = nullptr;
    for (int d = 0; d < v.rank; ++d) out[d + v.ax0] = in ? in[d] : nullptr;
}
static void remap3w(const GridView& v, void* const in[3], void* out[3]) {
    out[0] = out[1] = out[2] = nullptr;
    for (int d = 0; d < v.rank; ++d) out[d + v.ax0] = in ? in[d] : nullptr;
}

// the 16-byte vector paths of the marching kernels need 16-byte aligned fields (and 4-byte aligned flags): anything else -> scalar path
static void note_align(GridView& v, const void* p, unsigned mask = 15u) {
    if (p && ((uintptr_t)p & mask)) v.unaligned = true;
}

static int check_ptrs(const GridView& v, const void* const p[3], const char* what) {
    PHIHIP_REQUIRE(p != nullptr, "%s is NULL", what);
    for (int d = 0; d < v.rank; ++d) PHIHIP_REQUIRE(p[d] != nullptr, "%s[%d] is NULL", what, d);
    return PHIHIP_OK;
}

}  // namespace phihip

using namespace phihip;

extern "C" {

int phihip_version(void) { return PHIHIP_VERSION; }

#ifndef PHIHIP_BUILD_ID
#define PHIHIP_BUILD_ID "unknown src:unknown"
#endif
const char* phihip_build_id(void) { return PHIHIP_BUILD_ID; }

const char* phihip_last_error(void) { return g_err; }

int phihip_ctx_create(int device, phihip_ctx** out) {
    PHIHIP_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error("no HIP device available (%s); libphihip has no CPU fallback", e != hipSuccess ? hipGetErrorString(e) : "count = 0");
        return PHIHIP_ERR_NO_DEVICE;
    }
    PHIHIP_REQUIRE(device >= 0 && device < count, "device %d out of range (have %d)", device, count);
    PHIHIP_CHECK_HIP(hipSetDevice(device));
    phihip_ctx* ctx = new phihip_ctx();
    ctx->device = device;
    const char* at = getenv("PHIHIP_AUTOTUNE");
    if (at && at[0] == '0') ctx->autotune = false;
    const char* wsc = getenv("PHIHIP_WS_CANDIDATES");    // candidate allocations of the CG workspace (cg.hip place_workspace); 0 / 1: the first allocation is kept
    if (wsc && wsc[0]) { const int k = atoi(wsc); ctx->ws_candidates = k < 1 ? 1 : (k > 16 ? 16 : k); }
    const char* wsm = getenv("PHIHIP_WS_PLACE_MIN_BYTES");
    if (wsm && wsm[0]) ctx->ws_place_min_bytes = (size_t)atoll(wsm);
    const char* rc = getenv("PHIHIP_RESIDENT_CG");       // 0 / 1 / 2: phihip_set_resident_cg mode at creation (r6 default 1)
    if (rc && rc[0] >= '0' && rc[0] <= '2') ctx->resident_cg = rc[0] - '0';
    const char* coop = getenv("PHIHIP_RESIDENT_COOP");   // 0: the resident solver as a plain launch (A/B of the cooperative launch's cost)
    if (coop && (coop[0] == '0' || coop[0] == '1')) ctx->res_coop = coop[0] - '0';
    const char* coopc = getenv("PHIHIP_RESIDENT_COOP_CAPTURE");
    if (coopc && (coopc[0] == '0' || coopc[0] == '1')) ctx->res_coop_capture = coopc[0] - '0';
    const char* dma = getenv("PHIHIP_ADVECT_DMA");      // A/B switch of the LDS-DMA fill of the tiled self-advection (advect_tile.hip)
    if (dma && (dma[0] == '0' || dma[0] == '1')) ctx->adv_dma = dma[0] - '0';
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ctx->num_cu = prop.multiProcessorCount;
    *out = ctx;
    return PHIHIP_OK;
}

int phihip_ctx_destroy(phihip_ctx* ctx) {
    if (!ctx) return PHIHIP_OK;
    (void)hipSetDevice(ctx->device);
    DeviceBuffer* bufs[] = {&ctx->ws_r, &ctx->ws_d0, &ctx->ws_d1, &ctx->ws_div, &ctx->ws_part, &ctx->ws_state, &ctx->ws_scalars, &ctx->ws_rhs, &ctx->ws_adv, &ctx->ws_adv_flags, &ctx->ws_adj_q, &ctx->ws_adj_l, &ctx->ws_cg1, &ctx->ws_adj_g, &ctx->ws_res, &ctx->ws_adv_const};
    for (DeviceBuffer* b : bufs)
        if (b->ptr) (void)hipFree(b->ptr);
    if (ctx->host_state) (void)hipHostFree(ctx->host_state);
    if (ctx->host_flags) (void)hipHostFree(ctx->host_flags);
    if (ctx->adv_host) (void)hipHostFree(ctx->adv_host);
    for (auto& K : ctx->adv_policy)
        for (auto& P : K.e)
            if (P.ev) (void)hipEventDestroy(P.ev);
    for (int i = 0; i < 2; ++i)
        if (ctx->poll_ev[i]) (void)hipEventDestroy(ctx->poll_ev[i]);
    for (auto& p : ctx->ev_pool) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    delete ctx;
    return PHIHIP_OK;
}

int phihip_workspace_bytes(const phihip_ctx* ctx, size_t* bytes) {
    PHIHIP_REQUIRE(ctx && bytes, "ctx / bytes is NULL");
    *bytes = ctx->ws_r.bytes + ctx->ws_d0.bytes + ctx->ws_d1.bytes + ctx->ws_div.bytes + ctx->ws_part.bytes + ctx->ws_state.bytes +
             ctx->ws_scalars.bytes + ctx->ws_rhs.bytes + ctx->ws_adv.bytes + ctx->ws_adv_flags.bytes + ctx->ws_adj_q.bytes + ctx->ws_adj_l.bytes + ctx->ws_cg1.bytes + ctx->ws_adj_g.bytes + ctx->ws_res.bytes;
    return PHIHIP_OK;
}

int phihip_component_shape(const phihip_grid* grid, int comp, int32_t shape[3]) {
    GridView v;
    PHIHIP_TRY(make_view(grid, &v));
    PHIHIP_REQUIRE(comp >= 0 && comp < v.rank, "component %d out of range", comp);
    PHIHIP_REQUIRE(shape != nullptr, "shape is NULL");
    shape[0] = shape[1] = shape[2] = 1;
    for (int d = 0; d < v.rank; ++d) shape[d] = v.cn[comp + v.ax0][d + v.ax0];
    return PHIHIP_OK;
}

#define PHIHIP_ENTER(ctx, grid)                        \
    PHIHIP_REQUIRE((ctx) != nullptr, "ctx is NULL");   \
    GridView v;                                        \
    PHIHIP_TRY(make_view((grid), &v));                 \
    PHIHIP_CHECK_HIP(hipSetDevice((ctx)->device));     \
    hipStream_t s = (hipStream_t)stream;

int phihip_advect_staggered(phihip_ctx* ctx, const phihip_grid* grid, const void* const field[3], const void* const velocity[3],
                            void* const out[3], double dt, void* stream) {
    SlowTrace whole("phihip_advect_staggered (whole call)");
    PHIHIP_ENTER(ctx, grid);
    PHIHIP_TRY(check_ptrs(v, field, "field"));
    PHIHIP_TRY(check_ptrs(v, velocity, "velocity"));
    PHIHIP_TRY(check_ptrs(v, (const void* const*)out, "out"));
    for (int d = 0; d < v.rank; ++d)
        PHIHIP_REQUIRE(out[d] != field[d] && out[d] != velocity[d], "advect: out[%d] must not alias an input", d);
    const void *f[3], *u[3];
    void* o[3];
    remap3(v, field, f);
    remap3(v, velocity, u);
    remap3w(v, out, o);
    return run_advect_staggered(ctx, v, f, u, o, dt, s);
}

static int check_scalar_bc(const GridView& v, const int32_t s_bc[3][2], const char* what);

int phihip_advect_centered(phihip_ctx* ctx, const phihip_grid* grid, const void* sfield, const int32_t s_bc[3][2],
                           const double s_val[3][2], const void* const velocity[3], void* out, double dt, void* stream) {
    PHIHIP_ENTER(ctx, grid);
    PHIHIP_REQUIRE(sfield && out && s_bc, "advect_centered: NULL argument");
    PHIHIP_REQUIRE(sfield != out, "advect_centered: out must not alias the input");
    PHIHIP_TRY(check_ptrs(v, velocity, "velocity"));
    PHIHIP_TRY(check_scalar_bc(v, s_bc, "advect_centered"));
    const void* u[3];
    remap3(v, velocity, u);
    return run_advect_centered(ctx, v, sfield, s_bc, s_val, u, out, dt, s);
}

// The sampler describes `values` with a phihip_grid (shape, extrapolation) but has no staggered layout: build the view from a copy
// whose non-periodic sides are OPEN (no face bookkeeping, e.g. a one-sample axis with a constant extrapolation is fine) and hand the
// real rule over separately.
static int sampler_view(const phihip_grid* grid, phihip_grid* plain, int32_t s_bc[3][2], double s_val[3][2]) {
    PHIHIP_REQUIRE(grid != nullptr, "grid is NULL");
    *plain = *grid;
    for (int d = 0; d < 3; ++d)
        for (int side = 0; side < 2; ++side) {
            s_bc[d][side] = grid->bc[d][side];
            s_val[d][side] = grid->bc_val[d][side][0];
            if (d < grid->rank) {
                PHIHIP_REQUIRE(s_bc[d][side] >= PHIHIP_BC_PERIODIC && s_bc[d][side] <= PHIHIP_BC_OPEN, "grid.bc[%d][%d] invalid", d, side);
                if (s_bc[d][side] != PHIHIP_BC_PERIODIC) plain->bc[d][side] = PHIHIP_BC_OPEN;
            }
            plain->lower[d] = 0.0;
            plain->upper[d] = 1.0;
        }
    return PHIHIP_OK;
}

int phihip_grid_sample(phihip_ctx* ctx, const phihip_grid* grid_in, const void* values, int values_batch, const void* const coords[3],
                       int64_t points, void* out, void* out_min, void* out_max, void* stream) {
    phihip_grid plain;
    int32_t s_bc[3][2];
    double s_val[3][2];
    PHIHIP_TRY(sampler_view(grid_in, &plain, s_bc, s_val));
    const phihip_grid* grid = &plain;
    PHIHIP_ENTER(ctx, grid);
    PHIHIP_REQUIRE(values != nullptr && points >= 0, "grid_sample: values is NULL or points < 0");
    PHIHIP_REQUIRE(values_batch == 1 || values_batch == v.batch, "grid_sample: values_batch must be 1 or grid.batch");
    PHIHIP_REQUIRE((out || out_min) && ((out_min == nullptr) == (out_max == nullptr)), "grid_sample: pass out and / or (out_min and out_max)");
    PHIHIP_TRY(check_ptrs(v, coords, "coords"));
    const void* c[3];
    remap3(v, coords, c);
    return run_grid_sample(ctx, v, s_bc, s_val, values, values_batch, c, (long long)points, out, out_min, out_max, s);
}

int phihip_grid_sample_backward(phihip_ctx* ctx, const phihip_grid* grid_in, const void* values, int values_batch, const void* const coords[3],
                                int64_t points, const void* grad_out, void* grad_values, void* const grad_coords[3], void* stream) {
    phihip_grid plain;
    int32_t s_bc[3][2];
    double s_val[3][2];
    PHIHIP_TRY(sampler_view(grid_in, &plain, s_bc, s_val));
    const phihip_grid* grid = &plain;
    PHIHIP_ENTER(ctx, grid);
    PHIHIP_REQUIRE(values && grad_out && points >= 0, "grid_sample_backward: NULL argument or points < 0");
    PHIHIP_REQUIRE(values_batch == 1 || values_batch == v.batch, "grid_sample_backward: values_batch must be 1 or grid.batch");
    PHIHIP_TRY(check_ptrs(v, coords, "coords"));
    if (grad_coords) PHIHIP_TRY(check_ptrs(v, (const void* const*)grad_coords, "grad_coords"));
    const void* c[3];
    void* gc[3];
    remap3(v, coords, c);
    remap3w(v, grad_coords, gc);
    return run_grid_sample_bwd(ctx, v, s_bc, s_val, values, values_batch, c, (long long)points, grad_out, grad_values, grad_coords ? gc : nullptr, s);
}

static int check_scalar_bc(const GridView& v, const int32_t s_bc[3][2], const char* what) {
    for (int d = 0; d < v.rank; ++d) {
        for (int side = 0; side < 2; ++side)
            PHIHIP_REQUIRE(s_bc[d][side] >= PHIHIP_BC_PERIODIC && s_bc[d][side] <= PHIHIP_BC_OPEN, "%s: s_bc[%d][%d] invalid", what, d, side);
        PHIHIP_REQUIRE((s_bc[d][0] == PHIHIP_BC_PERIODIC) == (s_bc[d][1] == PHIHIP_BC_PERIODIC) &&
                           (s_bc[d][0] == PHIHIP_BC_PERIODIC) == (v.bc[d + v.ax0][0] == PHIHIP_BC_PERIODIC),
                       "%s: periodicity of the scalar must match the grid along axis %d", what, d);
    }
    return PHIHIP_OK;
}

int phihip_mac_cormack_staggered(phihip_ctx* ctx, const phihip_grid* grid, const void* const field[3], const void* const velocity[3],
                                 void* const out[3], double dt, double correction_strength, void* stream) {
    PHIHIP_ENTER(ctx, grid);
    PHIHIP_TRY(check_ptrs(v, field, "field"));
    PHIHIP_TRY(check_ptrs(v, velocity, "velocity"));
    PHIHIP_TRY(check_ptrs(v, (const void* const*)out, "out"));
    for (int d = 0; d < v.rank; ++d)
        PHIHIP_REQUIRE(out[d] != field[d] && out[d] != velocity[d], "mac_cormack: out[%d] must not alias an input", d);
    const void *f[3], *u[3];
    void* o[3];
    remap3(v, field, f);
    remap3(v, velocity, u);
    remap3w(v, out, o);
    return run_mac_cormack_staggered(ctx, v, f, u, o, dt, correction_strength, s);
}

int phihip_mac_cormack_centered(phihip_ctx* ctx, const phihip_grid* grid, const void* sfield, const int32_t s_bc[3][2],
                                const double s_val[3][2], const void* const velocity[3], void* out, double dt,
                                double correction_strength, void* stream) {
    PHIHIP_ENTER(ctx, grid);
    PHIHIP_REQUIRE(sfield && out && s_bc, "mac_cormack_centered: NULL argument");
    PHIHIP_REQUIRE(sfield != out, "mac_cormack_centered: out must not alias the input");
    PHIHIP_TRY(check_ptrs(v, velocity, "velocity"));
    PHIHIP_TRY(check_scalar_bc(v, s_bc, "mac_cormack_centered"));
    const void* u[3];
    remap3(v, velocity, u);
    return run_mac_cormack_centered(ctx, v, sfield, s_bc, s_val, u, out, dt, correction_strength, s);
}

int phihip_centered_to_staggered(phihip_ctx* ctx, const phihip_grid* grid, const void* sfield, const int32_t s_bc[3][2],
                                 const double s_val[3][2], const double vector[3], int accumulate, void* const out[3], void* stream) {
    PHIHIP_ENTER(ctx, grid);
    PHIHIP_REQUIRE(sfield && s_bc && vector, "centered_to_staggered: NULL argument");
    PHIHIP_TRY(check_ptrs(v, (const void* const*)out, "out"));
    PHIHIP_TRY(check_scalar_bc(v, s_bc, "centered_to_staggered"));
    double vec[3] = {0, 0, 0};
    for (int d = 0; d < v.rank; ++d) vec[d + v.ax0] = vector[d];
    void* o[3];
    remap3w(v, out, o);
    return run_centered_to_staggered(ctx, v, sfield, s_bc, s_val, vec, accumulate, o, s);
}

static_assert(sizeof(phihip_obstacle) == 4 * 4 + 8 * 21, "phihip_obstacle layout is part of the ABI (phiflow_amd/_capi.py mirrors it)");
static_assert(sizeof(phihip_solve) == 32 && sizeof(phihip_solve_info) == 32, "ABI structs");

static int check_obstacles(const GridView& v, const phihip_obstacle* obstacles, int count) {
    PHIHIP_REQUIRE(count >= 0, "obstacle count must be >= 0");
    PHIHIP_REQUIRE(count == 0 || obstacles != nullptr, "obstacles is NULL");
    for (int k = 0; k < count; ++k) {
        PHIHIP_REQUIRE(obstacles[k].kind == PHIHIP_OBSTACLE_BOX || obstacles[k].kind == PHIHIP_OBSTACLE_SPHERE, "obstacle %d: unknown kind %d", k,
                       obstacles[k].kind);
        const int nh = obstacles[k].kind == PHIHIP_OBSTACLE_SPHERE ? 1 : v.rank;
        for (int d = 0; d < nh; ++d) PHIHIP_REQUIRE(obstacles[k].half_size[d] >= 0, "obstacle %d: negative size", k);
        PHIHIP_REQUIRE(obstacles[k].group >= 0, "obstacle %d: group must be >= 0", k);
        PHIHIP_REQUIRE(obstacles[k].embed_mask >= 0 && obstacles[k].embed_mask < (1 << v.rank) - 1, "obstacle %d: embed_mask must leave one axis", k);
        if (obstacles[k].embed_mask != 0) {   // embedded geometries are not rotated (phi/geom/_embed.py:96-103)
            bool plain = true;
            for (int d = 0; d < 3; ++d) plain = plain && obstacles[k].angular_velocity[d] == 0.0;
            for (int a = 0; a < v.rank; ++a)
                for (int c = 0; c < v.rank; ++c) {
                    const double r = obstacles[k].rotation[a * 3 + c];
                    plain = plain && (r == 0.0 || (a == c && r == 1.0));
                }
            PHIHIP_REQUIRE(plain, "obstacle %d: an embedded geometry (embed_mask != 0) cannot rotate", k);
        }
        if (obstacles[k].group > 0) {   // member of a union: one rigid body without rotation
            for (int d = 0; d < 3; ++d)
                PHIHIP_REQUIRE(obstacles[k].angular_velocity[d] == 0.0, "obstacle %d: members of a union (group > 0) cannot rotate", k);
            if (k > 0 && obstacles[k - 1].group == obstacles[k].group)
                for (int d = 0; d < v.rank; ++d)
                    PHIHIP_REQUIRE(obstacles[k].velocity[d] == obstacles[k - 1].velocity[d], "obstacle %d: members of a union must share their velocity", k);
        }
    }
    return PHIHIP_OK;
}

int phihip_obstacle_accessible(phihip_ctx* ctx, const phihip_grid* grid, const phihip_obstacle* obstacles, int count, uint8_t* accessible,
                               void* stream) {
    PHIHIP_ENTER(ctx, grid);
    PHIHIP_REQUIRE(accessible != nullptr, "accessible is NULL");
    PHIHIP_TRY(check_obstacles(v, obstacles, count));
    return run_obstacle_accessible(ctx, v, obstacles, count, accessible, s);
}

int phihip_apply_obstacles(phihip_ctx* ctx, const phihip_grid* grid, const phihip_obstacle* obstacles, int count, void* const velocity[3],
                           void* stream) {
    PHIHIP_ENTER(ctx, grid);
    PHIHIP_TRY(check_ptrs(v, (const void* const*)velocity, "velocity"));
    PHIHIP_TRY(check_obstacles(v, obstacles, count));
    if (count == 0) return PHIHIP_OK;
    void* u[3];
    remap3w(v, velocity, u);
    return run_apply_obstacles(ctx, v, obstacles, count, u, s);
}

int phihip_build_cellflags(phihip_ctx* ctx, const phihip_grid* grid, const uint8_t* accessible, const uint8_t* active, int mask_batch,
                           uint8_t* flags, void* stream) {
    PHIHIP_ENTER(ctx, grid);
    PHIHIP_REQUIRE(flags != nullptr, "flags is NULL");
    PHIHIP_REQUIRE(mask_batch == 1 || mask_batch == v.batch, "mask_batch must be 1 or grid.batch");
    return run_build_cellflags(ctx, v, accessible, active, mask_batch, flags, s);
}

int phihip_divergence(phihip_ctx* ctx, const phihip_grid* grid, const void* const velocity[3], const uint8_t* flags, int mask_batch,
                      int balance, void* div, void* stream) {
    PHIHIP_ENTER(ctx, grid);
    PHIHIP_TRY(check_ptrs(v, velocity, "velocity"));
    PHIHIP_REQUIRE(div != nullptr, "div is NULL");
    PHIHIP_REQUIRE(mask_batch == 1 || mask_batch == v.batch, "mask_batch must be 1 or grid.batch");
    const void* u[3];
    remap3(v, velocity, u);
    // public bit set only: 2 is the library's internal "leave the shift in ws_scalars" mode of run_divergence
    return run_divergence(ctx, v, u, flags, mask_batch, balance & (PHIHIP_DIV_BALANCE | PHIHIP_DIV_FINITE_GUARD), div, s);
}

int phihip_laplace_apply(phihip_ctx* ctx, const phihip_grid* grid, const uint8_t* flags, int mask_batch, const void* p, void* out,
                         void* stream) {
    PHIHIP_ENTER(ctx, grid);
    PHIHIP_REQUIRE(p && out && p != out, "laplace_apply: p / out NULL or aliased");
    PHIHIP_REQUIRE(mask_batch == 1 || mask_batch == v.batch, "mask_batch must be 1 or grid.batch");
    note_align(v, p); note_align(v, out); note_align(v, flags, 3u);
    return run_laplace_apply(ctx, v, flags, mask_batch, p, out, s);
}

static int check_solve(const phihip_solve* solve) {
    PHIHIP_REQUIRE(solve != nullptr, "solve is NULL");
    PHIHIP_REQUIRE(solve->max_iterations >= 0, "solve.max_iterations must be >= 0");
    PHIHIP_REQUIRE(solve->rel_tol >= 0 && solve->abs_tol >= 0, "solve tolerances must be >= 0");
    PHIHIP_REQUIRE(solve->refresh_every >= 0 && solve->check_every >= 0, "solve.refresh_every / check_every must be >= 0");
    PHIHIP_REQUIRE(solve->method == PHIHIP_METHOD_CG || solve->method == PHIHIP_METHOD_CG_ADAPTIVE, "solve.method must be a phihip_method");
    return PHIHIP_OK;
}

// the slab phases implement 'CG' only
static int check_slab_solve(const phihip_solve* solve) {
    PHIHIP_TRY(check_solve(solve));
    if (solve->method != PHIHIP_METHOD_CG) {
        set_error("slab-decomposed solve: only PHIHIP_METHOD_CG is available");
        return PHIHIP_ERR_UNSUPPORTED;
    }
    return PHIHIP_OK;
}

int phihip_cg_solve(phihip_ctx* ctx, const phihip_grid* grid, const uint8_t* flags, int mask_batch, const void* rhs, void* x,
                    const phihip_solve* solve, phihip_solve_info* info, void* stream) {
    PHIHIP_ENTER(ctx, grid);
    PHIHIP_REQUIRE(rhs && x && rhs != x, "cg_solve: rhs / x NULL or aliased");
    PHIHIP_REQUIRE(mask_batch == 1 || mask_batch == v.batch, "mask_batch must be 1 or grid.batch");
    PHIHIP_TRY(check_solve(solve));
    note_align(v, rhs); note_align(v, x); note_align(v, flags, 3u);
    return run_cg(ctx, v, flags, mask_batch, rhs, x, solve, info, s);
}

int phihip_cg_solve_shifted(phihip_ctx* ctx, const phihip_grid* grid, double identity, double scale, const void* rhs, void* x,
                            const phihip_solve* solve, phihip_solve_info* info, void* stream) {
    PHIHIP_ENTER(ctx, grid);
    PHIHIP_REQUIRE(rhs && x && rhs != x, "cg_solve_shifted: rhs / x NULL or aliased");
    PHIHIP_REQUIRE(scale != 0.0, "cg_solve_shifted: scale must not be 0");
    PHIHIP_TRY(check_solve(solve));
    note_align(v, rhs); note_align(v, x);
    v.op_custom = 1;
    v.op_ident = identity;
    v.op_scale = scale;
    for (int ax = 0; ax < 3; ++ax)
        for (int side = 0; side < 2; ++side) {
            const int code = v.bc[ax][side];      // the pressure's rule for the velocity codes, as phihip_cg_solve applies it
            v.op_rule[ax][side] = code == PHIHIP_BC_PERIODIC ? 0 : (code == PHIHIP_BC_CLOSED ? 1 : 2);      // NB_WRAP / NB_CLAMP / NB_ZERO
        }
    return run_cg(ctx, v, nullptr, 1, rhs, x, solve, info, s);
}

// ---- f4: slab-decomposed CG phases -----------------------------------------------------------------------------------
static int slab_view(const phihip_grid* grid, int halo_lo, int halo_hi, GridView* v) {
    PHIHIP_TRY(make_view(grid, v));
    PHIHIP_REQUIRE(v->rank == 3, "slab decomposition is implemented for 3-D grids (slabs along x)");
    v->halo[0] = halo_lo != 0;
    v->halo[1] = halo_hi != 0;
    return PHIHIP_OK;
}

int phihip_slab_residual(phihip_ctx* ctx, const phihip_grid* grid, int halo_lo, int halo_hi, const uint8_t* flags, const void* x,
                         const void* x_lo, const void* x_hi, const void* rhs, void* r, double* sums, int keep_going, void* stream) {
    PHIHIP_REQUIRE(ctx != nullptr, "ctx is NULL");
    GridView v;
    PHIHIP_TRY(slab_view(grid, halo_lo, halo_hi, &v));
    PHIHIP_CHECK_HIP(hipSetDevice(ctx->device));
    PHIHIP_REQUIRE(x && rhs && r && sums, "slab_residual: NULL argument");
    PHIHIP_REQUIRE((!halo_lo || x_lo) && (!halo_hi || x_hi), "slab_residual: halo plane missing");
    note_align(v, x); note_align(v, x_lo); note_align(v, x_hi); note_align(v, rhs); note_align(v, r); note_align(v, flags, 3u);
    return run_slab_residual(ctx, v, flags, 1, x, x_lo, x_hi, rhs, r, sums, keep_going, (hipStream_t)stream);
}

int phihip_slab_matvec(phihip_ctx* ctx, const phihip_grid* grid, int halo_lo, int halo_hi, const uint8_t* flags, int first,
                       const double* sums_in, const void* r, const void* r_lo, const void* r_hi, const void* d_old, const void* d_lo,
                       const void* d_hi, void* d_new, double* sum_out, const phihip_solve* solve, void* stream) {
    PHIHIP_REQUIRE(ctx != nullptr, "ctx is NULL");
    GridView v;
    PHIHIP_TRY(slab_view(grid, halo_lo, halo_hi, &v));
    PHIHIP_CHECK_HIP(hipSetDevice(ctx->device));
    PHIHIP_REQUIRE(sums_in && r && d_old && d_new && sum_out && d_old != d_new, "slab_matvec: NULL or aliased argument");
    PHIHIP_REQUIRE((!halo_lo || (r_lo && d_lo)) && (!halo_hi || (r_hi && d_hi)), "slab_matvec: halo plane missing");
    PHIHIP_TRY(check_slab_solve(solve));
    note_align(v, r); note_align(v, r_lo); note_align(v, r_hi); note_align(v, d_old); note_align(v, d_lo); note_align(v, d_hi);
    note_align(v, d_new); note_align(v, flags, 3u);
    return run_slab_matvec(ctx, v, flags, 1, first, sums_in, r, r_lo, r_hi, d_old, d_lo, d_hi, d_new, sum_out, solve, (hipStream_t)stream);
}

int phihip_slab_update(phihip_ctx* ctx, const phihip_grid* grid, int halo_lo, int halo_hi, const uint8_t* flags, const double* sum_in,
                       const void* d, const void* d_lo, const void* d_hi, void* x, void* r, double* sum_out, int x_only,
                       const phihip_solve* solve, void* stream) {
    PHIHIP_REQUIRE(ctx != nullptr, "ctx is NULL");
    GridView v;
    PHIHIP_TRY(slab_view(grid, halo_lo, halo_hi, &v));
    PHIHIP_CHECK_HIP(hipSetDevice(ctx->device));
    PHIHIP_REQUIRE(sum_in && d && x && (x_only || (r && sum_out)), "slab_update: NULL argument");
    PHIHIP_REQUIRE(x_only || ((!halo_lo || d_lo) && (!halo_hi || d_hi)), "slab_update: halo plane missing");
    PHIHIP_TRY(check_slab_solve(solve));
    note_align(v, d); note_align(v, d_lo); note_align(v, d_hi); note_align(v, x); note_align(v, r); note_align(v, flags, 3u);
    return run_slab_update(ctx, v, flags, 1, sum_in, d, d_lo, d_hi, x, r, sum_out, x_only, solve, (hipStream_t)stream);
}

int phihip_slab_state(phihip_ctx* ctx, const phihip_grid* grid, int first, const double* sums_in, const phihip_solve* solve,
                      phihip_solve_info* info, int peek, void* stream) {
    PHIHIP_REQUIRE(ctx != nullptr, "ctx is NULL");
    GridView v;
    PHIHIP_TRY(make_view(grid, &v));
    PHIHIP_CHECK_HIP(hipSetDevice(ctx->device));
    PHIHIP_REQUIRE(sums_in != nullptr, "slab_state: sums_in is NULL");
    PHIHIP_TRY(check_slab_solve(solve));
    return run_slab_finish(ctx, v, first, sums_in, solve, info, peek, (hipStream_t)stream);
}

int phihip_solve_residuals(phihip_ctx* ctx, int batch, double* out_device, void* stream) {
    PHIHIP_REQUIRE(ctx != nullptr && out_device != nullptr && batch >= 1, "solve_residuals: bad argument");
    PHIHIP_CHECK_HIP(hipSetDevice(ctx->device));
    return run_export_residuals(ctx, batch, out_device, (hipStream_t)stream);
}

int phihip_solve_relative_residual(phihip_ctx* ctx, int batch, double* out_device, void* stream) {
    PHIHIP_REQUIRE(ctx != nullptr && out_device != nullptr && batch >= 1, "solve_relative_residual: bad argument");
    PHIHIP_CHECK_HIP(hipSetDevice(ctx->device));
    return run_export_relative_residual(ctx, batch, out_device, (hipStream_t)stream);
}

// RCCL's ncclAllReduce(sendbuff, recvbuff, count, datatype, op, comm, stream); ncclDouble = 8 (rccl.h). Resolved lazily: the library
// that created the caller's communicator is already in the process (dlsym over the global scope finds it, e.g. torch's bundled
// librccl); only a process without one gets a fresh librccl.so.1.
typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
static nccl_allreduce_fn resolve_nccl_allreduce(std::string* why) {
    static nccl_allreduce_fn fn = nullptr;
    static std::string reason;
    static std::once_flag once;
    std::call_once(once, [] {
        fn = (nccl_allreduce_fn)dlsym(RTLD_DEFAULT, "ncclAllReduce");
        if (!fn) {
            void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (h) fn = (nccl_allreduce_fn)dlsym(h, "ncclAllReduce");
            if (!fn) {
                const char* e = dlerror();          // (one call: dlerror() clears the message it returns)
                reason = e ? e : "not found";
            }
        }
    });
    if (why) *why = reason;
    return fn;
}

int phihip_allreduce_residual(phihip_ctx* ctx, void* comm, double* values_device, int count, int op, void* stream) {
    PHIHIP_REQUIRE(ctx != nullptr && comm != nullptr && values_device != nullptr, "allreduce_residual: NULL argument");
    PHIHIP_REQUIRE(count > 0, "allreduce_residual: count must be > 0");
    PHIHIP_REQUIRE(op == 0 || op == 2, "allreduce_residual: op must be 0 (sum) or 2 (max)");
    std::string why;
    nccl_allreduce_fn fn = resolve_nccl_allreduce(&why);
    if (!fn) {
        set_error("allreduce_residual: no RCCL in this process and librccl.so.1 cannot be loaded (%s)", why.c_str());
        return PHIHIP_ERR_UNSUPPORTED;
    }
    PHIHIP_CHECK_HIP(hipSetDevice(ctx->device));
    const int rc = fn(values_device, values_device, (size_t)count, /*ncclDouble*/ 8, op, comm, (hipStream_t)stream);
    if (rc != 0) {
        set_error("ncclAllReduce failed with ncclResult_t %d", rc);
        return PHIHIP_ERR_HIP;
    }
    return PHIHIP_OK;
}

int phihip_grad_subtract(phihip_ctx* ctx, const phihip_grid* grid, const uint8_t* flags, int mask_batch, const void* p,
                         void* const velocity[3], void* stream) {
    PHIHIP_ENTER(ctx, grid);
    PHIHIP_REQUIRE(p != nullptr, "p is NULL");
    PHIHIP_TRY(check_ptrs(v, (const void* const*)velocity, "velocity"));
    PHIHIP_REQUIRE(mask_batch == 1 || mask_batch == v.batch, "mask_batch must be 1 or grid.batch");
    void* u[3];
    remap3w(v, velocity, u);
    return run_grad_subtract(ctx, v, flags, mask_batch, p, u, s);
}

int phihip_make_incompressible(phihip_ctx* ctx, const phihip_grid* grid, void* const velocity[3], const void* const soft_mask[3],
                               const uint8_t* flags, int mask_batch, int balance, void* pressure, void* div_out,
                               const phihip_solve* solve, phihip_solve_info* info, void* stream) {
    PHIHIP_ENTER(ctx, grid);
    PHIHIP_TRY(check_ptrs(v, (const void* const*)velocity, "velocity"));
    PHIHIP_REQUIRE(pressure != nullptr, "pressure is NULL");
    PHIHIP_REQUIRE(mask_batch == 1 || mask_batch == v.batch, "mask_batch must be 1 or grid.batch");
    PHIHIP_TRY(check_solve(solve));
    void* u[3];
    remap3w(v, velocity, u);
    if (soft_mask) {
        PHIHIP_TRY(check_ptrs(v, soft_mask, "soft_mask"));
        const void* m[3];
        remap3(v, soft_mask, m);
        PHIHIP_TRY(run_scale_faces(ctx, v, u, m, s));
    }
    note_align(v, pressure); note_align(v, div_out); note_align(v, flags, 3u);
    void* div = div_out;
    if (!div) {
        const size_t bytes = (size_t)v.batch * v.cells * (v.dtype == PHIHIP_F64 ? 8 : 4);
        PHIHIP_TRY(ensure_buffer(ctx->ws_rhs, bytes));
        div = ctx->ws_rhs.ptr;
    }
    const void* cu[3] = {u[0], u[1], u[2]};
    const int guard = balance & PHIHIP_DIV_FINITE_GUARD;
    balance = (balance & ~PHIHIP_DIV_FINITE_GUARD) ? PHIHIP_DIV_BALANCE : 0;
    if (balance && cg_uses_marching(ctx, v) && !v.unaligned) {
        // divergence + partial sums in one pass; the mean is subtracted inside the solver's initial residual (no extra pass over div)
        PHIHIP_TRY(run_divergence(ctx, v, cu, flags, mask_batch, 2 | guard, div, s));
        PHIHIP_TRY(run_cg_balancing(ctx, v, flags, mask_batch, div, pressure, solve, info, (const double*)ctx->ws_scalars.ptr, s));
    } else {
        PHIHIP_TRY(run_divergence(ctx, v, cu, flags, mask_batch, balance | guard, div, s));
        PHIHIP_TRY(run_cg(ctx, v, flags, mask_batch, div, pressure, solve, info, s));
    }
    PHIHIP_TRY(run_grad_subtract(ctx, v, flags, mask_batch, pressure, u, s));
    return PHIHIP_OK;
}

int phihip_advect_staggered_backward(phihip_ctx* ctx, const phihip_grid* grid, const void* const field[3], const void* const velocity[3],
                                     const void* const grad_out[3], double dt, void* const grad_field[3], void* const grad_velocity[3],
                                     void* stream) {
    PHIHIP_ENTER(ctx, grid);
    PHIHIP_TRY(check_ptrs(v, field, "field"));
    PHIHIP_TRY(check_ptrs(v, velocity, "velocity"));
    PHIHIP_TRY(check_ptrs(v, grad_out, "grad_out"));
    if (grad_field) PHIHIP_TRY(check_ptrs(v, (const void* const*)grad_field, "grad_field"));
    if (grad_velocity) PHIHIP_TRY(check_ptrs(v, (const void* const*)grad_velocity, "grad_velocity"));
    const void *f[3], *u[3], *go[3];
    void *gf[3], *gu[3];
    remap3(v, field, f);
    remap3(v, velocity, u);
    remap3(v, grad_out, go);
    remap3w(v, grad_field, gf);
    remap3w(v, grad_velocity, gu);
    return run_advect_staggered_bwd(ctx, v, f, u, go, grad_field ? gf : nullptr, grad_velocity ? gu : nullptr, dt, s);
}

int phihip_advect_centered_backward(phihip_ctx* ctx, const phihip_grid* grid, const void* sfield, const int32_t s_bc[3][2],
                                    const double s_val[3][2], const void* const velocity[3], const void* grad_out, double dt, void* grad_s,
                                    void* const grad_velocity[3], void* stream) {
    PHIHIP_ENTER(ctx, grid);
    PHIHIP_REQUIRE(sfield && grad_out && s_bc, "advect_centered_backward: NULL argument");
    PHIHIP_TRY(check_ptrs(v, velocity, "velocity"));
    PHIHIP_TRY(check_scalar_bc(v, s_bc, "advect_centered_backward"));
    if (grad_velocity) PHIHIP_TRY(check_ptrs(v, (const void* const*)grad_velocity, "grad_velocity"));
    const void* u[3];
    void* gu[3];
    remap3(v, velocity, u);
    remap3w(v, grad_velocity, gu);
    return run_advect_centered_bwd(ctx, v, sfield, s_bc, s_val, u, grad_out, grad_s, grad_velocity ? gu : nullptr, dt, s);
}

int phihip_mac_cormack_staggered_backward(phihip_ctx* ctx, const phihip_grid* grid, const void* const field[3], const void* const velocity[3],
                                          const void* const grad_out[3], double dt, double correction_strength, void* const grad_field[3],
                                          void* const grad_velocity[3], void* stream) {
    PHIHIP_ENTER(ctx, grid);
    PHIHIP_TRY(check_ptrs(v, field, "field"));
    PHIHIP_TRY(check_ptrs(v, velocity, "velocity"));
    PHIHIP_TRY(check_ptrs(v, grad_out, "grad_out"));
    PHIHIP_TRY(check_ptrs(v, (const void* const*)grad_field, "grad_field"));
    if (grad_velocity) PHIHIP_TRY(check_ptrs(v, (const void* const*)grad_velocity, "grad_velocity"));
    const void *f[3], *u[3], *go[3];
    void *gf[3], *gu[3];
    remap3(v, field, f);
    remap3(v, velocity, u);
    remap3(v, grad_out, go);
    remap3w(v, grad_field, gf);
    remap3w(v, grad_velocity, gu);
    return run_mac_cormack_staggered_bwd(ctx, v, f, u, go, gf, grad_velocity ? gu : nullptr, dt, correction_strength, s);
}

int phihip_mac_cormack_centered_backward(phihip_ctx* ctx, const phihip_grid* grid, const void* sfield, const int32_t s_bc[3][2],
                                         const double s_val[3][2], const void* const velocity[3], const void* grad_out, double dt,
                                         double correction_strength, void* grad_s, void* const grad_velocity[3], void* stream) {
    PHIHIP_ENTER(ctx, grid);
    PHIHIP_REQUIRE(sfield && grad_out && s_bc && grad_s, "mac_cormack_centered_backward: NULL argument");
    PHIHIP_TRY(check_ptrs(v, velocity, "velocity"));
    PHIHIP_TRY(check_scalar_bc(v, s_bc, "mac_cormack_centered_backward"));
    if (grad_velocity) PHIHIP_TRY(check_ptrs(v, (const void* const*)grad_velocity, "grad_velocity"));
    const void* u[3];
    void* gu[3];
    remap3(v, velocity, u);
    remap3w(v, grad_velocity, gu);
    return run_mac_cormack_centered_bwd(ctx, v, sfield, s_bc, s_val, u, grad_out, grad_s, grad_velocity ? gu : nullptr, dt, correction_strength, s);
}

int phihip_diffuse_explicit_backward(phihip_ctx* ctx, const phihip_grid* grid, const void* const grad_out[3], void* const grad_in[3],
                                     double diffusivity_dt, void* stream) {
    PHIHIP_ENTER(ctx, grid);
    PHIHIP_TRY(check_ptrs(v, grad_out, "grad_out"));
    PHIHIP_TRY(check_ptrs(v, (const void* const*)grad_in, "grad_in"));
    const void* go[3];
    void* gi[3];
    remap3(v, grad_out, go);
    remap3w(v, grad_in, gi);
    return run_diffuse_bwd(ctx, v, go, gi, diffusivity_dt, s);
}

int phihip_diffuse_explicit_centered(phihip_ctx* ctx, const phihip_grid* grid, const void* sfield, const int32_t s_bc[3][2],
                                     const double s_val[3][2], void* out, double diffusivity_dt, int adjoint, void* stream) {
    PHIHIP_ENTER(ctx, grid);
    PHIHIP_REQUIRE(sfield && out && s_bc && sfield != out, "diffuse_explicit_centered: NULL or aliased argument");
    PHIHIP_TRY(check_scalar_bc(v, s_bc, "diffuse_explicit_centered"));
    note_align(v, sfield); note_align(v, out);
    return run_diffuse_centered(ctx, v, sfield, s_bc, s_val, out, diffusivity_dt, adjoint, s);
}

int phihip_centered_to_staggered_backward(phihip_ctx* ctx, const phihip_grid* grid, const int32_t s_bc[3][2], const double vector[3],
                                          const void* const grad_out[3], void* grad_s, void* stream) {
    PHIHIP_ENTER(ctx, grid);
    PHIHIP_REQUIRE(s_bc && vector && grad_s, "centered_to_staggered_backward: NULL argument");
    PHIHIP_TRY(check_ptrs(v, grad_out, "grad_out"));
    PHIHIP_TRY(check_scalar_bc(v, s_bc, "centered_to_staggered_backward"));
    double vec[3] = {0, 0, 0};
    for (int d = 0; d < v.rank; ++d) vec[d + v.ax0] = vector[d];
    const void* go[3];
    remap3(v, grad_out, go);
    return run_centered_to_staggered_bwd(ctx, v, s_bc, vec, go, grad_s, s);
}

int phihip_make_incompressible_backward(phihip_ctx* ctx, const phihip_grid* grid, const uint8_t* flags, int mask_batch, int balance,
                                        void* const grad_velocity[3], const void* grad_pressure, const phihip_solve* solve,
                                        phihip_solve_info* info, void* stream) {
    PHIHIP_ENTER(ctx, grid);
    PHIHIP_TRY(check_ptrs(v, (const void* const*)grad_velocity, "grad_velocity"));
    PHIHIP_REQUIRE(mask_batch == 1 || mask_batch == v.batch, "mask_batch must be 1 or grid.batch");
    PHIHIP_TRY(check_solve(solve));
    void* gu[3];
    remap3w(v, grad_velocity, gu);
    note_align(v, grad_pressure); note_align(v, flags, 3u);
    return run_project_bwd(ctx, v, flags, mask_batch, (balance & PHIHIP_DIV_BALANCE) ? 1 : 0, gu, grad_pressure, solve, info, s);
}

int phihip_diffuse_explicit(phihip_ctx* ctx, const phihip_grid* grid, const void* const velocity[3], void* const out[3],
                            double diffusivity_dt, void* stream) {
    PHIHIP_ENTER(ctx, grid);
    PHIHIP_TRY(check_ptrs(v, velocity, "velocity"));
    PHIHIP_TRY(check_ptrs(v, (const void* const*)out, "out"));
    for (int d = 0; d < v.rank; ++d) PHIHIP_REQUIRE(out[d] != velocity[d], "diffuse: out[%d] must not alias the input", d);
    const void* u[3];
    void* o[3];
    remap3(v, velocity, u);
    remap3w(v, out, o);
    for (int d = v.ax0; d < 3; ++d) { note_align(v, u[d]); note_align(v, o[d]); }      // (the marching kernels' vector paths need 16-byte aligned lattices)
    return run_diffuse(ctx, v, u, o, diffusivity_dt, s);
}

int phihip_diffuse_implicit(phihip_ctx* ctx, const phihip_grid* grid, const void* const velocity[3], void* const out[3], double diffusivity_dt,
                            const phihip_solve* solve, phihip_solve_info* info, void* stream) {
    PHIHIP_ENTER(ctx, grid);
    PHIHIP_TRY(check_ptrs(v, velocity, "velocity"));
    PHIHIP_TRY(check_ptrs(v, (const void* const*)out, "out"));
    for (int d = 0; d < v.rank; ++d) PHIHIP_REQUIRE(out[d] != velocity[d], "diffuse_implicit: out[%d] must not alias the input", d);
    PHIHIP_REQUIRE(diffusivity_dt >= 0.0, "diffuse_implicit: diffusivity * dt must be >= 0 (the operator I - k dt L is not positive definite otherwise)");
    PHIHIP_TRY(check_solve(solve));
    const void* u[3];
    void* o[3];
    remap3(v, velocity, u);
    remap3w(v, out, o);
    for (int d = v.ax0; d < 3; ++d) { note_align(v, u[d]); note_align(v, o[d]); }
    return run_diffuse_implicit(ctx, v, u, o, diffusivity_dt, solve, info, s);
}

int phihip_diffuse_implicit_centered(phihip_ctx* ctx, const phihip_grid* grid, const void* sfield, const int32_t s_bc[3][2], const double s_val[3][2],
                                     void* out, double diffusivity_dt, const phihip_solve* solve, phihip_solve_info* info, void* stream) {
    PHIHIP_ENTER(ctx, grid);
    PHIHIP_REQUIRE(sfield && out && s_bc && sfield != out, "diffuse_implicit_centered: NULL or aliased argument");
    PHIHIP_REQUIRE(diffusivity_dt >= 0.0, "diffuse_implicit_centered: diffusivity * dt must be >= 0");
    PHIHIP_TRY(check_scalar_bc(v, s_bc, "diffuse_implicit_centered"));
    PHIHIP_TRY(check_solve(solve));
    note_align(v, sfield); note_align(v, out);
    return run_diffuse_implicit_centered(ctx, v, sfield, s_bc, s_val, out, diffusivity_dt, solve, info, s);
}

int phihip_query_plan(phihip_ctx* ctx, const phihip_grid* grid, int has_flags, int family, int32_t out[6]) {
    PHIHIP_REQUIRE(ctx != nullptr && out != nullptr, "query_plan: NULL argument");
    PHIHIP_REQUIRE(family >= 0 && family < 5, "tuning family must be 0 (apply / residual), 1 (matvec), 2 (update), 3 (r-only update) or 4 (fused single-reduction iteration)");
    GridView v;
    PHIHIP_TRY(make_view(grid, &v));
    PHIHIP_CHECK_HIP(hipSetDevice(ctx->device));
    MarchConfig c;
    MarchGrid g;
    PHIHIP_TRY(plan_march(ctx, v, 1, has_flags != 0, family, &c, &g));
    out[0] = march_one_tile(c.vec) ? 1 : kTileShapes[c.id].rows;
    out[1] = march_one_tile(c.vec) ? 64 : kTileShapes[c.id].tpr;
    out[2] = c.chunk;
    out[3] = g.nblk;
    out[4] = march_occupancy_any(v, c.id, c.vec, family_mode(family), has_flags != 0);
    out[5] = c.vec;
    return PHIHIP_OK;
}

int phihip_profile_enable(phihip_ctx* ctx, int enable) {
    PHIHIP_REQUIRE(ctx != nullptr, "ctx is NULL");
    ctx->profiling = enable != 0;
    return PHIHIP_OK;
}

int phihip_profile_read(phihip_ctx* ctx, int32_t launches[PHIHIP_K_COUNT], double total_ms[PHIHIP_K_COUNT], int reset) {
    PHIHIP_REQUIRE(ctx && launches && total_ms, "profile_read: NULL argument");
    PHIHIP_CHECK_HIP(hipSetDevice(ctx->device));
    PHIHIP_TRY(profile_collect(ctx));
    for (int k = 0; k < PHIHIP_K_COUNT; ++k) {
        launches[k] = ctx->prof_launches[k];
        total_ms[k] = ctx->prof_ms[k];
        if (reset) {
            ctx->prof_launches[k] = 0;
            ctx->prof_ms[k] = 0;
        }
    }
    return PHIHIP_OK;
}

int phihip_set_tuning(phihip_ctx* ctx, int rows_per_thread, int threads_per_row, int chunk_planes) {
    PHIHIP_REQUIRE(ctx != nullptr, "ctx is NULL");
    PHIHIP_REQUIRE(rows_per_thread >= 0 && threads_per_row >= 0 && chunk_planes >= 0, "tuning values must be >= 0");
    for (int f = 0; f < 4; ++f) {
        ctx->tuning[f].rows = rows_per_thread;
        ctx->tuning[f].tpr = threads_per_row;
        ctx->tuning[f].chunk = chunk_planes;
    }
    return PHIHIP_OK;
}

int phihip_set_small_grid_solver(phihip_ctx* ctx, int enable) {
    PHIHIP_REQUIRE(ctx != nullptr, "ctx is NULL");
    ctx->small_cg = enable != 0;
    ctx->small_cg_cells = enable > 1 ? enable : 0;   // > 1: use the single-kernel solver up to that many cells (<= 16384 fp32, 8192 fp64)
    return PHIHIP_OK;
}

int phihip_set_advect_halo(phihip_ctx* ctx, int halo) {
    PHIHIP_REQUIRE(ctx != nullptr, "ctx is NULL");
    PHIHIP_REQUIRE(halo >= -1 && halo <= 3, "advect halo must be -1 (adaptive), 0 (gather kernels), 1, 2 or 3 (experimental: halo 1 with 16-row tiles, 3-D only)");
    ctx->adv_halo = halo;
    for (auto& K : ctx->adv_policy)
        for (auto& P : K.e) { P.mode = 1; P.calls = 0; P.pending = false; }
    return PHIHIP_OK;
}

int phihip_workspace_placement(phihip_ctx* ctx, int candidates, int32_t* last_candidates, double last_us[2]) {
    PHIHIP_REQUIRE(ctx != nullptr, "ctx is NULL");
    PHIHIP_REQUIRE(candidates <= 16, "workspace_placement: at most 16 candidates (got %d)", candidates);
    if (candidates >= 0) ctx->ws_candidates = candidates < 1 ? 1 : candidates;
    if (last_candidates) *last_candidates = ctx->ws_place_count;
    if (last_us) { last_us[0] = ctx->ws_place_first_us; last_us[1] = ctx->ws_place_best_us; }
    return PHIHIP_OK;
}

int phihip_query_advect_chunk(phihip_ctx* ctx, int32_t* planes) {
    PHIHIP_REQUIRE(ctx != nullptr && planes != nullptr, "query_advect_chunk: NULL argument");
    *planes = ctx->adv_last_chunk;
    return PHIHIP_OK;
}

int phihip_set_advect_dma(phihip_ctx* ctx, int enable, int32_t* last_was_dma) {
    PHIHIP_REQUIRE(ctx != nullptr, "ctx is NULL");
    if (enable >= 0) ctx->adv_dma = enable != 0;
    if (last_was_dma) *last_was_dma = ctx->adv_last_dma;
    return PHIHIP_OK;
}

int phihip_set_advect_windows_2d(phihip_ctx* ctx, int enable) {
    PHIHIP_REQUIRE(ctx != nullptr, "ctx is NULL");
    ctx->adv_win_2d = enable != 0;
    return PHIHIP_OK;
}

int phihip_set_advect_chunk(phihip_ctx* ctx, int planes) {
    PHIHIP_REQUIRE(ctx != nullptr && planes >= 0, "ctx is NULL or planes < 0");
    ctx->adv_chunk = planes;
    return PHIHIP_OK;
}

int phihip_advect_fallback_stats(phihip_ctx* ctx, int32_t out[2], void* stream) {
    PHIHIP_REQUIRE(ctx != nullptr && out != nullptr, "ctx / out is NULL");
    out[0] = out[1] = 0;
    if (ctx->adv_last_nblk <= 0 || !ctx->ws_adv_flags.ptr) return PHIHIP_OK;
    PHIHIP_CHECK_HIP(hipSetDevice(ctx->device));
    int host[4] = {0, 0, 0, 0};         // the work list's counters (advect_common.hpp): the most recent launch used the one of its parity ([3]: captured launches)
    PHIHIP_CHECK_HIP(hipMemcpyAsync(host, ctx->ws_adv_flags.ptr, sizeof(host), hipMemcpyDeviceToHost, (hipStream_t)stream));
    PHIHIP_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    const int n = ctx->adv_seq_captured ? host[3] : host[ctx->adv_seq & 1u];
    out[0] = n < ctx->adv_last_nblk ? n : ctx->adv_last_nblk;
    out[1] = ctx->adv_last_nblk;
    return PHIHIP_OK;
}

int phihip_set_resident_cg(phihip_ctx* ctx, int mode, long long max_cells) {
    PHIHIP_REQUIRE(ctx != nullptr && mode >= 0 && mode <= 2 && max_cells >= 0, "set_resident_cg: mode must be 0, 1 or 2, max_cells >= 0");
    ctx->resident_cg = mode;
    if (max_cells > 0) ctx->resident_cg_cells = max_cells;
    return PHIHIP_OK;
}

int phihip_set_single_reduction_cg(phihip_ctx* ctx, int mode, long long max_cells) {
    PHIHIP_REQUIRE(ctx != nullptr && mode >= 0 && mode <= 2 && max_cells >= 0, "set_single_reduction_cg: mode must be 0, 1 or 2, max_cells >= 0");
    ctx->cg1_mode = mode;
    ctx->cg1_cells = max_cells;
    return PHIHIP_OK;
}

int phihip_set_autotune(phihip_ctx* ctx, int enable) {
    PHIHIP_REQUIRE(ctx != nullptr, "ctx is NULL");
    ctx->autotune = enable != 0;
    if (!enable) ctx->tuned.clear();
    return PHIHIP_OK;
}

int phihip_set_deferred_x_update(phihip_ctx* ctx, int enable) {
    PHIHIP_REQUIRE(ctx != nullptr, "ctx is NULL");
    ctx->defer_x = enable != 0;
    return PHIHIP_OK;
}

int phihip_set_tuning_kernel(phihip_ctx* ctx, int family, int rows_per_thread, int threads_per_row, int chunk_planes) {
    PHIHIP_REQUIRE(ctx != nullptr, "ctx is NULL");
    PHIHIP_REQUIRE(family >= 0 && family < 5, "tuning family must be 0 (apply / residual), 1 (matvec), 2 (update), 3 (r-only update) or 4 (fused single-reduction iteration)");
    PHIHIP_REQUIRE(rows_per_thread >= 0 && threads_per_row >= 0 && chunk_planes >= 0, "tuning values must be >= 0");
    ctx->tuning[family].rows = rows_per_thread;
    ctx->tuning[family].tpr = threads_per_row;
    ctx->tuning[family].chunk = chunk_planes;
    return PHIHIP_OK;
}

}  // extern "C"
