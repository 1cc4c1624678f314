// common.hpp -- context, error reporting, launch + profiling helpers of libphihip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <array>
#include <map>
#include <string>
#include <vector>

#include "../../include/phihip.h"
#include "stencil_march.hpp"

namespace phihip {

void set_error(const char* fmt, ...);

#define PHIHIP_CHECK_HIP(expr)                                                                   \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            phihip::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return PHIHIP_ERR_HIP;                                                               \
        }                                                                                        \
    } while (0)

#define PHIHIP_REQUIRE(cond, ...)               \
    do {                                        \
        if (!(cond)) {                          \
            phihip::set_error(__VA_ARGS__);     \
            return PHIHIP_ERR_BAD_ARG;          \
        }                                       \
    } while (0)

#define PHIHIP_TRY(expr)             \
    do {                             \
        int _s = (expr);             \
        if (_s != PHIHIP_OK) return _s; \
    } while (0)

// Internal view of a phihip_grid: always three axes (a0 slow .. a2 fast). Rank-2 grids (x, y) map to (a1, a2), n0 = 1.
struct GridView {
    int rank, ax0;             // ax0 = 3 - rank : first used internal axis
    int dtype, batch;
    int n[3];                  // cells
    int bc[3][2];              // velocity boundary codes (unused axis: periodic)
    double bcv[3][2][3];       // [axis][side][component axis]
    double lower[3], dx[3];
    int cn[3][3];              // [component axis][axis] stored faces
    int off[3];                // physical face number of stored index 0 per component
    long long cells;           // n0*n1*n2
    long long ccells[3];       // cells per stored component
    int halo[2];               // slab decomposition: planes beyond the lower / upper a0 side come from a neighbour rank
    bool unaligned;            // some caller buffer of this call is not 16-byte aligned: the marching kernels take the scalar path
    // Linear operator of the CG kernels. Default (op_custom == 0): the pressure operator L = masked_laplace with the pressure's neighbour
    // rule derived from bc. Custom: op_ident * I + op_scale * L on a lattice of n cells with an explicit NeighbourRule per side
    // (implicit diffusion: I - k dt L on the lattice of a centred scalar or of one staggered component, the field's own extrapolation)
    int op_custom;
    double op_ident, op_scale;
    int op_rule[3][2];
};

int make_view(const phihip_grid* grid, GridView* out);

// by-value kernel parameter describing the staggered layout
struct VelGrid {
    int n[3];
    int cn[3][3];
    int off[3];
    int bc[3][2];
    double bcv[3][2][3];
    double dx[3];
    double rdx[3];     // 1 / dx (the back-trace works in index space: shift = u * (dt / dx))
    int ax0;
    long long cells;
    long long ccells[3];
};

VelGrid make_velgrid(const GridView& v);

// extrapolation of a centred scalar per internal axis / side: PERIODIC wrap, OPEN = zero-gradient, CLOSED = constant val
struct ScalarBc {
    int bc[3][2];
    double val[3][2];
};

ScalarBc make_scalar_bc(const GridView& v, const int32_t s_bc[3][2], const double s_val[3][2]);

}  // namespace phihip

// Dynamic LDS (kernels that need more than the 64 KB a static allocation may have: hipFuncSetAttribute(...MaxDynamicSharedMemorySize) up to
// the 160 KB of a gfx950 CU). The g++ emulation build of the tests (tests/hipemu) runs one workgroup at a time on one buffer.
#if defined(__HIPCC__)
#define PHIHIP_DYNAMIC_LDS(T, name)                                                  \
    extern __shared__ __attribute__((aligned(16))) unsigned char phihip_dyn_lds_raw[]; \
    T* const name = reinterpret_cast<T*>(phihip_dyn_lds_raw)
#else
#define PHIHIP_DYNAMIC_LDS(T, name) T* const name = reinterpret_cast<T*>(hipemu::dynamic_lds())
#endif

namespace phihip {
// true when the predicate holds for any lane of the wavefront: lets interior wavefronts skip the constant-side selects with a
// SCALAR branch (a per-lane `if` makes the compiler predicate both sides). The CPU emulation build has no wavefronts; there the
// per-thread predicate selects the same values.
__device__ __forceinline__ bool wave_any(bool pred) {
#ifdef __HIP_DEVICE_COMPILE__
    return __builtin_amdgcn_ballot_w64(pred) != 0ull;
#else
    return pred;
#endif
}

struct Tuning {
    int rows = 0, tpr = 0, chunk = 0;   // 0 = auto
};

// launch plan measured on this device for one (grid, kernel family): tile configuration + planes per workgroup (cg.hip autotune)
struct TunedPlan {
    int id, chunk;
    float us;            // measured launch time of the winner
    float us_model;      // measured launch time of the plan the analytic model would have chosen
};
typedef std::array<int, 10> PlanKey;   // dtype, rank, n0, n1, n2, batch, flags, flags per batch, family, vector path

struct DeviceBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;
};

}  // namespace phihip

struct phihip_ctx {
    int device = 0;
    int num_cu = 256;
    phihip::Tuning tuning[5];   // per kernel family: 0 = APPLY / RESID, 1 = MATVEC, 2 = UPDATE, 3 = UPDATE_R, 4 = CG1 (fused iteration)
    // workspace (grown on demand, reused between calls)
    phihip::DeviceBuffer ws_r, ws_d0, ws_d1, ws_div, ws_part, ws_state, ws_scalars, ws_rhs, ws_adv, ws_adv_flags, ws_adj_q, ws_adj_l, ws_cg1, ws_adj_g, ws_res, ws_adv_const;
    int adv_last_nblk = 0;        // (tile, plane) units of the most recent LDS-staged advection launch (capacity of its fix-up work list)
    bool adv_ctl_clear = false;   // the work list's control block in ws_adv_flags has been zeroed
    // Adaptive reach (r4). Each LDS-staged pass publishes how many (tile, plane) units fell back to the gather path (the fix-up launch writes
    // the count into pinned, device-mapped host memory); the NEXT pass of the same kind waits for that launch's event -- it is one time step
    // old, the wait is free unless the host runs more than a step ahead -- and picks its reach from the fraction: narrow (1 cell: fastest
    // while the flow stays below CFL 1) -> wide (2 cells: +10-25 % time, immune below CFL 2) -> gather kernels (no windows: flat cost at any
    // CFL), with a probe of the cheaper form every 64 calls. The decision depends on data only (not on timing): replicas that advance the
    // same state take the same path -- r5 (second step): the count a decision uses is that of ONE named pass, read a fixed number of passes (kAdvMaxLag + 1 = 3) after it
    // (every pass publishes into its own slot of a ring; the first non-blocking version took whatever pass had completed when the host looked and made
    // the path, hence the last bits of a CFL > 1 run, depend on host timing). Measured (profiles/r04_time_frow_session_f.jsonl, 256^3 fp32): semi_lagrangian(s, v) at CFL 0.5 / 1.3 /
    // 1.8: narrow 0.078 / 0.200 / 0.396 ms, wide 0.086 / 0.085 / 0.086, gather 0.105 / 0.106 / 0.107.
    struct AdvPolicy {
        hipEvent_t ev = nullptr;
        long long fp = 0;        // fingerprint of the grid the state below belongs to (0: free entry)
        int mode = 1;            // 0 gather, 1 narrow, 2 wide
        int last = 0;            // reach of the pass that `pending` refers to
        int calls = 0;
        long long units = 0;     // (tile, plane) units of that pass
        bool pending = false;    // an event + a published count are outstanding
        int age = 0;             // passes of this kind enqueued since that event was recorded
        unsigned obs_seq = 0;    // the pass (number among this KIND's passes, AdvKindState::seq) `ev` / `pending` belong to
        unsigned used = 0;       // AdvKindState::clock of the most recent pass on this grid (least recently used entry is replaced)
    };
    // r6: one policy PER GRID, a few grids per kind of pass. Until r5 a kind had ONE policy that restarted whenever the grid changed: a SlabFluid in
    // overlap mode (whole-slab pass and cut-side window passes alternate on one context) or two simulations sharing a context changed it on every call -- no
    // observation was ever resolved, the reach never left narrow, and the restart of `seq` reused publish slots of passes still in flight.
    static constexpr int kAdvGrids = 4;
    struct AdvKindState {
        AdvPolicy e[kAdvGrids];
        unsigned seq = 0;        // number of this kind's eager passes (all grids): pass `seq` publishes its count into slot seq % kAdvSlots
        unsigned clock = 0;
        int cur = 0;             // the entry of the pass being enqueued (adv_choose -> adv_record / prepare_fixlist)
    };
    static constexpr int kAdvSlots = 8, kAdvSlotBase = 16, kAdvCaptureBase = 16 + 4 * 8, kAdvHostInts = 64;   // layout of adv_host (ints); [15]: the resident solver's abort word
    AdvKindState adv_policy[4];   // AdvKind: self-advection, staggered MacCormack correction, centred semi-Lagrangian, centred MacCormack correction
    int* adv_host = nullptr;      // pinned, device-mapped: fallback count per kind
    int* adv_host_dev = nullptr;
    unsigned adv_seq = 0;         // launches of LDS-staged advection kernels so far: parity selects the work list's counter
    bool adv_seq_captured = false;   // the most recent such launch was captured into a hipGraph (its own counter, cleared by a memset node)
    double adv_const_key[19] = {0};   // the wall constants (+ element size) the table in ws_adv_const holds
    bool adv_const_valid = false;
    int adv_reach_now = 0;        // reach of the LDS-staged pass being enqueued (advect.hip pass_reach -> prepare_fixlist)
    int adv_last_dma = 0;         // the most recent tiled self-advection filled its ring by LDS-DMA
    int adv_dma = 1;              // r5: regular grids fill the self-advection's ring by LDS-DMA (PHIHIP_ADVECT_DMA=0: the register-staged kernel everywhere)
    int adv_chunk = 0;            // planes per workgroup of the tiled advection (0 = planned from the occupancy)
    int adv_last_chunk = 0;       // planes per workgroup the most recent tiled self-advection ran with (phihip_query_advect_chunk)
    int adv_halo = -1;            // phihip_set_advect_halo: reach of the LDS-staged advection kernels. -1 (default) = adaptive per kind of pass (AdvPolicy below),
                                  // 0 = the gather kernels of advect.hip, 1 / 2 = fixed (3 = experimental 16-row tiles of the self-advection)
    bool adv_win_2d = false;      // advect_win.hip on 2-D grids (slower than the gather kernels there; phihip_set_advect_halo(ctx, 4) switches it on for the parity tests)
    // first-call autotune of the CG marching kernels: the candidates of the plan model are timed once per (grid, family) on the
    // context's own workspace and the fastest is cached here. PHIHIP_AUTOTUNE=0 in the environment / phihip_set_autotune(ctx, 0)
    // keep the analytic plan (bit-reproducible launch geometry across processes).
    bool autotune = true;
    // r6: candidate allocations of the CG workspace (r, d0, d1) the first solve on a freshly grown workspace chooses from by timing the iteration loop
    // (cg.hip place_workspace; PHIHIP_WS_CANDIDATES / phihip_workspace_placement; <= 1: the first allocation is kept). The record of the last choice follows.
    int ws_candidates = 12;
    size_t ws_place_min_bytes = (size_t)72 << 20;      // vectors up to this size are not placed (PHIHIP_WS_PLACE_MIN_BYTES: the emulation test places small ones)
    int ws_place_count = 0;
    float ws_place_best_us = 0.f, ws_place_first_us = 0.f;
    std::map<phihip::PlanKey, phihip::TunedPlan> tuned;
    std::map<std::array<long long, 6>, int> adv_tuned;   // tiled advection: (dtype bytes, dim, halo, planes, tiles, batch) -> planes per workgroup
    // single-reduction (Chronopoulos-Gear) CG, one launch per iteration (stencil_march.hpp MODE_CG1): 0 = never, 1 (default since r3: the
    // five-sum closure of alpha reaches the accuracy of the two-launch form, tools/cg1_accuracy.py) = for solves whose iteration is bound
    // by the kernel boundaries (cg1_cells: cells x batch at most this), 2 = always ('CG' only)
    int cg1_mode = 1;
    long long cg1_cells = 0;      // 0 = built-in threshold
    // resident solver (cg_resident.hip): 0 = off, 1 = 2-D fp32 'CG' solves of at most resident_cg_cells cells x batch, 2 = whenever applicable
    // r6: ON by default (mode 1) -- the launch is cooperative (co-residency checked by the runtime, cooperative kernels of a device serialised), the solve number
    // lives on the device (capture-safe), a launch that does not fit falls back to the launch forms
    int resident_cg = 1;
    int res_coop = 0;             // 1 (PHIHIP_RESIDENT_COOP=1): launch the resident solver with hipLaunchCooperativeKernel. Default 0 since the last session of r6: the
                                  // cooperative launch synchronises with every queue of the device -- +0.04 ms per solve in a fresh process, +0.5 ms once the process has
                                  // created side streams (bench.py after its jit captures: the 128^2 plume step 0.36 -> 0.88 ms; profiles/r06_resident_coop_cost.txt)
    int res_coop_capture = 0;     // ... also while the stream is being captured (PHIHIP_RESIDENT_COOP_CAPTURE=1; default: the plain launch as a graph node)
    long long resident_cg_cells = 16LL << 20;      // (r6: 4 Mi -> 16 Mi with the sub-batch launches: 64 x 512^2)
    bool defer_x = true;          // CG: x is updated every other iteration only (UPDATE_R / UPDATE_X2, stencil_march.hpp)
    long long small_cg_cells = 0;   // experiment switch (phihip_set_small_grid_solver(ctx, n > 1)): cell limit instead of the built-in rule
    bool small_cg = true;         // grids that fit one CU's LDS are solved by the single-kernel CG (cg_small.hip)
    int slab_cur = 0;             // control-block slot of the running slab-decomposed solve
    void* last_state = nullptr;   // device control blocks of the most recent solve
    int last_state_batch = 0;
    void* host_state = nullptr;   // pinned readback buffer
    size_t host_state_bytes = 0;
    hipEvent_t poll_ev[2] = {nullptr, nullptr};   // throttle of the host's run-ahead in tolerance mode (cg.hip)
    // tolerance mode: the MATVEC prologue publishes (solve sequence number << 32 | continue flag) per batch entry straight into this
    // pinned, device-mapped array; the host reads it before every enqueue -- no peek kernel, no copy, no stream drain
    unsigned long long* host_flags = nullptr;
    unsigned long long* host_flags_dev = nullptr;
    size_t host_flags_count = 0;
    unsigned int solve_seq = 0;
    // profiling
    bool profiling = false;
    struct EventPair {
        hipEvent_t a, b;
        int kid;
    };
    std::vector<EventPair> ev_pool;
    size_t ev_used = 0;
    int32_t prof_launches[PHIHIP_K_COUNT] = {0};
    double prof_ms[PHIHIP_K_COUNT] = {0};
};

namespace phihip {

int ensure_buffer(DeviceBuffer& buf, size_t bytes);
int profile_begin(phihip_ctx* ctx, int kid, hipStream_t s, int* slot);
int profile_end(phihip_ctx* ctx, int slot, hipStream_t s);
int profile_collect(phihip_ctx* ctx);

// RAII-free helper: wraps one launch in an event pair when profiling is on
struct LaunchScope {
    phihip_ctx* ctx;
    hipStream_t s;
    int slot = -1;
    LaunchScope(phihip_ctx* c, int kid, hipStream_t st) : ctx(c), s(st) {
        if (ctx && ctx->profiling) profile_begin(ctx, kid, s, &slot);
    }
    ~LaunchScope() {
        if (slot >= 0) profile_end(ctx, slot, s);
    }
};

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// PHIHIP_TRACE_SLOW=<ms>: host-side sections that block longer than that many milliseconds report themselves on stderr (where does a call stall?)
struct SlowTrace {
    const char* what;
    std::chrono::steady_clock::time_point t0;
    bool open = true;
    static double limit_ms() { static const double l = [] { const char* e = getenv("PHIHIP_TRACE_SLOW"); return e ? atof(e) : -1.0; }(); return l; }
    explicit SlowTrace(const char* w) : what(w) { if (limit_ms() >= 0) t0 = std::chrono::steady_clock::now(); }
    void done() {
        if (!open || limit_ms() < 0) return;
        open = false;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms > limit_ms()) fprintf(stderr, "[phihip slow] %s: %.2f ms\n", what, ms);
    }
    ~SlowTrace() { done(); }
};

// work list of the LDS-staged advection kernels' fix-up pass (device side: advect_common.hpp)
struct FixItem {
    int wg, plane;
};
enum AdvKind { AK_SL_SELF = 0, AK_MC_STAG = 1, AK_SL_CEN = 2, AK_MC_CEN = 3, AK_NONE = -1 };
struct FixList {
    int* publish;        // host-mapped slot that receives this launch's count from the fix-up launch (adaptive reach), or nullptr
    int* count;          // entries appended by THIS launch's tile kernel (zero when it starts: cleared by the previous launch's tile kernel)
    int* next;           // the other counter: this launch's tile kernel clears it for the next launch (its last reader, the fix-up launch of
                         // the previous launch, has completed -- stream order); no atomics / tickets in the fix-up launch
    FixItem* items;
    int cap;
    int reach_tag;       // reach of THIS pass (1 / 2) << 28: published with the count, so the host attributes a count to the reach that produced it
};
// ws_adv_flags = [16 control ints | 64-byte dump slot | work list]; `units` = (tile, plane) pairs of the launch = capacity of the list.
// Call once per tile-kernel launch (the two counters alternate).
int prepare_fixlist(phihip_ctx* ctx, long long units, hipStream_t s, FixList* list, void** dump, int kind = AK_NONE);
// adaptive reach of the LDS-staged advection passes (advect.hip): reach for the next pass of `kind` (0 gather / 1 / 2), and the bookkeeping
// after a windowed pass was enqueued
int adv_choose(phihip_ctx* ctx, int kind, bool has_wide, long long grid_fp, hipStream_t s);
int ensure_adv_host_public(phihip_ctx* ctx);     // the pinned, device-mapped words (slot 15: "a resident solve gave up")
int adv_record(phihip_ctx* ctx, int kind, int reach, hipStream_t s);
constexpr int kFixupBlocks = 2048;       // fix-up grid (8 workgroups per CU), whatever the list holds

// The first-call autotunes time candidate launches with hipEventSynchronize -- illegal while `s` is being captured into a hipGraph (and
// the timings would be meaningless there): a capturing stream keeps the analytic plan, the next eager call on the grid tunes.
inline bool stream_is_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}

// ---- phases implemented in the .hip files ---------------------------------------------------------------------------
int run_advect_staggered(phihip_ctx*, const GridView&, const void* const f[3], const void* const v[3], void* const out[3], double dt, hipStream_t);
int run_advect_self_tiled(phihip_ctx*, const GridView&, const void* const v[3], void* const out[3], double dt, int halo, int kind, hipStream_t);
// LDS-windowed passes of advect_win.hip (PHIHIP_ERR_UNSUPPORTED: an axis with fewer than 4 samples -- the caller keeps the gather kernels)
int run_mc_correct_self_tiled(phihip_ctx*, const GridView&, const void* const v[3], const void* const fwd[3], void* const out[3], double dt, double ch, int kind, hipStream_t);
int run_advect_centered_tiled(phihip_ctx*, const GridView&, const void* s, const ScalarBc& sb, const void* const v[3], void* out, double dt, int halo, int kind, hipStream_t);
int run_mc_correct_centered_tiled(phihip_ctx*, const GridView&, const void* s, const ScalarBc& sb, const void* const v[3], const void* fwd, void* out, double dt,
                                  double ch, int halo, int kind, hipStream_t);
int run_grid_sample(phihip_ctx*, const GridView&, const int32_t s_bc[3][2], const double s_val[3][2], const void* values, int values_batch,
                    const void* const coords[3], long long npts, void* out, void* out_min, void* out_max, hipStream_t);
int run_grid_sample_bwd(phihip_ctx*, const GridView&, const int32_t s_bc[3][2], const double s_val[3][2], const void* values, int values_batch,
                        const void* const coords[3], long long npts, const void* gout, void* gvalues, void* const gcoords[3], hipStream_t);
int run_advect_centered(phihip_ctx*, const GridView&, const void* s, const int32_t s_bc[3][2], const double s_val[3][2],
                        const void* const v[3], void* out, double dt, hipStream_t);
int run_mac_cormack_staggered(phihip_ctx*, const GridView&, const void* const f[3], const void* const v[3], void* const out[3], double dt,
                              double strength, hipStream_t);
int run_mac_cormack_centered(phihip_ctx*, const GridView&, const void* s, const int32_t s_bc[3][2], const double s_val[3][2],
                             const void* const v[3], void* out, double dt, double strength, hipStream_t);
int run_centered_to_staggered(phihip_ctx*, const GridView&, const void* s, const int32_t s_bc[3][2], const double s_val[3][2],
                              const double vector[3], int accumulate, void* const out[3], hipStream_t);
int run_obstacle_accessible(phihip_ctx*, const GridView&, const phihip_obstacle* obs, int count, uint8_t* accessible, hipStream_t);
int run_apply_obstacles(phihip_ctx*, const GridView&, const phihip_obstacle* obs, int count, void* const v[3], hipStream_t);
int run_balance(phihip_ctx*, const GridView&, const uint8_t* flags, int mask_batch, void* x, hipStream_t);
int run_advect_staggered_bwd(phihip_ctx*, const GridView&, const void* const f[3], const void* const v[3], const void* const gout[3],
                             void* const gf[3], void* const gv[3], double dt, hipStream_t);
int run_advect_centered_bwd(phihip_ctx*, const GridView&, const void* s, const int32_t s_bc[3][2], const double s_val[3][2],
                            const void* const v[3], const void* gout, void* gs, void* const gv[3], double dt, hipStream_t);
int run_centered_to_staggered_bwd(phihip_ctx*, const GridView&, const int32_t s_bc[3][2], const double vector[3], const void* const gout[3],
                                  void* gs, hipStream_t);
int run_mac_cormack_staggered_bwd(phihip_ctx*, const GridView&, const void* const f[3], const void* const v[3], const void* const gout[3],
                                  void* const gf[3], void* const gv[3], double dt, double strength, hipStream_t);
int run_mac_cormack_centered_bwd(phihip_ctx*, const GridView&, const void* s, const int32_t s_bc[3][2], const double s_val[3][2],
                                 const void* const v[3], const void* gout, void* gs, void* const gv[3], double dt, double strength, hipStream_t);
int run_diffuse_bwd(phihip_ctx*, const GridView&, const void* const gout[3], void* const gin[3], double kdt, hipStream_t);
int run_diffuse_centered(phihip_ctx*, const GridView&, const void* s, const int32_t s_bc[3][2], const double s_val[3][2], void* out, double kdt,
                         int adjoint, hipStream_t);
int run_project_bwd(phihip_ctx*, const GridView&, const uint8_t* flags, int mask_batch, int balance, void* const gv[3], const void* gp,
                    const phihip_solve*, phihip_solve_info*, hipStream_t);
int run_slab_residual(phihip_ctx*, const GridView&, const uint8_t* flags, int mask_batch, const void* x, const void* x_lo, const void* x_hi,
                      const void* rhs, void* r, double* sums, int keep_going, hipStream_t);
int run_slab_matvec(phihip_ctx*, const GridView&, const uint8_t* flags, int mask_batch, int first, const double* sums_in, const void* r,
                    const void* r_lo, const void* r_hi, const void* d_old, const void* d_lo, const void* d_hi, void* d_new, double* sum_out,
                    const phihip_solve*, hipStream_t);
int run_slab_update(phihip_ctx*, const GridView&, const uint8_t* flags, int mask_batch, const double* sum_in, const void* d, const void* d_lo,
                    const void* d_hi, void* x, void* r, double* sum_out, int x_only, const phihip_solve*, hipStream_t);
int run_slab_finish(phihip_ctx*, const GridView&, int first, const double* sums_in, const phihip_solve*, phihip_solve_info*, int peek, hipStream_t);
int run_build_cellflags(phihip_ctx*, const GridView&, const uint8_t* accessible, const uint8_t* active, int mask_batch, uint8_t* flags, hipStream_t);
int run_divergence(phihip_ctx*, const GridView&, const void* const v[3], const uint8_t* flags, int mask_batch, int balance, void* div, hipStream_t);
int run_scale_faces(phihip_ctx*, const GridView&, void* const v[3], const void* const m[3], hipStream_t);
int run_grad_subtract(phihip_ctx*, const GridView&, const uint8_t* flags, int mask_batch, const void* p, void* const v[3], hipStream_t);
int run_diffuse(phihip_ctx*, const GridView&, const void* const v[3], void* const out[3], double kdt, hipStream_t);
int run_diffuse_implicit(phihip_ctx*, const GridView&, const void* const v[3], void* const out[3], double kdt, const phihip_solve*, phihip_solve_info*, hipStream_t);
int run_diffuse_implicit_centered(phihip_ctx*, const GridView&, const void* s, const int32_t s_bc[3][2], const double s_val[3][2], void* out, double kdt,
                                  const phihip_solve*, phihip_solve_info*, hipStream_t);
int run_laplace_apply(phihip_ctx*, const GridView&, const uint8_t* flags, int mask_batch, const void* p, void* out, hipStream_t);
int run_laplace_apply_multi(phihip_ctx*, const GridView* lattices, int count, const void* const* in, void* const* out, hipStream_t);   // MODE_APPLY, no flags: lattices of one tile configuration share a launch
int run_export_residuals(phihip_ctx*, int batch, double* out, hipStream_t);
int run_export_relative_residual(phihip_ctx*, int batch, double* out, hipStream_t);
int run_cg(phihip_ctx*, const GridView&, const uint8_t* flags, int mask_batch, const void* rhs, void* x, const phihip_solve*, phihip_solve_info*, hipStream_t);
// projection: rhs = unbalanced divergence, shift[b] = its mean over the active cells (run_divergence with balance = 2); balanced in place
bool cg_uses_marching(const phihip_ctx*, const GridView&);
int run_cg_balancing(phihip_ctx*, const GridView&, const uint8_t* flags, int mask_batch, void* rhs, void* x, const phihip_solve*, phihip_solve_info*,
                     const double* shift, hipStream_t);

}  // namespace phihip
