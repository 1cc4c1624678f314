// cg_resident.hip -- the WHOLE pressure solve of a batch of 2-D grids in ONE launch whose workgroups stay resident (r4).
//
// Why: a 512^2 CG iteration touches 7-10 MB -- it would take ~2 us from L2 -- but costs 7.8 us as one launch (single-reduction form,
// stencil_march.hpp MODE_CG1) and 10.7 us as two: the price is the dependent kernel boundary plus the prologue chain "partial sums ->
// control block -> first tile" (profiles/r03_cg1_sweep.jsonl). BASELINE configs[3] (8 x 512^2, one entry per GPU when sharded) lives there.
// Here the vectors never leave the chip: a batch entry is owned by G workgroups of 1024 threads, each holding 16 rows (one wavefront per
// row) of r, w = A r, s = A p, p and x in REGISTERS for the whole solve; per iteration a workgroup publishes its first / last row of
// (r, w, s) and five partial sums and reads its two neighbours' rows and the G x 5 sums of its entry back. Same recurrences as MODE_CG1
// (Chronopoulos & Gear single-reduction CG with the five-sum closure of alpha), same control logic as every other solver of the library
// (cg_advance: PhiML's tolerances, divergence test, true-residual refresh every `refresh_every` iterations) -- evaluated redundantly by
// every workgroup from identical sums, so the workgroups of an entry agree on when to stop without a word from the host; the kernel ends
// when its entries have converged.
//   halo of the stencil source: r_new = r - alpha (w + beta s) is RECOMPUTED on the neighbour's published rows (as MODE_CG1 does on its tile
//   halo) with the same expression as on own cells -- one exchange per iteration, and both sides of a cut hold the same bits.
//   exchange = data-tagged granules, no barrier and no fence: every published word travels as ONE naturally aligned 8-byte {value, tag}
//   written by one agent-scope relaxed atomic store (global_store_dwordx2 sc1: write-through) and read by agent-scope relaxed atomic loads
//   (sc1: past the L1) until its tag shows the phase the reader is in (MI355X_MICROARCH.md price list, "handoff-1to1" / "allgather":
//   granules need no ordering, 0.8-1.0 us per hop against 1.7 us for EACH of the release / acquire fences of a flag protocol -- the first
//   form of this kernel, one counter barrier per iteration with plain stores + fences, ran 14.7 us per 512^2 iteration, twice the launch
//   form: profiles/r04_sweep_resident_first_barrier_form.jsonl). tag = (solve number, phase); the all-to-all of the five sums is what orders
//   the phases: a workgroup overwrites a slot (two alternate) only after every workgroup of its entry has published the sums of the phase
//   in between, i.e. has consumed what the slot held.
//   The edge wavefronts (0 and 15) fetch the neighbours' rows -- all twelve granules of a vector in flight at once -- while wavefronts 8-12
//   poll the sums (one sum per wavefront, one workgroup per lane, added by the shuffle tree: the same order everywhere).
//   Measured (MI355X, tools/sweep_resident.py, profiles/r04_sweep_resident_granules_v3.jsonl, us per iteration, launch forms -> resident):
//   1 x 512^2 7.8 -> 7.6, 1 x 192^2 7.1 -> 6.2, 8 x 512^2 13.9-15.1 -> 11.0-11.7, 16 x 256^2 11.5 -> 7.4-8.5, 8 x 512x256 12.2 -> 8.0-9.2,
//   4 x 384^2 10.6 -> 7.8; tolerance-mode solves by the same factors (no host polling). With the granules moved in PAIRS (16-byte stores and
//   loads, gran2_store below -- the full chip was bound by the number of fabric transactions of the rows): 8 x 512^2 10.0, 16 x 256^2 8.2,
//   8 x 512x256 8.6, 4 x 384^2 7.6, i.e. 1.39-1.43x the launch forms (profiles/r04_sweep_resident_paired_granules_tree.jsonl). The floor is the all-to-all of the sums: ~2.5-3 us per
//   hop on this fabric (the guide's "allgather" row) + ~1.5 us of barriers / reductions inside the workgroup + the arithmetic; the 4 us per
//   iteration the round-3 verdict asked for one 512^2 entry is NOT reachable this way -- a single entry gains nothing, batches that fill
//   the chip gain 1.2-1.6x. Opt-in (phihip_set_resident_cg): the launch must be resident as a whole, which the library cannot promise when
//   other streams use the device.
//   Every wait is bounded: a workgroup that polls ~1 s raises `abort` (the host reports PHIHIP_ERR_HIP), nothing can hang the GPU.
//   Residency: batch x G <= number of CUs and one workgroup fills a CU's wave slots at <= 128 VGPRs, so the whole grid is resident on an
//   otherwise idle device (the library's stream); the workgroups of an entry share an XCD when the batch is a multiple of 8 (block id % 8).
// Scope: 2-D, fp32, no cell flags (obstacles keep the marching kernels), rows of whole 16-byte vectors up to 512 cells, 'CG'.
#include "common.hpp"
#include "stencil_march.hpp"

#include <mutex>

namespace phihip {

constexpr int kResBlock = 1024;
constexpr int kResRows = kResBlock / kWave;     // rows per workgroup: one wavefront per row
constexpr int kResMaxG = kWave;                 // workgroups per batch entry: one lane each of the wavefront that adds their partial sums up
constexpr unsigned kResSpinLimit = 3000000u;
typedef unsigned long long gran_t;              // {value : 32, tag : 32}

struct ResArgs {
    int n1, n2, G, batch, ns;     // ns: stride (granules) of a published row
    long long cells;
    int nb1_lo, nb1_hi, nb2_lo, nb2_hi;     // NeighbourRule per side of the two axes
    float w1, w2, ident;
    const uint8_t* flags;         // r6: packed cell flags (phihip_build_cellflags: bits 2, 3 / 4, 5 = the a1 / a2 faces are open for flux, bit 6 = active) or nullptr
    long long flag_bstride;       // cells (one flag array per batch entry) or 0 (shared)
    const float* y;               // right-hand side [batch][n1][n2]
    float* yout;                  // != nullptr: y - shift[b] is written back (fluid._balance_divergence folded in, as MODE_RESID_BAL)
    const double* shift;
    float* x;                     // x0 on entry, solution on exit
    gran_t* pub;                  // [2][batch][G][2 sides][3 arrays][ns] published boundary rows
    gran_t* part;                 // [2][batch][G][5][2] partial sums (low / high word of a double)
    int* abort_flag;              // zero at launch
    int* abort_host;              // pinned, device-mapped: set when the launch gave up (read by the next solve of a caller that passed no `info`)
    const unsigned* solve_ctr;    // r6: the solve number lives on the DEVICE (res_begin_kernel bumps it in front of every launch, also of every graph replay): tags =
                                  // (solve number << 20) | phase, granules of an earlier solve never match
    CgState* st_out;              // [batch]
    CgParams prm;
    int refresh_every;
};

// four consecutive cells as ONE value (clang / gcc vector extension: an SSA value, never an array in memory -- with a struct of four
// floats passed through the helpers by reference the compiler kept r in scratch)
typedef float f4 __attribute__((vector_size(16)));
__device__ __forceinline__ f4 f4_zero() { f4 r = {0.f, 0.f, 0.f, 0.f}; return r; }
__device__ __forceinline__ f4 f4_load(const float* p) { return *reinterpret_cast<const f4*>(p); }
__device__ __forceinline__ void f4_store(float* p, f4 a) { *reinterpret_cast<f4*>(p) = a; }
__device__ __forceinline__ f4 f4_splat(float a) { f4 r = {a, a, a, a}; return r; }
__device__ __forceinline__ f4 f4_fma(float a, f4 b, f4 c) {      // a * b + c with ONE rounding per element (an a * b + c expression may or may not contract)
    f4 r = {fmaf(a, b[0], c[0]), fmaf(a, b[1], c[1]), fmaf(a, b[2], c[2]), fmaf(a, b[3], c[3])};
    return r;
}
__device__ __forceinline__ float f4_dot(f4 a, f4 b) { return ((a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]) + a[3] * b[3]; }

// bit casts that the g++ emulation build knows as well
__device__ __forceinline__ unsigned bits_of(float a) { unsigned u; memcpy(&u, &a, 4); return u; }
__device__ __forceinline__ float float_of(unsigned u) { float a; memcpy(&a, &u, 4); return a; }
__device__ __forceinline__ unsigned long long bits_of(double a) { unsigned long long u; memcpy(&u, &a, 8); return u; }
__device__ __forceinline__ double double_of(unsigned long long u) { double a; memcpy(&a, &u, 8); return a; }

// ---- granules ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void gran_store(gran_t* p, unsigned value, unsigned tag) {
    const gran_t g = ((gran_t)tag << 32) | (gran_t)value;
#ifdef __HIP_DEVICE_COMPILE__
    __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    *reinterpret_cast<volatile gran_t*>(p) = g;
#endif
}
__device__ __forceinline__ gran_t gran_load(const gran_t* p) {
#ifdef __HIP_DEVICE_COMPILE__
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    return *reinterpret_cast<const volatile gran_t*>(p);
#endif
}
__device__ __forceinline__ void res_pause() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_s_sleep(1);
#elif !defined(__HIPCC__)
    hipemu::spin_yield();      // the g++ emulation of the tests (tests/hipemu): fibers, one at a time
#endif
}
__device__ __forceinline__ int res_load_flag(int* p) {
#ifdef __HIP_DEVICE_COMPILE__
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    return *reinterpret_cast<volatile int*>(p);
#endif
}
__device__ __forceinline__ void res_raise_flag(int* p) {
#ifdef __HIP_DEVICE_COMPILE__
    __hip_atomic_store(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    *reinterpret_cast<volatile int*>(p) = 1;
#endif
}
// two granules per 16-byte access: {value, tag, value, tag} -- every 8-byte half still validates itself, the fabric sees half the transactions
// (the boundary rows are 24.6 KB per workgroup and iteration: 256 workgroups issued ~1.6 M eight-byte writes per iteration; 8 x 512^2
// 11.0-11.7 -> 10.0-10.5 us per iteration with the stores alone, profiles/r04_sweep_resident_paired_granules*.jsonl). The store is inline
// assembly (no builtin emits a 16-byte sc1 store) and therefore INVISIBLE to the compiler's hazard recogniser: a VMEM store of more than 8
// bytes whose data registers a VALU instruction has just written needs wait states -- without the s_nop padding the tags never arrived
// (the first build of this form ran into the bound of every wait)
typedef unsigned u4 __attribute__((vector_size(16)));
__device__ __forceinline__ void gran2_store(gran_t* p, unsigned v0, unsigned v1, unsigned tag) {
#ifdef __HIP_DEVICE_COMPILE__
    const u4 g = {v0, tag, v1, tag};
    asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, off sc1\n\ts_nop 4" : : "v"(p), "v"(g) : "memory");
#else
    gran_store(p, v0, tag);
    gran_store(p + 1, v1, tag);
#endif
}
__device__ __forceinline__ void gran_put4(gran_t* p, f4 a, unsigned tag) {
    gran2_store(p, bits_of(a[0]), bits_of(a[1]), tag);
    gran2_store(p + 2, bits_of(a[2]), bits_of(a[3]), tag);
}
// four granules whose tags must read `tag`: polls (bounded) until they do. false = gave up (abort raised)
__device__ __forceinline__ bool gran_get4(const gran_t* p, unsigned tag, f4& out, int* abort_flag) {
    unsigned spins = 0;
    for (;;) {
        const gran_t g0 = gran_load(p), g1 = gran_load(p + 1), g2 = gran_load(p + 2), g3 = gran_load(p + 3);
        if ((unsigned)(g0 >> 32) == tag && (unsigned)(g1 >> 32) == tag && (unsigned)(g2 >> 32) == tag && (unsigned)(g3 >> 32) == tag) {
            out[0] = float_of((unsigned)g0); out[1] = float_of((unsigned)g1);
            out[2] = float_of((unsigned)g2); out[3] = float_of((unsigned)g3);
            return true;
        }
        if (++spins > kResSpinLimit || ((spins & 255u) == 0 && res_load_flag(abort_flag))) {
            res_raise_flag(abort_flag);
            out = f4_zero();
            return false;
        }
        res_pause();
    }
}

// the same vector of NARR arrays (stride `astride` granules): ALL loads are issued before the first tag is looked at -- one memory round trip
// per attempt instead of one per array
template <int NARR>
__device__ __forceinline__ bool gran_get4n(const gran_t* p, size_t astride, unsigned tag, f4 (&out)[NARR], int* abort_flag) {
    unsigned spins = 0;
    for (;;) {
        gran_t g[NARR][4];
#ifdef __HIP_DEVICE_COMPILE__
        u4 q[NARR][2];
        // (all loads and the wait in ONE asm block: its outputs are complete when the block ends, whatever the register allocator does with them)
        if constexpr (NARR == 3) {
            asm volatile("global_load_dwordx4 %0, %6, off sc1\n\tglobal_load_dwordx4 %1, %6, off offset:16 sc1\n\t"
                         "global_load_dwordx4 %2, %7, off sc1\n\tglobal_load_dwordx4 %3, %7, off offset:16 sc1\n\t"
                         "global_load_dwordx4 %4, %8, off sc1\n\tglobal_load_dwordx4 %5, %8, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(q[0][0]), "=&v"(q[0][1]), "=&v"(q[1][0]), "=&v"(q[1][1]), "=&v"(q[2][0]), "=&v"(q[2][1])
                         : "v"(p), "v"(p + astride), "v"(p + 2 * astride) : "memory");
        } else {
            asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(q[0][0]), "=&v"(q[0][1]) : "v"(p) : "memory");
        }
#pragma unroll
        for (int a = 0; a < NARR; ++a)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                // (the asm loads are invisible to the compiler's wait-count pass: the values are used only behind the explicit wait above)
                const u4 w = q[a][h];
                g[a][2 * h] = ((gran_t)w[1] << 32) | w[0];
                g[a][2 * h + 1] = ((gran_t)w[3] << 32) | w[2];
            }
#else
#pragma unroll
        for (int a = 0; a < NARR; ++a)
#pragma unroll
            for (int e = 0; e < 4; ++e) g[a][e] = gran_load(p + a * astride + e);
#endif
        bool good = true;
#pragma unroll
        for (int a = 0; a < NARR; ++a)
#pragma unroll
            for (int e = 0; e < 4; ++e) good = good && (unsigned)(g[a][e] >> 32) == tag;
        if (good) {
#pragma unroll
            for (int a = 0; a < NARR; ++a)
#pragma unroll
                for (int e = 0; e < 4; ++e) out[a][e] = float_of((unsigned)g[a][e]);
            return true;
        }
        if (++spins > kResSpinLimit || ((spins & 255u) == 0 && res_load_flag(abort_flag))) {
            res_raise_flag(abort_flag);
#pragma unroll
            for (int a = 0; a < NARR; ++a) out[a] = f4_zero();
            return false;
        }
        res_pause();
    }
}

template <int VPT, bool FLAGS = false>
__global__ __launch_bounds__(kResBlock) void cg_resident_kernel(ResArgs A) {
    constexpr int LS = 256 * VPT + 8;                 // LDS row stride; cell j sits at column 4 + j (vectors stay 16-byte aligned)
    PHIHIP_DYNAMIC_LDS(unsigned char, lds_raw);
    float* const L = reinterpret_cast<float*>(lds_raw);                                        // kResRows + 2 rows of the stencil source
    double* const red = reinterpret_cast<double*>(lds_raw + (size_t)(kResRows + 2) * LS * sizeof(float));   // [5][kResRows] this workgroup's sums per row
    double* const bc = red + 5 * kResRows;                                                     // [8] broadcast of the reduced sums
    int* const bci = reinterpret_cast<int*>(bc + 8);
    // the control block lives in LDS between iterations (two slots: every thread advances a copy of slot `cur`, thread 0 stores the result into
    // the other one): 24 registers per thread that the iteration body does not have to carry
    CgState* const stl = reinterpret_cast<CgState*>(bc + 10);
    // the neighbours' raw boundary rows [side][array][LS], fetched by the edge wavefronts WHILE other wavefronts poll the sums (each thread
    // reads back only what it wrote itself: no barrier in between)
    float* const hraw = reinterpret_cast<float*>(stl + 2);

    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const int G = A.G, n1 = A.n1, n2 = A.n2;
    const int b = (int)(blockIdx.x % (unsigned)A.batch), g = (int)(blockIdx.x / (unsigned)A.batch);   // entries of a batch of 8 sit on one XCD each
    const int row0 = g * kResRows;
    const int rows_here = min(kResRows, n1 - row0);
    const bool row_ok = wave < rows_here;
    const int lr = wave + 1;                            // LDS row of this wavefront's row
    const long long base = (long long)b * A.cells;
    const long long rowoff = base + (long long)(row0 + (row_ok ? wave : 0)) * n2;

    const unsigned tag_hi = (*A.solve_ctr & 0xFFFu) << 20;      // uniform (scalar load): the launch's solve number, bumped by res_begin_kernel in front of it
    bool ok[VPT], in_row[VPT], zl[VPT], zr[VPT];
    int j[VPT], jl[VPT], jr[VPT];
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        j[v] = (v * kWave + lane) * 4;
        in_row[v] = j[v] < n2;
        ok[v] = row_ok && in_row[v];
        zl[v] = zr[v] = false;
        jl[v] = nb_index(j[v] - 1, n2, A.nb2_lo, A.nb2_hi, zl[v]);
        jr[v] = nb_index(j[v] + 4, n2, A.nb2_lo, A.nb2_hi, zr[v]);
        if (!in_row[v]) { jl[v] = jr[v] = 0; j[v] = 0; }
    }
    // r6: cell flags (obstacles / `active` masks): the four flag bytes of every vector of the thread stay in ONE register for the whole solve. The neighbour VALUES
    // follow the boundary rule as without flags; whether a face carries flux, and whether the cell is solved at all, is the flag's (stencil_march.hpp, FLAGS form)
    unsigned fl[VPT];
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        fl[v] = 0u;
        if (FLAGS && ok[v]) fl[v] = *reinterpret_cast<const unsigned*>(A.flags + (long long)b * A.flag_bstride + (long long)(row0 + wave) * n2 + j[v]);
    }
    // y - shift on the ACTIVE cells only (fluid._balance_divergence subtracts the mean over the active cells from them)
    auto shifted = [&](int v, f4 y, float sh) -> f4 {
        if (!FLAGS) return y - f4_splat(sh);
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] -= ((fl[v] >> (8 * e)) & 64u) ? sh : 0.f;
        return y;
    };
    // the rows above / below this workgroup: 0 = the neighbouring workgroup's published row, 1 = clamp (own edge row), 2 = zero ghost
    int up_kind = 0, dn_kind = 0, up_g = g - 1, dn_g = g + 1;
    if (g == 0) { up_g = G - 1; up_kind = A.nb1_lo == NB_WRAP ? 0 : (A.nb1_lo == NB_CLAMP ? 1 : 2); }
    if (g == G - 1) { dn_g = 0; dn_kind = A.nb1_hi == NB_WRAP ? 0 : (A.nb1_hi == NB_CLAMP ? 1 : 2); }
    const bool first_row = wave == 0, last_row = wave == rows_here - 1;
    bool aborted = false;
    if (tid == 6) bci[1] = 0;         // "a poller of this workgroup gave up" (sticky; the barrier of the initial residual lies before the first use)

    auto pub_ptr = [&](int slot, int gg, int side, int arr) -> gran_t* {
        return A.pub + ((((size_t)slot * A.batch + b) * G + gg) * 2 + side) * 3 * (size_t)A.ns + (size_t)arr * A.ns;
    };
    auto part_ptr = [&](int slot, int gg) -> gran_t* { return A.part + (((size_t)slot * A.batch + b) * G + gg) * 10; };

    // vector v of this workgroup's first / last row -> the slot the neighbours read in phase `ph`
    auto publish1 = [&](unsigned ph, int v, f4 a0) {
        if (!in_row[v]) return;
        const unsigned tag = tag_hi | ph;
        if (first_row) gran_put4(pub_ptr(ph & 1, g, 0, 0) + j[v], a0, tag);
        if (last_row) gran_put4(pub_ptr(ph & 1, g, 1, 0) + j[v], a0, tag);
    };
    auto publish3 = [&](unsigned ph, int v, f4 a0, f4 a1, f4 a2) {
        if (!in_row[v]) return;
        const unsigned tag = tag_hi | ph;
        if (first_row) {
            gran_put4(pub_ptr(ph & 1, g, 0, 0) + j[v], a0, tag);
            gran_put4(pub_ptr(ph & 1, g, 0, 1) + j[v], a1, tag);
            gran_put4(pub_ptr(ph & 1, g, 0, 2) + j[v], a2, tag);
        }
        if (last_row) {
            gran_put4(pub_ptr(ph & 1, g, 1, 0) + j[v], a0, tag);
            gran_put4(pub_ptr(ph & 1, g, 1, 1) + j[v], a1, tag);
            gran_put4(pub_ptr(ph & 1, g, 1, 2) + j[v], a2, tag);
        }
    };
    // edge wavefronts: the neighbours' rows of phase `ph` (polled until their tags show it) -> hraw; three arrays (r, w, s) or one
    auto fetch_side = [&](unsigned ph, int gg, int side_of_neighbour, int slot_side, int v, bool three) {
        const unsigned tag = tag_hi | ph;
        const gran_t* q = pub_ptr(ph & 1, gg, side_of_neighbour, 0) + j[v];
        if (three) {
            f4 h[3];
            (void)gran_get4n<3>(q, (size_t)A.ns, tag, h, A.abort_flag);      // (a failed wait raised `abort`: get_sums reports it)
            f4_store(hraw + (slot_side * 3 + 0) * LS + j[v], h[0]);
            f4_store(hraw + (slot_side * 3 + 1) * LS + j[v], h[1]);
            f4_store(hraw + (slot_side * 3 + 2) * LS + j[v], h[2]);
        } else {
            f4 h[1];
            (void)gran_get4n<1>(q, (size_t)A.ns, tag, h, A.abort_flag);
            f4_store(hraw + (slot_side * 3 + 0) * LS + j[v], h[0]);
        }
    };
    auto prefetch = [&](unsigned ph, bool three) {
        if (!(first_row || last_row)) return;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            if (!in_row[v]) continue;
            if (first_row && up_kind == 0) fetch_side(ph, up_g, 1, 0, v, three);
            if (last_row && dn_kind == 0) fetch_side(ph, dn_g, 0, 1, v, three);
        }
    };
    // stencil source S of vector v of the own row -> LDS, and (first / last wavefront) of the row above / below the workgroup from the
    // prefetched rows. combine: h = a0 + ca1 a1 + ca2 a2 (the expression of the own cells)
    auto halo = [&](int side, int v, f4 S, int kind, bool combine, float ca1, float ca2) -> f4 {
        if (kind == 1) return S;                                  // clamp: the own edge row
        if (kind == 2) return f4_zero();
        f4 h = f4_load(hraw + (side * 3 + 0) * LS + j[v]);
        if (combine) h = f4_fma(ca2, f4_load(hraw + (side * 3 + 2) * LS + j[v]), f4_fma(ca1, f4_load(hraw + (side * 3 + 1) * LS + j[v]), h));
        return h;
    };
    auto stage = [&](int v, f4 S, bool combine, float ca1, float ca2) {
        if (ok[v]) f4_store(L + lr * LS + 4 + j[v], S);
        if (!in_row[v]) return;
        if (first_row) f4_store(L + 4 + j[v], halo(0, v, S, up_kind, combine, ca1, ca2));
        if (last_row) f4_store(L + (rows_here + 1) * LS + 4 + j[v], halo(1, v, S, dn_kind, combine, ca1, ca2));
    };
    // (ident I + w1 d^2_1 + w2 d^2_2) S for vector v from the staged rows, flux form like march_kernel (stencil_march.hpp)
    auto apply = [&](int v, f4 S) -> f4 {
        const f4 up = f4_load(L + (lr - 1) * LS + 4 + j[v]), dn = f4_load(L + (lr + 1) * LS + 4 + j[v]);
        const float lf = zl[v] ? 0.f : L[lr * LS + 4 + jl[v]];
        const float rt = zr[v] ? 0.f : L[lr * LS + 4 + jr[v]];
        f4 q;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float c = S[e];
            const float lo2 = e > 0 ? S[e > 0 ? e - 1 : 0] : lf;
            const float hi2 = e < 3 ? S[e < 3 ? e + 1 : e] : rt;
            if (FLAGS) {
                const unsigned f = (fl[v] >> (8 * e)) & 0xFFu;
                float acc = 0.f;
                if (f & 4u) acc += (up[e] - c) * A.w1;
                if (f & 8u) acc += (dn[e] - c) * A.w1;
                if (f & 16u) acc += (lo2 - c) * A.w2;
                if (f & 32u) acc += (hi2 - c) * A.w2;
                acc = fmaf(A.ident, c, acc);
                q[e] = (f & 64u) ? acc : c;        // inactive cell: identity row (fluid.py:202)
                continue;
            }
            const float t2 = ((hi2 - c) - (c - lo2)) * A.w2;
            const float t1 = ((dn[e] - c) - (c - up[e])) * A.w1;
            q[e] = fmaf(A.ident, c, t1 + t2);
        }
        return q;
    };
    // this workgroup's five partial sums of phase `ph` -> granules (two per double)
    auto put_partials = [&](unsigned ph, double a0, double a1, double a2, double a3, double a4) {
        a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3); a4 = wave_sum(a4);
        if (lane == 0) {
            red[0 * kResRows + wave] = a0; red[1 * kResRows + wave] = a1; red[2 * kResRows + wave] = a2;
            red[3 * kResRows + wave] = a3; red[4 * kResRows + wave] = a4;
        }
        __syncthreads();
        if (tid < 5) {
            double t = 0;
            for (int ww = 0; ww < kResRows; ++ww) t += red[tid * kResRows + ww];
            const unsigned long long bits = bits_of(t);
            gran_t* q = part_ptr(ph & 1, g) + 2 * tid;
            gran_store(q, (unsigned)bits, tag_hi | ph);
            gran_store(q + 1, (unsigned)(bits >> 32), tag_hi | ph);
        }
    };
    // the entry's sums of phase `ph`, added in a fixed order (every workgroup forms the same bits). This all-to-all is what orders the phases
    // (see the header). true = the launch was aborted
    double sum0 = 0, sum1 = 0, sum2 = 0, sum3 = 0, sum4 = 0;
    auto get_sums = [&](unsigned ph) -> bool {
        // the pollers are wavefronts 8 .. 12 (sum k = wavefront - 8, workgroup gg = lane): the edge wavefronts (0 and, in a full workgroup,
        // 15) fetch rows meanwhile. Added up by the wavefront's shuffle tree -- the same order in every workgroup
        if (wave >= 8 && wave < 13) {
            const int k = wave - 8;
            double val = 0;
            if (lane < G) {
                const gran_t* q = part_ptr(ph & 1, lane) + 2 * k;
                const unsigned tag = tag_hi | ph;
                unsigned spins = 0;
                for (;;) {
                    const gran_t lo = gran_load(q), hi = gran_load(q + 1);
                    if ((unsigned)(lo >> 32) == tag && (unsigned)(hi >> 32) == tag) {
                        val = double_of(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
                        break;
                    }
                    if (++spins > kResSpinLimit || ((spins & 255u) == 0 && res_load_flag(A.abort_flag))) {
                        // (thread 5 may have read the global flag before this poller gave up: the workgroup's own copy is what the barrier
                        // below orders -- without it a sum that lacks this lane's share could pass for the entry's sum)
                        res_raise_flag(A.abort_flag);
                        bci[1] = 1;
                        break;
                    }
                    res_pause();
                }
            }
            val = wave_sum(val);
            if (lane == 0) bc[k] = val;
        }
        if (tid == 5) bci[0] = res_load_flag(A.abort_flag);
        __syncthreads();
        sum0 = bc[0]; sum1 = bc[1]; sum2 = bc[2]; sum3 = bc[3]; sum4 = bc[4];
        return (bci[0] | bci[1]) != 0;
    };

    f4 x[VPT], r[VPT], w[VPT], s[VPT], p[VPT];
    int cur = 0;
    bool cont = false;
    unsigned ph = 0;                  // phase number: publish (rows, sums) of phase ph -> everybody's sums of ph -> neighbours' rows of ph
    const float yshift = A.shift ? (float)A.shift[b] : 0.f;

    // ---- r = y - A x0 (neighbour rows of x0 straight from the input array), |r|^2, |y|^2 ---------------------------------------------------
    {
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            x[v] = ok[v] ? f4_load(A.x + rowoff + j[v]) : f4_zero();
            w[v] = s[v] = p[v] = f4_zero();
            if (ok[v]) f4_store(L + lr * LS + 4 + j[v], x[v]);
        }
        if (first_row || last_row) {
            bool zu = false, zd = false;
            const int iu = nb_index(row0 - 1, n1, A.nb1_lo, A.nb1_hi, zu), id = nb_index(row0 + rows_here, n1, A.nb1_lo, A.nb1_hi, zd);
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                if (!in_row[v]) continue;
                if (first_row) f4_store(L + 4 + j[v], zu ? f4_zero() : f4_load(A.x + base + (long long)iu * n2 + j[v]));
                if (last_row) f4_store(L + (rows_here + 1) * LS + 4 + j[v], zd ? f4_zero() : f4_load(A.x + base + (long long)id * n2 + j[v]));
            }
        }
        __syncthreads();
        double a0 = 0, a1 = 0;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const f4 q = apply(v, x[v]);
            r[v] = f4_zero();
            if (!ok[v]) continue;
            f4 y = shifted(v, f4_load(A.y + rowoff + j[v]), yshift);
            r[v] = y - q;
            if (A.yout) f4_store(A.yout + rowoff + j[v], y);
            a0 += (double)f4_dot(r[v], r[v]);
            a1 += (double)f4_dot(y, y);
        }
        ++ph;
#pragma unroll
        for (int v = 0; v < VPT; ++v) publish1(ph, v, r[v]);
        put_partials(ph, a0, a1, 0, 0, 0);
        prefetch(ph, false);
        aborted = get_sums(ph);
        const CgState st = cg_advance(PRO_FIRST, CgState(), sum0, sum1, A.prm);
        if (tid == 0) stl[0] = st;
        cont = st.cont != 0;
    }
    // w = A r with gamma = |r|^2, delta = (A r).r and, against the standing p and s, mu = r.s, nu = (A r).p, sigma = p.s (start and refresh);
    // r's boundary rows were published in phase ph
    auto w_and_sums = [&]() {
#pragma unroll
        for (int v = 0; v < VPT; ++v) stage(v, r[v], false, 0.f, 0.f);
        __syncthreads();
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            w[v] = apply(v, r[v]);
            if (!ok[v]) continue;
            a0 += (double)f4_dot(r[v], r[v]);
            a1 += (double)f4_dot(w[v], r[v]);
            a2 += (double)f4_dot(r[v], s[v]);
            a3 += (double)f4_dot(w[v], p[v]);
            a4 += (double)f4_dot(p[v], s[v]);
        }
        ++ph;
#pragma unroll
        for (int v = 0; v < VPT; ++v) publish3(ph, v, r[v], w[v], s[v]);
        put_partials(ph, a0, a1, a2, a3, a4);
        prefetch(ph, true);
        aborted = get_sums(ph) || aborted;
    };
    if (cont && !aborted) w_and_sums();

    // ---- the iterations --------------------------------------------------------------------------------------------------------------------
    if (cont && !aborted) {
        for (int k = 1; k <= A.prm.max_iter; ++k) {
            float alpha, beta, ca1, ca2;
            {
                const CgState st = cg_advance(PRO_CG1, stl[cur], sum0, sum1, A.prm, sum2, sum3, sum4);
                if (tid == 0) stl[cur ^ 1] = st;
                cur ^= 1;
                cont = st.cont != 0;
                alpha = (float)st.alpha; beta = (float)st.beta;
                ca1 = (float)(-st.alpha); ca2 = (float)(-st.alpha * st.beta);      // r_new = r - alpha w - alpha beta s
            }
            if (!cont) break;
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                const f4 rn = f4_fma(ca2, s[v], f4_fma(ca1, w[v], r[v]));
                p[v] = f4_fma(beta, p[v], r[v]);          // p = r + beta p
                s[v] = f4_fma(beta, s[v], w[v]);          // s = w + beta s  (= A p)
                x[v] = f4_fma(alpha, p[v], x[v]);         // x += alpha p
                r[v] = rn;
                stage(v, rn, true, ca1, ca2);             // (the neighbours' prefetched r, w, s -> their r_new)
            }
            __syncthreads();
            double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                w[v] = apply(v, r[v]);                    // w = A r_new (the old w went into s)
                if (!ok[v]) continue;
                a0 += (double)f4_dot(r[v], r[v]);         // gamma' = |r_new|^2
                a1 += (double)f4_dot(w[v], r[v]);         // delta' = (A r_new).r_new
                a2 += (double)f4_dot(r[v], s[v]);         // mu'    = r_new.s
                a3 += (double)f4_dot(w[v], p[v]);         // nu'    = (A r_new).p
                a4 += (double)f4_dot(p[v], s[v]);         // sigma  = p.s
            }
            ++ph;
#pragma unroll
            for (int v = 0; v < VPT; ++v) publish3(ph, v, r[v], w[v], s[v]);
            put_partials(ph, a0, a1, a2, a3, a4);
            prefetch(ph, true);
            aborted = get_sums(ph);
            if (aborted) break;
            if (A.refresh_every > 0 && k % A.refresh_every == 0) {
                // true residual like PhiML: r = y - A x, then w, gamma, delta, mu, nu, sigma from it REPLACE the sums of iteration k
                ++ph;
#pragma unroll
                for (int v = 0; v < VPT; ++v) publish1(ph, v, x[v]);
                put_partials(ph, 0, 0, 0, 0, 0);          // (every phase carries the all-to-all that orders the slots)
                prefetch(ph, false);
                aborted = get_sums(ph);
#pragma unroll
                for (int v = 0; v < VPT; ++v) stage(v, x[v], false, 0.f, 0.f);
                __syncthreads();
#pragma unroll
                for (int v = 0; v < VPT; ++v) {
                    const f4 q = apply(v, x[v]);
                    if (!ok[v]) continue;
                    const f4 y = f4_load((A.yout ? A.yout : A.y) + rowoff + j[v]);      // (balanced by the first pass when a shift was given)
                    r[v] = (A.yout ? y : shifted(v, y, yshift)) - q;
                }
                ++ph;
#pragma unroll
                for (int v = 0; v < VPT; ++v) publish1(ph, v, r[v]);
                put_partials(ph, 0, 0, 0, 0, 0);          // (its __syncthreads also separates the reads of the staged x rows from the staging of r)
                prefetch(ph, false);
                aborted = get_sums(ph) || aborted;
                w_and_sums();
                if (aborted) break;
            }
        }     // (the __syncthreads of put_partials / get_sums separate this iteration's reads of the staged rows from the next staging)
    }
    __syncthreads();                                                     // (thread 0's last store of the control block)
    CgState st = stl[cur];
    if (!aborted) st = cg_advance(PRO_BETA, st, sum0, sum1, A.prm);      // the last (gamma, delta): converged / diverged / residual of the final iterate
#pragma unroll
    for (int v = 0; v < VPT; ++v)
        if (ok[v]) f4_store(A.x + rowoff + j[v], x[v]);
    if (g == 0 && tid == 0) {
        if (aborted) {      // -1: a wait gave up (the caller reports it)
            st.diverged = 1; st.converged = 0; st.cont = 0; st.iterations = -1;
            if (A.abort_host) {
#ifdef __HIP_DEVICE_COMPILE__
                __hip_atomic_store(A.abort_host, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#else
                *reinterpret_cast<volatile int*>(A.abort_host) = 1;
#endif
            }
        }
        A.st_out[b] = st;
    }
}

// In front of every resident launch (one workgroup; a node of the graph when the solve is captured, so every REPLAY gets a fresh number too -- until r5 the
// number was a kernel argument written by the host, a replay would have reused the capture's tags, and the resident solver was refused under capture):
// solve number += 1 (12 bits, 0 skipped: a zeroed buffer carries tag 0), the launch's abort flag cleared; when the number wraps, the granule buffers are
// cleared (a granule 4096 solves old could otherwise pass for a fresh one: ~13 MB for 8 x 512^2, once per 4095 solves).
__global__ __launch_bounds__(kResBlock) void res_begin_kernel(unsigned* ctr, int* abort_flag, unsigned long long* gran, size_t gran_words) {
    __shared__ unsigned next;
    if (threadIdx.x == 0) {
        unsigned n = (*ctr + 1u) & 0xFFFu;
        next = n;
        *ctr = n == 0u ? 1u : n;
        *abort_flag = 0;
    }
    __syncthreads();
    if (next == 0u)
        for (size_t i = threadIdx.x; i < gran_words; i += blockDim.x) gran[i] = 0ull;
}

static size_t resident_lds_bytes(int vpt) {
    const size_t ls = 256 * (size_t)vpt + 8;
    return (kResRows + 2) * ls * sizeof(float) + (5 * kResRows + 10) * sizeof(double) + 2 * sizeof(CgState) + 6 * ls * sizeof(float) + 16;
}

#if defined(__HIPCC__)
static int resident_blocks_per_cu(const phihip_ctx* ctx, int vpt, bool flags) {
    static int cached[16][3][2] = {{{0}}};
    int& c = cached[ctx->device >= 0 && ctx->device < 16 ? ctx->device : 0][vpt][flags ? 1 : 0];
    if (c == 0) {
        int n = 0;
        const size_t lds = resident_lds_bytes(vpt);
        hipError_t e;
        if (flags) e = vpt == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, cg_resident_kernel<1, true>, kResBlock, lds)
                                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, cg_resident_kernel<2, true>, kResBlock, lds);
        else e = vpt == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, cg_resident_kernel<1, false>, kResBlock, lds)
                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, cg_resident_kernel<2, false>, kResBlock, lds);
        c = (e == hipSuccess && n > 0) ? n : -1;
    }
    return c > 0 ? c : 0;
}
#endif

// How many batch entries ONE resident launch can take (0: the solver does not apply): 2-D fp32 'CG' -- r6: with or without cell flags --, rows of whole vectors up to 512
// cells; a launch has to be resident as a whole, i.e. entries x G workgroups <= what the occupancy calculator grants per CU x CUs (1024 threads at <= 128 VGPRs: one per CU
// today -- asked, not assumed, so that a compiler that needs more registers makes the solver fall back instead of stalling for ~1 s). r6: a batch of MORE entries runs as
// several launches one behind the other (entries are independent solves; in tolerance mode every launch ends when ITS entries have converged).
static int resident_entries_per_launch(const phihip_ctx* ctx, const GridView& v, const uint8_t* flags, const phihip_solve* solve) {
    if (v.rank != 2 || v.dtype != PHIHIP_F32 || v.unaligned || v.halo[0] || v.halo[1]) return 0;
    if (flags && ((uintptr_t)flags & 3u)) return 0;      // (the four flag bytes of a vector are read as one word)
    if (solve->method != PHIHIP_METHOD_CG || solve->max_iterations > 200000) return 0;      // (the phase number has 20 bits of the tag: up to 4 phases per iteration with refresh_every = 1)
    if (v.n[2] % 4 != 0 || v.n[2] > 512 || v.n[1] < 2) return 0;      // (rows up to 1024 cells would need VPT = 4: 80 state registers, spills at 128)
    const long long G = (v.n[1] + kResRows - 1) / kResRows;
    if (G > kResMaxG) return 0;      // (one lane per workgroup adds the entry's partial sums up: taller grids keep the launch forms)
#if defined(__HIPCC__)
    const long long capacity = (long long)resident_blocks_per_cu(ctx, v.n[2] <= 256 ? 1 : 2, flags != nullptr) * ctx->num_cu;
#else
    (void)ctx;
    const long long capacity = 16;    // the emulation of the tests keeps any grid "resident" (fibers); bounded by its memory: 1024 fibers of 256 KB stack per block
#endif
    const long long per = capacity / G;
    return (int)(per < v.batch ? per : v.batch);
}

bool cg_resident_applicable(const phihip_ctx* ctx, const GridView& v, const uint8_t* flags, const phihip_solve* solve) {
    return resident_entries_per_launch(ctx, v, flags, solve) >= 1;
}

// Mode 1's judgement of a batch that needs SEVERAL launches: worth it only where one launch is a chip-filling load of large grids (8 x 512^2: the launch forms cost >= 14 us
// per iteration for those eight). Measured, us per iteration launch forms -> sub-batches (profiles/r06_sweep_resident_subbatches.txt): 16 x 512^2 24.5 -> 19.7, 32 x 512^2
// 49.1 -> 39.6, 64 x 512^2 87.9 -> 79.2 (tolerance solves 1.25-1.4 x faster: a launch ends when ITS entries have converged) -- but 32 x 256^2 14.4 -> 16.1 and 64 x 256^2
// 24.8 -> 32.3: sixteen 256^2 entries per launch do not fill the chip's bandwidth, the launch forms over the whole batch do.
bool cg_resident_batch_pays(const phihip_ctx* ctx, const GridView& v, const uint8_t* flags, const phihip_solve* solve) {
    const int per = resident_entries_per_launch(ctx, v, flags, solve);
    if (per < 1) return false;
    return per >= v.batch || (long long)per * v.cells >= (2LL << 20);
}

int run_cg_resident(phihip_ctx* ctx, const GridView& v, const uint8_t* flags, int mask_batch, const void* rhs, void* x, const phihip_solve* solve, void* st_out,
                    const double* shift, hipStream_t s) {
    const int per = resident_entries_per_launch(ctx, v, flags, solve);
    if (per < 1) { set_error("cg (resident): the solve does not fit"); return PHIHIP_ERR_UNSUPPORTED; }
    if (per < v.batch) {      // r6: more entries than one launch holds -- sub-batches one behind the other on the stream (the exchange buffers are reused: stream order)
        for (int b0 = 0; b0 < v.batch; b0 += per) {
            GridView vc = v;
            vc.batch = v.batch - b0 < per ? v.batch - b0 : per;
            const size_t off = (size_t)b0 * v.cells;
            PHIHIP_TRY(run_cg_resident(ctx, vc, flags ? flags + (mask_batch > 1 ? off : 0) : nullptr, mask_batch > 1 ? vc.batch : 1, (const float*)rhs + off, (float*)x + off,
                                       solve, (CgState*)st_out + b0, shift ? shift + b0 : nullptr, s));
        }
        return PHIHIP_OK;
    }
    const int G = (v.n[1] + kResRows - 1) / kResRows;
    const int vpt = v.n[2] <= 256 ? 1 : 2;
    ResArgs A;
    memset(&A, 0, sizeof(A));
    A.n1 = v.n[1]; A.n2 = v.n[2]; A.G = G; A.batch = v.batch; A.ns = 256 * vpt;
    if (G > kResMaxG) { set_error("cg (resident): more than %d workgroups per entry", kResMaxG); return PHIHIP_ERR_UNSUPPORTED; }
    A.cells = v.cells;
    int rule[3][2];
    for (int ax = 1; ax < 3; ++ax)
        for (int side = 0; side < 2; ++side) {
            const int code = v.bc[ax][side];
            rule[ax][side] = v.op_custom ? v.op_rule[ax][side] : (code == PHIHIP_BC_PERIODIC ? NB_WRAP : (code == PHIHIP_BC_CLOSED ? NB_CLAMP : NB_ZERO));
        }
    A.nb1_lo = rule[1][0]; A.nb1_hi = rule[1][1]; A.nb2_lo = rule[2][0]; A.nb2_hi = rule[2][1];
    const double sc = v.op_custom ? v.op_scale : 1.0;
    A.w1 = (float)(sc / (v.dx[1] * v.dx[1])); A.w2 = (float)(sc / (v.dx[2] * v.dx[2]));
    A.ident = (float)(v.op_custom ? v.op_ident : 0.0);
    A.flags = flags;
    A.flag_bstride = mask_batch > 1 ? v.cells : 0;
    A.y = (const float*)rhs;
    A.yout = shift ? (float*)const_cast<void*>(rhs) : nullptr;
    A.shift = shift;
    A.x = (float*)x;
    const size_t pub_bytes = (size_t)2 * v.batch * G * 2 * 3 * A.ns * sizeof(gran_t);
    const size_t part_bytes = (size_t)2 * v.batch * G * 10 * sizeof(gran_t);
    // layout: [control block: abort flag at 0, solve number at 64 | published rows | partial sums]. The control block sits at a FIXED place (until the last session of r6 it
    // followed the granule areas, whose size depends on the batch: a context that alternated between two batch sizes -- the sub-batches above do, systematically -- read its
    // solve number from inside the other layout's granules). tags = (solve number, phase): granules of an earlier solve or of another layout never match, the number only
    // grows; a NEW buffer is zeroed (counter included), and when the 12-bit number wraps the WHOLE buffer's granules are cleared (res_begin_kernel).
    const size_t ctl_bytes = 256;
    const size_t had = ctx->ws_res.ptr ? ctx->ws_res.bytes : 0;
    PHIHIP_TRY(ensure_buffer(ctx->ws_res, ctl_bytes + pub_bytes + part_bytes));
    if (ctx->ws_res.bytes != had) PHIHIP_CHECK_HIP(hipMemsetAsync(ctx->ws_res.ptr, 0, ctx->ws_res.bytes, s));
    char* ws = (char*)ctx->ws_res.ptr;
    A.abort_flag = (int*)ws;
    A.solve_ctr = (const unsigned*)(ws + 64);
    A.pub = (gran_t*)(ws + ctl_bytes);
    A.part = (gran_t*)(ws + ctl_bytes + pub_bytes);
    hipLaunchKernelGGL(res_begin_kernel, dim3(1), dim3(kResBlock), 0, s, (unsigned*)(ws + 64), A.abort_flag, (unsigned long long*)(ws + ctl_bytes),
                       (ctx->ws_res.bytes - ctl_bytes) / sizeof(gran_t));
    PHIHIP_TRY(ensure_adv_host_public(ctx));
    A.abort_host = ctx->adv_host_dev + 15;
    A.st_out = (CgState*)st_out;
    A.prm.rtol = solve->rel_tol; A.prm.atol = solve->abs_tol; A.prm.max_iter = solve->max_iterations; A.prm.pad = 0;
    A.refresh_every = solve->refresh_every;
    const size_t lds = resident_lds_bytes(vpt);
    const dim3 grid((unsigned)(G * v.batch)), block(kResBlock);
    LaunchScope ls(ctx, PHIHIP_K_CG_UPDATE, s);
#if defined(__HIPCC__)
    // Resident solves of ONE process never overlap (r6): each waits for the event of the one before it, whatever the stream or context -- two launches that each hold half of
    // the CUs and wait for the rest is the one way two resident grids that fit the chip one at a time can stall each other (bounded: ~1 s, then PHIHIP_ERR_HIP). Not
    // under capture (an external event would become a node of the graph; replays on different streams are the application's to order) and not across processes.
    static std::mutex chain_mutex;
    static hipEvent_t chain_event[16] = {nullptr};
    static hipStream_t chain_stream[16] = {nullptr};
    static bool chain_any[16] = {false};
    const bool chained = !stream_is_capturing(s) && ctx->device >= 0 && ctx->device < 16;
    if (chained) {
        std::lock_guard<std::mutex> lock(chain_mutex);
        hipEvent_t& ev = chain_event[ctx->device];
        if (!ev) PHIHIP_CHECK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        else if (chain_any[ctx->device] && chain_stream[ctx->device] != s) PHIHIP_CHECK_HIP(hipStreamWaitEvent(s, ev, 0));
    }
    // (the optional) COOPERATIVE launch -- the runtime checks that the whole grid can be co-resident (fails cleanly otherwise: the caller falls back to the launch
    // forms) and runs cooperative kernels of a device one after the other, so two resident solves of different streams cannot each hold half of the CUs
    // and wait for the rest (until r5 co-residency was an assumption about an otherwise idle device, and the solver opt-in for that reason). The kernel itself
    // is unchanged: granule exchange, no grid barrier. ctx->res_coop = 0 (PHIHIP_RESIDENT_COOP=0): the plain launch.
    if (ctx->res_coop && (ctx->res_coop_capture || !stream_is_capturing(s))) {
        void* params[1] = {(void*)&A};
        const void* fn = flags ? (vpt == 1 ? (const void*)cg_resident_kernel<1, true> : (const void*)cg_resident_kernel<2, true>)
                               : (vpt == 1 ? (const void*)cg_resident_kernel<1, false> : (const void*)cg_resident_kernel<2, false>);
        const hipError_t e = hipLaunchCooperativeKernel(fn, grid, block, params, (unsigned)lds, s);
        if (e == hipErrorCooperativeLaunchTooLarge || e == hipErrorNotSupported) {
            (void)hipGetLastError();
            return PHIHIP_ERR_UNSUPPORTED;        // (cg.hip: the launch forms take the solve)
        }
        PHIHIP_CHECK_HIP(e);
    } else if (flags) {
        if (vpt == 1) hipLaunchKernelGGL((cg_resident_kernel<1, true>), grid, block, lds, s, A);
        else hipLaunchKernelGGL((cg_resident_kernel<2, true>), grid, block, lds, s, A);
    } else if (vpt == 1) hipLaunchKernelGGL((cg_resident_kernel<1, false>), grid, block, lds, s, A);
    else hipLaunchKernelGGL((cg_resident_kernel<2, false>), grid, block, lds, s, A);
    if (chained) {
        std::lock_guard<std::mutex> lock(chain_mutex);
        PHIHIP_CHECK_HIP(hipEventRecord(chain_event[ctx->device], s));
        chain_stream[ctx->device] = s;
        chain_any[ctx->device] = true;
    }
#else
    if (flags) {
        if (vpt == 1) hipemuLaunchResident((cg_resident_kernel<1, true>), grid, block, lds, s, A);
        else hipemuLaunchResident((cg_resident_kernel<2, true>), grid, block, lds, s, A);
    } else if (vpt == 1) hipemuLaunchResident((cg_resident_kernel<1, false>), grid, block, lds, s, A);
    else hipemuLaunchResident((cg_resident_kernel<2, false>), grid, block, lds, s, A);
#endif
    PHIHIP_CHECK_HIP(hipGetLastError());
    return PHIHIP_OK;
}

}  // namespace phihip
