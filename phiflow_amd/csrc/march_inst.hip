// march_inst.hip -- explicit instantiation + dispatch of march_kernel for one (element type, dimensionality) pair.
// Compiled four times (-DPHIHIP_INST_F64=0|1 -DPHIHIP_INST_DIM3=0|1) so the ~100 kernels per pair build in parallel.
#include "common.hpp"
#include "march_dispatch.hpp"

namespace phihip {

#if PHIHIP_INST_F64
using InstT = double;
#else
using InstT = float;
#endif
constexpr bool kInstDim3 = PHIHIP_INST_DIM3 != 0;
constexpr int kVmax = 16 / sizeof(InstT);

template <int V, int R, int TPR, int MODE, bool FLAGS, bool UNAL = false, bool ROWT = false>
static void launch_one(const MarchGrid& g, const MarchArgs<InstT>& a, dim3 grid, hipStream_t s) {
    if constexpr (UNAL || ROWT) {      // rows that are not whole vectors / unaligned buffers, and the row tile (stencil_march.hpp): one marching direction
        hipLaunchKernelGGL((march_kernel<InstT, V, R, TPR, MODE, FLAGS, kInstDim3, false, UNAL, ROWT>), grid, dim3(kBlock), 0, s, g, a);
    } else {
        // the bidirectional variant exists for 3-D MATVEC and UPDATE_R (the 3-word phases: the stencil source's halo planes weigh most there)
        constexpr bool mv = MODE == MODE_MATVEC || MODE == MODE_MATVEC_AD || MODE == MODE_UPDATE_R;
        if (mv && kInstDim3 && g.bidir)
            hipLaunchKernelGGL((march_kernel<InstT, V, R, TPR, MODE, FLAGS, kInstDim3, (mv && kInstDim3)>), grid, dim3(kBlock), 0, s, g, a);
        else
            hipLaunchKernelGGL((march_kernel<InstT, V, R, TPR, MODE, FLAGS, kInstDim3, false>), grid, dim3(kBlock), 0, s, g, a);
    }
}

template <int V, int R, int TPR, bool UNAL = false, bool ROWT = false>
static int launch_cfg(int mode, bool flags, const MarchGrid& g, const MarchArgs<InstT>& a, dim3 grid, hipStream_t s) {
#define PHIHIP_MODE_CASE(M)                                                \
    case M:                                                                \
        if (flags) launch_one<V, R, TPR, M, true, UNAL, ROWT>(g, a, grid, s);    \
        else launch_one<V, R, TPR, M, false, UNAL, ROWT>(g, a, grid, s);         \
        break;
    switch (mode) {
        PHIHIP_MODE_CASE(MODE_APPLY)
        PHIHIP_MODE_CASE(MODE_RESID)
        PHIHIP_MODE_CASE(MODE_MATVEC)
        PHIHIP_MODE_CASE(MODE_UPDATE)
        PHIHIP_MODE_CASE(MODE_MATVEC_AD)
        PHIHIP_MODE_CASE(MODE_UPDATE_AD)
        PHIHIP_MODE_CASE(MODE_UPDATE_R)
        PHIHIP_MODE_CASE(MODE_UPDATE_X2)
        PHIHIP_MODE_CASE(MODE_RESID_BAL)
        PHIHIP_MODE_CASE(MODE_APPLY_DOT)
        PHIHIP_MODE_CASE(MODE_CG1)
        default:
            set_error("march: bad mode %d", mode);
            return PHIHIP_ERR_BAD_ARG;
    }
#undef PHIHIP_MODE_CASE
    return PHIHIP_OK;
}

template <int V, int R, int TPR, int MODE, bool FLAGS, bool UNAL = false, bool ROWT = false>
static int occupancy_one() {
    // cached per process, not per device: occupancy is a property of (code object, architecture), and every device this library can run
    // on is a gfx950 with the same register file / LDS -- the devices of one node give the same answer
    static int cached = 0;
    if (cached == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, march_kernel<InstT, V, R, TPR, MODE, FLAGS, kInstDim3, false, UNAL, ROWT>, kBlock, 0) != hipSuccess || n < 1) n = 1;
        cached = n > 8 ? 8 : n;
    }
    return cached;
}

template <int V, int R, int TPR, bool UNAL = false, bool ROWT = false>
static int occupancy_cfg(int mode, bool flags) {
#define PHIHIP_OCC_CASE(M) \
    case M: return flags ? occupancy_one<V, R, TPR, M, true, UNAL, ROWT>() : occupancy_one<V, R, TPR, M, false, UNAL, ROWT>();
    switch (mode) {
        PHIHIP_OCC_CASE(MODE_APPLY)
        PHIHIP_OCC_CASE(MODE_RESID)
        PHIHIP_OCC_CASE(MODE_MATVEC)
        PHIHIP_OCC_CASE(MODE_UPDATE)
        PHIHIP_OCC_CASE(MODE_MATVEC_AD)
        PHIHIP_OCC_CASE(MODE_UPDATE_AD)
        PHIHIP_OCC_CASE(MODE_UPDATE_R)
        PHIHIP_OCC_CASE(MODE_UPDATE_X2)
        PHIHIP_OCC_CASE(MODE_RESID_BAL)
        PHIHIP_OCC_CASE(MODE_APPLY_DOT)
        PHIHIP_OCC_CASE(MODE_CG1)
        default: return 1;
    }
#undef PHIHIP_OCC_CASE
}

template <>
int march_occupancy<InstT, kInstDim3>(int id, int vec, int mode, bool flags) {
    if (vec == 1) return occupancy_cfg<1, 1, 64>(mode, flags);
    if (vec < 0) return occupancy_cfg<kVmax, 1, 64, true>(mode, flags);
    if (vec == 2 && kVmax == 4) {
        switch (id) {
            case 4: return occupancy_cfg<(kVmax == 4 ? 2 : kVmax), 4, 64>(mode, flags);
            case 6: return occupancy_cfg<(kVmax == 4 ? 2 : kVmax), 2, 64>(mode, flags);
            default: return occupancy_cfg<(kVmax == 4 ? 2 : kVmax), 1, 64>(mode, flags);
        }
    }
    switch (id) {
        case 0: return occupancy_cfg<kVmax, 1, 16>(mode, flags);
        case 1: return occupancy_cfg<kVmax, 2, 16>(mode, flags);
        case 2: return occupancy_cfg<kVmax, 2, 32>(mode, flags);
        case 3: return occupancy_cfg<kVmax, 4, 32>(mode, flags);
        case 4: return occupancy_cfg<kVmax, 4, 64>(mode, flags);
        case 5: return occupancy_cfg<kVmax, 1, 64>(mode, flags);
        case 6: return occupancy_cfg<kVmax, 2, 64>(mode, flags);
        case 7: return occupancy_cfg<kVmax, 1, 32>(mode, flags);
        case 8: return occupancy_cfg<kVmax, 1, 128, false, true>(mode, flags);
        case 9: return occupancy_cfg<kVmax, 2, 128, false, true>(mode, flags);
        case 10: return occupancy_cfg<kVmax, 4, 128, false, true>(mode, flags);
#if PHIHIP_INST_F64
        case 11: return occupancy_cfg<kVmax, 2, 256, false, true>(mode, flags);      // the WIDE row tiles (fp64 rows of 129 ... 256 vectors)
        case 12: return occupancy_cfg<kVmax, 4, 256, false, true>(mode, flags);
#endif
        default: return 1;
    }
}

// the slot that lanes outside the grid store into (march_kernel issues every store unconditionally); one per device and instantiation unit
static InstT* march_dump_slot() {
    static void* slot[16] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    if (!slot[dev] && hipMalloc(&slot[dev], 256) != hipSuccess) slot[dev] = nullptr;
    return (InstT*)slot[dev];
}

template <>
int launch_march<InstT, kInstDim3>(const MarchConfig& c, int mode, bool flags, const MarchGrid& g,
                                   const MarchArgs<InstT>& a_in, hipStream_t s) {
    MarchArgs<InstT> a = a_in;
    a.dump = march_dump_slot();
    if (!a.dump) {
        set_error("march: cannot allocate the dump slot");
        return PHIHIP_ERR_ALLOC;
    }
    dim3 grid(g.nblk, c.batch);
    if (c.vec == 1) return launch_cfg<1, 1, 64>(mode, flags, g, a, grid, s);
    if (c.vec < 0) return launch_cfg<kVmax, 1, 64, true>(mode, flags, g, a, grid, s);      // UNAL: ragged rows / unaligned buffers
    if (c.vec == 2 && kVmax == 4) {      // fp32 rows of even length (march_vector_width): the three 64-thread-row tiles
        switch (c.id) {
            case 4: return launch_cfg<(kVmax == 4 ? 2 : kVmax), 4, 64>(mode, flags, g, a, grid, s);
            case 5: return launch_cfg<(kVmax == 4 ? 2 : kVmax), 1, 64>(mode, flags, g, a, grid, s);
            case 6: return launch_cfg<(kVmax == 4 ? 2 : kVmax), 2, 64>(mode, flags, g, a, grid, s);
            default:
                set_error("march: tile config %d is not instantiated for 8-byte vectors", c.id);
                return PHIHIP_ERR_BAD_ARG;
        }
    }
    switch (c.id) {
        case 0: return launch_cfg<kVmax, 1, 16>(mode, flags, g, a, grid, s);
        case 1: return launch_cfg<kVmax, 2, 16>(mode, flags, g, a, grid, s);
        case 2: return launch_cfg<kVmax, 2, 32>(mode, flags, g, a, grid, s);
        case 3: return launch_cfg<kVmax, 4, 32>(mode, flags, g, a, grid, s);
        case 4: return launch_cfg<kVmax, 4, 64>(mode, flags, g, a, grid, s);
        case 5: return launch_cfg<kVmax, 1, 64>(mode, flags, g, a, grid, s);
        case 6: return launch_cfg<kVmax, 2, 64>(mode, flags, g, a, grid, s);
        case 7: return launch_cfg<kVmax, 1, 32>(mode, flags, g, a, grid, s);
        case 8: return launch_cfg<kVmax, 1, 128, false, true>(mode, flags, g, a, grid, s);      // the row tiles (lanes per row: g.tpr_rt)
        case 9: return launch_cfg<kVmax, 2, 128, false, true>(mode, flags, g, a, grid, s);
        case 10: return launch_cfg<kVmax, 4, 128, false, true>(mode, flags, g, a, grid, s);
#if PHIHIP_INST_F64
        case 11: return launch_cfg<kVmax, 2, 256, false, true>(mode, flags, g, a, grid, s);
        case 12: return launch_cfg<kVmax, 4, 256, false, true>(mode, flags, g, a, grid, s);
#endif
        default:
            set_error("march: bad tile config %d", c.id);
            return PHIHIP_ERR_BAD_ARG;
    }
}

// ---- MODE_APPLY on several lattices of one tile configuration in ONE launch (stencil_march.hpp march_apply_multi_kernel) -----------------------------
template <int V, int R, int TPR, bool UNAL = false, bool ROWT = false>
static int launch_multi_cfg(const MarchMulti<InstT>& m, const MarchArgs<InstT>& a, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL((march_apply_multi_kernel<InstT, V, R, TPR, kInstDim3, UNAL, ROWT>), grid, dim3(kBlock), 0, s, m, a);
    return PHIHIP_OK;
}

template <>
int launch_march_multi<InstT, kInstDim3>(const MarchConfig& c, int count, const MarchGrid* g, const MarchArgs<InstT>* args, hipStream_t s) {
    if (count < 1 || count > 3) {
        set_error("march (multi): %d lattices", count);
        return PHIHIP_ERR_BAD_ARG;
    }
    MarchArgs<InstT> a = args[0];
    a.dump = march_dump_slot();
    if (!a.dump) {
        set_error("march: cannot allocate the dump slot");
        return PHIHIP_ERR_ALLOC;
    }
    MarchMulti<InstT> m;
    memset(&m, 0, sizeof(m));
    unsigned nblk = 0;
    for (int l = 0; l < count; ++l) {
        m.g[l] = g[l];
        m.in[l] = args[l].a;
        m.out[l] = args[l].o1;
        nblk = (unsigned)g[l].nblk > nblk ? (unsigned)g[l].nblk : nblk;
    }
    dim3 grid(nblk, c.batch, count);
    if (c.vec == 1) return launch_multi_cfg<1, 1, 64>(m, a, grid, s);
    if (c.vec < 0) return launch_multi_cfg<kVmax, 1, 64, true>(m, a, grid, s);
    if (c.vec == 2 && kVmax == 4) {
        switch (c.id) {
            case 4: return launch_multi_cfg<(kVmax == 4 ? 2 : kVmax), 4, 64>(m, a, grid, s);
            case 5: return launch_multi_cfg<(kVmax == 4 ? 2 : kVmax), 1, 64>(m, a, grid, s);
            case 6: return launch_multi_cfg<(kVmax == 4 ? 2 : kVmax), 2, 64>(m, a, grid, s);
            default:
                set_error("march: tile config %d is not instantiated for 8-byte vectors", c.id);
                return PHIHIP_ERR_BAD_ARG;
        }
    }
    switch (c.id) {
        case 0: return launch_multi_cfg<kVmax, 1, 16>(m, a, grid, s);
        case 1: return launch_multi_cfg<kVmax, 2, 16>(m, a, grid, s);
        case 2: return launch_multi_cfg<kVmax, 2, 32>(m, a, grid, s);
        case 3: return launch_multi_cfg<kVmax, 4, 32>(m, a, grid, s);
        case 4: return launch_multi_cfg<kVmax, 4, 64>(m, a, grid, s);
        case 5: return launch_multi_cfg<kVmax, 1, 64>(m, a, grid, s);
        case 6: return launch_multi_cfg<kVmax, 2, 64>(m, a, grid, s);
        case 7: return launch_multi_cfg<kVmax, 1, 32>(m, a, grid, s);
        case 8: return launch_multi_cfg<kVmax, 1, 128, false, true>(m, a, grid, s);
        case 9: return launch_multi_cfg<kVmax, 2, 128, false, true>(m, a, grid, s);
        case 10: return launch_multi_cfg<kVmax, 4, 128, false, true>(m, a, grid, s);
#if PHIHIP_INST_F64
        case 11: return launch_multi_cfg<kVmax, 2, 256, false, true>(m, a, grid, s);
        case 12: return launch_multi_cfg<kVmax, 4, 256, false, true>(m, a, grid, s);
#endif
        default:
            set_error("march: bad tile config %d", c.id);
            return PHIHIP_ERR_BAD_ARG;
    }
}

}  // namespace phihip
