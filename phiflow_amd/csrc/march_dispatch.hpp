// march_dispatch.hpp -- tile configurations of march_kernel and the host-side dispatcher.
#pragma once
#include "common.hpp"

namespace phihip {

// (rows per thread R, threads per row TPR) per config id; tile = (256/TPR*R) rows x (TPR*V) columns
struct TileShape {
    int rows, tpr;
};
// (2, 64) (r3): full-width rows like (1, 64) -- no left / right halo columns when a row is 256 fp32 / 128 fp64 cells -- with 8 instead of
// 4 rows per tile: two halo rows per eight own rows (25 % extra L2 requests instead of 50 %) at ~105 instead of ~80 VGPRs
// (1, 32) (r3): 8 rows x 128 fp32 / 64 fp64 cells at the register cost of the one-row tiles -- for rows that are multiples of 128 but not of 256
// cells (384, 640), between (1, 16) (16 x 64: many halo columns) and (1, 64) (4 x 256: two halo rows per four own rows)
// (1, 128) (r4) is the ROW tile (stencil_march.hpp ROWT): whole rows of 65 ... 128 vectors (fp32 rows of 260 ... 512 cells that are not 256 or
// 512: 288, 320, 384, 448 ...), floor(256 / lanes per row) thread rows, no halo columns
// (2, 128), (4, 128): the same with 2 / 4 rows per thread (a 384-cell row leaves two thread rows: 2 own rows per 2 halo rows with one row per thread)
// (2, 256), (4, 256) (r6): the WIDE row tiles -- whole rows of 129 ... 256 vectors (384-cell fp64 rows = 192 lanes; BASELINE configs[4]): ONE thread row, 2 / 4 rows per
// thread, every lane fetches both halo vectors of its column. fp64 only (fp32 rows of more than 512 cells split into (., 64) tiles of 256 cells without remainder
// where it matters: 768, 1024).
constexpr int kNumTileConfigs = 13;
constexpr int kRowTile = 8;        // ids >= kRowTile are row tiles
constexpr int kWideRowTile = 11;   // ids >= kWideRowTile are the wide ones
constexpr TileShape kTileShapes[kNumTileConfigs] = {{1, 16}, {2, 16}, {2, 32}, {4, 32}, {4, 64}, {1, 64}, {2, 64}, {1, 32}, {1, 128}, {2, 128}, {4, 128}, {2, 256}, {4, 256}};

// Elements per thread along the fast axis: 16 bytes when the rows allow (n2 a multiple of 16 B / sizeof(T), 16-byte-aligned buffers); fp32 rows of
// EVEN length take 8-byte vectors (V = 2: the code path of the fp64 kernels; 250^3, 190^3 ... run 20-37 % slower on the scalar instantiation,
// profiles/r03_size_scan_ragged_rows.jsonl); everything else took V = 1 until r4:
// r4: rows that are not whole 16-byte vectors (255^3 ...) and buffers that are not 16-byte aligned take the UNAL instantiation -- 16-byte vectors
// at element alignment, the partial vector at the end of a row loaded early and rotated (stencil_march.hpp) -- reported as a NEGATIVE width
// (-4 fp32, -2 fp64); the scalar instantiation is left with rows shorter than two vectors. (255^3 ran 19 % below 256^3 on the scalar kernels.)
inline int march_vector_width(int n2, int esize, bool unaligned) {
    const int vmax = 16 / esize;
    if (!unaligned && n2 % vmax == 0) return vmax;
    if (!unaligned && esize == 4 && n2 % 2 == 0) return 2;
    // (the overlapping last vector must lie inside ONE tile -- its first cell's left neighbour has to be staged: a last tile of fewer than vmax
    // cells, n2 mod (64 vmax) in 1 .. vmax - 1, e.g. 257-259, keeps the scalar instantiation)
    const int rem = n2 % (64 * vmax);
    if (rem != 0 && rem < vmax) return 1;
    return n2 >= 2 * vmax ? -vmax : 1;
}
inline int march_vec_elems(int vec) { return vec < 0 ? -vec : vec; }          // elements per thread along the fast axis
inline bool march_one_tile(int vec) { return vec == 1 || vec < 0; }            // only the (1, 64) tile is instantiated
// which tile configurations are instantiated for a vector width: all for 16-byte vectors, (1,64) alone for V = 1 and for the UNAL kernels, the
// three 64-thread-row tiles (4,64), (1,64), (2,64) for the fp32 V = 2 kernels (128 cells per tile row)
inline bool march_tile_available(int vec, int esize, int id, int n2) {
    if (id >= kWideRowTile) return esize == 8 && vec == 2 && n2 % vec == 0 && n2 / vec > 128 && n2 / vec <= 256;
    if (id >= kRowTile) return vec == 16 / esize && n2 % vec == 0 && n2 / vec > 64 && n2 / vec <= 128;
    if (march_one_tile(vec)) return id == 5;
    if (vec == 2 && esize == 4) return id == 4 || id == 5 || id == 6;
    return true;
}
// extent of a tile configuration: rows (axis a1) x cells (axis a2)
inline void march_tile_extent(int vec, int id, int n2, int* t1, int* t2) {
    const int rows = march_one_tile(vec) ? 1 : kTileShapes[id].rows, tpr = march_one_tile(vec) ? 64 : kTileShapes[id].tpr;
    if (id >= kRowTile && !march_one_tile(vec)) {
        *t1 = 256 / (n2 / vec) * rows;
        *t2 = n2;
        return;
    }
    *t1 = 256 / tpr * rows;
    *t2 = tpr * march_vec_elems(vec);
}

struct MarchConfig {
    int id;      // index into kTileShapes (ignored when vec == 1)
    int vec;     // elements per thread along the fast axis (march_vector_width)
    int t1, t2;  // tile extent
    int chunk;   // planes per workgroup (pieces of the linearised (tile, plane) space)
    int batch;
};

inline int family_mode(int family) { return family == 1 ? MODE_MATVEC : (family == 2 ? MODE_UPDATE : (family == 3 ? MODE_UPDATE_R : (family == 4 ? MODE_CG1 : MODE_RESID))); }

// kernel families that may use different tile shapes (phihip_set_tuning_kernel)
enum MarchFamily { FAM_APPLY = 0, FAM_MATVEC = 1, FAM_UPDATE = 2, FAM_UPDATE_R = 3, FAM_CG1 = 4, FAM_COUNT = 5 };   // UPDATE_R: the r-only update (3 words); CG1: the fused single-reduction iteration

// choose tile + chunk for a grid (honours ctx->tuning[family]) and fill the decomposition fields of MarchGrid; force_id >= 0: that tile configuration if the
// grid's vector width has it (lattices that share ONE launch, run_diffuse)
int plan_march(const phihip_ctx* ctx, const GridView& v, int mask_batch, bool flags, int family, MarchConfig* cfg, MarchGrid* g, int force_id = -1);

// resident workgroups per CU of one march_kernel instantiation (hipOccupancyMaxActiveBlocksPerMultiprocessor, cached)
template <typename T, bool DIM3>
int march_occupancy(int id, int vec, int mode, bool flags);

template <> int march_occupancy<float, false>(int id, int vec, int mode, bool flags);
template <> int march_occupancy<float, true>(int id, int vec, int mode, bool flags);
template <> int march_occupancy<double, false>(int id, int vec, int mode, bool flags);
template <> int march_occupancy<double, true>(int id, int vec, int mode, bool flags);

inline int march_occupancy_any(const GridView& v, int id, int vec, int mode, bool flags) {
    if (v.dtype == PHIHIP_F64) return v.rank == 3 ? march_occupancy<double, true>(id, vec, mode, flags) : march_occupancy<double, false>(id, vec, mode, flags);
    return v.rank == 3 ? march_occupancy<float, true>(id, vec, mode, flags) : march_occupancy<float, false>(id, vec, mode, flags);
}

template <typename T, bool DIM3>
int launch_march(const MarchConfig& c, int mode, bool flags, const MarchGrid& g, const MarchArgs<T>& a, hipStream_t s);

// MODE_APPLY without cell flags on `count` <= 3 lattices that share the tile configuration c (id, vec, batch) in ONE launch; g[l] / a[l].a / a[l].o1 per
// lattice, weights / ident from a[0]
template <typename T, bool DIM3>
int launch_march_multi(const MarchConfig& c, int count, const MarchGrid* g, const MarchArgs<T>* a, hipStream_t s);

template <typename T>
inline int launch_march_multi_any(const GridView& v, const MarchConfig& c, int count, const MarchGrid* g, const MarchArgs<T>* a, hipStream_t s) {
    return v.rank == 3 ? launch_march_multi<T, true>(c, count, g, a, s) : launch_march_multi<T, false>(c, count, g, a, s);
}

template <typename T>
inline int launch_march_any(const GridView& v, const MarchConfig& c, int mode, bool flags, const MarchGrid& g,
                            const MarchArgs<T>& a, hipStream_t s) {
    return v.rank == 3 ? launch_march<T, true>(c, mode, flags, g, a, s) : launch_march<T, false>(c, mode, flags, g, a, s);
}

}  // namespace phihip
