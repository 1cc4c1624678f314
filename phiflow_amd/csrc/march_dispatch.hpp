// march_dispatch.hpp -- tile configurations of march_kernel and the host-side dispatcher.
#pragma once
#include "common.hpp"

namespace phihip {

// (rows per thread R, threads per row TPR) per config id; tile = (256/TPR*R) rows x (TPR*V) columns
struct TileShape {
    int rows, tpr;
};
constexpr int kNumTileConfigs = 6;
constexpr TileShape kTileShapes[kNumTileConfigs] = {{1, 16}, {2, 16}, {2, 32}, {4, 32}, {4, 64}, {1, 64}};

struct MarchConfig {
    int id;      // index into kTileShapes (ignored when vec == 1)
    int vec;     // elements per thread along the fast axis: 16 B / sizeof(T), or 1 when n2 is not a multiple of it
    int t1, t2;  // tile extent
    int chunk;   // planes per workgroup
    int batch;
};

// choose tile + chunk for a grid (honours ctx->tuning) and fill the decomposition fields of MarchGrid
int plan_march(const phihip_ctx* ctx, const GridView& v, int mask_batch, MarchConfig* cfg, MarchGrid* g);

template <typename T, bool DIM3>
int launch_march(const MarchConfig& c, int mode, bool flags, const MarchGrid& g, const MarchArgs<T>& a, hipStream_t s);

template <typename T>
inline int launch_march_any(const GridView& v, const MarchConfig& c, int mode, bool flags, const MarchGrid& g,
                            const MarchArgs<T>& a, hipStream_t s) {
    return v.rank == 3 ? launch_march<T, true>(c, mode, flags, g, a, s) : launch_march<T, false>(c, mode, flags, g, a, s);
}

}  // namespace phihip
