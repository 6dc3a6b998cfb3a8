// Region / window geometry shared by the LDS-staged forward and the LDS-accumulating backward.
#pragma once

namespace sdetr {

constexpr int kTX = 16, kTY = 8, kHalo = 4;  // level-0 region and window padding (pixels of each level)
constexpr int kTL = 4, kTP = 4, kTD = 32;    // levels, points, head dim

__host__ __device__ constexpr int tile_w_cap(int l) { return (kTX >> l) + 2 * kHalo + 2; }
__host__ __device__ constexpr int tile_h_cap(int l) { return ((kTY >> l) > 0 ? (kTY >> l) : 1) + 2 * kHalo + 2; }
__host__ __device__ constexpr int tile_base_px(int l)
{
    int s = 0;
    for (int i = 0; i < l; ++i) s += tile_w_cap(i) * tile_h_cap(i);
    return s;
}
constexpr int kTilePx = tile_base_px(kTL);   // 1020 pixels over the four windows

}  // namespace sdetr
