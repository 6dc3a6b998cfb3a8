// Shared definitions of the top-k kernels and the body of the rank-by-counting kernel (see topk.hip), as a device
// function so that another launch can carry it next to other work (fused_head_value.hip).
#pragma once
#include "common.h"

namespace sdetr {

constexpr int kRankThreads = 512;   // 8 waves share a workgroup's list scan (4 until round 2: the per-lane loop was the kernel's time)
constexpr int kRankWaves = kRankThreads / 64;
constexpr int kRankTile = 12288;  // keys staged per LDS round (48 KiB)

__device__ __forceinline__ uint32_t desc_bits(float s)
{
    if (s == 0.f) s = 0.f;  // -0 == +0
    const uint32_t u = __float_as_uint(s);
    const uint32_t asc = u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
    return ~asc;
}
__device__ __forceinline__ float undesc_bits(uint32_t d)
{
    const uint32_t asc = ~d;
    const uint32_t u = (asc & 0x80000000u) ? (asc ^ 0x80000000u) : ~asc;
    return __uint_as_float(u);
}

struct RankArgs {
    const float *score;
    const uint8_t *mask;
    int64_t mask_stride;  // bytes between mask rows
    const float *fill;  // device scalar or NULL
    const int64_t *payload;
    // candidate mode (after topk_prefilter): keys / positions / count per row instead of raw scores
    const uint32_t *cand_key;
    const uint32_t *cand_pos;
    const int32_t *cand_count;
    int N, k;
    int64_t index_offset;
    float *out_score;
    int64_t *out_index;
    int64_t out_stride;   // elements between output rows (>= k)
    // SLICED form (round 6, sdetr_masked_topk_sliced_f32): a "row" of the launch is slice s = row % slices of image
    // row / slices -- keys [s * slice_len, min(N, (s + 1) * slice_len)) of that image's score row, sorted COMPLETELY into
    // the same columns of the output row (position payload = column in the image row).  0 = off.
    int slices, slice_len;
};

constexpr int kRankLdsWords = kRankTile + kRankWaves * 64;   // tile | partial

// (`bx` / `b` = key block and row: the kernel's own block indices, or the position inside a launch that also carries
// other work -- fused_head_value.hip; `tile` [kRankTile] and `partial` [kRankWaves * 64] words of LDS.  The first 512
// threads of the workgroup take part.)
__device__ __forceinline__ void topk_rank_body(const RankArgs &p, int bx, int b, uint32_t *tile, uint32_t *partial)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int base = bx * 64;  // owned keys [base, base+64)
    const bool cand = p.cand_key != nullptr;
    const int img = p.slices ? b / p.slices : b;                          // image row the keys come from
    const int col0 = p.slices ? (b - img * p.slices) * p.slice_len : 0;   // first column of this launch row in it
    const int n_keys = cand ? p.cand_count[b] : (p.slices ? min(p.slice_len, p.N - col0) : p.N);  // length of the ranked list
    if (base >= n_keys) return;                        // uniform per workgroup
    const int k_row = p.slices ? n_keys : p.k;         // (a slice is sorted completely)
    const float *srow = p.score + (int64_t)img * p.N + col0;
    const uint8_t *mrow = p.mask ? p.mask + (int64_t)img * p.mask_stride + col0 : nullptr;
    const uint32_t *ckey = cand ? p.cand_key + (int64_t)b * p.N : nullptr;
    const float fill = p.fill ? *p.fill : 0.f;
    auto key_at = [&](int i) -> uint32_t {  // i < n_keys
        if (cand) return ckey[i];
        float s = srow[i];
        if (mrow && mrow[i]) s = fill;
        return desc_bits(s);
    };
    const int mypos = base + lane;
    // a list that fits one LDS round: the owned key is read from the staged tile (saves a dependent trip to memory in front
    // of the staging loads)
    const bool single = n_keys <= kRankTile;
    uint32_t mine = (!single && mypos < n_keys) ? key_at(mypos) : 0u;
    uint32_t rank = 0;

    for (int t0 = 0; t0 < n_keys; t0 += kRankTile) {
        const int tn = min(kRankTile, n_keys - t0);
        if (t0 > 0) __syncthreads();
        // stage: all global loads of this thread first (<= 48 scalars), then the LDS stores; padding keys
        // (positions >= N) are 0xffffffff, which no "<" test counts and whose positions fail the tie rule
        constexpr int kPer = kRankTile / kRankThreads;  // 24
        for (int c0 = 0; c0 < kPer; c0 += 12) {
            uint32_t kv[12];
            if (cand) {
#pragma unroll
                for (int c = 0; c < 12; ++c) kv[c] = ckey[min(t0 + (c0 + c) * kRankThreads + tid, n_keys - 1)];
            } else {
                float sv[12];
                uint8_t mk[12];
#pragma unroll
                for (int c = 0; c < 12; ++c) sv[c] = srow[min(t0 + (c0 + c) * kRankThreads + tid, n_keys - 1)];
#pragma unroll
                for (int c = 0; c < 12; ++c)
                    mk[c] = mrow ? mrow[min(t0 + (c0 + c) * kRankThreads + tid, n_keys - 1)] : (uint8_t)0;
#pragma unroll
                for (int c = 0; c < 12; ++c) kv[c] = desc_bits(mk[c] ? fill : sv[c]);
            }
#pragma unroll
            for (int c = 0; c < 12; ++c) {
                const int li = (c0 + c) * kRankThreads + tid;
                if (li < ((tn + 3) & ~3)) tile[li] = (t0 + li < n_keys) ? kv[c] : 0xffffffffu;
            }
            if ((c0 + 12) * kRankThreads >= tn) break;
        }
        __syncthreads();
        if (single && mypos < n_keys) mine = tile[mypos];
        // groups of 4 keys, round-robin over the wavefronts; the list splits into three ranges relative to
        // the owned block so every loop body is branch-free and the LDS reads pipeline (8 in flight)
        const int ngroups = (tn + 3) / 4;
        const uint4 *t4 = reinterpret_cast<const uint4 *>(tile);
        const int g_own0 = min(ngroups, max(0, (base - t0) / 4));           // first group inside the owned block
        const int g_own1 = min(ngroups, max(0, (base + 64 - t0 + 3) / 4));  // first group after it
        auto first_at_or_after = [&](int g0) { return g0 + ((wave - g0) % kRankWaves + kRankWaves) % kRankWaves; };
        int g = wave;
#pragma unroll 8
        for (; g < g_own0; g += kRankWaves) {  // before: ties sort before us
            const uint4 c = t4[g];
            rank += (c.x <= mine) + (c.y <= mine) + (c.z <= mine) + (c.w <= mine);
        }
        for (g = first_at_or_after(g_own0); g < g_own1; g += kRankWaves) {  // inside: exact positional tie rule
            const uint4 c = t4[g];
            const int j = t0 + g * 4;
            rank += (c.x < mine || (c.x == mine && j + 0 < mypos)) ? 1u : 0u;
            rank += (c.y < mine || (c.y == mine && j + 1 < mypos)) ? 1u : 0u;
            rank += (c.z < mine || (c.z == mine && j + 2 < mypos)) ? 1u : 0u;
            rank += (c.w < mine || (c.w == mine && j + 3 < mypos)) ? 1u : 0u;
        }
        g = first_at_or_after(g_own1);
#pragma unroll 8
        for (; g < ngroups; g += kRankWaves) {  // after: ties sort after us (padding keys 0xffffffff are never "<")
            const uint4 c = t4[g];
            rank += (c.x < mine) + (c.y < mine) + (c.z < mine) + (c.w < mine);
        }
    }
    partial[wave * 64 + lane] = rank;
    __syncthreads();
    if (wave == 0 && mypos < n_keys) {
        uint32_t r = 0;
#pragma unroll
        for (int w = 0; w < kRankWaves; ++w) r += partial[w * 64 + lane];
        if (r < (uint32_t)k_row) {
            const int pos = cand ? (int)p.cand_pos[(int64_t)b * p.N + mypos] : mypos + col0;
            if (p.out_score) p.out_score[(int64_t)img * p.out_stride + col0 + r] = undesc_bits(mine);
            p.out_index[(int64_t)img * p.out_stride + col0 + r] =
                p.payload ? p.payload[(int64_t)b * p.N + pos] : (int64_t)pos + p.index_offset;
        }
    }
}

}  // namespace sdetr
