// Per-layer row orders for the deformable attention (see encoder_rows.hip): the body as a device function, so that the
// layer-0 top-k launch (two workgroups on an otherwise empty chip) can carry the jobs (topk.hip).
#pragma once
#include "common.h"

namespace sdetr {

constexpr int kOrderThreads = 1024;
constexpr int kOrderMaxLayers = 8;
constexpr int kOrderSlotQuantum = 4096;     // slots per part are a multiple of 16 waves x 256 (aligned 8-byte slot reads)
constexpr int kOrderMaxParts = 64;
constexpr int kOrderMaxTokens = 64 * 65536; // (round 5) any pyramid: the tile positions are cut into parts
constexpr int kOrderPartSlots = 4096;       // preferred slots per part (8 KB of LDS): 6 parts at the benchmark pyramid, 22 at 5scale
constexpr int kOrderBatch = 12;            // rows a thread has in flight (two dependent loads each): the benchmark's 11 363 rows in one batch

// One workgroup per (image, layer, PART of the tile positions).  Round 4 ran one workgroup per (image, layer) over ALL
// positions (150 KB of 16-bit slots, pyramids beyond 76 800 tokens refused); round 5 cuts the positions into parts of
// `slot_cap` slots that run side by side: every part reads all the layer's rows (index -> tile position, two dependent
// loads), scatters those whose position falls into its range, COUNTS the rows below its range -- that count is where
// its compacted run starts in the layer's order -- and compacts its own slots.  The reference's 5scale pyramid (89 250
// tokens, 45 330 rows in the first layer) is 6 parts x 6 layers = 36 workgroups instead of 6 serial two-pass ones.
struct RowOrderArgs {
    const int64_t *sorted_index;
    int64_t index_batch_stride;
    const int32_t *tile_pos;
    int S, n0, nl, batch;
    int slot_cap;              // tile positions per part: a multiple of kOrderSlotQuantum
    int parts;                 // ceil(S / slot_cap)
    const int *counts_dev;
    int32_t *order;
    int64_t order_layer_stride, order_batch_stride;
};

// slots per part / number of parts for a pyramid of S tokens when a workgroup may use `lds_bytes` of dynamic LDS
static inline int order_slot_cap(int S, size_t lds_bytes)
{
    int cap = kOrderPartSlots;
    const int max_slots = (int)((lds_bytes / 2) / kOrderSlotQuantum) * kOrderSlotQuantum;
    if (max_slots <= 0 || S <= 0) return 0;
    if (cap > max_slots) cap = max_slots;
    // (few tokens: one part; many: at most kOrderMaxParts)
    while ((S + cap - 1) / cap > kOrderMaxParts) {
        if (cap + kOrderSlotQuantum > max_slots) return 0;
        cap += kOrderSlotQuantum;
    }
    return cap;
}

__device__ __forceinline__ int order_block_sum(int v, int *scratch /* [16] */, int tid)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    __syncthreads();                       // (scratch may still be read from the previous use)
    if ((tid & 63) == 0) scratch[tid >> 6] = v;
    __syncthreads();
    int t = 0;
#pragma unroll
    for (int w = 0; w < kOrderThreads / 64; ++w) t += scratch[w];
    return t;
}

// body of one (image b, layer k, part) job; `slot` = the workgroup's dynamic LDS (>= min(slot_cap, S rounded up to 8) * 2 bytes)
__device__ __forceinline__ void layer_row_orders_body(const RowOrderArgs &a, const int b, const int k, const int part,
                                                      uint16_t *slot)
{
    __shared__ int wave_tot[kOrderThreads / 64];
    __shared__ int red[kOrderThreads / 64];
    const int64_t *sorted_index = a.sorted_index;
    const int32_t *tile_pos = a.tile_pos;
    const int S = a.S, n0 = a.n0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = min(a.counts_dev[k], n0);              // rows of this layer
    const int64_t *idx = sorted_index + (int64_t)b * a.index_batch_stride;
    int32_t *out = a.order + k * a.order_layer_stride + b * a.order_batch_stride;
    const uint2 *slot2 = reinterpret_cast<const uint2 *>(slot);
    const int p0 = part * a.slot_cap;
    const int len = min(a.slot_cap, S - p0);
    const bool last_part = part == a.parts - 1;
    {
        uint4 *s4 = reinterpret_cast<uint4 *>(slot);
        const int n16 = (len + 7) >> 3;
        for (int p = tid; p < n16; p += kOrderThreads) s4[p] = make_uint4(~0u, ~0u, ~0u, ~0u);
    }
    __syncthreads();
    int below = 0, inside = 0, valid = 0;       // per thread: rows below this part's range / inside it / with a token of the pyramid
    for (int r0 = 0; r0 < c; r0 += kOrderThreads * kOrderBatch) {
        int64_t t[kOrderBatch];
        int pos[kOrderBatch];
        // (unconditional loads at clamped positions: behind a lane predicate hipcc issues them one round trip at a time)
#pragma unroll
        for (int i = 0; i < kOrderBatch; ++i) t[i] = idx[min(r0 + i * kOrderThreads + tid, c - 1)];
#pragma unroll
        for (int i = 0; i < kOrderBatch; ++i) pos[i] = tile_pos[min(max(t[i], (int64_t)0), (int64_t)S - 1)] - p0;
#pragma unroll
        for (int i = 0; i < kOrderBatch; ++i) {
            const int r = r0 + i * kOrderThreads + tid;
            const bool ok = r < c && t[i] >= 0 && t[i] < S;
            valid += ok;
            below += ok && pos[i] < 0;
            if (ok && pos[i] >= 0 && pos[i] < len) {
                ++inside;
                slot[pos[i]] = (uint16_t)r;     // (distinct tokens: one row per slot; a duplicate overwrites -- see below)
            }
        }
    }
    const int base0 = order_block_sum(below, red, tid);     // where this part's run starts (the barriers also publish the slots)
    const int n_inside = order_block_sum(inside, red, tid);
    const int n_valid = order_block_sum(valid, red, tid);
    // every wave compacts its own run of slots, 256 at a time (a lane reads four consecutive slots as one 8-byte word;
    // ranks from the ballots of the four sub-positions): pass 1 counts the run, the 16 totals are scanned, pass 2 writes
    const int per_wave = ((len + kOrderThreads / 64 - 1) / (kOrderThreads / 64) + 255) & ~255;
    const int w0 = min(len, wave * per_wave), w1 = min(len, w0 + per_wave);   // (w0 is a multiple of 256: 8-byte aligned
    const int s_round = (len + 7) & ~7;                                       //  reads; slots past len up to the rounded
    int total = 0;                                                            //  size hold 0xffff)
    for (int p = w0; p < w1; p += 256) {
        const int q = p + lane * 4;
        uint2 v = make_uint2(~0u, ~0u);
        if (q < s_round) v = slot2[q >> 2];
        const int n = ((v.x & 0xffffu) != 0xffffu) + ((v.x >> 16) != 0xffffu) + ((v.y & 0xffffu) != 0xffffu) + ((v.y >> 16) != 0xffffu);
        int acc = n;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d);
        total += acc;
    }
    if (lane == 0) wave_tot[wave] = total;
    __syncthreads();
    int base = base0, filled = 0;
    for (int w = 0; w < kOrderThreads / 64; ++w) {
        const int tw = wave_tot[w];
        if (w < wave) base += tw;
        filled += tw;
    }
    for (int p = w0; p < w1; p += 256) {
        const int q = p + lane * 4;
        uint2 v = make_uint2(~0u, ~0u);
        if (q < s_round) v = slot2[q >> 2];
        const uint32_t r0 = v.x & 0xffffu, r1 = v.x >> 16, r2 = v.y & 0xffffu, r3 = v.y >> 16;
        const int n = (r0 != 0xffffu) + (r1 != 0xffffu) + (r2 != 0xffffu) + (r3 != 0xffffu);
        // exclusive scan of n over the lanes
        int incl = n;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        int at = base + incl - n;
        if (r0 != 0xffffu) out[at++] = (int32_t)r0;
        if (r1 != 0xffffu) out[at++] = (int32_t)r1;
        if (r2 != 0xffffu) out[at++] = (int32_t)r2;
        if (r3 != 0xffffu) out[at++] = (int32_t)r3;
        base += __shfl(incl, 63);
    }
    // A caller-supplied list may hold a token twice or tokens outside the pyramid; the order must still be a
    // permutation of the rows -- the gather writes exactly the rows it lists.  Rows that lost their slot to a duplicate
    // follow their part's run, rows with a token outside the pyramid close the layer's order (written by the last part);
    // both in row order, by one wave (rare, and the common case pays two comparisons).
    const bool lost = filled < n_inside;
    const bool stray = last_part && n_valid < c;
    if ((lost || stray) && wave == 0) {
        int at_lost = base0 + filled, at_stray = n_valid;
        for (int r0 = 0; r0 < c; r0 += 64) {
            const int r = r0 + lane;
            bool is_lost = false, is_stray = false;
            if (r < c) {
                const int64_t t = idx[r];
                if (t < 0 || t >= S) is_stray = stray;
                else {
                    const int pos = tile_pos[t] - p0;
                    is_lost = lost && pos >= 0 && pos < len && slot[pos] != (uint16_t)r;
                }
            }
            const uint64_t ml = __ballot(is_lost), ms = __ballot(is_stray);
            const uint64_t before = (1ull << lane) - 1ull;
            if (is_lost) out[at_lost + __popcll(ml & before)] = r;
            if (is_stray) out[at_stray + __popcll(ms & before)] = r;
            at_lost += __popcll(ml);
            at_stray += __popcll(ms);
        }
    }
}

}  // namespace sdetr
