// Per-layer row orders for the deformable attention (see encoder_rows.hip): the body as a device function, so that the
// layer-0 top-k launch (two workgroups on an otherwise empty chip) can carry the jobs (topk.hip).
#pragma once
#include "common.h"

namespace sdetr {

constexpr int kOrderThreads = 1024;
constexpr int kOrderMaxLayers = 8;
constexpr int kOrderSlotQuantum = 4096;     // slots per pass are a multiple of 16 waves x 256 (aligned 8-byte slot reads)
constexpr int kOrderMaxPasses = 16;
constexpr int kOrderMaxTokens = 16 * 65536; // (round 5) any pyramid: positions beyond one pass's slots take further passes
constexpr int kOrderBatch = 12;            // rows a thread has in flight (two dependent loads each): the benchmark's 11 363 rows in one batch

// one workgroup per (image, layer): the first version did all layers in one workgroup per image -- 39 us, eleven
// dependent (index -> tile position) round trips per thread one after the other
struct RowOrderArgs {
    const int64_t *sorted_index;
    int64_t index_batch_stride;
    const int32_t *tile_pos;
    int S, n0, nl, batch;
    int slot_cap;              // 16-bit slots the workgroup's dynamic LDS holds: a multiple of kOrderSlotQuantum
    const int *counts_dev;
    int32_t *order;
    int64_t order_layer_stride, order_batch_stride;
};

// slots per pass for a pyramid of S tokens when the launch can give the job `lds_bytes` of dynamic LDS: the fewest passes
// that fit, evenly sized, rounded to the slot quantum (0: does not fit kOrderMaxPasses passes)
static inline int order_slot_cap(int S, size_t lds_bytes)
{
    const int max_slots = (int)((lds_bytes / 2) / kOrderSlotQuantum) * kOrderSlotQuantum;
    if (max_slots <= 0 || S <= 0) return 0;
    const int passes = (S + max_slots - 1) / max_slots;
    if (passes > kOrderMaxPasses) return 0;
    const int per = (S + passes - 1) / passes;
    return (per + kOrderSlotQuantum - 1) / kOrderSlotQuantum * kOrderSlotQuantum;
}

// body of one (image b, layer k) job; `slot` = the workgroup's dynamic LDS (>= min(slot_cap, S rounded up to 8) * 2 bytes)
__device__ __forceinline__ void layer_row_orders_body(const RowOrderArgs &a, const int b, const int k, uint16_t *slot)
{
    __shared__ int wave_tot[kOrderThreads / 64];
    const int64_t *sorted_index = a.sorted_index;
    const int64_t index_batch_stride = a.index_batch_stride;
    const int32_t *tile_pos = a.tile_pos;
    const int S = a.S, n0 = a.n0;
    const int *counts_dev = a.counts_dev;
    int32_t *order = a.order;
    const int64_t order_layer_stride = a.order_layer_stride, order_batch_stride = a.order_batch_stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = min(counts_dev[k], n0);              // rows of this layer
    const int64_t *idx = sorted_index + (int64_t)b * index_batch_stride;
    int32_t *out = order + k * order_layer_stride + b * order_batch_stride;
    const uint2 *slot2 = reinterpret_cast<const uint2 *>(slot);
    // Round 5: the slot array covers `cap` tile positions at a time.  A pyramid whose S 16-bit slots do not fit the LDS
    // (the reference's 5scale configuration: 89 250 tokens = 178 KB) takes ceil(S / cap) passes over the rows; a pass
    // scatters only the rows whose position falls into its range and appends its compacted run behind the previous one
    // (tile positions ascend across passes, so the concatenation is the tile-major order).
    const int cap = a.slot_cap;
    int done = 0;                                      // rows written by the passes so far (workgroup-uniform)
    for (int p0 = 0; p0 < S; p0 += cap) {
    const int len = min(cap, S - p0);
    {
        uint4 *s4 = reinterpret_cast<uint4 *>(slot);
        const int n16 = (len + 7) >> 3;
        for (int p = tid; p < n16; p += kOrderThreads) s4[p] = make_uint4(~0u, ~0u, ~0u, ~0u);
    }
    __syncthreads();
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
            // (distinct tokens: one row per slot; duplicates overwrite each other and are caught by the count below)
            if (r < c && t[i] >= 0 && t[i] < S && pos[i] >= 0 && pos[i] < len) slot[pos[i]] = (uint16_t)r;
        }
    }
    __syncthreads();
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
    int base = done, pass_total = 0;
    for (int w = 0; w < kOrderThreads / 64; ++w) {
        const int tw = wave_tot[w];
        if (w < wave) base += tw;
        pass_total += tw;
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
    done += pass_total;
    __syncthreads();            // the next pass rewrites the slots and the wave totals
    }   // passes
    // Duplicate or out-of-range tokens in a caller-supplied list leave fewer than c rows: the order must still be
    // a permutation -- the gather writes exactly the rows it lists -- so such a list is walked in list order instead.
    if (done != c)
        for (int r = tid; r < c; r += kOrderThreads) out[r] = r;
}

}  // namespace sdetr
