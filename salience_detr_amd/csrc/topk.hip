// Masked top-k (sorted descending, ties -> lower index first) for the salience filtering stage.
//
// Replaces the torch.topk / torch.sort calls of models/bricks/salience_transformer.py:146-150
// (per-level top-k under masked_fill(mask, score.min())), :156-158 (global sort of the selected
// scores + index gather) and :366-367 (per-layer top-300).
//
// A sorting network confined to one CU is LDS-bandwidth bound (a 16 K-key bitonic sort is 105 passes over
// 128 KiB of LDS: measured 190-350 us), and a radix/bitwise SELECT confined to one CU is instruction
// bound (measured 25-60 us for 4 K-22 K keys).  The sizes here (273 ... 22 323 keys per row) are small
// enough that brute-force RANK BY COUNTING spread over the whole chip beats both:
//
//   topk_min  (fill_mode 1 only, one workgroup): min over the whole [B,N] score array -> workspace[0],
//             the value the reference substitutes for masked scores (score.min(), masked entries included).
//   topk_rank (ceil(N/64) workgroups per row): workgroup w owns keys [64w, 64w+64) of its row.  All keys
//             of the row (key = descending-orderable score bits, masked -> fill) are staged in LDS in
//             12 K-key tiles; the eight wavefronts stream them as broadcast ds_read_b128 and count, per owned
//             key, the keys that sort before it.  The list is in index order, so the tie rule is positional:
//             "<=" for keys before the owned block, "<" after it, exact only inside it (two VALU ops per
//             comparison, branch-free loops, 8 LDS reads in flight).  rank < k  =>  out[rank] = (score, index):
//             the rank IS the output slot -- no select pass, no sorting network, no merge.
//   N^2 comparisons (282 M for the largest level, 16 800 keys) over 1024 SIMDs is ~10-20 us.
//
//   When k is well below N (the per-layer top-300 of 11 363, the 40 % budget of the finest level) a
//   PREFILTER shrinks the ranked set first: topk_prefilter (one 1024-thread workgroup per row) ranks a
//   1024-key strided sample in LDS, takes the sample key whose rank is k*S/N plus five standard deviations
//   as a conservative threshold, counts the row's keys passing it and -- only if at least k pass, otherwise
//   it lets everything pass -- compacts them STABLY (index order, block scan) into a scratch list.  The
//   rank kernel then counts among those ~1.2 k candidates only (any key that fails the threshold sorts
//   after every candidate, so candidate ranks are global ranks).  Exactness never depends on the sample.
#include <cstdlib>

#include "common.h"

#include "topk_core.h"
#include "row_orders_core.h"
#include "finalize_core.h"
#include "topk_attention_core.h"

namespace sdetr {

__global__ void __launch_bounds__(1024) topk_min_kernel(const float *score, int64_t total, float *out)
{
    __shared__ float red[16];
    const int tid = threadIdx.x;
    float mn = INFINITY;
    int64_t i = tid;
    for (; i + 3 * 1024 < total; i += 4 * 1024) {
        const float a0 = score[i], a1 = score[i + 1024], a2 = score[i + 2048], a3 = score[i + 3072];
        mn = fminf(fminf(mn, a0), fminf(fminf(a1, a2), a3));
    }
    for (; i < total; i += 1024) mn = fminf(mn, score[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o, 64));
    if ((tid & 63) == 0) red[tid >> 6] = mn;
    __syncthreads();
    if (tid == 0) {
        mn = red[0];
        for (int w = 1; w < 16; ++w) mn = fminf(mn, red[w]);
        *out = mn;
    }
}

constexpr int kPreThreads = 1024;
constexpr int kPreWaves = kPreThreads / 64;

struct PrefilterArgs {
    const float *score;
    const uint8_t *mask;
    int64_t mask_stride;
    const float *fill;
    int N, k;
    int stage;            // 1: stage the keys through (dynamic) LDS, N * 4 bytes
    uint32_t *cand_key;   // [B][N]
    uint32_t *cand_pos;   // [B][N]
    int32_t *cand_count;  // [B]
};

// exclusive prefix over the block's threads (thread order) + grand total
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *buf /*[kPreWaves]*/, int tid,
                                                         uint32_t &total)
{
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t n = __shfl_up(incl, o, 64);
        if ((tid & 63) >= o) incl += n;
    }
    if ((tid & 63) == 63) buf[tid >> 6] = incl;
    __syncthreads();
    uint32_t before = 0, t = 0;
#pragma unroll
    for (int w = 0; w < kPreWaves; ++w) {
        const uint32_t c = buf[w];
        if (w < (tid >> 6)) before += c;
        t += c;
    }
    total = t;
    return before + incl - v;
}

template <int KPT>
__global__ void __launch_bounds__(kPreThreads) topk_prefilter_kernel(PrefilterArgs p)
{
    __shared__ __attribute__((aligned(16))) uint32_t samp[kPreThreads];
    __shared__ uint32_t scan_buf[kPreWaves];
    __shared__ uint32_t thr_s;
    const int tid = threadIdx.x, b = blockIdx.x;
    const float *srow = p.score + (int64_t)b * p.N;
    const uint8_t *mrow = p.mask ? p.mask + (int64_t)b * p.mask_stride : nullptr;
    const float fill = p.fill ? *p.fill : 0.f;
    const int chunk = (p.N + kPreThreads - 1) / kPreThreads;  // <= KPT
    const int lo = tid * chunk, hi = min(p.N, lo + chunk);
    uint32_t keys[KPT];
    if (p.stage) {
        // Coalesced global reads (thread t takes elements t, t + 1024, ...) staged through LDS, from which every
        // thread then takes its CONTIGUOUS chunk -- the candidate list must stay in position order (the rank kernel's
        // tie rule), and reading the chunks straight from global memory makes every load instruction of a wave touch
        // ~48 cache lines: 16 waves x KPT loads on this one CU's address unit was ~4 us of the kernel's 11.
        extern __shared__ uint32_t keybuf[];
        // every load of the thread in flight at once (batches of 6 cost one trip to memory each: three for the
        // finest level's 17 keys per thread)
        float sv[KPT];
        uint8_t mk[KPT];
#pragma unroll
        for (int c = 0; c < KPT; ++c) sv[c] = srow[min(c * kPreThreads + tid, p.N - 1)];
#pragma unroll
        for (int c = 0; c < KPT; ++c) mk[c] = mrow ? mrow[min(c * kPreThreads + tid, p.N - 1)] : (uint8_t)0;
#pragma unroll
        for (int c = 0; c < KPT; ++c) {
            const int i = c * kPreThreads + tid;
            if (i < p.N) keybuf[i] = desc_bits(mk[c] ? fill : sv[c]);
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < KPT; ++c) keys[c] = (lo + c < hi) ? keybuf[lo + c] : 0xffffffffu;
    } else {
        float sv[KPT];
        uint8_t mk[KPT];
#pragma unroll
        for (int c = 0; c < KPT; ++c) sv[c] = srow[min(lo + c, p.N - 1)];
#pragma unroll
        for (int c = 0; c < KPT; ++c) mk[c] = mrow ? mrow[min(lo + c, p.N - 1)] : (uint8_t)0;
#pragma unroll
        for (int c = 0; c < KPT; ++c) keys[c] = (lo + c < hi) ? desc_bits(mk[c] ? fill : sv[c]) : 0xffffffffu;
    }
    // strided sample of 256 keys: the first key of every 4th thread's chunk (0xffffffff = no key: sorts last).
    // Ranking 256 x 256 on four wavefronts costs ~2 us; a 1024-key sample ranked by all 16 wavefronts of
    // this single workgroup took ~40 us and bought only a ~15 % smaller candidate set.
    constexpr int kSample = kPreThreads / 4;
    if ((tid & 3) == 0) samp[tid >> 2] = keys[0];
    if (tid == 0) thr_s = 0xffffffffu;
    __syncthreads();
    const int owners = (p.N + chunk - 1) / chunk;  // threads that own at least one key
    const int S = (owners + 3) / 4;                // samples that are real keys
    // target sample rank: k*S/N plus five standard deviations of the binomial sample count
    const float frac = (float)p.k / (float)p.N;
    const int target = min(S - 1, (int)(frac * S + 5.f * sqrtf(fmaxf(frac * (1.f - frac) * S, 1.f)) + 1.f));
    if (tid < kSample) {
        const uint32_t mine = samp[tid];
        uint32_t rank = 0;
        const uint4 *s4 = reinterpret_cast<const uint4 *>(samp);
#pragma unroll 8
        for (int g = 0; g < kSample / 4; ++g) {
            const uint4 c = s4[g];
            const int j = g * 4;
            rank += (c.x < mine || (c.x == mine && j + 0 < tid)) ? 1u : 0u;
            rank += (c.y < mine || (c.y == mine && j + 1 < tid)) ? 1u : 0u;
            rank += (c.z < mine || (c.z == mine && j + 2 < tid)) ? 1u : 0u;
            rank += (c.w < mine || (c.w == mine && j + 3 < tid)) ? 1u : 0u;
        }
        if ((int)rank == target) thr_s = mine;  // ranks are a permutation: exactly one writer
    }
    __syncthreads();
    uint32_t thr = thr_s;
    uint32_t n_pass = 0;
#pragma unroll
    for (int c = 0; c < KPT; ++c) n_pass += (lo + c < hi && keys[c] <= thr) ? 1u : 0u;
    uint32_t total;
    uint32_t out = block_exclusive_scan(n_pass, scan_buf, tid, total);
    if (total < (uint32_t)p.k) {  // the sample was unlucky: let every key pass (still exact, just slower)
        thr = 0xffffffffu;
        n_pass = (uint32_t)max(0, hi - lo);
        __syncthreads();
        out = block_exclusive_scan(n_pass, scan_buf, tid, total);
    }
    uint32_t *ck = p.cand_key + (int64_t)b * p.N;
    uint32_t *cp = p.cand_pos + (int64_t)b * p.N;
#pragma unroll
    for (int c = 0; c < KPT; ++c) {
        if (lo + c < hi && keys[c] <= thr) {
            ck[out] = keys[c];
            cp[out] = (uint32_t)(lo + c);
            ++out;
        }
    }
    if (tid == 0) p.cand_count[b] = (int)total;
}

// ---- one-launch sorted top-k by histogram sort (one workgroup per row) -----------------------------------------------
// The per-layer top-300 of ~2 000-11 000 class scores (models/bricks/salience_transformer.py:366-367) and the finest
// level's top-6680 of 16 800 salience scores (:146-150) used to be two launches each -- a sampled-threshold prefilter
// on one workgroup, then the chip-wide rank-by-counting of the candidates (n_cand^2 compares): 17 us per encoder layer
// and 48 us for level 0, for problems whose data is 45-67 KB.  Rank by counting is quadratic; with the keys spread
// over bins that are MONOTONE in the sort order, a key's rank is (keys in earlier bins) + (bin mates that sort before
// it), which is linear for any row that does not pile up in single bins.  One 1024-thread workgroup per row:
//   1. keys (descending-orderable score bits, masked -> fill) into registers, thread t holds positions t, t + 1024, ...
//   2. the FLOOR group -- keys equal to the row's smallest score (the fill value floods it when an image is padded) --
//      is set aside: it sorts after everything else, ties by position, so its ranks are position prefix counts (a
//      bitmap of the members' positions + a scan of its popcounts), never comparisons; only needed when k exceeds the
//      number of other keys;
//   3. the other keys: bin = floor((smax - score) * nbins / (smax - smin)) over the FINITE range of the row -- linear in
//      the score, so a bell-shaped row fills the bins evenly where a digit of the key bits would pile a row into a
//      handful (round 2's radix attempt: 2-4x slower than the two launches); infinities clamp to the end bins;
//      ds_add_u32 histogram, block scan -> bin offsets, the bin in which the running count crosses k is the CUT bin;
//   4. keys of bins <= cut are scattered into the LDS list of their bin (slot from a returning ds_add); then one thread
//      per LIST ENTRY (neighbouring lanes = the same few bins: similar trip counts, near-broadcast reads) counts the bin
//      mates that sort before it (key, then position: the tie rule of the rank kernel); rank < k is the output slot;
//   5. CROWDED bins (> 48 keys) are left out of 4 and taken one at a time by the whole workgroup.  They come from
//      near-ties: the ~700 border tokens per image whose zeroed rows (base_transformer.py:104-111) give salience scores
//      that differ in the last bits or not at all, all inside one bin next to a few ordinary scores.  Up to 2048
//      entries: a bitonic sort of the 48-bit (key, position) pairs in LDS (55 barrier-separated steps for 1024 entries,
//      ~4 us, whatever the keys are), the sorted index is the rank inside the bin.  Above that: comparisons with the
//      mates (correct for every input, quadratic in the bin).
// Bit-identical to prefilter + rank.
constexpr int kHsThreads = 1024;
constexpr int kHsWaves = kHsThreads / 64;
constexpr int kHsBins = 4096;
constexpr int kHsBinsPerThread = kHsBins / kHsThreads;   // 4
constexpr int kHsMaxKpt = 17;
constexpr int kHsMaxN = kHsThreads * kHsMaxKpt;          // 17 408 keys: the scatter list covers a whole row
constexpr int kHsCrowd = 48;                             // bins above this many keys are taken cooperatively
constexpr int kHsMaxCrowded = 64;                        // listed crowded bins (further ones are ranked entry by entry)
constexpr int kHsMapWords = (kHsMaxN + 31) / 32;         // 544
constexpr uint32_t kHsFlag = 0x80000000u;                // in off[bin]: the bin is on the crowded list
constexpr int kHsTmp = 2048;                             // entries of a crowded bin that are sorted in LDS

// (benchmarks/micro/hsort_phases.hip compiles this file with SDETR_HS_STAMPS: cycle stamps of workgroup 0 at the phase
// boundaries.  Empty in the library.)
#ifdef SDETR_HS_STAMPS
__device__ unsigned long long hs_stamps[16];
#define HS_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) hs_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define HS_STAMP(i) do { } while (0)
#endif

struct SelectArgs {
    const float *score;
    const uint8_t *mask;
    int64_t mask_stride;
    const float *fill;
    const int64_t *payload;
    int N, k;
    int64_t index_offset;
    float *out_score;
    int64_t *out_index;
    int64_t out_stride;
};

// `sel_lds` (optional, k entries of static LDS): the output indices once more, for a consumer inside the same workgroup
// (topk_hsort_inproj_kernel); complete behind the body's last barrier + one more.
template <int KPT>
__device__ __forceinline__ void topk_hsort_body(const SelectArgs &p, const int b, uint32_t *hs_lds, int32_t *sel_lds = nullptr,
                                                const bool to_memory = true)
{
    // [off: kHsBins + 1 (+3 pad)][cur: kHsBins][list keys: N][list positions (u16): N]
    uint32_t *off = hs_lds;
    uint32_t *cur = hs_lds + kHsBins + 4;
    uint32_t *lkey = cur + kHsBins;
    uint16_t *lpos = reinterpret_cast<uint16_t *>(lkey + p.N);
    __shared__ uint32_t scan_buf[kHsWaves];
    __shared__ uint32_t red_u[kHsWaves];
    __shared__ float red_f[2 * kHsWaves];
    __shared__ uint32_t misc[4];                       // cut bin, crowded bins listed, (per crowded bin) equal / less counts
    __shared__ uint32_t crowded[kHsMaxCrowded];
    __shared__ uint32_t bitmap[kHsMapWords];           // positions of a tie group's members
    __shared__ uint32_t bit_prefix[kHsMapWords];       // members in the words before
    __shared__ uint64_t tmp[kHsTmp];                   // a crowded bin's (key << 16 | position) pairs while they are sorted
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *srow = p.score + (int64_t)b * p.N;
    const uint8_t *mrow = p.mask ? p.mask + (int64_t)b * p.mask_stride : nullptr;
    const float fill = p.fill ? *p.fill : 0.f;
    uint32_t keys[KPT];
    HS_STAMP(0);
    {
        float sv[KPT];
        uint8_t mk[KPT];
#pragma unroll
        for (int c = 0; c < KPT; ++c) sv[c] = srow[min(c * kHsThreads + tid, p.N - 1)];
#pragma unroll
        for (int c = 0; c < KPT; ++c) mk[c] = mrow ? mrow[min(c * kHsThreads + tid, p.N - 1)] : (uint8_t)0;
#pragma unroll
        for (int c = 0; c < KPT; ++c) keys[c] = desc_bits(mk[c] ? fill : sv[c]);
    }
#pragma unroll
    for (int i = 0; i < kHsBinsPerThread; ++i) cur[tid * kHsBinsPerThread + i] = 0u;
    if (tid == 0) { misc[0] = 0u; misc[1] = 0u; }
    auto real = [&](int c) { return c * kHsThreads + tid < p.N; };
    auto emit = [&](uint32_t rank, uint32_t key, int pos) {
        if (p.out_score && to_memory) p.out_score[(int64_t)b * p.out_stride + rank] = undesc_bits(key);
        const int64_t index = p.payload ? p.payload[(int64_t)b * p.N + pos] : (int64_t)pos + p.index_offset;
        if (to_memory) p.out_index[(int64_t)b * p.out_stride + rank] = index;
        if (sel_lds) sel_lds[rank] = (int32_t)index;
    };
    // Ranks of a group of EQUAL keys `tie` (every thread finds its members in its registers): base + number of members at
    // earlier positions, emitted when < limit.  Uniform control flow (barriers inside).
    auto rank_ties_by_position = [&](uint32_t tie, uint32_t base, uint32_t limit) {
        const int words = (p.N + 31) >> 5;
        if (tid < words) bitmap[tid] = 0u;
        __syncthreads();
#pragma unroll
        for (int c = 0; c < KPT; ++c) {
            const int pos = c * kHsThreads + tid;
            if (real(c) && keys[c] == tie) atomicOr(&bitmap[pos >> 5], 1u << (pos & 31));
        }
        __syncthreads();
        uint32_t total;
        const uint32_t before = block_exclusive_scan(tid < words ? (uint32_t)__popc(bitmap[tid]) : 0u, scan_buf, tid, total);
        if (tid < words) bit_prefix[tid] = before;
        __syncthreads();
#pragma unroll
        for (int c = 0; c < KPT; ++c) {
            const int pos = c * kHsThreads + tid;
            if (real(c) && keys[c] == tie) {
                const uint32_t rank = base + bit_prefix[pos >> 5] + (uint32_t)__popc(bitmap[pos >> 5] & ((1u << (pos & 31)) - 1u));
                if (rank < limit) emit(rank, tie, pos);
            }
        }
        __syncthreads();   // the bitmap is free again
    };

    // ---- one reduction: the largest key (= smallest score: the FLOOR group) and the finite score range ----
    uint32_t kmax = 0u;
    float smin = INFINITY, smax = -INFINITY;
#pragma unroll
    for (int c = 0; c < KPT; ++c) {
        if (!real(c)) continue;
        kmax = max(kmax, keys[c]);
        const float v = undesc_bits(keys[c]);
        if (fabsf(v) < INFINITY) { smin = fminf(smin, v); smax = fmaxf(smax, v); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, o, 64));
        smin = fminf(smin, __shfl_xor(smin, o, 64));
        smax = fmaxf(smax, __shfl_xor(smax, o, 64));
    }
    if (lane == 0) { red_u[wave] = kmax; red_f[wave] = smin; red_f[kHsWaves + wave] = smax; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < kHsWaves; ++w) {
        kmax = max(kmax, red_u[w]);
        smin = fminf(smin, red_f[w]);
        smax = fmaxf(smax, red_f[kHsWaves + w]);
    }
    // (the range includes the floor value when it is finite: a fill value far below an image's own scores costs bin
    // resolution, never exactness)
    const float span = smax - smin;
    // (no finite key, or one value: everything lands in bin 0 and is ranked among its mates)
    const float scale = (span > 0.f && span < INFINITY) ? (float)kHsBins / span : 0.f;
    const float top = (span >= 0.f && span < INFINITY) ? smax : 0.f;
    auto bin_of = [&](uint32_t key) -> uint32_t {
        // monotone in the key order: fl(top - v) and the product are monotone, the conversion truncates
        // keys outside the finite range (infinities, NaN bit patterns) sort before or after every finite key: end bins
        if (key <= 0x007fffffu) return 0u;                         // +inf and what the key order puts in front of it
        if (key >= 0xff800000u) return (uint32_t)(kHsBins - 1);    // -inf and what it puts behind
        const float d = (top - undesc_bits(key)) * scale;
        return (uint32_t)fminf(fmaxf(d, 0.f), (float)(kHsBins - 1));
    };
    HS_STAMP(1);   // keys loaded, floor group and score range known
    // ---- histogram of the other keys ----
#pragma unroll
    for (int c = 0; c < KPT; ++c)
        if (real(c) && keys[c] != kmax) atomicAdd(&cur[bin_of(keys[c])], 1u);
    __syncthreads();
    HS_STAMP(2);   // histogram
    uint32_t h[kHsBinsPerThread], hsum = 0;
#pragma unroll
    for (int i = 0; i < kHsBinsPerThread; ++i) { h[i] = cur[tid * kHsBinsPerThread + i]; hsum += h[i]; }
    uint32_t n_other;
    const uint32_t before = block_exclusive_scan(hsum, scan_buf, tid, n_other);   // total = keys outside the floor group
    const uint32_t kk = min((uint32_t)p.k, n_other);           // ranks [0, kk) go to them
    __syncthreads();
    {
        uint32_t run = before;
#pragma unroll
        for (int i = 0; i < kHsBinsPerThread; ++i) {
            const int bin = tid * kHsBinsPerThread + i;
            uint32_t flag = 0u;
            if (run < kk && h[i] > (uint32_t)kHsCrowd) {   // a crowded bin that holds ranks below k
                const uint32_t slot = atomicAdd(&misc[1], 1u);
                if (slot < (uint32_t)kHsMaxCrowded) { crowded[slot] = (uint32_t)bin; flag = kHsFlag; }
            }
            off[bin] = run | flag;
            cur[bin] = run;          // the bin's scatter cursor
            if (run < kk && kk <= run + h[i]) misc[0] = (uint32_t)bin;   // exactly one bin: the counts partition the keys
            run += h[i];
        }
        if (tid == kHsThreads - 1) off[kHsBins] = run;
    }
    __syncthreads();
    const uint32_t cut = misc[0];
    const uint32_t n_crowded = min(misc[1], (uint32_t)kHsMaxCrowded);
    const uint32_t n_keep = kk > 0 ? (off[cut + 1] & ~kHsFlag) : 0u;   // keys of the bins <= cut
    HS_STAMP(3);   // offsets, cut bin
    // ---- scatter the keys of bins <= cut into their bins' lists ----
    if (kk > 0) {
#pragma unroll
        for (int c = 0; c < KPT; ++c) {
            if (!real(c) || keys[c] == kmax) continue;
            const uint32_t bin = bin_of(keys[c]);
            if (bin > cut) continue;
            const uint32_t slot = atomicAdd(&cur[bin], 1u);
            lkey[slot] = keys[c];
            lpos[slot] = (uint16_t)(c * kHsThreads + tid);   // N <= 17 408 < 2^16
        }
    }
    __syncthreads();
    HS_STAMP(4);   // scatter
    // ---- one thread per list entry: rank among the bin mates (bins on the crowded list are skipped) ----
    for (uint32_t e = (uint32_t)tid; e < n_keep; e += kHsThreads) {
        const uint32_t key = lkey[e];
        const int pos = (int)lpos[e];
        const uint32_t bin = bin_of(key);
        const uint32_t o = off[bin];
        if (o & kHsFlag) continue;
        const uint32_t lo = o, hi = off[bin + 1] & ~kHsFlag;
        uint32_t rank = lo;
        // four mates per round, all eight LDS reads issued before the first compare (clamped indices: a read past the
        // bin's end repeats its last entry and is masked out)
        for (uint32_t j = lo; j < hi; j += 4) {
            uint32_t kj[4];
            int pj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t jj = min(j + u, hi - 1u);
                kj[u] = lkey[jj];
                pj[u] = (int)lpos[jj];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                rank += (j + u < hi && (kj[u] < key || (kj[u] == key && pj[u] < pos))) ? 1u : 0u;
        }
        if (rank < kk) emit(rank, key, pos);
    }
    HS_STAMP(5);   // ranks of the ordinary bins (thread 0's share)
    // ---- crowded bins, one at a time ----
    for (uint32_t ci = 0; ci < n_crowded; ++ci) {
        const uint32_t bin = crowded[ci];
        const uint32_t lo = off[bin] & ~kHsFlag, hi = off[bin + 1] & ~kHsFlag, m = hi - lo;
        if (m <= (uint32_t)kHsThreads) {
            // one entry per thread: partners less than a wavefront apart exchange through lane permutes (no LDS, no
            // barrier: 45 of the 55 steps of a 1024-entry sort), the others through `tmp`
            uint32_t P = 64;
            while (P < m) P <<= 1;
            uint64_t mine = (uint32_t)tid < m ? (((uint64_t)lkey[lo + tid] << 16) | (uint64_t)lpos[lo + tid]) : ~0ull;
            for (uint32_t size = 2; size <= P; size <<= 1) {
                const bool asc = ((uint32_t)tid & size) == 0u;
                for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
                    uint64_t other;
                    if (stride >= 64u) {
                        tmp[tid] = mine;
                        __syncthreads();
                        other = tmp[(uint32_t)tid ^ stride];
                        __syncthreads();
                    } else {
                        const uint32_t olo = (uint32_t)__shfl_xor((int)(uint32_t)mine, (int)stride, 64);
                        const uint32_t ohi = (uint32_t)__shfl_xor((int)(uint32_t)(mine >> 32), (int)stride, 64);
                        other = ((uint64_t)ohi << 32) | olo;
                    }
                    const bool lower = ((uint32_t)tid & stride) == 0u;
                    const uint64_t mn = mine < other ? mine : other, mx = mine < other ? other : mine;
                    mine = (lower == asc) ? mn : mx;
                }
            }
            // (threads >= P hold ~0 and only ever meet each other: P is a multiple of 64 and, for the LDS steps, a power
            // of two that their partner index stays above)
            if ((uint32_t)tid < m) {
                const uint32_t rank = lo + (uint32_t)tid;
                if (rank < kk) emit(rank, (uint32_t)(mine >> 16), (int)(mine & 0xffffu));
            }
        } else if (m <= (uint32_t)kHsTmp) {
            uint32_t P = 64;
            while (P < m) P <<= 1;
            for (uint32_t t = tid; t < P; t += kHsThreads)
                tmp[t] = t < m ? (((uint64_t)lkey[lo + t] << 16) | (uint64_t)lpos[lo + t]) : ~0ull;
            for (uint32_t size = 2; size <= P; size <<= 1) {
                for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
                    __syncthreads();
                    for (uint32_t t = tid; t < (P >> 1); t += kHsThreads) {
                        const uint32_t i = ((t & ~(stride - 1u)) << 1) | (t & (stride - 1u));   // bit `stride` clear
                        const uint32_t j = i | stride;
                        const uint64_t a = tmp[i], c = tmp[j];
                        if ((a > c) == ((i & size) == 0u)) { tmp[i] = c; tmp[j] = a; }
                    }
                }
            }
            __syncthreads();
            for (uint32_t t = tid; t < m; t += kHsThreads) {
                const uint64_t e = tmp[t];
                const uint32_t rank = lo + t;
                if (rank < kk) emit(rank, (uint32_t)(e >> 16), (int)(e & 0xffffu));
            }
            __syncthreads();   // tmp is rewritten by the next crowded bin
        } else {
            for (uint32_t e = lo + tid; e < hi; e += kHsThreads) {
                const uint32_t key = lkey[e];
                const int pos = (int)lpos[e];
                uint32_t rank = lo;
                for (uint32_t j = lo; j < hi; ++j) {
                    const uint32_t kj = lkey[j];
                    rank += (kj < key || (kj == key && (int)lpos[j] < pos)) ? 1u : 0u;
                }
                if (rank < kk) emit(rank, key, pos);
            }
        }
    }
    HS_STAMP(6);   // crowded bins
    // ---- the floor group fills the ranks the other keys leave ----
    if ((uint32_t)p.k > n_other) rank_ties_by_position(kmax, n_other, (uint32_t)p.k);
    HS_STAMP(7);
}

__global__ void __launch_bounds__(kRankThreads) topk_rank_kernel(RankArgs p)
{
    __shared__ __attribute__((aligned(16))) uint32_t rank_lds[kRankLdsWords];
    topk_rank_body(p, (int)blockIdx.x, (int)blockIdx.y, rank_lds, rank_lds + kRankTile);
}

// ---- merge of descending-sorted segments ------------------------------------------------------------------------
// The global sort of salience_transformer.py:156-158 runs over the concatenation of the per-level top-k results,
// each of which is already sorted (descending, ties in position order).  A stable descending sort of that
// concatenation is therefore an L-way merge: an element's final rank is its position inside its own segment plus, for
// every other segment, the number of elements that precede it there -- strictly greater scores, and equal scores only
// if that segment comes earlier in the concatenation.  One thread per element, L-1 binary searches: O(n log n)
// compares in total instead of the rank kernel's n^2, and bit-identical to the stable sort.
constexpr int kMaxSegments = 8;
struct MergeArgs {
    const float *score;       // [B, n]
    const int64_t *payload;   // [B, n]
    int seg_start[kMaxSegments + 1];
    int nseg, n;
    int64_t *out_index;       // [B, limit], rows out_stride apart
    float *out_score;         // the same or NULL
    int limit;                // ranks below it are written (n: the whole merge)
    int64_t out_stride;
    int64_t index_offset;     // added to the payload
};

__global__ void __launch_bounds__(256) merge_sorted_kernel(MergeArgs p)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n) return;
    const float *row = p.score + (int64_t)blockIdx.y * p.n;
    const uint32_t key = desc_bits(row[i]);   // ascending in this key == descending in score (same total order as the rank kernel)
    int seg = 0;
    for (int s = 1; s < p.nseg; ++s) seg = i >= p.seg_start[s] ? s : seg;
    int rank = i - p.seg_start[seg];
    for (int s = 0; s < p.nseg; ++s) {
        if (s == seg) continue;
        // number of elements of segment s with key < mine (s after my segment) or key <= mine (s before it)
        int lo = p.seg_start[s], hi = p.seg_start[s + 1];
        const bool inclusive = s < seg;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const uint32_t k = desc_bits(row[mid]);
            if (inclusive ? k <= key : k < key) lo = mid + 1; else hi = mid;
        }
        rank += lo - p.seg_start[s];
    }
    if (rank >= p.limit) return;
    const int64_t o = (int64_t)blockIdx.y * p.out_stride + rank;
    p.out_index[o] = p.payload[(int64_t)blockIdx.y * p.n + i] + p.index_offset;
    if (p.out_score) p.out_score[o] = row[i];
}

// The same merge with the row's keys staged in LDS first (rows of up to kMergeLdsKeys keys): the binary searches are
// chains of ~40 dependent reads per element -- from L2 that was the kernel's whole 13 us; from LDS ~2.
constexpr int kMergeLdsKeys = 24576;   // 96 KB of keys (level 0 of the 800 x 1333 pyramid: 16 800)
__device__ __forceinline__ void merge_sorted_lds_body(const MergeArgs &p, int bx, int by, uint32_t *mkeys)
{
    const float *row = p.score + (int64_t)by * p.n;
    // eight loads in flight per thread, then their LDS stores (one load -> wait -> store per iteration was a chain of
    // n / 1024 trips to memory: 16 us at 16 800 keys, most of this kernel)
    for (int i0 = threadIdx.x; i0 < p.n; i0 += 8 * 1024) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = row[min(i0 + u * 1024, p.n - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + u * 1024 < p.n) mkeys[i0 + u * 1024] = desc_bits(v[u]);
    }
    __syncthreads();
    const int i = bx * 1024 + threadIdx.x;
    if (i >= p.n) return;
    const uint32_t key = mkeys[i];
    int seg = 0;
    for (int s = 1; s < p.nseg; ++s) seg = i >= p.seg_start[s] ? s : seg;
    int rank = i - p.seg_start[seg];
    for (int s = 0; s < p.nseg; ++s) {
        if (s == seg) continue;
        int lo = p.seg_start[s], hi = p.seg_start[s + 1];
        const bool inclusive = s < seg;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const uint32_t k = mkeys[mid];
            if (inclusive ? k <= key : k < key) lo = mid + 1; else hi = mid;
        }
        rank += lo - p.seg_start[s];
    }
    if (rank >= p.limit) return;
    const int64_t o = (int64_t)by * p.out_stride + rank;
    p.out_index[o] = p.payload[(int64_t)by * p.n + i] + p.index_offset;
    if (p.out_score) p.out_score[o] = row[i];
}

__global__ void __launch_bounds__(1024) merge_sorted_lds_kernel(MergeArgs p)
{
    extern __shared__ uint32_t mkeys[];
    merge_sorted_lds_body(p, (int)blockIdx.x, (int)blockIdx.y, mkeys);
}

// The merge of the sliced top-k (a few workgroups on an otherwise idle chip) carrying a pending rank job -- the deferred
// top-k of the next coarser level, whose indices nobody needs before the global merge behind this launch (round 6: with
// the salience head hoisted, the modulation launch that used to carry it is as long as its own traffic, the job 11.6 us)
// -- and / or the token-space pass of the encoder's output (FinalizeJob: needed at the very end of the encoder).
// Blocks [0, merge_bx * merge_rows): the merge; then the rank job's workgroups (the first 512 threads); then the pass.
__global__ void __launch_bounds__(1024) merge_sorted_lds_rank_kernel(MergeArgs p, int merge_bx, int merge_rows, RankArgs rk,
                                                                     int rk_bx, int rk_blocks, int rk_tile, FinalizeJob fin,
                                                                     int fin_blocks)
{
    extern __shared__ uint32_t mkeys[];
    const int blk = (int)blockIdx.x, nm = merge_bx * merge_rows;
    if (blk < nm) {
        merge_sorted_lds_body(p, blk % merge_bx, blk / merge_bx, mkeys);
    } else if (blk < nm + rk_blocks) {
        if (threadIdx.x >= kRankThreads) return;
        const int r = blk - nm;
        topk_rank_body(rk, r % rk_bx, r / rk_bx, mkeys, mkeys + rk_tile);
    } else {
        finalize_all_role(fin, blk - nm - rk_blocks, fin_blocks);
    }
}

}  // namespace sdetr

using namespace sdetr;

template <int KPT>
__global__ void __launch_bounds__(kHsThreads) topk_hsort_kernel(SelectArgs p)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t hs_dyn[];
    topk_hsort_body<KPT>(p, (int)blockIdx.x, hs_dyn);
}

// the same launch carrying the encoder's row-order jobs (row_orders_core.h): workgroups [0, nsel) sort their row, workgroup
// nsel + k * batch + b builds the order of (image b, layer k).  The selection runs on 2 workgroups of an otherwise empty
// chip for ~10 us; as a launch of their own the orders cost the step ~17 us.
template <int KPT>
__global__ void __launch_bounds__(kHsThreads) topk_hsort_orders_kernel(SelectArgs p, int nsel, RowOrderArgs o)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t hs_dyn[];
    const int blk = (int)blockIdx.x;
    if (blk < nsel) {
        topk_hsort_body<KPT>(p, blk, hs_dyn);
    } else {
        const int j = blk - nsel;   // (part, image, layer)
        layer_row_orders_body(o, (j / o.parts) % o.batch, j / (o.parts * o.batch), j % o.parts, reinterpret_cast<uint16_t *>(hs_dyn));
    }
}

// The top-300 selection of an encoder layer TOGETHER with the in-projection of the selected rows (topk_attention.hip,
// launch 1 of the layer's self-attention): the selection is one workgroup's dependent chain (~7 us on an empty chip) and
// the in-projection behind it was a launch of 8.3-9.1 us that starts with two dependent trips to memory (the selection,
// then the rows it names).  Here every one of the image's `wgs` workgroups runs the SAME selection (deterministic: the same
// list in each; the first writes it out) and keeps it in LDS, then its first kHsTileWaves waves take one tile each of the
// image's (Npad / 32) x 24 in-projection tiles: no launch boundary, no trip for the selection, the rows' loads start the
// moment the last rank is known.  (Four tile waves per workgroup, as in the stand-alone launch: the operand loads are
// row-strided -- 32 cache lines per instruction -- and sixteen waves of them on one CU took 12 us in its address unit.)
// Workgroups >= nsel carry the row-order jobs as in topk_hsort_orders_kernel.
constexpr int kHsTileWaves = 4;

template <int KPT>
__global__ void __launch_bounds__(kHsThreads) topk_hsort_inproj_kernel(SelectArgs p, TkInArgs q, int wgs, int nsel, RowOrderArgs o)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t hs_dyn[];
    __shared__ int32_t sel_lds[kTkMaxSel];
    const int blk = (int)blockIdx.x;
    if (blk >= nsel) {
        const int j = blk - nsel;   // (part, image, layer)
        layer_row_orders_body(o, (j / o.parts) % o.batch, j / (o.parts * o.batch), j % o.parts, reinterpret_cast<uint16_t *>(hs_dyn));
        return;
    }
    const int b = blk / wgs, part = blk - b * wgs;
    topk_hsort_body<KPT>(p, b, hs_dyn, sel_lds, part == 0);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= kHsTileWaves) return;
    const int idx = part * kHsTileWaves + wave;      // the wave's tile among the image's (Npad / 32) x 24
    const int tile = idx / 24, ftile = idx - tile * 24;
    if (tile >= q.Npad / 32) return;
    const int i = tile * 32 + (lane & 31);
    inproj_wave_body<4>(q, b, tile, ftile, lane, (int64_t)sel_lds[min(i, q.N - 1)]);
}

// Rows beyond one workgroup's histogram sort (round 6; the per-layer top-300 of the reference's 5scale pyramid: up to 45 330
// rows): workgroup (b, s) keeps the sorted top-k of slice s of row b -- candidate scores and their positions IN THE ROW
// (index_offset = the slice's start) -- and the selection of the slices x k candidates with the positions as payload follows
// (topk_hsort_inproj_kernel or topk_hsort_kernel).  The row's top-k are among its slices' top-k, and ties stay in position
// order: the slices are concatenated in row order and each is sorted ties-by-position, so equal scores keep ascending
// positions in the candidate list, whose sort breaks ties by candidate position.  (filter_ops._sliced_topk is the same in
// five framework launches around two of these kernels: padding fill + copy, slice sort, offset add, clamp, final sort.)
template <int KPT>
__global__ void __launch_bounds__(kHsThreads) topk_hsort_slices_kernel(SelectArgs p, int slices, int slice_len, float *cand_score,
                                                                       int64_t *cand_pos)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t hs_dyn[];
    const int b = (int)blockIdx.x / slices, sl = (int)blockIdx.x - b * slices;
    const int start = sl * slice_len;
    SelectArgs q = p;
    q.score = p.score + (int64_t)b * p.N + start;
    q.N = min(slice_len, p.N - start);
    q.index_offset = start;
    q.out_score = cand_score + ((int64_t)b * slices + sl) * p.k;
    q.out_index = cand_pos + ((int64_t)b * slices + sl) * p.k;
    q.out_stride = p.k;
    topk_hsort_body<KPT>(q, 0, hs_dyn);
}

static bool use_select(int n, int k)
{
    // the shapes that used to take prefilter + rank (k well below n) and fit one workgroup: histogram sort in ONE launch
    // (SDETR_TOPK_SELECT=0: the two-launch path, for A/B runs)
    static const bool enabled = [] { const char *e = ab_env("SDETR_TOPK_SELECT"); return !(e && e[0] == '0'); }();
    if (!enabled) return false;
    return n >= 1024 && n <= kHsMaxN && (int64_t)k * 5 <= (int64_t)n * 2;
}

static bool use_prefilter(int n, int k)
{
    // worth it when the ranked set shrinks at least ~2x and the row fits the register-resident prefilter
    return n >= 2048 && n <= kPreThreads * 24 && (int64_t)k * 5 <= (int64_t)n * 2;
}

// whether sdetr_masked_topk_desc_f32 runs something else than the plain rank kernel for this shape (the one-launch
// select, or the sampled-threshold prefilter in front of the rank kernel): such a call cannot ride in another launch
extern "C" int sdetr_topk_uses_prefilter(int n, int k) { return (use_select(n, k) || use_prefilter(n, k)) ? 1 : 0; }

extern "C" size_t sdetr_topk_workspace_bytes(int B, int n, int k)
{
    if (B <= 0 || n <= 0 || k <= 0) return 0;
    size_t bytes = 16;  // one float: the masked-fill value
    if (use_select(n, k)) return bytes;
    if (use_prefilter(n, k)) bytes += (size_t)B * n * 8 + (size_t)B * 4 + 16;  // candidate keys, positions, counts
    return bytes;
}

static int fill_order_args(RowOrderArgs &o, const sdetr_row_orders_job *job)
{
    if (job->batch <= 0 || job->spatial_size <= 0 || job->num_rows <= 0 || job->num_layers <= 0) return fail("row-orders job: bad sizes");
    if (!job->sorted_index || !job->tile_pos || !job->counts || !job->order) return fail("row-orders job: null pointer");
    if (job->num_layers > kOrderMaxLayers || job->spatial_size > kOrderMaxTokens || job->num_rows >= 0xffff)
        return fail("row-orders job: too many layers / tokens / rows");
    const int64_t ibs = job->index_batch_stride ? job->index_batch_stride : job->num_rows;
    const int64_t obs = job->order_batch_stride ? job->order_batch_stride : job->num_rows;
    if (ibs < job->num_rows || obs < job->num_rows) return fail("row-orders job: bad strides");
    o = RowOrderArgs{};
    o.sorted_index = job->sorted_index; o.index_batch_stride = ibs; o.tile_pos = job->tile_pos; o.S = job->spatial_size;
    o.n0 = job->num_rows; o.nl = job->num_layers; o.batch = job->batch; o.counts_dev = job->counts; o.order = job->order;
    o.order_layer_stride = (int64_t)job->batch * obs; o.order_batch_stride = obs;
    return 0;
}

static int masked_topk_impl(sdetr_stream_t stream, const float *score, const uint8_t *mask, int64_t mask_row_stride,
                            int fill_mode, const float *fill_value, const int64_t *payload, int B, int n, int k,
                            int64_t index_offset, float *out_score, int64_t *out_index, int64_t out_row_stride,
                            void *workspace, size_t workspace_bytes, const sdetr_row_orders_job *job, int *carried);

extern "C" int sdetr_masked_topk_desc_f32(sdetr_stream_t stream, const float *score, const uint8_t *mask,
                                          int64_t mask_row_stride, int fill_mode, const float *fill_value,
                                          const int64_t *payload, int B, int n, int k,
                                          int64_t index_offset, float *out_score, int64_t *out_index,
                                          int64_t out_row_stride, void *workspace, size_t workspace_bytes)
{
    return masked_topk_impl(stream, score, mask, mask_row_stride, fill_mode, fill_value, payload, B, n, k, index_offset,
                            out_score, out_index, out_row_stride, workspace, workspace_bytes, nullptr, nullptr);
}

// sdetr_masked_topk_desc_f32 + a row-orders job (sdetr_layer_row_orders' arguments) that the selection's launch carries
// when it is the one-workgroup-per-row histogram sort; otherwise the job runs as a launch of its own behind it.
extern "C" int sdetr_masked_topk_desc_with_orders_f32(sdetr_stream_t stream, const float *score, const uint8_t *mask,
                                                      int64_t mask_row_stride, int fill_mode, const float *fill_value,
                                                      const int64_t *payload, int B, int n, int k, int64_t index_offset,
                                                      float *out_score, int64_t *out_index, int64_t out_row_stride,
                                                      void *workspace, size_t workspace_bytes,
                                                      const sdetr_row_orders_job *job)
{
    if (!job) return fail("masked_topk_with_orders: no job");
    int carried = 0;
    if (int rc = masked_topk_impl(stream, score, mask, mask_row_stride, fill_mode, fill_value, payload, B, n, k, index_offset,
                                  out_score, out_index, out_row_stride, workspace, workspace_bytes, job, &carried))
        return rc;
    if (carried) return 0;
    return sdetr_layer_row_orders(stream, job->sorted_index, job->index_batch_stride, job->tile_pos, job->batch,
                                  job->spatial_size, job->num_rows, job->num_layers, job->counts, job->order,
                                  job->order_batch_stride);
}

static int masked_topk_impl(sdetr_stream_t stream, const float *score, const uint8_t *mask, int64_t mask_row_stride,
                            int fill_mode, const float *fill_value, const int64_t *payload, int B, int n, int k,
                            int64_t index_offset, float *out_score, int64_t *out_index, int64_t out_row_stride,
                            void *workspace, size_t workspace_bytes, const sdetr_row_orders_job *job, int *carried)
{
    if (out_row_stride == 0) out_row_stride = k;
    if (out_row_stride < k) return fail("masked_topk: output row stride too small");
    if (B < 0 || n < 0 || k < 0) return fail("masked_topk: negative size");
    if (k > n) return fail("masked_topk: k (%d) out of range for a row of %d scores", k, n);
    if (fill_mode < 0 || fill_mode > 2) return fail("masked_topk: bad fill_mode %d", fill_mode);
    if (fill_mode == 0 && mask) return fail("masked_topk: a mask needs a fill mode");
    if (fill_mode == 2 && !fill_value) return fail("masked_topk: fill_mode 2 needs the device scalar fill_value");
    if (mask && mask_row_stride == 0) mask_row_stride = n;
    if (mask && mask_row_stride < n) return fail("masked_topk: mask row stride too small");
    if (B == 0 || k == 0) return 0;
    if (!score || !out_index) return fail("masked_topk: null pointer");
    if (n >= (1 << 30) || B > 65535) return fail("masked_topk: row too long / too many rows");
    RankArgs r{};
    r.score = score; r.mask = mask; r.mask_stride = mask_row_stride; r.payload = payload; r.N = n; r.k = k;
    r.index_offset = index_offset; r.out_score = out_score; r.out_index = out_index; r.out_stride = out_row_stride;
    if (fill_mode == 1) {
        if (!workspace || workspace_bytes < sizeof(float))
            return fail("masked_topk: needs %zu bytes of workspace, got %zu", sizeof(float), workspace_bytes);
        hipLaunchKernelGGL(topk_min_kernel, dim3(1), dim3(1024), 0, stream, score, (int64_t)B * n,
                           reinterpret_cast<float *>(workspace));
        if (int e = check_launch("topk_min")) return e;
        r.fill = reinterpret_cast<const float *>(workspace);
    } else if (fill_mode == 2) {
        r.fill = fill_value;
    }
    if (use_select(n, k)) {
        SelectArgs a{};
        a.score = score; a.mask = mask; a.mask_stride = mask_row_stride; a.fill = r.fill; a.payload = payload; a.N = n;
        a.k = k; a.index_offset = index_offset; a.out_score = out_score; a.out_index = out_index;
        a.out_stride = out_row_stride;
        const int chunk = (n + kHsThreads - 1) / kHsThreads;
        size_t dyn = ((size_t)(2 * kHsBins + 4) + (size_t)n) * 4 + (((size_t)n * 2 + 15) & ~(size_t)15);
        RowOrderArgs o{};
        int order_blocks = 0;
        if (job) {
            if (int rc = fill_order_args(o, job)) return rc;
            // (the launch's dynamic LDS limit next to the sort's static arrays)
            o.slot_cap = order_slot_cap(o.S, 136 * 1024);
            o.parts = o.slot_cap > 0 ? (o.S + o.slot_cap - 1) / o.slot_cap : 0;
            const size_t need = (((size_t)(o.slot_cap < o.S ? o.slot_cap : o.S) + 7) & ~(size_t)7) * 2;
            if (o.slot_cap > 0 && need <= 136 * 1024) {
                order_blocks = o.batch * o.nl * o.parts;
                if (need > dyn) dyn = need;
                if (carried) *carried = 1;
            }
        }
#define SDETR_HS(KPT)                                                                                               \
    do {                                                                                                            \
        if (order_blocks) {                                                                                         \
            static DeviceOnce lds_once2;                                                                            \
            allow_dynamic_lds(topk_hsort_orders_kernel<KPT>, lds_once2, 136 * 1024);                                \
            hipLaunchKernelGGL(topk_hsort_orders_kernel<KPT>, dim3((unsigned)(B + order_blocks)), dim3(kHsThreads), dyn, \
                               stream, a, B, o);                                                                    \
        } else {                                                                                                    \
            static DeviceOnce lds_once;                                                                             \
            allow_dynamic_lds(topk_hsort_kernel<KPT>, lds_once, 136 * 1024);   /* + ~14 KB of static LDS */         \
            hipLaunchKernelGGL(topk_hsort_kernel<KPT>, dim3((unsigned)B), dim3(kHsThreads), dyn, stream, a);        \
        }                                                                                                           \
    } while (0)
        if (chunk <= 3) SDETR_HS(3);
        else if (chunk <= 5) SDETR_HS(5);
        else if (chunk <= 7) SDETR_HS(7);
        else if (chunk <= 9) SDETR_HS(9);
        else if (chunk <= 12) SDETR_HS(12);
        else SDETR_HS(17);
#undef SDETR_HS
        return check_launch("topk_hsort");
    }
    int ranked = n;
    if (use_prefilter(n, k)) {
        const size_t need = sdetr_topk_workspace_bytes(B, n, k);
        if (!workspace || workspace_bytes < need)
            return fail("masked_topk: needs %zu bytes of workspace, got %zu", need, workspace_bytes);
        PrefilterArgs f{};
        f.score = score; f.mask = mask; f.mask_stride = mask_row_stride; f.fill = r.fill; f.N = n; f.k = k;
        f.cand_key = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(workspace) + 16);
        f.cand_pos = f.cand_key + (size_t)B * n;
        f.cand_count = reinterpret_cast<int32_t *>(f.cand_pos + (size_t)B * n);
        const int chunk = (n + kPreThreads - 1) / kPreThreads;
        // keys staged through LDS when they fit next to the kernel's static arrays
        f.stage = (size_t)n * 4 <= 150 * 1024 ? 1 : 0;
        const size_t dyn = f.stage ? (size_t)n * 4 : 0;
#define SDETR_PRE(KPT)                                                                                              \
    do {                                                                                                            \
        static DeviceOnce lds_once;                                                                                 \
        allow_dynamic_lds(topk_prefilter_kernel<KPT>, lds_once, 152 * 1024);                                        \
        hipLaunchKernelGGL(topk_prefilter_kernel<KPT>, dim3((unsigned)B), dim3(kPreThreads), dyn, stream, f);       \
    } while (0)
        if (chunk <= 5) SDETR_PRE(5);
        else if (chunk <= 12) SDETR_PRE(12);
        else if (chunk <= 17) SDETR_PRE(17);
        else SDETR_PRE(24);
#undef SDETR_PRE
        if (int e = check_launch("topk_prefilter")) return e;
        r.cand_key = f.cand_key; r.cand_pos = f.cand_pos; r.cand_count = f.cand_count;
        // expected candidates ~ k + 5 sigma (+ one sample stride); the grid covers the worst case (n) and
        // surplus workgroups exit on their first instruction
        ranked = n;
    }
    hipLaunchKernelGGL(topk_rank_kernel, dim3((unsigned)((ranked + 63) / 64), (unsigned)B), dim3(kRankThreads), 0, stream, r);
    return check_launch("topk_rank");
}

static int launch_merge(hipStream_t stream, const MergeArgs &a, int B, const sdetr_rank_job *rank = nullptr,
                        const sdetr_finalize_job *finalize = nullptr, bool *carried = nullptr)
{
    const int n = a.n;
    if (carried) *carried = false;
    if ((rank || finalize) && n <= kMergeLdsKeys) {
        RankArgs r{};
        int rk_bx = 1, rk_blocks = 0, rk_tile = kRankTile;
        size_t lds = (size_t)n * 4;
        if (rank) {
            if (rank->batch <= 0 || rank->n <= 0 || rank->k <= 0 || rank->k > rank->n || !rank->score || !rank->out_index)
                return fail("merge_sorted: bad rank job");
            if (rank->n >= (1 << 30) || rank->batch > 65535) return fail("merge_sorted: rank row too long / too many rows");
            const int64_t ors = rank->out_row_stride ? rank->out_row_stride : rank->k;
            const int64_t mrs = rank->mask && !rank->mask_row_stride ? rank->n : rank->mask_row_stride;
            if (ors < rank->k || (rank->mask && mrs < rank->n)) return fail("merge_sorted: rank job strides too small");
            if (rank->mask && !rank->fill_value) return fail("merge_sorted: a masked rank job needs its fill value");
            r.score = rank->score; r.mask = rank->mask; r.mask_stride = mrs; r.fill = rank->fill_value; r.N = rank->n;
            r.k = rank->k; r.index_offset = rank->index_offset; r.out_score = rank->out_score; r.out_index = rank->out_index;
            r.out_stride = ors;
            rk_bx = (rank->n + 63) / 64;
            rk_blocks = rk_bx * rank->batch;
            rk_tile = rk_bx * 64 < kRankTile ? rk_bx * 64 : kRankTile;
            if ((size_t)(rk_tile + kRankWaves * 64) * 4 > lds) lds = (size_t)(rk_tile + kRankWaves * 64) * 4;
        }
        FinalizeJob fj{};
        int fin_blocks = 0;
        if (finalize) {
            if (int rc = fill_finalize_job(fj, finalize, "merge_sorted")) return rc;
            fin_blocks = 192;   // x 1024 threads: ~14 pieces per thread
        }
        const int mbx = (n + 1023) / 1024;
        static DeviceOnce lds_once2;
        allow_dynamic_lds(merge_sorted_lds_rank_kernel, lds_once2, kMergeLdsKeys * 4);
        hipLaunchKernelGGL(merge_sorted_lds_rank_kernel, dim3((unsigned)(mbx * B + rk_blocks + fin_blocks)), dim3(1024), lds, stream,
                           a, mbx, B, r, rk_bx, rk_blocks, rk_tile, fj, fin_blocks);
        if (carried) *carried = true;
        return check_launch("merge_sorted (+ jobs)");
    }
    if (n <= kMergeLdsKeys) {
        static DeviceOnce lds_once;
        allow_dynamic_lds(merge_sorted_lds_kernel, lds_once, kMergeLdsKeys * 4);
        hipLaunchKernelGGL(merge_sorted_lds_kernel, dim3((unsigned)((n + 1023) / 1024), (unsigned)B), dim3(1024), (size_t)n * 4,
                           stream, a);
        return check_launch("merge_sorted");
    }
    hipLaunchKernelGGL(merge_sorted_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)B), dim3(256), 0, stream, a);
    return check_launch("merge_sorted");
}

extern "C" int sdetr_merge_sorted_desc(sdetr_stream_t stream, const float *score, const int64_t *payload,
                                       const int *segment_start, int num_segments, int B, int n, int64_t *out_index,
                                       float *out_score)
{
    if (B < 0 || n < 0 || num_segments <= 0 || num_segments > kMaxSegments) return fail("merge_sorted: bad sizes");
    if ((int64_t)B * n == 0) return 0;
    if (!score || !payload || !segment_start || !out_index) return fail("merge_sorted: null pointer");
    MergeArgs a{};
    a.score = score; a.payload = payload; a.nseg = num_segments; a.n = n; a.out_index = out_index; a.out_score = out_score;
    a.limit = n; a.out_stride = n; a.index_offset = 0;
    for (int s = 0; s < num_segments; ++s) {
        a.seg_start[s] = segment_start[s];
        if (segment_start[s] < 0 || segment_start[s] > n || (s > 0 && segment_start[s] < segment_start[s - 1]))
            return fail("merge_sorted: segment starts must be non-decreasing inside [0, n]");
    }
    if (segment_start[0] != 0) return fail("merge_sorted: the first segment starts at 0");
    a.seg_start[num_segments] = n;
    return launch_merge(static_cast<hipStream_t>(stream), a, B);
}

// Top-k with k a sizeable fraction of a long row, in two chip-wide launches (round 6; rounds 1-5: one histogram-sort
// workgroup per row -- 36 us on 2 workgroups for the finest level of the 800 x 1333 pyramid, 6680 of 16 800 --, round 5's
// python form of this for the 5scale pyramid: six launches).  The row is cut into `slices` <= 8 slices; launch 1 sorts
// every slice completely by rank counting (the rank kernel on rows 1/slices as long: 1/slices of the comparisons, the
// position payload and the strided mask handled in the kernel), launch 2 merges the sorted slices (stable in slice order,
// i.e. ties stay in position order -- the tie rule of the one-launch forms) and writes the first k.  Masked entries
// compete with *fill_value exactly as in sdetr_masked_topk_desc_f32 with fill_mode 2.
extern "C" size_t sdetr_topk_sliced_workspace_bytes(int B, int n) { return B > 0 && n > 0 ? (size_t)B * n * 12 + 16 : 0; }

extern "C" int sdetr_masked_topk_sliced_with_rank_f32(sdetr_stream_t stream, const float *score, const uint8_t *mask,
                                                      int64_t mask_row_stride, const float *fill_value, int B, int n, int k,
                                                      int slices, int64_t index_offset, float *out_score,
                                                      int64_t *out_index, int64_t out_row_stride, void *workspace,
                                                      size_t workspace_bytes, const sdetr_rank_job *rank,
                                                      const sdetr_finalize_job *finalize, int *jobs_carried);

extern "C" int sdetr_masked_topk_sliced_f32(sdetr_stream_t stream, const float *score, const uint8_t *mask,
                                            int64_t mask_row_stride, const float *fill_value, int B, int n, int k,
                                            int slices, int64_t index_offset, float *out_score, int64_t *out_index,
                                            int64_t out_row_stride, void *workspace, size_t workspace_bytes)
{
    return sdetr_masked_topk_sliced_with_rank_f32(stream, score, mask, mask_row_stride, fill_value, B, n, k, slices,
                                                  index_offset, out_score, out_index, out_row_stride, workspace,
                                                  workspace_bytes, nullptr, nullptr, nullptr);
}

// ... and a pending rank job and / or the finalize pass (NULL: none) in the merge launch; *jobs_carried (may be NULL) = 1
// when the launch took them (rows of up to 24 576 keys: the LDS form of the merge), else 0 and the caller launches them.
extern "C" int sdetr_masked_topk_sliced_with_rank_f32(sdetr_stream_t stream, const float *score, const uint8_t *mask,
                                                      int64_t mask_row_stride, const float *fill_value, int B, int n, int k,
                                                      int slices, int64_t index_offset, float *out_score,
                                                      int64_t *out_index, int64_t out_row_stride, void *workspace,
                                                      size_t workspace_bytes, const sdetr_rank_job *rank,
                                                      const sdetr_finalize_job *finalize, int *jobs_carried)
{
    if (jobs_carried) *jobs_carried = 0;
    if (B < 0 || n < 0 || k < 0) return fail("masked_topk_sliced: negative size");
    if (k > n) return fail("masked_topk_sliced: k (%d) out of range for a row of %d scores", k, n);
    if (slices < 2 || slices > kMaxSegments) return fail("masked_topk_sliced: 2 .. %d slices (got %d)", kMaxSegments, slices);
    if (out_row_stride == 0) out_row_stride = k;
    if (out_row_stride < k) return fail("masked_topk_sliced: output row stride too small");
    if (mask && !fill_value) return fail("masked_topk_sliced: a mask needs the device scalar fill_value");
    if (mask && mask_row_stride == 0) mask_row_stride = n;
    if (mask && mask_row_stride < n) return fail("masked_topk_sliced: mask row stride too small");
    if (B == 0 || k == 0) return 0;
    if (!score || !out_index) return fail("masked_topk_sliced: null pointer");
    if (n >= (1 << 30) || (int64_t)B * slices > 65535) return fail("masked_topk_sliced: row too long / too many rows");
    if (!workspace || workspace_bytes < sdetr_topk_sliced_workspace_bytes(B, n))
        return fail("masked_topk_sliced: needs %zu bytes of workspace, got %zu", sdetr_topk_sliced_workspace_bytes(B, n),
                    workspace_bytes);
    const int c = (n + slices - 1) / slices;
    const int used = (n + c - 1) / c;   // slices that hold keys
    int64_t *ws_pos = reinterpret_cast<int64_t *>(workspace);                    // [B, n] column of every sorted key
    float *ws_score = reinterpret_cast<float *>(ws_pos + (size_t)B * n);         // [B, n] the sorted slices, back to back
    hipStream_t hs = static_cast<hipStream_t>(stream);
    RankArgs r{};
    r.score = score; r.mask = mask; r.mask_stride = mask_row_stride; r.fill = mask ? fill_value : nullptr; r.N = n; r.k = c;
    r.index_offset = 0; r.out_score = ws_score; r.out_index = ws_pos; r.out_stride = n; r.slices = used; r.slice_len = c;
    hipLaunchKernelGGL(topk_rank_kernel, dim3((unsigned)((c + 63) / 64), (unsigned)(B * used)), dim3(kRankThreads), 0, hs, r);
    if (int e = check_launch("topk_rank (slices)")) return e;
    MergeArgs a{};
    a.score = ws_score; a.payload = ws_pos; a.nseg = used; a.n = n; a.out_index = out_index; a.out_score = out_score;
    a.limit = k; a.out_stride = out_row_stride; a.index_offset = index_offset;
    for (int s = 0; s < used; ++s) a.seg_start[s] = s * c;
    a.seg_start[used] = n;
    bool carried = false;
    const int rc = launch_merge(hs, a, B, rank, finalize, &carried);
    if (jobs_carried) *jobs_carried = carried ? 1 : 0;
    return rc;
}

// Slices of a long row for topk_hsort_slices_kernel: as few as cover the row with slices one workgroup sorts, but enough
// candidates (slices x k) for the one-workgroup selection that follows; balanced lengths (every slice holds >= k keys).
static bool select_slices(int n, int k, int *slices, int *slice_len)
{
    int s = (n + 8191) / 8192;
    while (s * k < 1024) ++s;
    const int len = (n + s - 1) / s;
    *slices = s; *slice_len = len;
    // every slice and the candidate row must be rows the histogram sort takes; the last slice is the shortest
    const int last = n - (s - 1) * len;
    return len <= kHsMaxN && last >= 1024 && (int64_t)k * 5 <= (int64_t)last * 2 && s * k <= kHsMaxN && k <= last;
}

extern "C" int64_t sdetr_topk_select_candidate_bytes(int batch_size, int n, int k)
{
    int slices = 0, slice_len = 0;
    if (n <= kHsMaxN || batch_size <= 0 || k <= 0 || !select_slices(n, k, &slices, &slice_len)) return 0;
    return (((int64_t)batch_size * slices * k * 4 + 15) & ~(int64_t)15) + (int64_t)batch_size * slices * k * 8;
}

// The layer's top-k selection (no mask, positions as indices) and the in-projection of the selected rows in ONE launch
// (topk_hsort_inproj_kernel); sdetr_topk_attention_with_projection_bf16(..., in_projection_done = 1) follows.  Rows the
// one-workgroup histogram sort covers (1024 <= n <= 17 408, 5 k <= 2 n) with k <= 384 -- and longer rows in TWO launches
// (their slices' top-k first, topk_hsort_slices_kernel, into `candidate_workspace` of sdetr_topk_select_candidate_bytes
// bytes; that query returns 0 for a row the one launch takes or the sliced form does not); `job` as in
// sdetr_masked_topk_desc_with_orders_f32 (carried by the launch when its slots fit, launched behind it otherwise).
extern "C" int sdetr_topk_select_inproj_bf16(sdetr_stream_t stream, const float *score, int batch_size, int n, int k,
                                             int64_t *out_index, const void *query, int64_t query_batch_stride,
                                             const void *pos, int64_t pos_batch_stride, const void *in_proj_weight,
                                             const void *in_proj_bias, void *workspace, int64_t workspace_bytes,
                                             int32_t *hint, int64_t hint_batch_stride, const sdetr_row_orders_job *job,
                                             void *candidate_workspace, int64_t candidate_bytes)
{
    if (batch_size <= 0 || n <= 0 || k <= 0 || k > n) return fail("topk_select_inproj: bad sizes (n %d, k %d)", n, k);
    if (!score || !out_index || !query || !pos || !in_proj_weight || !in_proj_bias || !workspace)
        return fail("topk_select_inproj: null pointer");
    const int rows = n;                    // the layer's rows (strides, hint); n becomes the keys of the final selection
    const float *sel_score = score;
    const int64_t *sel_payload = nullptr;
    if (n > kHsMaxN) {
        // a long row: its slices' top-k first (one more launch), the selection below then runs on the candidates
        int slices = 0, slice_len = 0;
        if (!select_slices(n, k, &slices, &slice_len))
            return fail("topk_select_inproj: rows of %d scores with k = %d are beyond the sliced form", n, k);
        const int64_t need = sdetr_topk_select_candidate_bytes(batch_size, n, k);
        if (!candidate_workspace || candidate_bytes < need)
            return fail("topk_select_inproj: %lld bytes of candidate workspace needed", (long long)need);
        float *cs = static_cast<float *>(candidate_workspace);
        int64_t *cp = reinterpret_cast<int64_t *>(static_cast<char *>(candidate_workspace) +
                                                  (((int64_t)batch_size * slices * k * 4 + 15) & ~(int64_t)15));
        SelectArgs sa{};
        sa.score = score; sa.N = n; sa.k = k;
        const int schunk = (slice_len + kHsThreads - 1) / kHsThreads;
        const size_t sdyn = ((size_t)(2 * kHsBins + 4) + (size_t)slice_len) * 4 + (((size_t)slice_len * 2 + 15) & ~(size_t)15);
#define SDETR_HSS(KPT)                                                                                              \
    do {                                                                                                            \
        static DeviceOnce lds_once4;                                                                                \
        allow_dynamic_lds(topk_hsort_slices_kernel<KPT>, lds_once4, 136 * 1024);                                    \
        hipLaunchKernelGGL(topk_hsort_slices_kernel<KPT>, dim3((unsigned)(batch_size * slices)), dim3(kHsThreads), sdyn, \
                           static_cast<hipStream_t>(stream), sa, slices, slice_len, cs, cp);                        \
    } while (0)
        if (schunk <= 3) SDETR_HSS(3);
        else if (schunk <= 5) SDETR_HSS(5);
        else if (schunk <= 7) SDETR_HSS(7);
        else if (schunk <= 9) SDETR_HSS(9);
        else if (schunk <= 12) SDETR_HSS(12);
        else SDETR_HSS(17);
#undef SDETR_HSS
        if (int rc = check_launch("topk_select_slices")) return rc;
        sel_score = cs; sel_payload = cp; n = slices * k;
    }
    if (!use_select(n, k) || k > kTkMaxSel)
        return fail("topk_select_inproj: rows of 1024..%d scores with 5 k <= 2 n and k <= %d (got n %d, k %d)", kHsMaxN, kTkMaxSel, n, k);
    const int npad = (k + 31) / 32 * 32;
    if (workspace_bytes < (int64_t)batch_size * npad * (512 + 256) * 2) return fail("topk_select_inproj: workspace too small");
    if (query_batch_stride < (int64_t)rows * kTkE || pos_batch_stride < (int64_t)rows * kTkE || (query_batch_stride & 7) || (pos_batch_stride & 7))
        return fail("topk_select_inproj: bad batch strides");
    if (hint && hint_batch_stride < rows) return fail("topk_select_inproj: hint rows shorter than the layer");
    SelectArgs a{};
    a.score = sel_score; a.payload = sel_payload; a.N = n; a.k = k; a.out_index = out_index; a.out_stride = k;
    TkInArgs q{};
    q.query = (const bf16_t *)query; q.q_bs = query_batch_stride; q.pos = (const bf16_t *)pos; q.p_bs = pos_batch_stride;
    q.sel = out_index; q.w = (const bf16_t *)in_proj_weight; q.bias = (const bf16_t *)in_proj_bias;
    q.qk = (bf16_t *)workspace; q.vt = q.qk + (int64_t)batch_size * npad * 512;
    q.B = batch_size; q.N = k; q.Npad = npad; q.hint = hint; q.hint_bs = hint_batch_stride;
    const int wgs = ((npad / 32) * 24 + kHsTileWaves - 1) / kHsTileWaves;
    const int chunk = (n + kHsThreads - 1) / kHsThreads;
    size_t dyn = ((size_t)(2 * kHsBins + 4) + (size_t)n) * 4 + (((size_t)n * 2 + 15) & ~(size_t)15);
    RowOrderArgs o{};
    int order_blocks = 0;
    if (job) {
        if (int rc = fill_order_args(o, job)) return rc;
        o.slot_cap = order_slot_cap(o.S, 136 * 1024);
        o.parts = o.slot_cap > 0 ? (o.S + o.slot_cap - 1) / o.slot_cap : 0;
        const size_t need = (((size_t)(o.slot_cap < o.S ? o.slot_cap : o.S) + 7) & ~(size_t)7) * 2;
        if (o.slot_cap > 0 && need <= 136 * 1024) {
            order_blocks = o.batch * o.nl * o.parts;
            if (need > dyn) dyn = need;
        }
    }
    const int nsel = batch_size * wgs;
#define SDETR_HSI(KPT)                                                                                              \
    do {                                                                                                            \
        static DeviceOnce lds_once3;                                                                                \
        allow_dynamic_lds(topk_hsort_inproj_kernel<KPT>, lds_once3, 136 * 1024);                                    \
        hipLaunchKernelGGL(topk_hsort_inproj_kernel<KPT>, dim3((unsigned)(nsel + order_blocks)), dim3(kHsThreads), dyn, \
                           static_cast<hipStream_t>(stream), a, q, wgs, nsel, o);                                   \
    } while (0)
    if (chunk <= 3) SDETR_HSI(3);
    else if (chunk <= 5) SDETR_HSI(5);
    else if (chunk <= 7) SDETR_HSI(7);
    else if (chunk <= 9) SDETR_HSI(9);
    else if (chunk <= 12) SDETR_HSI(12);
    else SDETR_HSI(17);
#undef SDETR_HSI
    if (int rc = check_launch("topk_select_inproj")) return rc;
    if (job && !order_blocks)   // (a pyramid whose orders do not fit the launch: on their own)
        return sdetr_layer_row_orders(stream, job->sorted_index, job->index_batch_stride, job->tile_pos, job->batch,
                                      job->spatial_size, job->num_rows, job->num_layers, job->counts, job->order,
                                      job->order_batch_stride);
    return 0;
}
