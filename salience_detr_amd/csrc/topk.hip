// Masked top-k (sorted descending, ties -> lower index first) for the salience filtering stage.
//
// Replaces the torch.topk / torch.sort calls of models/bricks/salience_transformer.py:146-150
// (per-level top-k under masked_fill(mask, score.min())), :156-158 (global sort of the selected
// scores + index gather) and :366-367 (per-layer top-300).
//
// A sorting network confined to one CU is LDS-bandwidth bound (a 16 K-key bitonic sort is 105
// passes over 128 KiB of LDS; measured 190-350 us), so the sort is split into two launches that
// use the whole chip and contain no sorting network at all:
//
//  1. topk_select (one 1024-thread workgroup per row): every thread keeps a CONTIGUOUS chunk of
//     the row's keys in registers (key = descending-orderable score bits, masked entries replaced
//     by the whole-array minimum as the reference does).  A 32-step bitwise search finds the k-th
//     key (one block-wide count per bit, no LDS atomics), then a stable block scan compacts the k
//     survivors -- in index order, ties at the threshold resolved to the lowest indices -- into a
//     global scratch list (u32 key, u32 position).
//  2. topk_rank (ceil(k/64) workgroups per row): rank by counting.  Workgroup w owns survivors
//     [64w, 64w+64); its four wavefronts stream the survivor key list (staged in LDS, read as
//     broadcast ds_read_b128) and count, per owned key, the keys that sort before it.  Because
//     the list is in index order the tie rule is positional: "<=" for keys listed before the
//     owned block, "<" after it, exact only inside it -- two VALU ops per comparison.  The rank
//     IS the output slot, so results are written directly, no merge step.
//     Work is k^2 comparisons (45 M for k = 6 680) spread over k/64 workgroups: ~3 us of VALU
//     per wavefront instead of a 91-pass single-CU sort.
#include "common.h"

namespace sdetr {

constexpr int kSelThreads = 1024;
constexpr int kSelWaves = kSelThreads / 64;
constexpr int kMaxKeysPerThread = 24;    // register-resident rows up to 24 576 scores
constexpr int kRankThreads = 256;
constexpr int kRankTile = 12288;         // survivor keys staged per LDS round (48 KiB)

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

struct SelectArgs {
    const float *score;
    const uint8_t *mask;
    int fill_mode;
    int B, N, k;
    uint32_t *cand_key;  // [B][k]
    uint32_t *cand_pos;  // [B][k]
};

__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t *buf /*[kSelWaves]*/, int tid)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((tid & 63) == 0) buf[tid >> 6] = v;
    __syncthreads();
    uint32_t t = 0;
#pragma unroll
    for (int w = 0; w < kSelWaves; ++w) t += buf[w];
    return t;
}

// exclusive prefix over the block's threads (thread order) + grand total
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *buf /*[kSelWaves]*/, int tid,
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
    for (int w = 0; w < kSelWaves; ++w) {
        const uint32_t c = buf[w];
        if (w < (tid >> 6)) before += c;
        t += c;
    }
    total = t;
    return before + incl - v;
}

// KPT = keys held in registers per thread (0: row too long, re-read it from global every pass)
template <int KPT>
__global__ void __launch_bounds__(kSelThreads) topk_select_kernel(SelectArgs p)
{
    constexpr bool IN_REGS = KPT > 0;
    constexpr int kKeysPerThread = KPT > 0 ? KPT : 1;
    __shared__ uint32_t bufA[kSelWaves], bufB[kSelWaves];
    __shared__ float redf[kSelWaves];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const float *srow = p.score + (int64_t)b * p.N;
    const uint8_t *mrow = p.mask ? p.mask + (int64_t)b * p.N : nullptr;

    float fill = 0.f;
    if (p.fill_mode == 1) {
        float mn = INFINITY;
        const int64_t total = (int64_t)p.B * p.N;
        for (int64_t i = tid; i < total; i += kSelThreads) mn = fminf(mn, p.score[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o, 64));
        if ((tid & 63) == 0) redf[tid >> 6] = mn;
        __syncthreads();
        mn = redf[0];
#pragma unroll
        for (int w = 1; w < kSelWaves; ++w) mn = fminf(mn, redf[w]);
        fill = mn;
    }

    // contiguous chunk per thread => compaction in thread order is compaction in index order
    const int chunk = (p.N + kSelThreads - 1) / kSelThreads;
    const int lo = tid * chunk;
    const int hi = min(p.N, lo + chunk);
    auto load_key = [&](int i) -> uint32_t {
        float s = srow[i];
        if (mrow && mrow[i]) s = fill;
        return desc_bits(s);
    };
    uint32_t keys[kKeysPerThread];
    if (IN_REGS) {
#pragma unroll
        for (int c = 0; c < kKeysPerThread; ++c) keys[c] = (lo + c < hi) ? load_key(lo + c) : 0xffffffffu;
    }

    uint32_t threshold = 0xffffffffu, need_eq = 0;
    if (p.k < p.N) {
        uint32_t prefix = 0, rem = p.k;
        for (int bit = 31; bit >= 0; --bit) {
            // keys matching the decided prefix whose current bit is 0  <=>  (key >> bit) == (prefix >> bit)
            // (prefix has bit `bit` and everything below still clear).  Out-of-range slots hold 0xffffffff and
            // can only match an all-ones prefix, which is excluded by the range test.
            const uint32_t want = prefix >> bit;
            // wave-wide count without any cross-lane data movement: the compare mask IS the ballot
            // (v_cmp -> SGPR pair), popcounted on the scalar unit
            uint32_t cnt = 0;
            if (IN_REGS) {
#pragma unroll
                for (int c = 0; c < kKeysPerThread; ++c)
                    cnt += (uint32_t)__popcll(__ballot(((keys[c] >> bit) == want) && (lo + c < hi)));
            } else {
                const int steps = chunk;  // uniform trip count so every lane reaches the ballot
                for (int c = 0; c < steps; ++c) {
                    const int i = lo + c;
                    cnt += (uint32_t)__popcll(__ballot(i < hi && (load_key(min(i, p.N - 1)) >> bit) == want));
                }
            }
            uint32_t *buf = (bit & 1) ? bufA : bufB;  // alternate buffers: one barrier per bit
            if ((tid & 63) == 0) buf[tid >> 6] = cnt;
            __syncthreads();
            uint32_t zeros = 0;
#pragma unroll
            for (int w = 0; w < kSelWaves; ++w) zeros += buf[w];
            if (rem > zeros) {
                prefix |= 1u << bit;
                rem -= zeros;
            }
        }
        threshold = prefix;  // the k-th key in sorted order (with multiplicity)
        need_eq = rem;       // how many keys == threshold belong to the top k (lowest indices first)
    }

    // stable compaction: all keys < threshold, plus the first need_eq keys == threshold
    uint32_t n_lt = 0, n_eq = 0;
    if (p.k < p.N) {
        if (IN_REGS) {
#pragma unroll
            for (int c = 0; c < kKeysPerThread; ++c) {
                if (lo + c < hi) {
                    n_lt += keys[c] < threshold;
                    n_eq += keys[c] == threshold;
                }
            }
        } else {
            for (int i = lo; i < hi; ++i) {
                const uint32_t key = load_key(i);
                n_lt += key < threshold;
                n_eq += key == threshold;
            }
        }
    }
    __syncthreads();
    uint32_t tot_eq, tot_sel;
    const uint32_t eq_before = (p.k < p.N) ? block_exclusive_scan(n_eq, bufA, tid, tot_eq) : 0u;
    uint32_t my_sel;
    if (p.k < p.N) {
        const uint32_t eq_take = eq_before >= need_eq ? 0u : min(n_eq, need_eq - eq_before);
        my_sel = n_lt + eq_take;
    } else {
        my_sel = (uint32_t)max(0, hi - lo);
    }
    __syncthreads();
    uint32_t out = block_exclusive_scan(my_sel, bufB, tid, tot_sel);
    uint32_t eq_seen = eq_before;
    uint32_t *ck = p.cand_key + (int64_t)b * p.k;
    uint32_t *cp = p.cand_pos + (int64_t)b * p.k;
    auto emit = [&](uint32_t key, int i) {
        bool take = true;
        if (p.k < p.N) {
            take = key < threshold;
            if (key == threshold) take = (eq_seen++ < need_eq);
        }
        if (take) {
            ck[out] = key;
            cp[out] = (uint32_t)i;
            ++out;
        }
    };
    if (IN_REGS) {
#pragma unroll
        for (int c = 0; c < kKeysPerThread; ++c)
            if (lo + c < hi) emit(keys[c], lo + c);
    } else {
        for (int i = lo; i < hi; ++i) emit(load_key(i), i);
    }
}

struct RankArgs {
    const uint32_t *cand_key;
    const uint32_t *cand_pos;
    const int64_t *payload;
    int N, k;
    int64_t index_offset;
    float *out_score;
    int64_t *out_index;
};

__global__ void __launch_bounds__(kRankThreads) topk_rank_kernel(RankArgs p)
{
    __shared__ __attribute__((aligned(16))) uint32_t tile[kRankTile];
    __shared__ uint32_t partial[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int base = blockIdx.x * 64;  // owned survivors [base, base+64)
    const uint32_t *ck = p.cand_key + (int64_t)b * p.k;
    const int mypos = base + lane;
    const uint32_t mine = mypos < p.k ? ck[mypos] : 0u;
    uint32_t rank = 0;

    for (int t0 = 0; t0 < p.k; t0 += kRankTile) {
        const int tn = min(kRankTile, p.k - t0);
        if (t0 > 0) __syncthreads();
        for (int i = tid; i < (tn + 3) / 4; i += kRankThreads) {
            uint4 v;
            const int j = t0 + i * 4;
            if (j + 3 < p.k && ((reinterpret_cast<uintptr_t>(ck + j) & 15) == 0)) {
                v = *reinterpret_cast<const uint4 *>(ck + j);
            } else {  // pad with the worst key: never counted by "<" or "<=" against a real key ... except
                      // equal 0xffffffff keys, which the position test below excludes (j >= k)
                v.x = j + 0 < p.k ? ck[j + 0] : 0xffffffffu;
                v.y = j + 1 < p.k ? ck[j + 1] : 0xffffffffu;
                v.z = j + 2 < p.k ? ck[j + 2] : 0xffffffffu;
                v.w = j + 3 < p.k ? ck[j + 3] : 0xffffffffu;
            }
            reinterpret_cast<uint4 *>(tile)[i] = v;
        }
        __syncthreads();
        // groups of 4 keys, round-robin over the 4 wavefronts
        const int ngroups = (tn + 3) / 4;
        for (int g = wave; g < ngroups; g += 4) {
            const uint4 c = reinterpret_cast<const uint4 *>(tile)[g];  // same address in all lanes: broadcast
            const int j = t0 + g * 4;
            if (j + 3 < base) {  // entirely before the owned block: ties sort before us
                rank += (c.x <= mine) + (c.y <= mine) + (c.z <= mine) + (c.w <= mine);
            } else if (j >= base + 64) {  // entirely after: ties sort after us (padding keys are > or == -> excluded)
                rank += (c.x < mine) + (c.y < mine) + (c.z < mine) + (c.w < mine);
            } else {  // inside the owned block: exact positional tie rule
                rank += (c.x < mine || (c.x == mine && j + 0 < mypos)) ? 1u : 0u;
                rank += (c.y < mine || (c.y == mine && j + 1 < mypos)) ? 1u : 0u;
                rank += (c.z < mine || (c.z == mine && j + 2 < mypos)) ? 1u : 0u;
                rank += (c.w < mine || (c.w == mine && j + 3 < mypos)) ? 1u : 0u;
            }
        }
    }
    partial[wave][lane] = rank;
    __syncthreads();
    if (wave == 0 && mypos < p.k) {
        const uint32_t r = partial[0][lane] + partial[1][lane] + partial[2][lane] + partial[3][lane];
        const uint32_t pos = p.cand_pos[(int64_t)b * p.k + mypos];
        if (p.out_score) p.out_score[(int64_t)b * p.k + r] = undesc_bits(mine);
        p.out_index[(int64_t)b * p.k + r] =
            p.payload ? p.payload[(int64_t)b * p.N + pos] : (int64_t)pos + p.index_offset;
    }
}

}  // namespace sdetr

using namespace sdetr;

extern "C" size_t sdetr_topk_workspace_bytes(int B, int n, int k)
{
    if (B <= 0 || n <= 0 || k <= 0) return 0;
    return (size_t)B * k * 2 * sizeof(uint32_t);
}

extern "C" int sdetr_masked_topk_desc_f32(sdetr_stream_t stream, const float *score, const uint8_t *mask,
                                          int fill_mode, const int64_t *payload, int B, int n, int k,
                                          int64_t index_offset, float *out_score, int64_t *out_index, void *workspace,
                                          size_t workspace_bytes)
{
    if (B < 0 || n < 0 || k < 0) return fail("masked_topk: negative size");
    if (k > n) return fail("masked_topk: k (%d) out of range for a row of %d scores", k, n);
    if (fill_mode != 0 && fill_mode != 1) return fail("masked_topk: bad fill_mode %d", fill_mode);
    if (fill_mode == 0 && mask) return fail("masked_topk: a mask needs fill_mode 1");
    if (B == 0 || k == 0) return 0;
    if (!score || !out_index) return fail("masked_topk: null pointer");
    if (n >= (1 << 30) || B > 65535) return fail("masked_topk: row too long / too many rows");
    const size_t need = sdetr_topk_workspace_bytes(B, n, k);
    if (!workspace || workspace_bytes < need)
        return fail("masked_topk: needs %zu bytes of workspace, got %zu", need, workspace_bytes);
    // padding keys of the rank kernel (0xffffffff) tie with a real key only for score == -NaN patterns;
    // the "<"/position rules keep them out of every count because their positions are >= k.
    SelectArgs s{};
    s.score = score; s.mask = mask; s.fill_mode = fill_mode; s.B = B; s.N = n; s.k = k;
    s.cand_key = reinterpret_cast<uint32_t *>(workspace);
    s.cand_pos = s.cand_key + (size_t)B * k;
    const int chunk = (n + kSelThreads - 1) / kSelThreads;
#define SDETR_SEL(KPT) hipLaunchKernelGGL(topk_select_kernel<KPT>, dim3((unsigned)B), dim3(kSelThreads), 0, stream, s)
    if (chunk <= 2) SDETR_SEL(2);
    else if (chunk <= 5) SDETR_SEL(5);
    else if (chunk <= 12) SDETR_SEL(12);
    else if (chunk <= 17) SDETR_SEL(17);
    else if (chunk <= kMaxKeysPerThread) SDETR_SEL(kMaxKeysPerThread);
    else SDETR_SEL(0);
#undef SDETR_SEL
    if (int e = check_launch("topk_select")) return e;
    RankArgs r{};
    r.cand_key = s.cand_key; r.cand_pos = s.cand_pos; r.payload = payload; r.N = n; r.k = k;
    r.index_offset = index_offset; r.out_score = out_score; r.out_index = out_index;
    hipLaunchKernelGGL(topk_rank_kernel, dim3((unsigned)((k + 63) / 64), (unsigned)B), dim3(kRankThreads), 0, stream, r);
    return check_launch("topk_rank");
}
