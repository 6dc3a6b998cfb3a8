// Masked top-k (sorted descending, ties -> lower index first) for the salience filtering stage.
//
// Replaces the torch.topk / torch.sort calls of models/bricks/salience_transformer.py:146-150
// (per-level top-k under masked_fill(mask, score.min())), :156-158 (global sort of the selected
// scores + index gather) and :366-367 (per-layer top-300).
//
// The problem sizes are LDS-sized (largest segment 16 800 scores, k <= 6 680; global sort of
// 11 363 keys), so ONE 1024-thread workgroup owns one batch row end to end and nothing but the
// final k (score, index) pairs ever leaves the CU:
//   1. (fill_mode 1) min over the whole [B,N] score array -- every workgroup recomputes it, which
//      is cheaper than a separate launch + dependency for <= 134 KB of scores;
//   2. composite 64-bit key = (descending-orderable score bits << 32) | position, so an ascending
//      sort is "score descending, index ascending" and all keys are distinct;
//   3. if k < N: 8-pass MSD radix SELECT of the k-th smallest key straight from global memory
//      (256-bin LDS histogram per pass), then compaction of the k keys <= threshold into LDS;
//   4. bitonic sort of next_pow2(k) keys in LDS (global scratch only when that exceeds 16 384
//      keys, e.g. the reference's "5scale" pyramid);
//   5. write out_score / out_index (optionally through an int64 payload gather).
#include "common.h"

namespace sdetr {

constexpr int kTopkThreads = 1024;
constexpr int kLdsKeys = 16384;  // 128 KiB of u64 keys in LDS

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

struct TopkArgs {
    const float *score;
    const uint8_t *mask;
    int fill_mode;
    const int64_t *payload;
    int B, N, k, npad;
    int64_t index_offset;
    float *out_score;
    int64_t *out_index;
    uint64_t *workspace;  // [B][npad] when npad > kLdsKeys
};

__device__ __forceinline__ uint64_t make_key(const TopkArgs &p, const float *srow, const uint8_t *mrow, int i,
                                             float fill)
{
    float s = srow[i];
    if (mrow && mrow[i]) s = fill;
    return ((uint64_t)desc_bits(s) << 32) | (uint32_t)i;
}

__global__ void __launch_bounds__(kTopkThreads) masked_topk_kernel(TopkArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem_raw);         // [256]
    uint32_t *misc = hist + 256;                                     // [8]
    float *red = reinterpret_cast<float *>(misc + 8);                // [16] wave partials
    uint64_t *lds_keys = reinterpret_cast<uint64_t *>(smem_raw + 2048);

    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const float *srow = p.score + (int64_t)b * p.N;
    const uint8_t *mrow = p.mask ? p.mask + (int64_t)b * p.N : nullptr;
    uint64_t *keys = (p.npad <= kLdsKeys) ? lds_keys : p.workspace + (int64_t)b * p.npad;

    // ---- 1. fill value = min over the whole [B,N] array (masked entries included) ----
    float fill = 0.f;
    if (p.fill_mode == 1) {
        float mn = INFINITY;
        const int64_t total = (int64_t)p.B * p.N;
        for (int64_t i = tid; i < total; i += kTopkThreads) mn = fminf(mn, p.score[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o, 64));
        if ((tid & 63) == 0) red[tid >> 6] = mn;
        __syncthreads();
        mn = red[0];
#pragma unroll
        for (int w = 1; w < kTopkThreads / 64; ++w) mn = fminf(mn, red[w]);
        fill = mn;
    }

    // ---- 2/3. threshold = k-th smallest composite key (radix select), then compaction ----
    if (p.k < p.N) {
        uint64_t prefix = 0;      // decided high bits
        uint32_t remaining = p.k; // rank (1-based) of the wanted key inside the current prefix class
        for (int shift = 56; shift >= 0; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const uint64_t hi_mask = (shift == 56) ? 0ull : (~0ull << (shift + 8));
            for (int i = tid; i < p.N; i += kTopkThreads) {
                const uint64_t key = make_key(p, srow, mrow, i, fill);
                if ((key & hi_mask) == prefix) atomicAdd(&hist[(key >> shift) & 0xff], 1u);
            }
            __syncthreads();
            if (tid < 64) {
                // wave 0: inclusive scan of the 256 bins (4 per lane) to find the crossing bin
                uint32_t c[4], local = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    c[u] = hist[tid * 4 + u];
                    local += c[u];
                }
                uint32_t incl = local;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t n = __shfl_up(incl, o, 64);
                    if (tid >= o) incl += n;
                }
                uint32_t run = incl - local;  // exclusive prefix of this lane's 4 bins
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (remaining > run && remaining <= run + c[u]) {
                        misc[0] = tid * 4 + u;
                        misc[1] = remaining - run;
                    }
                    run += c[u];
                }
            }
            __syncthreads();
            prefix |= (uint64_t)misc[0] << shift;
            remaining = misc[1];
            __syncthreads();
        }
        const uint64_t threshold = prefix;
        if (tid == 0) misc[2] = 0;
        __syncthreads();
        for (int i = tid; i < p.N; i += kTopkThreads) {
            const uint64_t key = make_key(p, srow, mrow, i, fill);
            if (key <= threshold) keys[atomicAdd(&misc[2], 1u)] = key;
        }
        for (int i = p.k + tid; i < p.npad; i += kTopkThreads) keys[i] = ~0ull;
    } else {
        for (int i = tid; i < p.N; i += kTopkThreads) keys[i] = make_key(p, srow, mrow, i, fill);
        for (int i = p.N + tid; i < p.npad; i += kTopkThreads) keys[i] = ~0ull;
    }
    __syncthreads();

    // ---- 4. bitonic sort, ascending ----
    const int half = p.npad >> 1;
    for (int size = 2; size <= p.npad; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < half; i += kTopkThreads) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool asc = (lo & size) == 0;
                const uint64_t a = keys[lo], c = keys[hi];
                if ((a > c) == asc) {
                    keys[lo] = c;
                    keys[hi] = a;
                }
            }
            __syncthreads();
        }
    }

    // ---- 5. output ----
    for (int i = tid; i < p.k; i += kTopkThreads) {
        const uint64_t key = keys[i];
        const uint32_t pos = (uint32_t)key;
        if (p.out_score) p.out_score[(int64_t)b * p.k + i] = undesc_bits((uint32_t)(key >> 32));
        p.out_index[(int64_t)b * p.k + i] =
            p.payload ? p.payload[(int64_t)b * p.N + pos] : (int64_t)pos + p.index_offset;
    }
}

static int next_pow2(int v)
{
    int n = 2;
    while (n < v) n <<= 1;
    return n;
}

}  // namespace sdetr

using namespace sdetr;

extern "C" size_t sdetr_topk_workspace_bytes(int B, int n, int k)
{
    if (B <= 0 || n <= 0 || k <= 0) return 0;
    const int npad = next_pow2(k < n ? k : n);
    return npad <= kLdsKeys ? 0 : (size_t)B * npad * sizeof(uint64_t);
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
    if (n >= (1 << 30)) return fail("masked_topk: row too long");
    TopkArgs a{};
    a.score = score; a.mask = mask; a.fill_mode = fill_mode; a.payload = payload;
    a.B = B; a.N = n; a.k = k; a.npad = next_pow2(k < n ? k : n);
    a.index_offset = index_offset; a.out_score = out_score; a.out_index = out_index;
    size_t lds = 2048;
    if (a.npad <= kLdsKeys) {
        lds += (size_t)a.npad * sizeof(uint64_t);
    } else {
        const size_t need = (size_t)B * a.npad * sizeof(uint64_t);
        if (!workspace || workspace_bytes < need)
            return fail("masked_topk: needs %zu bytes of workspace, got %zu", need, workspace_bytes);
        a.workspace = reinterpret_cast<uint64_t *>(workspace);
    }
    hipLaunchKernelGGL(masked_topk_kernel, dim3((unsigned)B), dim3(kTopkThreads), lds, stream, a);
    return check_launch("masked_topk");
}
