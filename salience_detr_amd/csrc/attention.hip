// Dense multi-head self-attention over a few hundred tokens with 32-channel heads, bf16, no mask: the 300 selected
// queries of an encoder layer (models/bricks/salience_transformer.py:371-376) and the 900-1100 object queries of a
// decoder layer (:565-570), after the in-projection:  out[b, i, 32 h + :] = softmax(Q_h K_h^T / sqrt(32)) V_h.
//
// The framework's flash kernel takes 14 us for 2 x 300 and 33 us for 2 x 900 tokens -- latency, not work (0.2 GFLOP).
// Here a workgroup owns (image, head, block of up to 256 queries):
//   1. all waves build the head's K and V^T MFMA fragments for ALL keys in LDS (4 KB per 32 keys): K rows are A-operand
//      fragments as they lie in memory; V^T fragments are gathered column-wise (2-byte loads) in the k order of the
//      score accumulator's rows, so that
//   2. each wave runs a flash loop for its 32 queries: S^T = K Q^T (lane = query, registers = keys), online softmax in
//      fp32 (the two lanes that share a query combine max / sum with one cross-half shuffle), and P^T -- S^T's own
//      accumulator after exp() and bf16 rounding -- is the B operand of O^T += V^T P^T.  No LDS round trip for Q or P.
//   3. O^T / l is stored as bf16 into the concatenated-heads buffer [B, N, 256] (the input of out_proj).
#include "common.h"

namespace sdetr {

constexpr int kAttHeadDim = 32;
constexpr int kAttMaxBlocks = 36;        // 36 x 32 = 1152 keys: 144 KB of fragments
constexpr int kAttWaves = 8;             // 4 query groups of 32 x 2 key halves: 128 queries per workgroup

typedef __bf16 at_bf16x8_t __attribute__((ext_vector_type(8)));
typedef float at_f32x16_t __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) const char *at_lds_cptr_t;

struct AttArgs {
    const bf16_t *q, *k, *v;            // element [b][i][32 h + d] at base + b * batch_stride + i * row_stride + 32 h + d
    int64_t q_batch, q_row, k_batch, k_row, v_batch, v_row;
    bf16_t *out;                        // [B, N, heads * 32]
    int N, heads, qchunks;
    float scale;
};

__device__ __forceinline__ at_f32x16_t at_mfma(uint4 a, uint4 b, at_f32x16_t c)
{
    return mfma_act_32x32x16(a, b, c);
}
__device__ __forceinline__ uint4 at_lds_read16(at_lds_cptr_t p)
{
    const u32x4_t v = *reinterpret_cast<const __attribute__((address_space(3))) u32x4_t *>(p);
    return make_uint4(v.x, v.y, v.z, v.w);
}
// registers 8m .. 8m+7 of an accumulator -> one packed bf16 operand fragment
__device__ __forceinline__ uint4 at_pack_half(const at_f32x16_t &c, int m)
{
    return make_uint4(pack_act2(c[8 * m], c[8 * m + 1]), pack_act2(c[8 * m + 2], c[8 * m + 3]),
                      pack_act2(c[8 * m + 4], c[8 * m + 5]), pack_act2(c[8 * m + 6], c[8 * m + 7]));
}
// accumulator row of register i for lane half h
__device__ __forceinline__ int at_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// Round 5: (a) the fragment build requests the rows of ALL of a wave's key blocks before it stores the first one (a block
// per load round trip had made the build 4-5 round trips long); (b) the eight waves are 4 query groups x 2 key halves: a
// wave runs the flash loop over half the key blocks, the halves' (max, sum, O) states meet in LDS (exact online-softmax
// merge, fixed order: low keys then high keys).  2 x 8 x 900 tokens: 28 -> see profiles.
constexpr int kAttQueryWaves = 4, kAttKeySplits = kAttWaves / kAttQueryWaves;
constexpr int kAttBuildPerWave = (kAttMaxBlocks + kAttWaves - 1) / kAttWaves;      // 5

__global__ void __launch_bounds__(64 * kAttWaves) attention_heads_kernel(AttArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int nblk = (p.N + 31) / 32;
    char *kf = lds;                      // [nblk][2][1 KB]  K fragments (k-steps 0, 1 of the 32 head channels)
    char *vf = lds + nblk * 2048;        // [nblk][2][1 KB]  V^T fragments (k-blocks 0, 1 of the 32 keys)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = lane & 31, h = lane >> 5;
    const int qw = wave % kAttQueryWaves, ks = wave / kAttQueryWaves;
    const int chunk = blockIdx.x % p.qchunks, bh = blockIdx.x / p.qchunks;
    const int b = bh / p.heads, head = bh - b * p.heads;
    const bf16_t *kbase = p.k + (int64_t)b * p.k_batch + head * kAttHeadDim;
    const bf16_t *vbase = p.v + (int64_t)b * p.v_batch + head * kAttHeadDim;

    // my query rows first (their latency hides behind the fragment build)
    const int q0 = chunk * (32 * kAttQueryWaves) + qw * 32;
    const int qi = q0 + t;
    const bool has_query = q0 < p.N;                                        // (wave-uniform)
    uint4 qfrag[2];
    {
        const bf16_t *qr = p.q + (int64_t)b * p.q_batch + (int64_t)min(qi, p.N - 1) * p.q_row + head * kAttHeadDim + 8 * h;
        qfrag[0] = *reinterpret_cast<const uint4 *>(qr);
        qfrag[1] = *reinterpret_cast<const uint4 *>(qr + 16);
    }
    // ---- K and V^T fragments of every key block, built by all waves: all loads, then all stores ----
    {
        uint4 k0[kAttBuildPerWave], k1[kAttBuildPerWave];
        uint32_t vv[kAttBuildPerWave][2][4];
#pragma unroll
        for (int i = 0; i < kAttBuildPerWave; ++i) {
            const int kb = wave + kAttWaves * i;
            if (kb < nblk) {                                                    // (wave-uniform)
                const int key = min(kb * 32 + t, p.N - 1);                      // (rows past N are masked in the loop)
                const bf16_t *kr = kbase + (int64_t)key * p.k_row + 8 * h;
                k0[i] = *reinterpret_cast<const uint4 *>(kr);
                k1[i] = *reinterpret_cast<const uint4 *>(kr + 16);
                // V^T: lane (t = head channel, h) gathers channel t of the 8 keys that slot (m, h) of P^T's operand stands for
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2) {
                        const int r0 = min(kb * 32 + at_row(8 * m + 2 * s2, h), p.N - 1);
                        const int r1 = min(kb * 32 + at_row(8 * m + 2 * s2 + 1, h), p.N - 1);
                        const uint32_t lo = vbase[(int64_t)r0 * p.v_row + t], hi = vbase[(int64_t)r1 * p.v_row + t];
                        vv[i][m][s2] = lo | (hi << 16);
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < kAttBuildPerWave; ++i) {
            const int kb = wave + kAttWaves * i;
            if (kb < nblk) {
                *reinterpret_cast<uint4 *>(kf + (kb * 2 + 0) * 1024 + lane * 16) = k0[i];
                *reinterpret_cast<uint4 *>(kf + (kb * 2 + 1) * 1024 + lane * 16) = k1[i];
                *reinterpret_cast<uint4 *>(vf + (kb * 2 + 0) * 1024 + lane * 16) = make_uint4(vv[i][0][0], vv[i][0][1], vv[i][0][2], vv[i][0][3]);
                *reinterpret_cast<uint4 *>(vf + (kb * 2 + 1) * 1024 + lane * 16) = make_uint4(vv[i][1][0], vv[i][1][1], vv[i][1][2], vv[i][1][3]);
            }
        }
    }
    __syncthreads();

    // ---- flash loop over my half of the key blocks: S^T = K Q^T (lane = my query, registers = keys), O^T += V^T P^T,
    //      five key blocks per round ----
    const int per_split = (nblk + kAttKeySplits - 1) / kAttKeySplits;
    const int kb_begin = ks * per_split, kb_end = min(nblk, kb_begin + per_split);
    at_f32x16_t o;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = 0.f;
    float run_max = -INFINITY, run_sum = 0.f;
    constexpr int G = 5;
    if (has_query) {
        for (int k0 = kb_begin; k0 < kb_end; k0 += G) {
            at_f32x16_t s[G];
            uint4 kfr[G][2];
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const int kb = min(k0 + j, kb_end - 1);
                const at_lds_cptr_t kp = (at_lds_cptr_t)kf + kb * 2048 + lane * 16;
                kfr[j][0] = at_lds_read16(kp);
                kfr[j][1] = at_lds_read16(kp + 1024);
            }
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < G; ++j) {
#pragma unroll
                for (int i = 0; i < 16; ++i) s[j][i] = 0.f;
                s[j] = at_mfma(kfr[j][0], qfrag[0], s[j]);
                s[j] = at_mfma(kfr[j][1], qfrag[1], s[j]);
            }
#pragma unroll
            for (int j = 0; j < G; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = (k0 + j) * 32 + at_row(i, h);
                    s[j][i] = (k0 + j < kb_end && key < p.N) ? s[j][i] * p.scale : -INFINITY;
                    mx = fmaxf(mx, s[j][i]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 32));            // the other half of my query's keys
            const float new_max = fmaxf(run_max, mx);      // (finite: every round holds at least one real key)
            const float corr = __expf(run_max - new_max);
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < G; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    s[j][i] = __expf(s[j][i] - new_max);
                    sum += s[j][i];
                }
            sum += __shfl_xor(sum, 32);
            run_sum = run_sum * corr + sum;
            run_max = new_max;
#pragma unroll
            for (int i = 0; i < 16; ++i) o[i] *= corr;
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const int kb = min(k0 + j, kb_end - 1);        // (a repeated block carries P = 0)
                const at_lds_cptr_t vp = (at_lds_cptr_t)vf + kb * 2048 + lane * 16;
                o = at_mfma(at_lds_read16(vp), at_pack_half(s[j], 0), o);
                o = at_mfma(at_lds_read16(vp + 1024), at_pack_half(s[j], 1), o);
            }
        }
    }
    // ---- the key halves meet: the high half's state through LDS (over the fragments, which nobody reads any more) ----
    __syncthreads();
    float *mg = reinterpret_cast<float *>(lds) + (qw * 64 + lane) * 20;     // 18 floats per lane, 80-byte pitch
    if (ks == 1 && has_query) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4 *>(mg + 4 * g) = make_float4(o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]);
        mg[16] = run_max;
        mg[17] = run_sum;
    }
    __syncthreads();
    if (ks != 0 || !has_query) return;
    {
        const float m1 = mg[16], s1 = mg[17];
        // (a half without blocks -- N <= 32 * per_split -- left max = -inf, sum = 0: its factor is 0)
        const float new_max = fmaxf(run_max, m1);
        const float c0 = __expf(run_max - new_max), c1 = m1 == -INFINITY ? 0.f : __expf(m1 - new_max);
        run_sum = run_sum * c0 + s1 * c1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = *reinterpret_cast<const float4 *>(mg + 4 * g);
            o[4 * g] = o[4 * g] * c0 + v.x * c1; o[4 * g + 1] = o[4 * g + 1] * c0 + v.y * c1;
            o[4 * g + 2] = o[4 * g + 2] * c0 + v.z * c1; o[4 * g + 3] = o[4 * g + 3] * c0 + v.w * c1;
        }
    }
    if (qi < p.N) {
        const float inv = 1.f / run_sum;
        bf16_t *orow = p.out + ((int64_t)b * p.N + qi) * (p.heads * kAttHeadDim) + head * kAttHeadDim + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<uint2 *>(orow + 8 * g) =
                make_uint2(pack_act2(o[4 * g] * inv, o[4 * g + 1] * inv), pack_act2(o[4 * g + 2] * inv, o[4 * g + 3] * inv));
    }
}

}  // namespace sdetr

using namespace sdetr;

extern "C" int sdetr_attention_heads_bf16(sdetr_stream_t stream, const void *q, int64_t q_batch_stride, int64_t q_row_stride,
                                          const void *k, int64_t k_batch_stride, int64_t k_row_stride, const void *v,
                                          int64_t v_batch_stride, int64_t v_row_stride, int batch_size, int num_tokens,
                                          int num_heads, int head_dim, float scale, void *out)
{
    if (head_dim != kAttHeadDim) return fail("attention_heads: built for 32-channel heads (got %d)", head_dim);
    if (batch_size < 0 || num_tokens < 0 || num_heads <= 0) return fail("attention_heads: bad sizes");
    if (num_tokens > kAttMaxBlocks * 32) return fail("attention_heads: at most %d tokens (got %d)", kAttMaxBlocks * 32, num_tokens);
    if ((int64_t)batch_size * num_tokens == 0) return 0;
    if (!q || !k || !v || !out) return fail("attention_heads: null pointer");
    if ((q_row_stride & 7) || (k_row_stride & 7) || (q_batch_stride & 7) || (k_batch_stride & 7))
        return fail("attention_heads: q / k strides must be multiples of 8 elements (16-byte rows pieces)");
    AttArgs a{};
    a.q = (const bf16_t *)q; a.k = (const bf16_t *)k; a.v = (const bf16_t *)v;
    a.q_batch = q_batch_stride; a.q_row = q_row_stride; a.k_batch = k_batch_stride; a.k_row = k_row_stride;
    a.v_batch = v_batch_stride; a.v_row = v_row_stride; a.out = (bf16_t *)out; a.N = num_tokens; a.heads = num_heads;
    a.qchunks = (num_tokens + 32 * kAttQueryWaves - 1) / (32 * kAttQueryWaves); a.scale = scale;
    const int nblk = (num_tokens + 31) / 32;
    const size_t lds = std::max((size_t)nblk * 4096, (size_t)kAttQueryWaves * 64 * 80);   // fragments, then the merge states
    static DeviceOnce lds_once1;
    allow_dynamic_lds(attention_heads_kernel, lds_once1, kAttMaxBlocks * 4096);
    hipLaunchKernelGGL(attention_heads_kernel, dim3((unsigned)(batch_size * num_heads * a.qchunks)), dim3(64 * kAttWaves), lds,
                       static_cast<hipStream_t>(stream), a);
    return check_launch("attention_heads");
}
