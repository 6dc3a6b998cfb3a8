// The dense self-attention over the top-k selected queries of an encoder layer (models/bricks/
// salience_transformer.py:366-376: gather tgt / pos, q = k = tgt + pos, v = tgt, nn.MultiheadAttention) up to the
// concatenated heads, in ONE launch: one workgroup per (image, head), one wave per 32 selected tokens.
//
// The framework path is a gather kernel, the [600,256]x[256,768] in-projection GEMM and a fused attention kernel
// (5 + 7 + 14 us for 2 x 300 tokens -- all launch / latency bound).  Here a wave
//   1. gathers its 32 tokens' rows straight from the layer's query / position buffers into MFMA operand fragments
//      (x_qk = tgt + pos, x_v = tgt: salience_transformer.py:371-374),
//   2. projects them with this head's 32 rows of Wq, Wk, Wv (three 16 KB tiles copied to LDS by LDS-DMA):
//        Q^T = Wq X_qk^T, K^T = Wk X_qk^T   (lane = token, registers = head channels)
//        V   = X_v Wv^T                      (lane = head channel, registers = tokens)
//      -- chosen so that every accumulator IS an MFMA operand of the next product after bias + bf16 rounding
//      (the contraction index order is the accumulator's row order in both operands, so it cancels):
//      K^T is the A operand of S^T = K Q^T for this wave's 32 keys, Q^T its B operand, V the A operand of
//      O^T = V^T P^T, and S^T's accumulator (lane = query, registers = keys) after exp() the B operand P^T;
//   3. publishes its K and V fragments in LDS, and after one barrier runs a flash-style loop over the key blocks
//      (online softmax in fp32; the two lanes that share a query combine max / sum with one cross-half shuffle),
//   4. stores O^T / l as bf16 into the concatenated-heads buffer [B, N, 256].
// 88 MFMAs per wave; N <= 320 (ten waves).  The out-projection + residual + norm + scatter that follow stay separate.
//
// Measured on MI355X (2 x 300 tokens, in-kernel s_memtime): 27 us -- 23k cycles for the prologue, 11.6k for the
// projections, 12k for the attention loop.  The prologue is the row gather: each of the 16 (image, head) workgroups
// pulls all 300 token rows (2 x 150 KB, 16-byte pieces at a 32-byte stride) through ONE CU's L1 at 64 B/clk, and the
// eight heads of an image repeat it.  That equals the three-launch framework path (gather on the whole chip 5 us +
// GEMM 7 + attention 14), so the encoder keeps that path by default (`fused_topk_attention`).
#include "common.h"

namespace sdetr {

constexpr int kMhaMaxTokens = 320;
constexpr int kMhaHeadDim = 32;

typedef __bf16 mh_bf16x8_t __attribute__((ext_vector_type(8)));
typedef float mh_f32x16_t __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) const char *mh_lds_cptr_t;

struct MhaArgs {
    const bf16_t *q;          // [B, c, 256], images q_batch_stride elements apart
    int64_t q_batch_stride;
    const bf16_t *pos;        // same row indexing
    int64_t pos_batch_stride;
    const int64_t *sel;       // [B, N]
    const char *pw;           // packed in_proj weight: 24 tiles (tile = part * heads + head; part 0/1/2 = q/k/v)
    const float *bias;        // [768]
    bf16_t *out;              // [B, N, 256]
    int N, heads;
    float scale;
};

__device__ __forceinline__ mh_f32x16_t mh_mfma(uint4 a, uint4 b, mh_f32x16_t c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mh_bf16x8_t, a), __builtin_bit_cast(mh_bf16x8_t, b),
                                                   c, 0, 0, 0);
}

__device__ __forceinline__ uint4 mh_lds_read16(mh_lds_cptr_t p)
{
    const u32x4_t v = *reinterpret_cast<const __attribute__((address_space(3))) u32x4_t *>(p);
    return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ uint32_t mh_add_bf16x2(uint32_t a, uint32_t b)
{
    return pack_bf16x2(bf16_lo(a) + bf16_lo(b), bf16_hi(a) + bf16_hi(b));
}

// registers 8m .. 8m+7 of an accumulator -> one packed bf16 operand fragment
__device__ __forceinline__ uint4 mh_pack_half(const mh_f32x16_t &c, int m)
{
    return make_uint4(pack_bf16x2(c[8 * m], c[8 * m + 1]), pack_bf16x2(c[8 * m + 2], c[8 * m + 3]),
                      pack_bf16x2(c[8 * m + 4], c[8 * m + 5]), pack_bf16x2(c[8 * m + 6], c[8 * m + 7]));
}

// accumulator row of register i for lane half h
__device__ __forceinline__ int mh_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES) mha_topk_kernel(MhaArgs p)
{
    // LDS: this head's Wq | Wk | Wv tiles (48 KB), then per key block two K fragments and two V fragments (1 KB each)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *wts = lds;
    char *kf = lds + 3 * 16384;
    char *vf = kf + WAVES * 2048;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = lane & 31, h = lane >> 5;
    const int b = blockIdx.x / p.heads, head = blockIdx.x - b * p.heads;
    const int nblk = (p.N + 31) / 32;   // == WAVES used; waves past it only help with the copies

    const int tok = min(wave * 32 + t, p.N - 1);          // (padding lanes repeat the last token; masked below)
    const bool active = wave < nblk;
    // Ten waves per block leave ~168 registers per wave: the token rows are fetched twice (x_qk = tgt + pos for Q / K,
    // then tgt alone for V) instead of keeping both fragment sets live next to three accumulators.
    uint4 xb[16];
    const bf16_t *qr = p.q, *pr = p.pos;
    uint4 pb[16];
    if (active) {
        const int64_t r = p.sel[(int64_t)b * p.N + tok];
        qr = p.q + (int64_t)b * p.q_batch_stride + r * 256 + 8 * h;
        pr = p.pos + (int64_t)b * p.pos_batch_stride + r * 256 + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) xb[ks] = *reinterpret_cast<const uint4 *>(qr + 16 * ks);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) pb[ks] = *reinterpret_cast<const uint4 *>(pr + 16 * ks);
    }
    // ---- weights -> LDS: 48 pieces of 1 KB over the waves ----
    {
        const uint32_t wts_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)wts;
        for (int piece = wave; piece < 48; piece += WAVES) {
            const int part = piece >> 4, f = piece & 15;
            const char *src = p.pw + ((int64_t)(part * p.heads + head) * 16 + f) * 1024 + lane * 16;
            const uint32_t d = __builtin_amdgcn_readfirstlane(wts_lds + piece * 1024);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(d) : "memory", "m0");
        }
    }

    if (active) {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
            xb[ks] = make_uint4(mh_add_bf16x2(xb[ks].x, pb[ks].x), mh_add_bf16x2(xb[ks].y, pb[ks].y),
                                mh_add_bf16x2(xb[ks].z, pb[ks].z), mh_add_bf16x2(xb[ks].w, pb[ks].w));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    uint4 qfrag[2];
    if (active) {
        const mh_lds_cptr_t wq = (mh_lds_cptr_t)wts + lane * 16, wk = wq + 16384, wv = wq + 32768;
        const float *bq = p.bias + head * kMhaHeadDim, *bk = bq + 256, *bv = bq + 512;
        mh_f32x16_t aq, ak, av;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            aq[i] = bq[mh_row(i, h)];     // Q^T / K^T: registers = head channels
            ak[i] = bk[mh_row(i, h)];
            av[i] = bv[t];                // V: lane = head channel
        }
        // One stream of 48 weight fragments (Wq, Wk alternating per k-step, then Wv) through an 8-deep register ring
        // (requested 8 MFMAs before use: with ~3 waves per SIMD an MFMA waiting on the LDS read issued just before
        // it is the whole run time).  A k-step's x_qk fragment is replaced by the same k-step of tgt alone as soon
        // as its second MFMA has been issued, so the V product starts without waiting for a reload.
        constexpr int R = 8;
        auto frag_at = [&](int f) { return f < 32 ? ((f & 1) ? wk : wq) + (f >> 1) * 1024 : wv + (f - 32) * 1024; };
        uint4 ring[R];
#pragma unroll
        for (int f = 0; f < R; ++f) ring[f] = mh_lds_read16(frag_at(f));
#pragma unroll
        for (int f = 0; f < 48; ++f) {
            if (f < 32) {
                if (f & 1) {
                    ak = mh_mfma(ring[f % R], xb[f >> 1], ak);
                    xb[f >> 1] = *reinterpret_cast<const uint4 *>(qr + 16 * (f >> 1));   // tgt alone, for V
                } else {
                    aq = mh_mfma(ring[f % R], xb[f >> 1], aq);
                }
            } else {
                av = mh_mfma(xb[f - 32], ring[f % R], av);
            }
            if (f + R < 48) ring[f % R] = mh_lds_read16(frag_at(f + R));
            __builtin_amdgcn_sched_barrier(0);
        }
        qfrag[0] = mh_pack_half(aq, 0);
        qfrag[1] = mh_pack_half(aq, 1);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            *reinterpret_cast<uint4 *>(kf + (wave * 2 + m) * 1024 + lane * 16) = mh_pack_half(ak, m);
            *reinterpret_cast<uint4 *>(vf + (wave * 2 + m) * 1024 + lane * 16) = mh_pack_half(av, m);
        }
    }
    __syncthreads();
    if (!active) return;

    // ---- flash loop: S^T = K Q^T (lane = my query, registers = keys), O^T += V^T P^T, five key blocks per round ----
    // (one rescale per 160 keys instead of per 32, and ten independent MFMAs back to back instead of dependent pairs
    // separated by the softmax arithmetic)
    mh_f32x16_t o;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = 0.f;
    float run_max = -INFINITY, run_sum = 0.f;
    constexpr int G = 5;
    for (int k0 = 0; k0 < nblk; k0 += G) {
        mh_f32x16_t s[G];
        uint4 kfr[G][2];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int kb = min(k0 + j, nblk - 1);
            const mh_lds_cptr_t kp = (mh_lds_cptr_t)kf + kb * 2048 + lane * 16;
            kfr[j][0] = mh_lds_read16(kp);
            kfr[j][1] = mh_lds_read16(kp + 1024);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < G; ++j) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s[j][i] = 0.f;
            s[j] = mh_mfma(kfr[j][0], qfrag[0], s[j]);
            s[j] = mh_mfma(kfr[j][1], qfrag[1], s[j]);
        }
#pragma unroll
        for (int j = 0; j < G; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = (k0 + j) * 32 + mh_row(i, h);
                s[j][i] = (k0 + j < nblk && key < p.N) ? s[j][i] * p.scale : -INFINITY;
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
            const int kb = min(k0 + j, nblk - 1);        // (a repeated block carries P = 0)
            const mh_lds_cptr_t vp = (mh_lds_cptr_t)vf + kb * 2048 + lane * 16;
            o = mh_mfma(mh_lds_read16(vp), mh_pack_half(s[j], 0), o);
            o = mh_mfma(mh_lds_read16(vp + 1024), mh_pack_half(s[j], 1), o);
        }
    }
    if (wave * 32 + t < p.N) {
        const float inv = 1.f / run_sum;
        bf16_t *orow = p.out + ((int64_t)b * p.N + wave * 32 + t) * 256 + head * kMhaHeadDim + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<uint2 *>(orow + 8 * g) =
                make_uint2(pack_bf16x2(o[4 * g] * inv, o[4 * g + 1] * inv), pack_bf16x2(o[4 * g + 2] * inv, o[4 * g + 3] * inv));
    }
}

}  // namespace sdetr

using namespace sdetr;

extern "C" int sdetr_topk_attention_heads_bf16(sdetr_stream_t stream, const void *query, int64_t query_batch_stride,
                                               const void *pos, int64_t pos_batch_stride, const int64_t *index,
                                               int batch_size, int num_select, int embed_dim, int num_heads,
                                               const void *packed_in_proj, const float *in_proj_bias, void *out)
{
    if (embed_dim != 256 || num_heads * kMhaHeadDim != 256)
        return fail("topk_attention: built for embed_dim 256 and 8 heads of 32 (got %d, %d)", embed_dim, num_heads);
    if (batch_size < 0 || num_select < 0 || num_select > kMhaMaxTokens)
        return fail("topk_attention: at most %d selected tokens (got %d)", kMhaMaxTokens, num_select);
    if ((int64_t)batch_size * num_select == 0) return 0;
    if (!query || !pos || !index || !packed_in_proj || !in_proj_bias || !out) return fail("topk_attention: null pointer");
    if ((query_batch_stride % 8) || (pos_batch_stride % 8)) return fail("topk_attention: strides must be multiples of 8");
    MhaArgs a{};
    a.q = (const bf16_t *)query; a.q_batch_stride = query_batch_stride; a.pos = (const bf16_t *)pos;
    a.pos_batch_stride = pos_batch_stride; a.sel = index; a.pw = (const char *)packed_in_proj; a.bias = in_proj_bias;
    a.out = (bf16_t *)out; a.N = num_select; a.heads = num_heads; a.scale = 0.17677669529663687f;   // 1/sqrt(32)
    constexpr int W = kMhaMaxTokens / 32;
    const size_t lds = 3 * 16384 + 2 * (size_t)W * 2048;
    static DeviceOnce lds_once1;
    allow_dynamic_lds(mha_topk_kernel<W>, lds_once1, (int)lds);
    hipLaunchKernelGGL(mha_topk_kernel<W>, dim3((unsigned)(batch_size * num_heads)), dim3(64 * W), lds,
                       static_cast<hipStream_t>(stream), a);
    return check_launch("topk_attention");
}
