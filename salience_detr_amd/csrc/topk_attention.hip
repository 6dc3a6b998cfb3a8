// The encoder layer's dense self-attention over its top-k rows (models/bricks/salience_transformer.py:366-379:
// gather select_tgt / select_pos, nn.MultiheadAttention(q = k = x + pos, v = x), residual + pre_norm, scatter back),
// bf16, 8 heads x 32 channels, embed dim 256 -- in TWO launches, no library GEMM:
//
//  1. topk_inproj_kernel: gather + position add + in-projection.  One wave per (32 selected rows, 32 output
//     features); both operands straight from global memory in MFMA fragment order (a lane's 8 consecutive k of its
//     row are 16 contiguous bytes of the row), Y^T = W X^T so that a lane owns one token; the position add of the
//     q / k features is a second MFMA into the same accumulator (W x + W pos: exact in fp32, no bf16 re-rounding of
//     x + pos).  q and k go to a row-major [B, Npad, 512] slab, v to a TRANSPOSED [B, 8, 32, Npad] slab: exactly the
//     A-operand order of the O^T += V^T P^T product below (the strided 2-byte gathers of round 1's kernel are gone).
//  2. topk_attn_out_kernel: one 8-wave workgroup per (image, 32 queries), wave = head.  Flash loop as in
//     attention.hip (S^T = K Q^T, online softmax, P^T = the score accumulator itself after exp and bf16 rounding),
//     the heads' outputs meet in LDS, every wave then computes 32 features of out_proj for the 32 queries, adds the
//     bias and the residual row, LayerNorm (two-pass, statistics exchanged through LDS) and writes the rows back to
//     their places in the layer's query buffer.
//
// Replaces select_stack + library GEMM + attention_heads + library GEMM + layernorm/scatter (five launches).
#include "common.h"

namespace sdetr {

typedef __bf16 tk_bf16x8_t __attribute__((ext_vector_type(8)));
typedef float tk_f32x16_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ tk_f32x16_t tk_mfma(uint4 a, uint4 b, tk_f32x16_t c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(tk_bf16x8_t, a), __builtin_bit_cast(tk_bf16x8_t, b),
                                                   c, 0, 0, 0);
}
// accumulator row of register i for lane half h (v_mfma_f32_32x32x16: C[row][col = lane & 31])
__device__ __forceinline__ int tk_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }
__device__ __forceinline__ uint4 tk_pack_half(const tk_f32x16_t &c, int m)
{
    return make_uint4(pack_bf16x2(c[8 * m], c[8 * m + 1]), pack_bf16x2(c[8 * m + 2], c[8 * m + 3]),
                      pack_bf16x2(c[8 * m + 4], c[8 * m + 5]), pack_bf16x2(c[8 * m + 6], c[8 * m + 7]));
}
__device__ __forceinline__ float tk_bf16(bf16_t v) { return __uint_as_float((uint32_t)v << 16); }

constexpr int kTkE = 256, kTkHeads = 8, kTkHd = 32;

struct TkInArgs {
    const bf16_t *query;   // [B, rows, 256] the layer's rows; images q_bs elements apart
    int64_t q_bs;
    const bf16_t *pos;     // [B, n0, 256] position rows in the same (sorted) order; images p_bs elements apart
    int64_t p_bs;
    const int64_t *sel;    // [B, N] selected row numbers
    const bf16_t *w;       // in_proj_weight [768, 256]
    const bf16_t *bias;    // in_proj_bias [768]
    bf16_t *qk;            // [B, Npad, 512]
    bf16_t *vt;            // [B, 8, 32, Npad]
    int B, N, Npad;
};

// one 32-feature x 32-token tile; QK = the tile holds q or k features (the position rows are added).  Straight-line
// code per variant: with a branch inside, hipcc sinks the operand loads into the MFMA sequence (two in flight)
template <bool QK>
__device__ __forceinline__ tk_f32x16_t inproj_tile(const bf16_t *wr, const bf16_t *xr, const bf16_t *pr)
{
    tk_f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {               // 8 k-steps of 16 at a time: 16 / 24 fragments in flight
        uint4 a[8], bx[8], bp[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a[j] = *reinterpret_cast<const uint4 *>(wr + (half * 8 + j) * 16);
            bx[j] = *reinterpret_cast<const uint4 *>(xr + (half * 8 + j) * 16);
            if (QK) bp[j] = *reinterpret_cast<const uint4 *>(pr + (half * 8 + j) * 16);
        }
        __builtin_amdgcn_sched_barrier(0);   // every load of the half issued before its first MFMA
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc = tk_mfma(a[j], bx[j], acc);
            if (QK) acc = tk_mfma(a[j], bp[j], acc);   // W (x + pos) = W x + W pos
        }
    }
    return acc;
}

__global__ void __launch_bounds__(256) topk_inproj_kernel(TkInArgs p)
{
    // the wave number must be PROVABLY uniform: the two tile variants are chosen by a branch on it, and an MFMA under
    // a branch the compiler takes for divergent is merely exec-masked -- which matrix instructions ignore
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t = lane & 31, h = lane >> 5;
    const int tiles = p.Npad / 32;
    const int fgroup = blockIdx.x % 6, rest = blockIdx.x / 6;
    const int tile = rest % tiles, b = rest / tiles;
    const int ftile = fgroup * 4 + wave;                 // 24 tiles of 32 output features
    const bool qk_tile = ftile < 16;
    const int i = tile * 32 + t;                          // my token (as B-operand / accumulator column)
    const bool valid = i < p.N;
    const int64_t row = p.sel[(int64_t)b * p.N + min(i, p.N - 1)];
    const bf16_t *xr = p.query + (int64_t)b * p.q_bs + row * kTkE + 8 * h;
    const bf16_t *pr = p.pos + (int64_t)b * p.p_bs + row * kTkE + 8 * h;
    const bf16_t *wr = p.w + (int64_t)(ftile * 32 + t) * kTkE + 8 * h;

    // bias of my 16 features (groups of 4 consecutive ones: accumulator rows 8g + 4h + 0..3), loaded up front as four
    // 8-byte pieces -- per-element loads behind the `valid` test were serialised by the compiler, one round trip each
    float bias[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint2 bv = *reinterpret_cast<const uint2 *>(p.bias + ftile * 32 + 8 * g + 4 * h);
        bias[4 * g] = bf16_lo(bv.x); bias[4 * g + 1] = bf16_hi(bv.x);
        bias[4 * g + 2] = bf16_lo(bv.y); bias[4 * g + 3] = bf16_hi(bv.y);
    }
    // rows of the accumulator = features, column = my token; padded tokens are written as zeros (finite keys / values)
    if (qk_tile) {
        const tk_f32x16_t acc = inproj_tile<true>(wr, xr, pr);
        bf16_t *out = p.qk + ((int64_t)b * p.Npad + i) * 512 + ftile * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = valid ? acc[4 * g + r] + bias[4 * g + r] : 0.f;
            *reinterpret_cast<uint2 *>(out + 8 * g) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        }
    } else {
        const tk_f32x16_t acc = inproj_tile<false>(wr, xr, pr);
        const int head = ftile - 16;
        bf16_t *out = p.vt + ((int64_t)b * kTkHeads + head) * kTkHd * p.Npad + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = tk_row(r, h);
            const float v = valid ? acc[r] + bias[r] : 0.f;
            out[(int64_t)ch * p.Npad] = (bf16_t)(pack_bf16x2(v, 0.f) & 0xffffu);
        }
    }
}

struct TkOutArgs {
    const bf16_t *qk;      // [B, Npad, 512]
    const bf16_t *vt;      // [B, 8, 32, Npad]
    const int64_t *sel;    // [B, N]
    bf16_t *query;         // [B, rows, 256]: residual rows are read from it, results written back to it
    int64_t q_bs;
    const bf16_t *wo;      // out_proj.weight [256, 256]
    const bf16_t *bo;      // out_proj.bias [256]
    const bf16_t *gamma, *beta;   // pre_norm
    float eps, scale;
    int B, N, Npad;
};

constexpr int kTkORow = 528;   // bytes per query row of the heads' outputs in LDS (512 + 16: bank spread)

__global__ void __launch_bounds__(512) topk_attn_out_kernel(TkOutArgs p)
{
    __shared__ __attribute__((aligned(16))) char o_lds[32 * kTkORow];
    __shared__ float part[2][8][32];
    const int lane = threadIdx.x & 63, head = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave = head
    const int t = lane & 31, h = lane >> 5;
    const int tiles = p.Npad / 32;
    const int tile = blockIdx.x % tiles, b = blockIdx.x / tiles;
    const int qi = tile * 32 + t;
    const bool valid = qi < p.N;
    const int nblk = (p.N + 31) / 32;
    const bf16_t *qkb = p.qk + (int64_t)b * p.Npad * 512;
    const bf16_t *vtb = p.vt + ((int64_t)b * kTkHeads + head) * kTkHd * p.Npad + (int64_t)t * p.Npad;   // my channel's row

    // out_proj bias, LayerNorm gain / bias of my 16 features (8g + 4h + 0..3 of the wave's 32), four 8-byte pieces each
    uint2 bo_v[4], gamma_v[4], beta_v[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        bo_v[g] = *reinterpret_cast<const uint2 *>(p.bo + head * 32 + 8 * g + 4 * h);
        gamma_v[g] = *reinterpret_cast<const uint2 *>(p.gamma + head * 32 + 8 * g + 4 * h);
        beta_v[g] = *reinterpret_cast<const uint2 *>(p.beta + head * 32 + 8 * g + 4 * h);
    }
    const int64_t row = p.sel[(int64_t)b * p.N + min(qi, p.N - 1)];
    bf16_t *xrow = p.query + (int64_t)b * p.q_bs + row * kTkE + head * 32 + 4 * h;
    uint2 res_v[4];   // the residual: my 16 features of my query's row
#pragma unroll
    for (int g = 0; g < 4; ++g) res_v[g] = *reinterpret_cast<const uint2 *>(xrow + 8 * g);
    uint4 qfrag[2];
    {
        const bf16_t *qr = qkb + (int64_t)qi * 512 + head * kTkHd + 8 * h;   // (qi < Npad: padded rows are zeros)
        qfrag[0] = *reinterpret_cast<const uint4 *>(qr);
        qfrag[1] = *reinterpret_cast<const uint4 *>(qr + 16);
    }
    // ---- flash loop over the key blocks, five per round ----
    tk_f32x16_t o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    float run_max = -INFINITY, run_sum = 0.f;
    constexpr int G = 5;
    for (int k0 = 0; k0 < nblk; k0 += G) {
        uint4 kfr[G][2], vfr[G][2];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int kb = min(k0 + j, nblk - 1);
            const bf16_t *kr = qkb + (int64_t)(kb * 32 + t) * 512 + 256 + head * kTkHd + 8 * h;
            kfr[j][0] = *reinterpret_cast<const uint4 *>(kr);
            kfr[j][1] = *reinterpret_cast<const uint4 *>(kr + 16);
#pragma unroll
            for (int m = 0; m < 2; ++m) {   // keys {0-3, 8-11} + 16 m + 4 h of the block: the rows of registers 8m..8m+7
                const uint2 lo = *reinterpret_cast<const uint2 *>(vtb + kb * 32 + 16 * m + 4 * h);
                const uint2 hi = *reinterpret_cast<const uint2 *>(vtb + kb * 32 + 16 * m + 8 + 4 * h);
                vfr[j][m] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // the round's 30 loads are issued as one batch
        tk_f32x16_t s[G];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < G; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[j][r] = 0.f;
            s[j] = tk_mfma(kfr[j][0], qfrag[0], s[j]);
            s[j] = tk_mfma(kfr[j][1], qfrag[1], s[j]);
        }
        // only the last key block of the last round can hold padded keys / be a repeat (wave-uniform test); the raw
        // scores are compared, the 1/sqrt(32) scale goes into the exponent: exp2((s - max) * scale * log2 e)
#pragma unroll
        for (int j = 0; j < G; ++j) {
            if (k0 + j >= nblk - 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = (k0 + j) * 32 + tk_row(r, h);
                    if (k0 + j >= nblk || key >= p.N) s[j][r] = -INFINITY;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[j][r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));            // the other half of my query's keys
        const float new_max = fmaxf(run_max, mx);      // (finite: every round holds at least one real key)
        const float c2 = p.scale * 1.4426950408889634f;
        const float corr = __builtin_amdgcn_exp2f((run_max - new_max) * c2);
        const float shift = -new_max * c2;
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < G; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[j][r] = __builtin_amdgcn_exp2f(fmaf(s[j][r], c2, shift));
                sum += s[j][r];
            }
        sum += __shfl_xor(sum, 32);
        run_sum = run_sum * corr + sum;
        run_max = new_max;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= corr;
#pragma unroll
        for (int j = 0; j < G; ++j) {                    // (a repeated block carries P = 0)
            o = tk_mfma(vfr[j][0], tk_pack_half(s[j], 0), o);
            o = tk_mfma(vfr[j][1], tk_pack_half(s[j], 1), o);
        }
    }
    // out_proj fragments of my 32 features (A operand: lane = feature, 8 consecutive k): issued here so that their
    // round trip overlaps the LDS exchange below (held across the flash loop they cost 64 registers and spills)
    uint4 wfrag[16];
    {
        const bf16_t *wr = p.wo + (int64_t)(head * 32 + t) * kTkE + 8 * h;
#pragma unroll
        for (int j = 0; j < 16; ++j) wfrag[j] = *reinterpret_cast<const uint4 *>(wr + j * 16);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- the heads meet in LDS: O[query][32 head + channel] bf16 ----
    {
        const float inv = 1.f / run_sum;
        char *orow = o_lds + t * kTkORow + (head * kTkHd + 4 * h) * 2;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<uint2 *>(orow + 16 * g) =
                make_uint2(pack_bf16x2(o[4 * g] * inv, o[4 * g + 1] * inv), pack_bf16x2(o[4 * g + 2] * inv, o[4 * g + 3] * inv));
    }
    __syncthreads();
    // ---- out_proj: Z^T[feature][query] = Wo O^T, my 32 features ----
    tk_f32x16_t z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint4 of = *reinterpret_cast<const uint4 *>(o_lds + t * kTkORow + (j * 16 + 8 * h) * 2);
        z = tk_mfma(wfrag[j], of, z);
    }
    // + bias + residual row; LayerNorm over the 256 features of a query (16 here, 32 per wave, 8 waves)
    float sum = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint2 res = res_v[g];
        const float rv[4] = {bf16_lo(res.x), bf16_hi(res.x), bf16_lo(res.y), bf16_hi(res.y)};
        const float bv[4] = {bf16_lo(bo_v[g].x), bf16_hi(bo_v[g].x), bf16_lo(bo_v[g].y), bf16_hi(bo_v[g].y)};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            z[4 * g + r] += bv[r] + rv[r];
            sum += z[4 * g + r];
        }
    }
    sum += __shfl_xor(sum, 32);
    if (h == 0) part[0][head][t] = sum;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) mean += part[0][w][t];
    mean *= (1.f / kTkE);
    float sq = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float d = z[r] - mean;
        sq += d * d;
    }
    sq += __shfl_xor(sq, 32);
    if (h == 0) part[1][head][t] = sq;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) var += part[1][w][t];
    const float rstd = rsqrtf(var * (1.f / kTkE) + p.eps);
    if (valid) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float y[4];
            const float gm[4] = {bf16_lo(gamma_v[g].x), bf16_hi(gamma_v[g].x), bf16_lo(gamma_v[g].y), bf16_hi(gamma_v[g].y)};
            const float bt[4] = {bf16_lo(beta_v[g].x), bf16_hi(beta_v[g].x), bf16_lo(beta_v[g].y), bf16_hi(beta_v[g].y)};
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = (z[4 * g + r] - mean) * rstd * gm[r] + bt[r];
            *reinterpret_cast<uint2 *>(xrow + 8 * g) = make_uint2(pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]));
        }
    }
}

}  // namespace sdetr

using namespace sdetr;

extern "C" int64_t sdetr_topk_attention_workspace_bytes(int batch_size, int num_selected)
{
    const int64_t npad = (num_selected + 31) / 32 * 32;
    return (int64_t)batch_size * npad * (512 + 256) * 2;
}

extern "C" int sdetr_topk_attention_bf16(sdetr_stream_t stream, void *query, int64_t query_batch_stride, const void *pos,
                                         int64_t pos_batch_stride, const int64_t *selected, int batch_size,
                                         int num_rows, int num_selected, const void *in_proj_weight,
                                         const void *in_proj_bias, const void *out_proj_weight, const void *out_proj_bias,
                                         const void *norm_weight, const void *norm_bias, float norm_eps, int embed_dim,
                                         int num_heads, void *workspace, int64_t workspace_bytes)
{
    if (embed_dim != kTkE || num_heads != kTkHeads)
        return fail("topk_attention: built for embed_dim 256 with 8 heads of 32 channels (got %d / %d)", embed_dim, num_heads);
    if (batch_size < 0 || num_rows < 0 || num_selected < 0 || num_selected > num_rows)
        return fail("topk_attention: bad sizes (rows %d, selected %d)", num_rows, num_selected);
    if (num_selected > 1152) return fail("topk_attention: at most 1152 selected rows (got %d)", num_selected);
    if ((int64_t)batch_size * num_selected == 0) return 0;
    if (!query || !pos || !selected || !in_proj_weight || !in_proj_bias || !out_proj_weight || !out_proj_bias ||
        !norm_weight || !norm_bias || !workspace)
        return fail("topk_attention: null pointer");
    if (workspace_bytes < sdetr_topk_attention_workspace_bytes(batch_size, num_selected))
        return fail("topk_attention: workspace too small");
    if (query_batch_stride < (int64_t)num_rows * kTkE || pos_batch_stride < (int64_t)num_rows * kTkE ||
        (query_batch_stride & 7) || (pos_batch_stride & 7))
        return fail("topk_attention: bad batch strides");
    const int npad = (num_selected + 31) / 32 * 32;
    TkInArgs a{};
    a.query = (const bf16_t *)query; a.q_bs = query_batch_stride; a.pos = (const bf16_t *)pos; a.p_bs = pos_batch_stride;
    a.sel = selected; a.w = (const bf16_t *)in_proj_weight; a.bias = (const bf16_t *)in_proj_bias;
    a.qk = (bf16_t *)workspace; a.vt = a.qk + (int64_t)batch_size * npad * 512;
    a.B = batch_size; a.N = num_selected; a.Npad = npad;
    hipLaunchKernelGGL(topk_inproj_kernel, dim3((unsigned)(batch_size * (npad / 32) * 6)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a);
    if (int e = check_launch("topk_inproj")) return e;
    TkOutArgs o{};
    o.qk = a.qk; o.vt = a.vt; o.sel = selected; o.query = (bf16_t *)query; o.q_bs = query_batch_stride;
    o.wo = (const bf16_t *)out_proj_weight; o.bo = (const bf16_t *)out_proj_bias;
    o.gamma = (const bf16_t *)norm_weight; o.beta = (const bf16_t *)norm_bias; o.eps = norm_eps;
    o.scale = 0.17677669529663687f;   // 1 / sqrt(32)
    o.B = batch_size; o.N = num_selected; o.Npad = npad;
    hipLaunchKernelGGL(topk_attn_out_kernel, dim3((unsigned)(batch_size * (npad / 32))), dim3(512), 0,
                       static_cast<hipStream_t>(stream), o);
    return check_launch("topk_attn_out");
}
