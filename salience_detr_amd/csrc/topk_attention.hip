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

#include "topk_attention_core.h"

namespace sdetr {

__global__ void __launch_bounds__(256) topk_inproj_kernel(TkInArgs p)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles = p.Npad / 32;
    const int fgroup = blockIdx.x % 6, rest = blockIdx.x / 6;
    const int tile = rest % tiles, b = rest / tiles;
    const int i = tile * 32 + (lane & 31);
    inproj_wave_body(p, b, tile, fgroup * 4 + wave, lane, p.sel[(int64_t)b * p.N + min(i, p.N - 1)]);
}

template <int KT>
__global__ void __launch_bounds__(512) topk_attn_out_kernel(TkOutArgs p)
{
    topk_attn_out_body<KT>(p, (int)blockIdx.x);
}

}  // namespace sdetr

using namespace sdetr;

// (internal, for fused_head_value.hip: the in-projection launch with a filled TkInArgs)
extern "C" int sdetr_topk_inproj_launch(sdetr_stream_t stream, const void *tk_in_args)
{
    const TkInArgs &a = *static_cast<const TkInArgs *>(tk_in_args);
    hipLaunchKernelGGL(topk_inproj_kernel, dim3((unsigned)(a.B * (a.Npad / 32) * 6)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a);
    return check_launch("topk_inproj");
}

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
    if (num_selected > kTkMaxKeyTiles * 16)
        return fail("topk_attention: at most %d selected rows (got %d)", kTkMaxKeyTiles * 16, num_selected);
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
    const unsigned grid = (unsigned)(batch_size * ((num_selected + kTkQ - 1) / kTkQ));
    hipStream_t hs = static_cast<hipStream_t>(stream);
    switch (npad / 32) {   // key tiles come in pairs (32-key blocks of the second product)
#define SDETR_TK_CASE(BLK) case BLK: hipLaunchKernelGGL((topk_attn_out_kernel<2 * BLK>), dim3(grid), dim3(512), 0, hs, o); break;
        SDETR_TK_CASE(1) SDETR_TK_CASE(2) SDETR_TK_CASE(3) SDETR_TK_CASE(4) SDETR_TK_CASE(5) SDETR_TK_CASE(6)
        SDETR_TK_CASE(7) SDETR_TK_CASE(8) SDETR_TK_CASE(9) SDETR_TK_CASE(10) SDETR_TK_CASE(11) SDETR_TK_CASE(12)
#undef SDETR_TK_CASE
        default: return fail("topk_attention: unsupported key count");
    }
    return check_launch("topk_attn_out");
}
