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
    bf16_t *qk;            // q rows [B, Npad, 256], then the K fragments [B, 8, Npad/16, 64 lanes, 8]
    bf16_t *vt;            // V^T fragments [B, 8, Npad/32, 2, 64 lanes, 8]
    int B, N, Npad;
};
// Fragment-major K and V^T: the attention kernel's operand fragments are stored exactly as its lanes hold them, so
// each of its loads is one contiguous kilobyte per wave (row-major slabs made every K load touch 16 rows and every
// V^T load 64 separate 8-byte pieces: the address unit, not the latency, was 40 % of that kernel's time).
//   K  (A operand of S^T = K Q^T, 16 keys x 32 channels per tile): lane = 16 (ch / 8) + key % 16, slot = ch % 8
//   V^T (A operand of O^T += V^T P^T, 16 channels x 32 keys per block and channel half c): lane = 16 g + ch % 16 where
//       the lane group g holds keys {4g..4g+3} (slots 0-3) and {16+4g..16+4g+3} (slots 4-7) of the block
__device__ __forceinline__ int64_t tk_k_index(int b, int head, int key, int ch, int Npad)
{
    return ((((int64_t)b * kTkHeads + head) * (Npad / 16) + key / 16) * 64 + 16 * (ch / 8) + key % 16) * 8 + ch % 8;
}
__device__ __forceinline__ int64_t tk_vt_index(int b, int head, int key, int ch, int Npad)
{
    const int kk = key % 32, hi = kk / 16, g = (kk % 16) / 4, pos = 4 * hi + kk % 4;
    return (((((int64_t)b * kTkHeads + head) * (Npad / 32) + key / 32) * 2 + ch / 16) * 64 + 16 * g + ch % 16) * 8 + pos;
}

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
        const int head = ftile & 7;                           // tiles 0-7: q of head 0-7, tiles 8-15: k
        bf16_t *kbase = p.qk + (int64_t)p.B * p.Npad * 256;
#pragma unroll
        for (int g = 0; g < 4; ++g) {                         // my 4 consecutive channels 8g + 4h + 0..3 of the head
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = valid ? acc[4 * g + r] + bias[4 * g + r] : 0.f;
            bf16_t *out = ftile < 8 ? p.qk + ((int64_t)b * p.Npad + i) * 256 + head * 32 + 8 * g + 4 * h
                                    : kbase + tk_k_index(b, head, i, 8 * g + 4 * h, p.Npad);
            *reinterpret_cast<uint2 *>(out) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        }
    } else {
        const tk_f32x16_t acc = inproj_tile<false>(wr, xr, pr);
        const int head = ftile - 16;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = tk_row(r, h);
            const float v = valid ? acc[r] + bias[r] : 0.f;
            p.vt[tk_vt_index(b, head, i, ch, p.Npad)] = (bf16_t)(pack_bf16x2(v, 0.f) & 0xffffu);
        }
    }
}

struct TkOutArgs {
    const bf16_t *qk;      // q rows [B, Npad, 256], then the K fragments (see tk_k_index)
    const bf16_t *vt;      // V^T fragments (see tk_vt_index)
    const int64_t *sel;    // [B, N]
    bf16_t *query;         // [B, rows, 256]: residual rows are read from it, results written back to it
    int64_t q_bs;
    const bf16_t *wo;      // out_proj.weight [256, 256]
    const bf16_t *bo;      // out_proj.bias [256]
    const bf16_t *gamma, *beta;   // pre_norm
    float eps, scale;
    int B, N, Npad;
};

typedef float tk_f32x4_t __attribute__((ext_vector_type(4)));
// v_mfma_f32_16x16x32_bf16: A lane l = row l & 15, k = 8 (l >> 4) .. +7; B lane l = column l & 15, same k;
// C lane l = column l & 15, rows 4 (l >> 4) + 0..3
__device__ __forceinline__ tk_f32x4_t tk_mfma16(uint4 a, uint4 b, tk_f32x4_t c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tk_bf16x8_t, a), __builtin_bit_cast(tk_bf16x8_t, b),
                                                   c, 0, 0, 0);
}

constexpr int kTkQ = 16;            // queries per workgroup
constexpr int kTkORow = 528;        // bytes per query row of the heads' outputs in LDS (512 + 16: bank spread)
constexpr int kTkMaxKeyTiles = 24;  // 16-key tiles held in registers at once: up to 384 selected rows

// One 8-wave workgroup per (image, 16 queries), wave = head.  The head dimension (32) is ONE k-step of the 16x16x32
// MFMA, so S^T = K Q^T is one instruction per 16 keys and all of a query's scores (<= 384 keys: 96 registers) stay in
// registers: a plain two-pass softmax (max, exp2 with the 1/sqrt(32) scale folded in, sum), no running rescale.  The
// scores of two neighbouring key tiles ARE the B operand of O^T += V^T P^T after bf16 rounding (lane group g holds keys
// {4g..4g+3, 16+4g..16+4g+3} of a 32-key block; the V^T fragments are loaded in that key order, two 8-byte pieces).
// 40 workgroups for 2 x 300 rows (the 32-query version: 20, with four times the per-SIMD softmax arithmetic).
// A load whose result is discarded: brings the line into this XCD's L2. The compiler does not know the write to
// `sink` is still pending when the statement ends, so the caller keeps that register reserved (tk_touch_done) until
// the loads must have landed.
__device__ __forceinline__ void tk_touch(const void *ptr, uint32_t &sink)
{
    asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(ptr) : "memory");
}
__device__ __forceinline__ void tk_touch_done(uint32_t &sink) { asm volatile("" : "+v"(sink)); }

template <int KT>   // key tiles (of 16) = Npad / 16, compile time: the score array must live in registers
__global__ void __launch_bounds__(512) topk_attn_out_kernel(TkOutArgs p)
{
    __shared__ __attribute__((aligned(16))) char o_lds[kTkQ * kTkORow];
    __shared__ float part[2][8][kTkQ];
    const int lane = threadIdx.x & 63, head = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave = head
    const int t = lane & 15, g = lane >> 4;
    const int tiles = (p.N + kTkQ - 1) / kTkQ;
    const int tile = blockIdx.x % tiles, b = blockIdx.x / tiles;
    const int qi = tile * kTkQ + t;                 // < Npad (rows past N are zero rows of the slab)
    const bool valid = qi < p.N;
    const bf16_t *kfb = p.qk + (int64_t)p.B * p.Npad * 256 + (((int64_t)b * kTkHeads + head) * KT) * 512 + lane * 8;
    const bf16_t *vfb = p.vt + (((int64_t)b * kTkHeads + head) * (KT / 2)) * 1024 + lane * 8;

    // ---- everything this wave reads from global memory, issued up front ----
    // oldest load in flight: the residual row's index (loads return in order, so its consumer waits for it alone)
    const int64_t row = p.sel[(int64_t)b * p.N + min(qi, p.N - 1)];
    uint32_t sink = 0;
    // warm the out_proj rows this wave will want after the softmax (their registers are not free until then; untouched
    // they cost a second exposed trip to memory in the middle of the kernel)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) tk_touch(p.wo + (int64_t)(head * 32 + 16 * c + t) * kTkE + 32 * j + 8 * g, sink);
    const uint4 qfrag = *reinterpret_cast<const uint4 *>(p.qk + ((int64_t)b * p.Npad + qi) * 256 + head * kTkHd + 8 * g);
    uint4 kfr[KT];
#pragma unroll
    for (int j = 0; j < KT; ++j) kfr[j] = *reinterpret_cast<const uint4 *>(kfb + j * 512);   // one contiguous KB per wave
    // V^T fragments (A operand of the second product): per 32-key block and 16-channel half, keys {4g..4g+3} and
    // {16+4g..16+4g+3} of channel t (+16).  Requested together with K: one round trip for both (the scores take over
    // the K fragments' registers tile by tile)
    uint4 vfr[KT / 2][2];
#pragma unroll
    for (int m = 0; m < KT / 2; ++m)
#pragma unroll
        for (int c = 0; c < 2; ++c) vfr[m][c] = *reinterpret_cast<const uint4 *>(vfb + (m * 2 + c) * 512);
    bf16_t *xrow = p.query + (int64_t)b * p.q_bs + row * kTkE + head * 32 + 4 * g;   // my features: 32 head + 16 c + 4 g + r
    uint2 res_v[2], bo_v[2], gamma_v[2], beta_v[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        res_v[c] = *reinterpret_cast<const uint2 *>(xrow + 16 * c);
        bo_v[c] = *reinterpret_cast<const uint2 *>(p.bo + head * 32 + 16 * c + 4 * g);
        gamma_v[c] = *reinterpret_cast<const uint2 *>(p.gamma + head * 32 + 16 * c + 4 * g);
        beta_v[c] = *reinterpret_cast<const uint2 *>(p.beta + head * 32 + 16 * c + 4 * g);
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- scores: S^T[key][query], one MFMA per 16 keys ----
    tk_f32x4_t s[KT];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        s[j] = tk_f32x4_t{0.f, 0.f, 0.f, 0.f};
        s[j] = tk_mfma16(kfr[j], qfrag, s[j]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < KT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (j * 16 + 4 * g + r >= p.N) s[j][r] = -INFINITY;    // padded keys (only the last tiles: folds for full ones)
            mx = fmaxf(mx, s[j][r]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float c2 = p.scale * 1.4426950408889634f, shift = -mx * c2;
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < KT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s[j][r] = __builtin_amdgcn_exp2f(fmaf(s[j][r], c2, shift));
            sum += s[j][r];
        }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    // ---- O^T[channel][query] += V^T P^T, 32 keys per MFMA, two channel halves ----
    tk_f32x4_t o[2] = {tk_f32x4_t{0.f, 0.f, 0.f, 0.f}, tk_f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int m = 0; m < KT / 2; ++m) {
        const uint4 pf = make_uint4(pack_bf16x2(s[2 * m][0], s[2 * m][1]), pack_bf16x2(s[2 * m][2], s[2 * m][3]),
                                    pack_bf16x2(s[2 * m + 1][0], s[2 * m + 1][1]), pack_bf16x2(s[2 * m + 1][2], s[2 * m + 1][3]));
        o[0] = tk_mfma16(vfr[m][0], pf, o[0]);
        o[1] = tk_mfma16(vfr[m][1], pf, o[1]);
    }
    // out_proj fragments of my 32 features (two 16-feature tiles x 8 k-steps of 32): issued once the score and V^T
    // registers are free (together they would not fit in 256), their round trip overlaps the LDS exchange
    tk_touch_done(sink);   // every load issued before the scores has returned by now (the score MFMAs waited for them)
    uint4 wfrag[2][8];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            wfrag[c][j] = *reinterpret_cast<const uint4 *>(p.wo + (int64_t)(head * 32 + 16 * c + t) * kTkE + 32 * j + 8 * g);
    __builtin_amdgcn_sched_barrier(0);
    // ---- the heads meet in LDS: O[query][32 head + channel] bf16 (my channels: 16 c + 4 g + r) ----
    {
        const float inv = 1.f / sum;
        char *orow = o_lds + t * kTkORow + (head * kTkHd + 4 * g) * 2;
#pragma unroll
        for (int c = 0; c < 2; ++c)
            *reinterpret_cast<uint2 *>(orow + 32 * c) =
                make_uint2(pack_bf16x2(o[c][0] * inv, o[c][1] * inv), pack_bf16x2(o[c][2] * inv, o[c][3] * inv));
    }
    __syncthreads();
    // ---- out_proj: Z^T[feature][query] = Wo O^T, my 32 features as two 16-feature tiles ----
    tk_f32x4_t z[2] = {tk_f32x4_t{0.f, 0.f, 0.f, 0.f}, tk_f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint4 of = *reinterpret_cast<const uint4 *>(o_lds + t * kTkORow + (32 * j + 8 * g) * 2);
        z[0] = tk_mfma16(wfrag[0][j], of, z[0]);
        z[1] = tk_mfma16(wfrag[1][j], of, z[1]);
    }
    // + bias + residual; LayerNorm over the 256 features of a query (8 here, 32 per wave, 8 waves)
    float tot = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float rv[4] = {bf16_lo(res_v[c].x), bf16_hi(res_v[c].x), bf16_lo(res_v[c].y), bf16_hi(res_v[c].y)};
        const float bv[4] = {bf16_lo(bo_v[c].x), bf16_hi(bo_v[c].x), bf16_lo(bo_v[c].y), bf16_hi(bo_v[c].y)};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            z[c][r] += bv[r] + rv[r];
            tot += z[c][r];
        }
    }
    tot += __shfl_xor(tot, 16);
    tot += __shfl_xor(tot, 32);
    if (g == 0) part[0][head][t] = tot;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) mean += part[0][w][t];
    mean *= (1.f / kTkE);
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = z[c][r] - mean;
            sq += d * d;
        }
    sq += __shfl_xor(sq, 16);
    sq += __shfl_xor(sq, 32);
    if (g == 0) part[1][head][t] = sq;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) var += part[1][w][t];
    const float rstd = rsqrtf(var * (1.f / kTkE) + p.eps);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float gm[4] = {bf16_lo(gamma_v[c].x), bf16_hi(gamma_v[c].x), bf16_lo(gamma_v[c].y), bf16_hi(gamma_v[c].y)};
        const float bt[4] = {bf16_lo(beta_v[c].x), bf16_hi(beta_v[c].x), bf16_lo(beta_v[c].y), bf16_hi(beta_v[c].y)};
        float y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = (z[c][r] - mean) * rstd * gm[r] + bt[r];
        if (valid) *reinterpret_cast<uint2 *>(xrow + 16 * c) = make_uint2(pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]));
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
