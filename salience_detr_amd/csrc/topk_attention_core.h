// Shared definitions of the top-k attention kernels and the body of the attention + out-projection + LayerNorm kernel
// (see topk_attention.hip), as a device function so that another launch can carry it next to other work
// (fused_head_value.hip).
#pragma once
#include "common.h"

namespace sdetr {

typedef __bf16 tk_bf16x8_t __attribute__((ext_vector_type(8)));
typedef float tk_f32x16_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ tk_f32x16_t tk_mfma(uint4 a, uint4 b, tk_f32x16_t c)
{
    return mfma_act_32x32x16(a, b, c);
}
// accumulator row of register i for lane half h (v_mfma_f32_32x32x16: C[row][col = lane & 31])
__device__ __forceinline__ int tk_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }
__device__ __forceinline__ uint4 tk_pack_half(const tk_f32x16_t &c, int m)
{
    return make_uint4(pack_act2(c[8 * m], c[8 * m + 1]), pack_act2(c[8 * m + 2], c[8 * m + 3]),
                      pack_act2(c[8 * m + 4], c[8 * m + 5]), pack_act2(c[8 * m + 6], c[8 * m + 7]));
}
constexpr int kTkMaxSel = 384;   // selected rows per image the two-launch attention is built for
__device__ __forceinline__ float tk_bf16(bf16_t v) { return act_lo((uint32_t)v); }   // (one activation element -> f32)

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
    int32_t *hint;         // optional [B, hint_bs]: hint[b][sel[b][i]] = i + 1 (see TLArgs::skip_hint)
    int64_t hint_bs;
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
// `bias4` / `bv`: the lane's four 8-byte bias pieces (stride 8 elements) are requested inside the LAST round's MFMA chain,
// once some of its fragments are dead (they arrive under the remaining MFMAs) -- requested up front they were eight more
// live registers under 24 fragments, and the 1024-thread selection + in-projection launch (topk.hip: 128 registers per
// wave) spilled.
// `R` = k-steps (of 16) per round of operand requests: 8 = two rounds of 16 / 24 fragments (the stand-alone launch); 4 =
// four rounds through TWO register sets, round r + 2 requested when round r's MFMAs are issued (the selection +
// in-projection launch: 125 registers, no spill; one exposed trip to memory instead of two -- measured the same 7-8 us
// as the two- and three-round forms: the in-projection behind a selection is not bound by how its operands are requested).
template <bool QK, int R = 8>
__device__ __forceinline__ tk_f32x16_t inproj_tile(const bf16_t *wr, const bf16_t *xr, const bf16_t *pr, const bf16_t *bias4,
                                                   uint2 (&bv)[4])
{
    tk_f32x16_t acc = {};   // (a constant zero: the first MFMA takes it as its inline C operand, no registers until then)
    if constexpr (R == 4) {
        // four rounds of four k-steps through TWO register sets: round r + 2 is requested when round r's MFMAs are issued,
        // so a wave waits for one trip to memory (and what of the second the first round's MFMAs do not cover), with 24
        // fragments in flight at most -- and only 12 while its MFMAs need the accumulator
        uint4 a[2][4], bx[2][4], bp[2][4];
        auto request = [&](int rd, int s) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[s][j] = *reinterpret_cast<const uint4 *>(wr + (rd * 4 + j) * 16);
                bx[s][j] = *reinterpret_cast<const uint4 *>(xr + (rd * 4 + j) * 16);
                if (QK) bp[s][j] = *reinterpret_cast<const uint4 *>(pr + (rd * 4 + j) * 16);
            }
        };
        request(0, 0);
        request(1, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rd = 0; rd < 4; ++rd) {
            const int s = rd & 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc = tk_mfma(a[s][j], bx[s][j], acc);
                if (QK) acc = tk_mfma(a[s][j], bp[s][j], acc);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (rd + 2 < 4) request(rd + 2, s);
            if (rd == 2) {
#pragma unroll
                for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const uint2 *>(bias4 + 8 * g);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        return acc;
    }
    constexpr int ROUNDS = (16 + R - 1) / R;
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        uint4 a[R], bx[R], bp[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (rd * R + j >= 16) continue;
            a[j] = *reinterpret_cast<const uint4 *>(wr + (rd * R + j) * 16);
            bx[j] = *reinterpret_cast<const uint4 *>(xr + (rd * R + j) * 16);
            if (QK) bp[j] = *reinterpret_cast<const uint4 *>(pr + (rd * R + j) * 16);
        }
        __builtin_amdgcn_sched_barrier(0);   // every load of the round issued before its first MFMA
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (rd * R + j >= 16) continue;
            acc = tk_mfma(a[j], bx[j], acc);
            if (QK) acc = tk_mfma(a[j], bp[j], acc);   // W (x + pos) = W x + W pos
            if (rd == ROUNDS - 1 && j == 1) {          // (two k-steps' fragments are dead: room for the bias pieces)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const uint2 *>(bias4 + 8 * g);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    return acc;
}

// One wave's tile of the in-projection: 32 selected rows (tile `tile` of image b) x 32 output features (`ftile` of 24);
// `row` = the lane's selected row, sel[b][min(32 tile + (lane & 31), N - 1)] (from memory or from the selection's own LDS
// list).  `ftile` must be PROVABLY wave-uniform: the two tile variants are chosen by a branch on it, and an MFMA under a
// branch the compiler takes for divergent is merely exec-masked -- which matrix instructions ignore.
template <int R = 8>
__device__ __forceinline__ void inproj_wave_body(const TkInArgs &p, const int b, const int tile, const int ftile, const int lane,
                                                 const int64_t row)
{
    const int t = lane & 31, h = lane >> 5;
    const bool qk_tile = ftile < 16;
    const int i = tile * 32 + t;                          // my token (as B-operand / accumulator column)
    const bool valid = i < p.N;
    if (p.hint && ftile == 0 && h == 0 && valid) p.hint[(int64_t)b * p.hint_bs + row] = i + 1;
    // (wave-uniform bases + 32-bit lane offsets: the loads take the scalar-base form, one address register each)
    const uint32_t ro = (uint32_t)row * (uint32_t)kTkE + 8u * (uint32_t)h;
    const bf16_t *xr = p.query + (int64_t)b * p.q_bs + ro;
    const bf16_t *pr = p.pos + (int64_t)b * p.p_bs + ro;
    const bf16_t *wr = p.w + (int64_t)ftile * 32 * kTkE + (uint32_t)(t * kTkE + 8 * h);
    // bias of my 16 features (groups of 4 consecutive ones: accumulator rows 8g + 4h + 0..3) as four 8-byte pieces --
    // per-element loads behind the `valid` test were serialised by the compiler, one round trip each
    const bf16_t *bias4 = p.bias + ftile * 32 + 4 * h;
    uint2 bv[4];
    float bias[16];
    auto unpack_bias = [&]() {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bias[4 * g] = act_lo(bv[g].x); bias[4 * g + 1] = act_hi(bv[g].x);
            bias[4 * g + 2] = act_lo(bv[g].y); bias[4 * g + 3] = act_hi(bv[g].y);
        }
    };
    // rows of the accumulator = features, column = my token; padded tokens are written as zeros (finite keys / values)
    if (qk_tile) {
        const tk_f32x16_t acc = inproj_tile<true, R>(wr, xr, pr, bias4, bv);
        unpack_bias();
        const int head = ftile & 7;                           // tiles 0-7: q of head 0-7, tiles 8-15: k
        bf16_t *kbase = p.qk + (int64_t)p.B * p.Npad * 256;
#pragma unroll
        for (int g = 0; g < 4; ++g) {                         // my 4 consecutive channels 8g + 4h + 0..3 of the head
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = valid ? acc[4 * g + r] + bias[4 * g + r] : 0.f;
            bf16_t *out = ftile < 8 ? p.qk + ((int64_t)b * p.Npad + i) * 256 + head * 32 + 8 * g + 4 * h
                                    : kbase + tk_k_index(b, head, i, 8 * g + 4 * h, p.Npad);
            *reinterpret_cast<uint2 *>(out) = make_uint2(pack_act2(v[0], v[1]), pack_act2(v[2], v[3]));
        }
    } else {
        const tk_f32x16_t acc = inproj_tile<false, R>(wr, xr, pr, bias4, bv);
        unpack_bias();
        const int head = ftile - 16;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = tk_row(r, h);
            const float v = valid ? acc[r] + bias[r] : 0.f;
            p.vt[tk_vt_index(b, head, i, ch, p.Npad)] = (bf16_t)(pack_act2(v, 0.f) & 0xffffu);
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
    // optional: the deformable attention's offset | weight projection of the UPDATED rows, written into the
    // head-major slab the MSDA kernel reads ([B, 8, fx_rows, 48]; a launch that projected every row from the
    // not-yet-updated queries runs next to this kernel -- fused_head_value.hip -- and these rows overwrite its result)
    const bf16_t *fx_w;    // [384, 256] projection weight, rows in head-major order (48 per head), or NULL
    const float *fx_b;     // [384]
    const bf16_t *fx_pos;  // [B, n0, 256] position rows, images fx_p_bs elements apart
    int64_t fx_p_bs;
    bf16_t *fx_slab;       // [B, 8, fx_rows, 48]
    int fx_rows;
    int fx_by_selection;   // 0: row index = the query's row in the layer (writes the MSDA slab itself); 1: its position
                           // in the selection (a side buffer [B, 8, N, 48])
    // optional (round 6): `wo` / `fx_w` once more in MFMA-fragment order -- [head][16-feature tile][k-step of 32][lane][8]:
    // a wave's fragment load is then one contiguous KB (8 cache lines) where the row-major weight gives 16 pieces of 64
    // bytes from 16 rows per instruction.  Cycle stamps: of a workgroup's 44 000 cycles 14 500 went into the out_proj
    // phase and 12 300 into the projection of the updated rows -- their 16 + 24 fragment loads per lane.
    const bf16_t *wo_frag;    // [8][2][8][64][8] or NULL
    const bf16_t *fx_w_frag;  // [8][3][8][64][8] or NULL
};

typedef float tk_f32x4_t __attribute__((ext_vector_type(4)));
// v_mfma_f32_16x16x32_bf16: A lane l = row l & 15, k = 8 (l >> 4) .. +7; B lane l = column l & 15, same k;
// C lane l = column l & 15, rows 4 (l >> 4) + 0..3
__device__ __forceinline__ tk_f32x4_t tk_mfma16(uint4 a, uint4 b, tk_f32x4_t c)
{
    return mfma_act_16x16x32(a, b, c);
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

// (`block` = the kernel's own block index, or the position inside a launch that also carries other work.)
template <int KT>   // key tiles (of 16) = Npad / 16, compile time: the score array must live in registers
__device__ __forceinline__ void topk_attn_out_body(const TkOutArgs &p, int block)
{
    __shared__ __attribute__((aligned(16))) char o_lds[kTkQ * kTkORow];
    __shared__ float part[2][8][kTkQ];
    const int lane = threadIdx.x & 63, head = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave = head
    const int t = lane & 15, g = lane >> 4;
#ifdef TK_STAMPS   // (benchmark builds: cycle stamps per phase, printed by workgroup 0)
    long long tk_t[10];
    int tk_n = 0;
#define TK_STAMP() tk_t[tk_n++] = clock64()
#else
#define TK_STAMP()
#endif
    TK_STAMP();
    const int tiles = (p.N + kTkQ - 1) / kTkQ;
    const int tile = block % tiles, b = block / tiles;
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
    if (p.wo_frag) {   // the wave's 16 KB of out_proj fragments, 24 KB of projection fragments: a line per lane and touch
        tk_touch(p.wo_frag + head * 8192 + lane * 64, sink);
        tk_touch(p.wo_frag + head * 8192 + 4096 + lane * 64, sink);
        if (p.fx_w_frag) {
#pragma unroll
            for (int i = 0; i < 3; ++i) tk_touch(p.fx_w_frag + head * 12288 + i * 4096 + lane * 64, sink);
        }
    } else {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) tk_touch(p.wo + (int64_t)(head * 32 + 16 * c + t) * kTkE + 32 * j + 8 * g, sink);
    }
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
    uint2 pos_v[2] = {make_uint2(0u, 0u), make_uint2(0u, 0u)};
    if (p.fx_w) {
        const bf16_t *prow = p.fx_pos + (int64_t)b * p.fx_p_bs + row * kTkE + head * 32 + 4 * g;
        pos_v[0] = *reinterpret_cast<const uint2 *>(prow);
        pos_v[1] = *reinterpret_cast<const uint2 *>(prow + 16);
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
    TK_STAMP();   // 1: operands back, score products issued
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
        const uint4 pf = make_uint4(pack_act2(s[2 * m][0], s[2 * m][1]), pack_act2(s[2 * m][2], s[2 * m][3]),
                                    pack_act2(s[2 * m + 1][0], s[2 * m + 1][1]), pack_act2(s[2 * m + 1][2], s[2 * m + 1][3]));
        o[0] = tk_mfma16(vfr[m][0], pf, o[0]);
        o[1] = tk_mfma16(vfr[m][1], pf, o[1]);
    }
    // out_proj fragments of my 32 features (two 16-feature tiles x 8 k-steps of 32): issued once the score and V^T
    // registers are free (together they would not fit in 256), their round trip overlaps the LDS exchange
    TK_STAMP();   // 2: softmax, P V
    tk_touch_done(sink);   // every load issued before the scores has returned by now (the score MFMAs waited for them)
    uint4 wfrag[2][8];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            wfrag[c][j] = p.wo_frag
                              ? *reinterpret_cast<const uint4 *>(p.wo_frag + ((head * 2 + c) * 8 + j) * 512 + lane * 8)
                              : *reinterpret_cast<const uint4 *>(p.wo + (int64_t)(head * 32 + 16 * c + t) * kTkE + 32 * j + 8 * g);
    __builtin_amdgcn_sched_barrier(0);
    // ---- the heads meet in LDS: O[query][32 head + channel] bf16 (my channels: 16 c + 4 g + r) ----
    {
        const float inv = 1.f / sum;
        char *orow = o_lds + t * kTkORow + (head * kTkHd + 4 * g) * 2;
#pragma unroll
        for (int c = 0; c < 2; ++c)
            *reinterpret_cast<uint2 *>(orow + 32 * c) =
                make_uint2(pack_act2(o[c][0] * inv, o[c][1] * inv), pack_act2(o[c][2] * inv, o[c][3] * inv));
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
    TK_STAMP();   // 3: heads exchanged through LDS, out_proj
    // + bias + residual; LayerNorm over the 256 features of a query (8 here, 32 per wave, 8 waves)
    float tot = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float rv[4] = {act_lo(res_v[c].x), act_hi(res_v[c].x), act_lo(res_v[c].y), act_hi(res_v[c].y)};
        const float bv[4] = {act_lo(bo_v[c].x), act_hi(bo_v[c].x), act_lo(bo_v[c].y), act_hi(bo_v[c].y)};
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
        const float gm[4] = {act_lo(gamma_v[c].x), act_hi(gamma_v[c].x), act_lo(gamma_v[c].y), act_hi(gamma_v[c].y)};
        const float bt[4] = {act_lo(beta_v[c].x), act_hi(beta_v[c].x), act_lo(beta_v[c].y), act_hi(beta_v[c].y)};
        float y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = (z[c][r] - mean) * rstd * gm[r] + bt[r];
        const uint2 yb = make_uint2(pack_act2(y[0], y[1]), pack_act2(y[2], y[3]));
        if (valid) *reinterpret_cast<uint2 *>(xrow + 16 * c) = yb;
        if (p.fx_w) {
            // the projection's input row: bf16(updated row + position row), as the token-resident kernel forms it
            const uint2 xp = make_uint2(pack_act2(act_lo(yb.x) + act_lo(pos_v[c].x), act_hi(yb.x) + act_hi(pos_v[c].x)),
                                        pack_act2(act_lo(yb.y) + act_lo(pos_v[c].y), act_hi(yb.y) + act_hi(pos_v[c].y)));
            *reinterpret_cast<uint2 *>(o_lds + t * kTkORow + (head * kTkHd + 16 * c + 4 * g) * 2) = xp;
        }
    }
    TK_STAMP();   // 4: residual, LayerNorm, row stores
    if (!p.fx_w) return;
    // ---- sampling_offsets | attention_weights of the 16 updated rows: wave = head, 48 features = three 16-feature
    // tiles, P^T[feature][query] = W (x + pos)^T + b; rows of the head-major slab [B, 8, rows, 48] ----
    uint4 pw[3][8];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            pw[c][j] = p.fx_w_frag
                           ? *reinterpret_cast<const uint4 *>(p.fx_w_frag + ((head * 3 + c) * 8 + j) * 512 + lane * 8)
                           : *reinterpret_cast<const uint4 *>(p.fx_w + (int64_t)(head * 48 + 16 * c + t) * kTkE + 32 * j + 8 * g);
    tk_f32x4_t pa[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float4 bv = *reinterpret_cast<const float4 *>(p.fx_b + head * 48 + 16 * c + 4 * g);
        pa[c] = tk_f32x4_t{bv.x, bv.y, bv.z, bv.w};
    }
    __syncthreads();   // every wave has written its 32 channels of the 16 rows
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint4 xf = *reinterpret_cast<const uint4 *>(o_lds + t * kTkORow + (32 * j + 8 * g) * 2);
#pragma unroll
        for (int c = 0; c < 3; ++c) pa[c] = tk_mfma16(pw[c][j], xf, pa[c]);
    }
    if (valid) {
        bf16_t *srow = p.fx_slab + (((int64_t)b * kTkHeads + head) * p.fx_rows + (p.fx_by_selection ? qi : row)) * 48 + 4 * g;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            *reinterpret_cast<uint2 *>(srow + 16 * c) = make_uint2(pack_act2(pa[c][0], pa[c][1]), pack_act2(pa[c][2], pa[c][3]));
    }
#ifdef TK_STAMPS
    TK_STAMP();   // 5: projection of the updated rows
    if (threadIdx.x == 0 && (block == 0 || block == 20))
        printf("attn blk=%d cycles: operands %lld | softmax+PV %lld | exchange+out_proj %lld | LN+stores %lld | projection %lld | total %lld\n", block,
               tk_t[1] - tk_t[0], tk_t[2] - tk_t[1], tk_t[3] - tk_t[2], tk_t[4] - tk_t[3], tk_t[5] - tk_t[4], tk_t[5] - tk_t[0]);
#endif
}

}  // namespace sdetr
