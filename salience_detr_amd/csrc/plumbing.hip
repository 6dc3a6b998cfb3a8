// HBM-bound plumbing kernels of the hot path that replace long chains of tiny framework launches.
//
//  * pyramid_flatten_kernel: F0 of the path (reference models/bricks/base_transformer.py:22-33
//    flatten_multi_level / get_lvl_pos_embed and :74-112 the validity geometry of
//    gen_encoder_output_proposals).  One launch per level turns the NCHW feature and position maps into
//    the token-major tensors the transformer consumes -- feat_flatten, pos + level_embed, the masked
//    sum (feat + pos) * keep that feeds enc_output, the flattened padding mask, and (optionally) bf16
//    copies of the first two for the bf16 encoder -- with a 32x32 LDS transpose so both the NCHW reads
//    and the token-major writes are coalesced.  The reference does this with ~110 launches
//    (cat / transpose / arange / compare chains); the keep flag (token not padding AND its proposal box
//    inside (0.01, 0.99)) is evaluated per token from the valid extents, which each block recounts from
//    row 0 / column 0 of the mask exactly like the reference.
//  * class_max_times_kernel: mc_score = score_tgt.max(-1)[0] * foreground_pre_layer
//    (models/bricks/salience_transformer.py:366): one pass over the [B*Nq, num_classes] logits.
#include "common.h"
#include "class_head_core.h"

namespace sdetr {

struct FlattenArgs {
    const float *feat;   // [B,C,H,W]
    const float *pos;    // [B,C,H,W]
    const uint8_t *mask; // [B,H,W]
    const float *level_embed;  // [C]
    int B, C, H, W, S, start;
    float box_wh;  // 0.05 * 2^level
    float *feat_out;     // [B,S,C]
    float *pos_out;      // [B,S,C]
    float *sum_out;      // [B,S,C]  (feat + pos) * keep
    uint8_t *mask_out;   // [B,S]
    bf16_t *feat_bf16;   // [B,S,C] or NULL
    bf16_t *pos_bf16;    // [B,S,C] or NULL
    float *valid_ratio;  // this level's (w, h) pair of image 0; stride between images below
    int valid_ratio_stride;
};

// 256 threads as (32, 8): tile of 32 tokens x 32 channels.  `lds`: 2 * 32 * 33 floats + 2 ints.
__device__ __forceinline__ void pyramid_flatten_body(const FlattenArgs &p, int bx, int by, int b, float *lds)
{
    float (*tf)[33] = reinterpret_cast<float (*)[33]>(lds);
    float (*tp)[33] = reinterpret_cast<float (*)[33]>(lds + 32 * 33);
    int *valid_hw = reinterpret_cast<int *>(lds + 2 * 32 * 33);
    const int HW = p.H * p.W;
    const int tok0 = bx * 32, ch0 = by * 32;
    const int tid = (int)threadIdx.x;
    const int tx = tid & 31, ty = tid >> 5;
    const uint8_t *mb = p.mask + (int64_t)b * HW;

    // valid extents: number of unmasked entries in column 0 (height) and row 0 (width)
    if (tid < 64) {
        int cnt = 0;
        if (tid < 32) { for (int i = tid; i < p.H; i += 32) cnt += mb[(int64_t)i * p.W] == 0; }
        else          { for (int i = tid - 32; i < p.W; i += 32) cnt += mb[i] == 0; }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 32);
        if ((tid & 31) == 0) valid_hw[tid >> 5] = cnt;
    }
    // load: threads along tokens (contiguous in NCHW)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = ch0 + ty + 8 * r, t = tok0 + tx;
        float f = 0.f, q = 0.f;
        if (c < p.C && t < HW) {
            const int64_t i = ((int64_t)b * p.C + c) * HW + t;
            f = p.feat[i];
            q = p.pos[i] + p.level_embed[c];
        }
        tf[ty + 8 * r][tx] = f;
        tp[ty + 8 * r][tx] = q;
    }
    __syncthreads();
    const float vh = (float)valid_hw[0], vw = (float)valid_hw[1];
    if (p.valid_ratio && bx == 0 && by == 0 && tid == 0) {  // get_valid_ratios: (w, h)
        p.valid_ratio[(int64_t)b * p.valid_ratio_stride + 0] = vw / (float)p.W;
        p.valid_ratio[(int64_t)b * p.valid_ratio_stride + 1] = vh / (float)p.H;
    }
    // store: threads along channels (contiguous in token-major)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = tok0 + ty + 8 * r, c = ch0 + tx;
        if (t < HW && c < p.C) {
            const int y = t / p.W, x = t - y * p.W;
            const bool pad = mb[t] != 0;
            const float cx = ((float)x + 0.5f) / vw, cy = ((float)y + 0.5f) / vh;
            const bool keep = !pad && cx > 0.01f && cx < 0.99f && cy > 0.01f && cy < 0.99f &&
                              p.box_wh > 0.01f && p.box_wh < 0.99f;
            const float f = tf[tx][ty + 8 * r], q = tp[tx][ty + 8 * r];
            const int64_t o = ((int64_t)b * p.S + p.start + t) * p.C + c;
            if (p.feat_out) p.feat_out[o] = f;
            if (p.pos_out) p.pos_out[o] = q;
            p.sum_out[o] = keep ? f + q : 0.f;
            // bf16 copies: even lanes store channel pairs (4-byte stores; C is even)
            if (!(tx & 1) && c + 1 < p.C) {
                if (p.feat_bf16) *reinterpret_cast<uint32_t *>(p.feat_bf16 + o) = pack_act2(f, tf[tx + 1][ty + 8 * r]);
                if (p.pos_bf16) *reinterpret_cast<uint32_t *>(p.pos_bf16 + o) = pack_act2(q, tp[tx + 1][ty + 8 * r]);
            }
            if (c == 0) p.mask_out[(int64_t)b * p.S + p.start + t] = pad ? 1 : 0;
        }
    }
}

// Wide form for levels whose pixel count is a multiple of 4 (and C of 64): tiles of 64 tokens x 64 channels, 16-byte
// loads along the pixels of a channel row (a wave reads 4 rows x 256 contiguous bytes instead of 2 x 128) and 16-byte
// fp32 / 8-byte bf16 stores along the channels of a token.  Same arithmetic, same outputs.
// `lds`: 2 * 64 * 65 floats + 2 ints.
__device__ __forceinline__ void pyramid_flatten_wide_body(const FlattenArgs &p, int bx, int by, int b, float *lds)
{
    float (*tf)[65] = reinterpret_cast<float (*)[65]>(lds);   // [token][channel]; odd stride: both phases at most 2-way conflicts
    float (*tp)[65] = reinterpret_cast<float (*)[65]>(lds + 64 * 65);
    int *valid_hw = reinterpret_cast<int *>(lds + 2 * 64 * 65);
    const int HW = p.H * p.W;
    const int tok0 = bx * 64, ch0 = by * 64;
    const int tid = (int)threadIdx.x;
    const uint8_t *mb = p.mask + (int64_t)b * HW;
    if (tid < 128) {
        int cnt = 0;
        if (tid < 64) { for (int i = tid; i < p.H; i += 64) cnt += mb[(int64_t)i * p.W] == 0; }
        else          { for (int i = tid - 64; i < p.W; i += 64) cnt += mb[i] == 0; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        if ((tid & 63) == 0) valid_hw[tid >> 6] = cnt;
    }
    // load: 16 lanes x 4 pixels along a channel row, 16 channel rows per pass
    const int g = tid & 15, cr = tid >> 4;
    float4 f[4], q[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = ch0 + cr + 16 * r, t = tok0 + 4 * g;
        f[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        q[r] = f[r];
        if (t < HW) {   // HW % 4 == 0: a piece is inside the level or outside it as a whole
            const int64_t i = ((int64_t)b * p.C + c) * HW + t;
            f[r] = *reinterpret_cast<const float4 *>(p.feat + i);
            q[r] = *reinterpret_cast<const float4 *>(p.pos + i);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int cl = cr + 16 * r;
        const float e = p.level_embed[ch0 + cl];
        tf[4 * g][cl] = f[r].x; tf[4 * g + 1][cl] = f[r].y; tf[4 * g + 2][cl] = f[r].z; tf[4 * g + 3][cl] = f[r].w;
        tp[4 * g][cl] = q[r].x + e; tp[4 * g + 1][cl] = q[r].y + e; tp[4 * g + 2][cl] = q[r].z + e; tp[4 * g + 3][cl] = q[r].w + e;
    }
    __syncthreads();
    const float vh = (float)valid_hw[0], vw = (float)valid_hw[1];
    if (p.valid_ratio && bx == 0 && by == 0 && tid == 0) {
        p.valid_ratio[(int64_t)b * p.valid_ratio_stride + 0] = vw / (float)p.W;
        p.valid_ratio[(int64_t)b * p.valid_ratio_stride + 1] = vh / (float)p.H;
    }
    // store: 16 lanes x 4 channels along a token, 16 tokens per pass
    const int c4 = 4 * (tid & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int tl = (tid >> 4) + 16 * r, t = tok0 + tl;
        if (t >= HW) continue;
        const int y = t / p.W, x = t - y * p.W;
        const bool pad = mb[t] != 0;
        const float cx = ((float)x + 0.5f) / vw, cy = ((float)y + 0.5f) / vh;
        const bool keep = !pad && cx > 0.01f && cx < 0.99f && cy > 0.01f && cy < 0.99f && p.box_wh > 0.01f && p.box_wh < 0.99f;
        const float4 fv = make_float4(tf[tl][c4], tf[tl][c4 + 1], tf[tl][c4 + 2], tf[tl][c4 + 3]);
        const float4 qv = make_float4(tp[tl][c4], tp[tl][c4 + 1], tp[tl][c4 + 2], tp[tl][c4 + 3]);
        const int64_t o = ((int64_t)b * p.S + p.start + t) * p.C + ch0 + c4;
        if (p.feat_out) *reinterpret_cast<float4 *>(p.feat_out + o) = fv;
        if (p.pos_out) *reinterpret_cast<float4 *>(p.pos_out + o) = qv;
        *reinterpret_cast<float4 *>(p.sum_out + o) =
            keep ? make_float4(fv.x + qv.x, fv.y + qv.y, fv.z + qv.z, fv.w + qv.w) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.feat_bf16) *reinterpret_cast<uint2 *>(p.feat_bf16 + o) = make_uint2(pack_act2(fv.x, fv.y), pack_act2(fv.z, fv.w));
        if (p.pos_bf16) *reinterpret_cast<uint2 *>(p.pos_bf16 + o) = make_uint2(pack_act2(qv.x, qv.y), pack_act2(qv.z, qv.w));
        if (c4 == 0 && ch0 == 0) p.mask_out[(int64_t)b * p.S + p.start + t] = pad ? 1 : 0;
    }
}

constexpr int kFlattenLdsFloats = 2 * 64 * 65 + 2;
constexpr int kFlattenMaxLevels = 8;

__global__ void __launch_bounds__(256) pyramid_flatten_kernel(FlattenArgs p)
{
    __shared__ float lds[2 * 32 * 33 + 2];
    pyramid_flatten_body(p, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, lds);
}

__global__ void __launch_bounds__(256) pyramid_flatten_wide_kernel(FlattenArgs p)
{
    __shared__ float lds[kFlattenLdsFloats];
    pyramid_flatten_wide_body(p, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, lds);
}

// All levels in one launch: the blocks of level 0 first (the long tail of small levels runs beside them instead of
// after them -- three of the four per-level launches of the 800x1333 pyramid are latency, not bandwidth).
struct FlattenAllArgs {
    FlattenArgs lv[kFlattenMaxLevels];
    int first_block[kFlattenMaxLevels + 1];   // prefix sums of the levels' block counts
    int blocks_x[kFlattenMaxLevels], blocks_y[kFlattenMaxLevels];
    int wide[kFlattenMaxLevels];
    int levels;
};

__global__ void __launch_bounds__(256) pyramid_flatten_all_kernel(FlattenAllArgs a)
{
    __shared__ float lds[kFlattenLdsFloats];
    const int blk = (int)blockIdx.x;
    int l = 0;
    while (l + 1 < a.levels && blk >= a.first_block[l + 1]) ++l;
    int r = blk - a.first_block[l];
    const int bx = r % a.blocks_x[l];
    r /= a.blocks_x[l];
    const int by = r % a.blocks_y[l], b = r / a.blocks_y[l];
    if (a.wide[l]) pyramid_flatten_wide_body(a.lv[l], bx, by, b, lds);
    else pyramid_flatten_body(a.lv[l], bx, by, b, lds);
}

template <typename T>
__device__ __forceinline__ float to_f32(T v);
template <>
__device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return act_lo((uint32_t)v); }   // (a 16-bit activation element)

// 16 lanes per row; rows = B*Nq
template <typename T>
__global__ void __launch_bounds__(256) class_max_times_kernel(const T *score, const float *fg, int64_t rows, int C,
                                                              int rows_per_batch, int64_t fg_batch_stride, float *out)
{
    const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int l = threadIdx.x & 15;
    float mx = -INFINITY;
    if (row < rows) {
        const T *s = score + row * C;
        for (int c = l; c < C; c += 16) mx = fmaxf(mx, to_f32<T>(s[c]));
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
    if (row < rows && l == 0) {
        const int64_t b = row / rows_per_batch;
        out[row] = mx * fg[b * fg_batch_stride + (row - b * rows_per_batch)];
    }
}

// out = mask ? min(mins[0..L)) : score   (foreground_score of salience_transformer.py:164-168: the per-level
// minima are already known, torch would re-reduce the whole [B,S] array)
__global__ void __launch_bounds__(256) masked_fill_min_kernel(const float *score, const uint8_t *mask, const float *mins,
                                                              int L, int64_t total, float *out)
{
    float m = mins[0];
    for (int l = 1; l < L; ++l) m = fminf(m, mins[l]);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = mask[i] ? m : score[i];
}

// reference points of the encoder (salience_transformer.py:418-432) for the tokens index[b][i] (or token i):
// out[b][i][j] = ((x + 0.5) / (vr[b][l][0] * W_l), (y + 0.5) / (vr[b][l][1] * H_l)) * vr[b][j], l = the token's level
__global__ void __launch_bounds__(256) reference_points_kernel(const float *vr, const int64_t *shapes, const int64_t *lsi,
                                                               const int64_t *index, int64_t index_batch_stride, int B,
                                                               int n, int L, float *out)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= (int64_t)B * n) return;
    const int b = (int)(r / n), i = (int)(r - (int64_t)b * n);
    const int64_t tok = index ? index[(int64_t)b * index_batch_stride + i] : i;
    int l = 0;
    for (int j = 1; j < L; ++j) l = tok >= lsi[j] ? j : l;
    const int W = (int)shapes[2 * l + 1], H = (int)shapes[2 * l];
    const int t = (int)(tok - lsi[l]);
    const int y = t / W, x = t - y * W;
    const float *v = vr + (int64_t)b * L * 2;
    const float cx = ((float)x + 0.5f) / (v[2 * l] * (float)W), cy = ((float)y + 0.5f) / (v[2 * l + 1] * (float)H);
    float *o = out + r * L * 2;
    for (int j = 0; j < L; ++j) {
        o[2 * j] = cx * v[2 * j];
        o[2 * j + 1] = cy * v[2 * j + 1];
    }
}

// Entry of the encoder's sorted-order loop (salience_transformer.py:453-461 gathers + :418-432 reference points) in one
// launch: for the tokens index[b][i] copy their query and position rows, gather their foreground score and evaluate
// their reference points -- four launches of 5-7 us before.  One wave per row pair: 32 lanes x 16 bytes = a 512-byte row.
struct PrepareArgs {
    const uint4 *tokens, *pos;       // [B, S, vec_per_row]
    const float *score;              // [B, S] or NULL
    const int64_t *index;            // [B, n], images index_batch_stride apart
    int64_t index_batch_stride;
    const float *vr;                 // [B, L, 2]
    const int64_t *shapes, *lsi;
    int B, S, n, L, vec_per_row;
    uint4 *q_out, *pos_out;          // [B, n, vec_per_row]
    float *score_out;                // [B, n] or NULL
    float *ref_out;                  // [B, n, L, 2]
    const uint8_t *score_mask;       // [B, S] or NULL: masked tokens take min(score_mins) instead of their score
    const float *score_mins;         // [num_mins]
    int num_mins;
};

// One row per 16-byte-piece group and pass.  Round 4: 32-bit row arithmetic (the 64-bit division per lane was ~150
// instructions), level geometry from LDS, and the reference points of a row written by 2 L lanes of its group, one element
// each, their operands requested before the row stores: 16.3 -> 13.3 us for 22 726 rows (46 MB: ~9 us of bandwidth).  More
// rows per thread (kPrepRows = 4: the index loads, then the row pieces, then the stores, 711 workgroups instead of 2841)
// measured SLOWER, 27.9 us: the launch lives on the number of independent index -> row chains in flight.
constexpr int kPrepRows = 1;
__global__ void __launch_bounds__(256) encoder_prepare_kernel(PrepareArgs p)
{
    // (32-bit row arithmetic: the launcher checks B * n and B * S * vec_per_row against 2^31 -- a 64-bit division per row and
    // lane is ~150 instructions)
    __shared__ int geo[3 * kMaxLevels];   // level start | W | H
    if ((int)threadIdx.x < p.L) {
        geo[threadIdx.x] = (int)p.lsi[threadIdx.x];
        geo[kMaxLevels + threadIdx.x] = (int)p.shapes[2 * threadIdx.x + 1];
        geo[2 * kMaxLevels + threadIdx.x] = (int)p.shapes[2 * threadIdx.x];
    }
    __syncthreads();
    float fill = 0.f;   // masked_fill(mask, score.min()) of salience_transformer.py:168: the minimum of the level minima
    if (p.score_mask) {
        fill = p.score_mins[0];
        for (int l = 1; l < p.num_mins; ++l) fill = fminf(fill, p.score_mins[l]);
    }
    const uint32_t total = (uint32_t)p.B * (uint32_t)p.n, n = (uint32_t)p.n;
    const uint32_t per_block = 256u / (uint32_t)p.vec_per_row;      // rows per block pass (vec_per_row divides 256)
    const uint32_t sub = threadIdx.x / (uint32_t)p.vec_per_row, c = threadIdx.x - sub * (uint32_t)p.vec_per_row;
    const uint32_t stride = gridDim.x * per_block;
    const bool ref_lane = c < 2u * (uint32_t)p.L;   // this lane writes element c = (level c / 2, x | y) of its rows' reference points
    for (uint32_t r0 = blockIdx.x * per_block + sub; r0 < total; r0 += kPrepRows * stride) {
        uint32_t row[kPrepRows], bi[kPrepRows];
        int tok[kPrepRows];
        bool ok[kPrepRows];
#pragma unroll
        for (int u = 0; u < kPrepRows; ++u) {
            const uint32_t r = r0 + u * stride;
            ok[u] = r < total && r >= r0;
            row[u] = ok[u] ? r : r0;
            bi[u] = row[u] / n;
            tok[u] = (int)p.index[(int64_t)bi[u] * p.index_batch_stride + (row[u] - bi[u] * n)];
        }
        uint4 qv[kPrepRows], pv[kPrepRows];
        float sc[kPrepRows], centre_px[kPrepRows], size[kPrepRows], va[kPrepRows], vb[kPrepRows];
        uint8_t mk[kPrepRows];
#pragma unroll
        for (int u = 0; u < kPrepRows; ++u) {
            const uint32_t src = (bi[u] * (uint32_t)p.S + (uint32_t)tok[u]) * (uint32_t)p.vec_per_row + c;
            qv[u] = p.tokens[src];
            pv[u] = p.pos[src];
            sc[u] = (c == 0 && p.score_out) ? p.score[bi[u] * (uint32_t)p.S + (uint32_t)tok[u]] : 0.f;
            mk[u] = (c == 0 && p.score_out && p.score_mask) ? p.score_mask[bi[u] * (uint32_t)p.S + (uint32_t)tok[u]] : (uint8_t)0;
            centre_px[u] = size[u] = va[u] = vb[u] = 1.f;
            if (ref_lane) {
                int l = 0;
                for (int j = 1; j < p.L; ++j) l = tok[u] >= geo[j] ? j : l;
                const uint32_t W = (uint32_t)geo[kMaxLevels + l], H = (uint32_t)geo[2 * kMaxLevels + l];
                const uint32_t t = (uint32_t)(tok[u] - geo[l]);
                const uint32_t y = t / W, x = t - y * W;
                const bool is_y = c & 1u;
                centre_px[u] = (float)(is_y ? y : x) + 0.5f;
                size[u] = (float)(is_y ? H : W);
                const float *v = p.vr + (int64_t)bi[u] * p.L * 2;
                va[u] = v[2 * l + (is_y ? 1 : 0)];
                vb[u] = v[c];
            }
        }
#pragma unroll
        for (int u = 0; u < kPrepRows; ++u) {
            if (!ok[u]) continue;
            p.q_out[(int64_t)row[u] * p.vec_per_row + c] = qv[u];
            p.pos_out[(int64_t)row[u] * p.vec_per_row + c] = pv[u];
            if (c == 0 && p.score_out) p.score_out[row[u]] = mk[u] ? fill : sc[u];
            if (ref_lane) p.ref_out[(int64_t)row[u] * p.L * 2 + c] = centre_px[u] / (va[u] * size[u]) * vb[u];
        }
    }
}

// The same entry with the FIRST layer's class score of the gathered rows (salience_transformer.py:462, 366:
// max_c(class_head(q)) * foreground) -- the class head used to be the next launch (8.2 us at 2 x 11 363 rows, most of it
// the launch's own start-up).  512-byte rows only; 32 rows per 1024-thread block, a row per 32-lane group exactly as
// above (the launch lives on the number of independent index -> row chains in flight: unchanged), the gathered query rows
// also go to an LDS tile and three waves take one 32-class tile each (class_head_core.h).
struct PrepareClsArgs {
    const char *cls_pw;      // the class head's packed fragments (sdetr_class_head_pack_bf16)
    const float *cls_bias;   // [96], -inf on the padded classes
    float *cmax;             // [B, n]
};

__global__ void __launch_bounds__(1024) encoder_prepare_cls_kernel(PrepareArgs p, PrepareClsArgs k)
{
    __shared__ int geo[3 * kMaxLevels];   // level start | W | H
    __shared__ __attribute__((aligned(16))) unsigned char ytile[kClsTileRows * kClsRowBytes];
    __shared__ float fgs[kClsTileRows];
    __shared__ float red[3][kClsTileRows];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint4 af[16];
    if (wave < 3) class_frag_load<0, 16>(af, k.cls_pw, wave, lane);   // (requested first: they arrive under the gather)
    if (tid < p.L) {
        geo[tid] = (int)p.lsi[tid];
        geo[kMaxLevels + tid] = (int)p.shapes[2 * tid + 1];
        geo[2 * kMaxLevels + tid] = (int)p.shapes[2 * tid];
    }
    __syncthreads();
    float fill = 0.f;
    if (p.score_mask) {
        fill = p.score_mins[0];
        for (int l = 1; l < p.num_mins; ++l) fill = fminf(fill, p.score_mins[l]);
    }
    const uint32_t total = (uint32_t)p.B * (uint32_t)p.n, n = (uint32_t)p.n;
    const uint32_t sub = (uint32_t)tid >> 5, c = (uint32_t)tid & 31u;
    const uint32_t r = blockIdx.x * (uint32_t)kClsTileRows + sub;
    const bool ok = r < total;
    const uint32_t row = ok ? r : total - 1u;
    const uint32_t bi = row / n;
    const int tok = (int)p.index[(int64_t)bi * p.index_batch_stride + (row - bi * n)];
    const uint32_t src = (bi * (uint32_t)p.S + (uint32_t)tok) * 32u + c;
    const uint4 qv = p.tokens[src];
    const uint4 pv = p.pos[src];
    const float sc = c == 0 ? p.score[bi * (uint32_t)p.S + (uint32_t)tok] : 0.f;
    const uint8_t mk = (c == 0 && p.score_mask) ? p.score_mask[bi * (uint32_t)p.S + (uint32_t)tok] : (uint8_t)0;
    const bool ref_lane = c < 2u * (uint32_t)p.L;
    float centre_px = 1.f, size = 1.f, va = 1.f, vb = 1.f;
    if (ref_lane) {
        int l = 0;
        for (int j = 1; j < p.L; ++j) l = tok >= geo[j] ? j : l;
        const uint32_t W = (uint32_t)geo[kMaxLevels + l], H = (uint32_t)geo[2 * kMaxLevels + l];
        const uint32_t t = (uint32_t)(tok - geo[l]);
        const uint32_t y = t / W, x = t - y * W;
        const bool is_y = c & 1u;
        centre_px = (float)(is_y ? y : x) + 0.5f;
        size = (float)(is_y ? H : W);
        const float *v = p.vr + (int64_t)bi * p.L * 2;
        va = v[2 * l + (is_y ? 1 : 0)];
        vb = v[c];
    }
    if (ok) {
        p.q_out[(int64_t)row * 32 + c] = qv;
        p.pos_out[(int64_t)row * 32 + c] = pv;
        if (c == 0 && p.score_out) p.score_out[row] = mk ? fill : sc;
        if (ref_lane) p.ref_out[(int64_t)row * p.L * 2 + c] = centre_px / (va * size) * vb;
    }
    *reinterpret_cast<uint4 *>(ytile + sub * kClsRowBytes + 16 * c) = qv;
    if (c == 0) fgs[sub] = mk ? fill : sc;
    __syncthreads();
    if (wave < 3) {
        const float mx = class_tile_max(af, ytile, k.cls_bias, wave, lane);
        if (lane < 32) red[wave][lane] = mx;
    }
    __syncthreads();
    if (tid < kClsTileRows) {
        const uint32_t rr = blockIdx.x * (uint32_t)kClsTileRows + (uint32_t)tid;
        if (rr < total) k.cmax[rr] = fmaxf(fmaxf(red[0][tid], red[1][tid]), red[2][tid]) * fgs[tid];
    }
}

}  // namespace sdetr

using namespace sdetr;

static int encoder_prepare_impl(sdetr_stream_t stream, const void *tokens, const void *pos, int row_bytes, const float *score,
                                const int64_t *index, int64_t index_batch_stride, int batch_size, int spatial_size, int rows,
                                const float *valid_ratios, const int64_t *shapes, const int64_t *level_start_index,
                                int num_levels, void *query_out, void *pos_out, float *score_out, float *reference_points_out,
                                const uint8_t *score_mask, const float *score_mins, int num_mins, const void *class_packed,
                                const float *class_bias_padded, float *class_score_out);

extern "C" int sdetr_encoder_prepare_sorted_scored(sdetr_stream_t stream, const void *tokens, const void *pos, int row_bytes,
                                                   const float *score, const int64_t *index, int64_t index_batch_stride,
                                                   int batch_size, int spatial_size, int rows, const float *valid_ratios,
                                                   const int64_t *shapes, const int64_t *level_start_index, int num_levels,
                                                   void *query_out, void *pos_out, float *score_out,
                                                   float *reference_points_out, const uint8_t *score_mask,
                                                   const float *score_mins, int num_mins, const void *class_packed,
                                                   const float *class_bias_padded, float *class_score_out)
{
    if (row_bytes != 512) return fail("encoder_prepare_scored: 512-byte rows (256 16-bit channels) only");
    if (!class_packed || !class_bias_padded || !class_score_out || !score)
        return fail("encoder_prepare_scored: the class head's fragments, bias, a score output and the foreground score are needed");
    return encoder_prepare_impl(stream, tokens, pos, row_bytes, score, index, index_batch_stride, batch_size, spatial_size, rows,
                                valid_ratios, shapes, level_start_index, num_levels, query_out, pos_out, score_out,
                                reference_points_out, score_mask, score_mins, num_mins, class_packed, class_bias_padded,
                                class_score_out);
}

extern "C" int sdetr_encoder_prepare_sorted(sdetr_stream_t stream, const void *tokens, const void *pos, int row_bytes,
                                            const float *score, const int64_t *index, int64_t index_batch_stride,
                                            int batch_size, int spatial_size, int rows, const float *valid_ratios,
                                            const int64_t *shapes, const int64_t *level_start_index, int num_levels,
                                            void *query_out, void *pos_out, float *score_out, float *reference_points_out,
                                            const uint8_t *score_mask, const float *score_mins, int num_mins)
{
    return encoder_prepare_impl(stream, tokens, pos, row_bytes, score, index, index_batch_stride, batch_size, spatial_size, rows,
                                valid_ratios, shapes, level_start_index, num_levels, query_out, pos_out, score_out,
                                reference_points_out, score_mask, score_mins, num_mins, nullptr, nullptr, nullptr);
}

static int encoder_prepare_impl(sdetr_stream_t stream, const void *tokens, const void *pos, int row_bytes, const float *score,
                                const int64_t *index, int64_t index_batch_stride, int batch_size, int spatial_size, int rows,
                                const float *valid_ratios, const int64_t *shapes, const int64_t *level_start_index,
                                int num_levels, void *query_out, void *pos_out, float *score_out, float *reference_points_out,
                                const uint8_t *score_mask, const float *score_mins, int num_mins, const void *class_packed,
                                const float *class_bias_padded, float *class_score_out)
{
    if (batch_size < 0 || spatial_size < 0 || rows < 0 || num_levels <= 0 || num_levels > kMaxLevels)
        return fail("encoder_prepare: bad sizes");
    if (row_bytes <= 0 || (row_bytes & 15) || 256 % (row_bytes / 16)) return fail("encoder_prepare: row_bytes must be 16 * a divisor of 256");
    if ((int64_t)batch_size * rows == 0) return 0;
    if (!tokens || !pos || !index || !valid_ratios || !shapes || !level_start_index || !query_out || !pos_out || !reference_points_out)
        return fail("encoder_prepare: null pointer");
    if (score_out && !score) return fail("encoder_prepare: score_out without score");
    if ((score_mask != nullptr) != (score_mins != nullptr) || (score_mask && (num_mins <= 0 || !score_out)))
        return fail("encoder_prepare: score_mask and score_mins (num_mins > 0) come together, with a score output");
    if (index_batch_stride < rows) return fail("encoder_prepare: index batch stride too small");
    if (2 * num_levels > row_bytes / 16) return fail("encoder_prepare: rows of at least 32 bytes per level expected");
    if ((int64_t)batch_size * rows >= ((int64_t)1 << 30) || (int64_t)batch_size * spatial_size * (row_bytes / 16) >= ((int64_t)1 << 31))
        return fail("encoder_prepare: too many rows for 32-bit row arithmetic");
    PrepareArgs a{};
    a.tokens = (const uint4 *)tokens; a.pos = (const uint4 *)pos; a.score = score; a.index = index;
    a.index_batch_stride = index_batch_stride; a.vr = valid_ratios; a.shapes = shapes; a.lsi = level_start_index;
    a.B = batch_size; a.S = spatial_size; a.n = rows; a.L = num_levels; a.vec_per_row = row_bytes / 16;
    a.q_out = (uint4 *)query_out; a.pos_out = (uint4 *)pos_out; a.score_out = score_out; a.ref_out = reference_points_out;
    a.score_mask = score_mask; a.score_mins = score_mins; a.num_mins = num_mins;
    if (class_score_out) {
        PrepareClsArgs k{static_cast<const char *>(class_packed), class_bias_padded, class_score_out};
        const int64_t cblocks = ((int64_t)batch_size * rows + kClsTileRows - 1) / kClsTileRows;
        hipLaunchKernelGGL(encoder_prepare_cls_kernel, dim3((unsigned)cblocks), dim3(1024), 0, static_cast<hipStream_t>(stream), a, k);
        return check_launch("encoder_prepare_scored");
    }
    const int per_block = 256 / a.vec_per_row;
    int64_t blocks = ((int64_t)batch_size * rows + per_block * kPrepRows - 1) / (per_block * kPrepRows);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(encoder_prepare_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return check_launch("encoder_prepare");
}

extern "C" int sdetr_masked_fill_min(sdetr_stream_t stream, const float *score, const uint8_t *mask, const float *mins,
                                     int num_mins, int64_t total, float *out)
{
    if (total < 0 || num_mins <= 0) return fail("masked_fill_min: bad sizes");
    if (total == 0) return 0;
    if (!score || !mask || !mins || !out) return fail("masked_fill_min: null pointer");
    int64_t blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(masked_fill_min_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), score,
                       mask, mins, num_mins, total, out);
    return check_launch("masked_fill_min");
}

extern "C" int sdetr_encoder_reference_points(sdetr_stream_t stream, const float *valid_ratios, const int64_t *shapes,
                                              const int64_t *level_start_index, const int64_t *index,
                                              int64_t index_batch_stride, int batch_size, int rows, int num_levels,
                                              float *out)
{
    if (batch_size < 0 || rows < 0 || num_levels <= 0 || num_levels > kMaxLevels) return fail("reference_points: bad sizes");
    if ((int64_t)batch_size * rows == 0) return 0;
    if (!valid_ratios || !shapes || !level_start_index || !out) return fail("reference_points: null pointer");
    if (index && index_batch_stride < rows) return fail("reference_points: index batch stride too small");
    const int64_t total = (int64_t)batch_size * rows;
    hipLaunchKernelGGL(reference_points_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), valid_ratios, shapes, level_start_index, index,
                       index_batch_stride, batch_size, rows, num_levels, out);
    return check_launch("reference_points");
}


extern "C" int sdetr_pyramid_flatten_level(sdetr_stream_t stream, const float *feat, const float *pos,
                                           const uint8_t *mask, const float *level_embed, int B, int C, int H, int W,
                                           int level, int level_start, int S, float *feat_out, float *pos_out,
                                           float *sum_out, uint8_t *mask_out, void *feat_bf16, void *pos_bf16,
                                           float *valid_ratio, int valid_ratio_stride)
{
    if (B < 0 || C <= 0 || H <= 0 || W <= 0 || level < 0 || level_start < 0 || S < level_start + H * W)
        return fail("pyramid_flatten_level: bad dims");
    if (!feat || !pos || !mask || !level_embed || !sum_out || !mask_out)
        return fail("pyramid_flatten_level: null pointer");
    if ((feat_bf16 || pos_bf16) && (C & 1)) return fail("pyramid_flatten_level: bf16 copies need an even channel count");
    if (B == 0) return 0;
    FlattenArgs a{};
    a.feat = feat; a.pos = pos; a.mask = mask; a.level_embed = level_embed;
    a.B = B; a.C = C; a.H = H; a.W = W; a.S = S; a.start = level_start;
    a.box_wh = 0.05f * (float)(1u << level);
    a.feat_out = feat_out; a.pos_out = pos_out; a.sum_out = sum_out; a.mask_out = mask_out;
    a.feat_bf16 = reinterpret_cast<bf16_t *>(feat_bf16); a.pos_bf16 = reinterpret_cast<bf16_t *>(pos_bf16);
    a.valid_ratio = valid_ratio; a.valid_ratio_stride = valid_ratio_stride;
    if ((C & 63) == 0 && ((H * W) & 3) == 0 && ((reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(pos)) & 15) == 0) {
        const dim3 grid((unsigned)((H * W + 63) / 64), (unsigned)(C / 64), (unsigned)B);
        hipLaunchKernelGGL(pyramid_flatten_wide_kernel, grid, dim3(256), 0, stream, a);
        return check_launch("pyramid_flatten_level");
    }
    const dim3 grid((unsigned)((H * W + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)B);
    hipLaunchKernelGGL(pyramid_flatten_kernel, grid, dim3(256), 0, stream, a);
    return check_launch("pyramid_flatten_level");
}

// The whole pyramid in one launch: the same per-level arithmetic and outputs as `num_levels` calls of
// sdetr_pyramid_flatten_level with level_start = the running pixel count.  feats / pos / masks: HOST arrays of
// `num_levels` device pointers; heights / widths: host arrays; level_embeds [num_levels, C] and valid_ratios
// [B, num_levels, 2] (may be NULL) on the device.
extern "C" int sdetr_pyramid_flatten(sdetr_stream_t stream, int num_levels, const float *const *feats,
                                     const float *const *pos, const uint8_t *const *masks, const int *heights,
                                     const int *widths, const float *level_embeds, int B, int C, int S, float *feat_out,
                                     float *pos_out, float *sum_out, uint8_t *mask_out, void *feat_bf16, void *pos_bf16,
                                     float *valid_ratios)
{
    if (num_levels <= 0 || num_levels > kFlattenMaxLevels) return fail("pyramid_flatten: 1..%d levels", kFlattenMaxLevels);
    if (B < 0 || C <= 0 || S <= 0) return fail("pyramid_flatten: bad dims");
    if (!feats || !pos || !masks || !heights || !widths || !level_embeds || !sum_out || !mask_out)
        return fail("pyramid_flatten: null pointer");
    if ((feat_bf16 || pos_bf16) && (C & 1)) return fail("pyramid_flatten: bf16 copies need an even channel count");
    FlattenAllArgs a{};
    a.levels = num_levels;
    int64_t start = 0, blocks = 0;
    for (int l = 0; l < num_levels; ++l) {
        const int H = heights[l], W = widths[l];
        if (H <= 0 || W <= 0 || !feats[l] || !pos[l] || !masks[l]) return fail("pyramid_flatten: bad level %d", l);
        FlattenArgs &v = a.lv[l];
        v.feat = feats[l]; v.pos = pos[l]; v.mask = masks[l]; v.level_embed = level_embeds + (int64_t)l * C;
        v.B = B; v.C = C; v.H = H; v.W = W; v.S = S; v.start = (int)start;
        v.box_wh = 0.05f * (float)(1u << l);
        v.feat_out = feat_out; v.pos_out = pos_out; v.sum_out = sum_out; v.mask_out = mask_out;
        v.feat_bf16 = reinterpret_cast<bf16_t *>(feat_bf16); v.pos_bf16 = reinterpret_cast<bf16_t *>(pos_bf16);
        v.valid_ratio = valid_ratios ? valid_ratios + 2 * l : nullptr;
        v.valid_ratio_stride = 2 * num_levels;
        const int64_t HW = (int64_t)H * W;
        a.wide[l] = (C & 63) == 0 && (HW & 3) == 0 &&
                    ((reinterpret_cast<uintptr_t>(feats[l]) | reinterpret_cast<uintptr_t>(pos[l])) & 15) == 0;
        a.blocks_x[l] = (int)(a.wide[l] ? (HW + 63) / 64 : (HW + 31) / 32);
        a.blocks_y[l] = a.wide[l] ? C / 64 : (C + 31) / 32;
        a.first_block[l] = (int)blocks;
        blocks += (int64_t)a.blocks_x[l] * a.blocks_y[l] * B;
        start += HW;
        if (blocks > 0x7fffffff || start > S) return fail("pyramid_flatten: sizes out of range");
    }
    a.first_block[num_levels] = (int)blocks;
    if (start != S) return fail("pyramid_flatten: spatial_size %d is not the pixel count %lld", S, (long long)start);
    if (B == 0) return 0;
    hipLaunchKernelGGL(pyramid_flatten_all_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    return check_launch("pyramid_flatten");
}

extern "C" int sdetr_class_max_times(sdetr_stream_t stream, const void *score, int score_dtype, const float *scale,
                                     int64_t scale_batch_stride, int batch_size, int rows_per_batch, int num_classes,
                                     float *out)
{
    const int64_t rows = (int64_t)batch_size * rows_per_batch;
    if (batch_size < 0 || rows_per_batch < 0 || num_classes <= 0 || scale_batch_stride < rows_per_batch)
        return fail("class_max_times: bad dims");
    if (rows == 0) return 0;
    if (!score || !scale || !out) return fail("class_max_times: null pointer");
    const dim3 grid((unsigned)((rows + 15) / 16)), block(256);
    if (score_dtype == SDETR_F32)
        hipLaunchKernelGGL(class_max_times_kernel<float>, grid, block, 0, stream, (const float *)score, scale, rows,
                           num_classes, rows_per_batch, scale_batch_stride, out);
    else if (score_dtype == kActCode)
        hipLaunchKernelGGL(class_max_times_kernel<bf16_t>, grid, block, 0, stream, (const bf16_t *)score, scale, rows,
                           num_classes, rows_per_batch, scale_batch_stride, out);
    else
        return fail("class_max_times: bad dtype %d", score_dtype);
    return check_launch("class_max_times");
}
