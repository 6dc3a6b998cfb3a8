// Row N3: the RepVGGPluX neck on TOKEN-MAJOR feature maps (include/salience_hip.h section (13)).
//
// The reference reshapes the encoder memory [B, sum H*W, C] to NCHW, runs the neck's convolutions and flattens back
// (models/bricks/salience_transformer.py:185-192, models/necks/repnet.py).  Token-major IS channels-last, so nothing
// is transposed here: a pixel's C channels are one contiguous row.  In eval mode every BatchNorm folds into the
// convolution before it and the 3x3 + 1x1 pair of a RepVGG block folds into one 3x3 (host side, salience_neck.py);
// what is left for the device is
//   conv3x3_tokens_kernel   grouped / dense 3x3, stride 1 or 2, + bias (+ SiLU)
//   neck_combine_kernel     act(a + nearest_upsample(b) + bias): the epilogue of the 1x1 convolutions, whose
//                           GEMMs are plain library GEMMs on the token rows
//   se_context / se_gate / se_apply   the attention-pooled gate of models/bricks/basic.py:29-54 and the shortcut
// fp32 VALU arithmetic on fp32 or bf16 storage (LDS-tiled conv3x3_tokens_kernel: any channel count, the parity path),
// and conv3x3_mfma_kernel for bf16 maps with the real neck's channel counts.
#include "common.h"

namespace sdetr {
namespace {

template <typename T>
struct Store;
template <>
struct Store<float> {
    static __device__ __forceinline__ float4 load4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
    static __device__ __forceinline__ void store4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
};
template <>
struct Store<bf16_t> {
    static __device__ __forceinline__ float4 load4(const bf16_t *p)
    {
        const uint2 u = *reinterpret_cast<const uint2 *>(p);
        return make_float4(act_lo(u.x), act_hi(u.x), act_lo(u.y), act_hi(u.y));
    }
    static __device__ __forceinline__ void store4(bf16_t *p, float4 v)
    {
        *reinterpret_cast<uint2 *>(p) = make_uint2(pack_act2(v.x, v.y), pack_act2(v.z, v.w));
    }
};

__device__ __forceinline__ float silu(float v) { return v / (1.0f + __expf(-v)); }

// ------------------------------------------------------------------------------------------------
// 3x3 convolution, padding 1.  A workgroup owns a 4 x 16 tile of output pixels and 64 output channels of one
// group; thread (tx = tid % 16, ty = tid / 16) owns output channels 4*tx .. 4*tx+3 of 4 pixels in a row.  The input
// channels arrive 16 at a time: the tile's input patch and the 9 x 16 x 64 weight block go through LDS.
constexpr int kTileH = 4, kTileW = 16, kCK = 16, kCB = 64;

struct ConvArgs {
    const void *x;
    const float *w;     // [G][3][3][CiG][CoG]
    const float *bias;  // [G * CoG] or NULL
    void *out;
    int B, H, W, Ho, Wo;
    int ldx;            // elements between consecutive pixels of x (>= G * CiG)
    int G, CiG, CoG;
    int act;            // 0 none, 1 SiLU
    int tiles_x;
};

template <typename T, int S>
__global__ void __launch_bounds__(kBlock) conv3x3_tokens_kernel(ConvArgs p)
{
    constexpr int PR = 3 + (kTileH - 1) * S, PC = 3 + (kTileW - 1) * S, NC = 3 + 3 * S;
    __shared__ __attribute__((aligned(16))) float in_s[PR * PC * kCK];
    __shared__ __attribute__((aligned(16))) float w_s[9 * kCK * kCB];

    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int r = ty >> 2, px0 = (ty & 3) * 4;
    const int tile_x = blockIdx.x % p.tiles_x, tile_y = blockIdx.x / p.tiles_x;
    const int cblocks = (p.CoG + kCB - 1) / kCB;
    const int g = blockIdx.y / cblocks, cb0 = (blockIdx.y % cblocks) * kCB;
    const int b = blockIdx.z;
    const int oy0 = tile_y * kTileH, ox0 = tile_x * kTileW;
    const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
    const T *x = reinterpret_cast<const T *>(p.x) + (int64_t)b * p.H * p.W * p.ldx + (int64_t)g * p.CiG;
    const float *w = p.w + (int64_t)g * 9 * p.CiG * p.CoG;

    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    for (int ck0 = 0; ck0 < p.CiG; ck0 += kCK) {
        __syncthreads();
        // input patch: [PR][PC][kCK], 4 channels per load, zero outside the image / past the group's channels
        for (int e = tid; e < PR * PC * (kCK / 4); e += kBlock) {
            const int c4 = e % (kCK / 4), pix = e / (kCK / 4);
            const int pc = pix % PC, pr = pix / PC;
            const int iy = iy0 + pr, ix = ix0 + pc, c = ck0 + c4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && c < p.CiG)
                v = Store<T>::load4(x + ((int64_t)iy * p.W + ix) * p.ldx + c);
            *reinterpret_cast<float4 *>(in_s + (pix * kCK + c4 * 4)) = v;
        }
        // weights: [9][kCK][kCB]
        for (int e = tid; e < 9 * kCK * (kCB / 4); e += kBlock) {
            const int co4 = e % (kCB / 4), rest = e / (kCB / 4);
            const int c = rest % kCK, tap = rest / kCK;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ck0 + c < p.CiG && cb0 + co4 * 4 < p.CoG)
                v = *reinterpret_cast<const float4 *>(w + ((int64_t)tap * p.CiG + ck0 + c) * p.CoG + cb0 + co4 * 4);
            *reinterpret_cast<float4 *>(w_s + (tap * kCK + c) * kCB + co4 * 4) = v;
        }
        __syncthreads();
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const float *row = in_s + ((r * S + ky) * PC + px0 * S) * kCK;
#pragma unroll
            for (int c4 = 0; c4 < kCK / 4; ++c4) {
                float4 iv[NC];
#pragma unroll
                for (int n = 0; n < NC; ++n) iv[n] = *reinterpret_cast<const float4 *>(row + n * kCK + c4 * 4);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float *wt = w_s + ((ky * 3 + kx) * kCK + c4 * 4) * kCB + tx * 4;
                    const float4 w0 = *reinterpret_cast<const float4 *>(wt);
                    const float4 w1 = *reinterpret_cast<const float4 *>(wt + kCB);
                    const float4 w2 = *reinterpret_cast<const float4 *>(wt + 2 * kCB);
                    const float4 w3 = *reinterpret_cast<const float4 *>(wt + 3 * kCB);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 a = iv[q * S + kx];
                        acc[q][0] = fmaf(a.x, w0.x, acc[q][0]); acc[q][1] = fmaf(a.x, w0.y, acc[q][1]);
                        acc[q][2] = fmaf(a.x, w0.z, acc[q][2]); acc[q][3] = fmaf(a.x, w0.w, acc[q][3]);
                        acc[q][0] = fmaf(a.y, w1.x, acc[q][0]); acc[q][1] = fmaf(a.y, w1.y, acc[q][1]);
                        acc[q][2] = fmaf(a.y, w1.z, acc[q][2]); acc[q][3] = fmaf(a.y, w1.w, acc[q][3]);
                        acc[q][0] = fmaf(a.z, w2.x, acc[q][0]); acc[q][1] = fmaf(a.z, w2.y, acc[q][1]);
                        acc[q][2] = fmaf(a.z, w2.z, acc[q][2]); acc[q][3] = fmaf(a.z, w2.w, acc[q][3]);
                        acc[q][0] = fmaf(a.w, w3.x, acc[q][0]); acc[q][1] = fmaf(a.w, w3.y, acc[q][1]);
                        acc[q][2] = fmaf(a.w, w3.z, acc[q][2]); acc[q][3] = fmaf(a.w, w3.w, acc[q][3]);
                    }
                }
            }
        }
    }

    const int co = cb0 + tx * 4;
    const int oy = oy0 + r;
    if (co >= p.CoG || oy >= p.Ho) return;
    const int cout = p.G * p.CoG, cglob = g * p.CoG + co;
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bias = *reinterpret_cast<const float4 *>(p.bias + cglob);
    T *out = reinterpret_cast<T *>(p.out) + ((int64_t)b * p.Ho + oy) * p.Wo * cout + cglob;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ox = ox0 + px0 + q;
        if (ox >= p.Wo) break;
        float4 v = make_float4(acc[q][0] + bias.x, acc[q][1] + bias.y, acc[q][2] + bias.z, acc[q][3] + bias.w);
        if (p.act) v = make_float4(silu(v.x), silu(v.y), silu(v.z), silu(v.w));
        Store<T>::store4(out + (int64_t)ox * cout, v);
    }
}

// ------------------------------------------------------------------------------------------------
// The same convolution on the matrix cores, bf16 in / fp32 accumulate, for channel counts of the real neck
// (in_per_group % 16 == 0, out_per_group % 64 == 0).  Y^T[out-ch x pixel] = sum over (tap, 16 input channels) of
// W[32 x 16] X^T[16 x 32] with v_mfma_f32_32x32x16_bf16: the B operand of lane (t = lane % 32, h = lane / 32) is 8
// consecutive channels of pixel t shifted by the tap -- 16 contiguous bytes of the token-major map, loaded straight
// from global memory (zero outside the image); the A operands are 1 KB lane-ordered fragments pre-packed by
// sdetr_neck_pack_conv3x3_bf16.  A wave owns 64 output pixels (linear index over batch x Ho x Wo, so no tile is
// wasted on narrow levels) x 64 output channels: 4 accumulators, 4 MFMAs per 4 loads.  No LDS, no barriers.
typedef __bf16 nk_bf16x8_t __attribute__((ext_vector_type(8)));
typedef float nk_f32x16_t __attribute__((ext_vector_type(16)));

struct ConvMfmaArgs {
    const bf16_t *x;
    const char *pw;     // fragments [G][CoG/32][9][CiG/16][64 lanes][8] bf16
    const float *bias;  // [G * CoG] or NULL
    bf16_t *out;
    int64_t P;          // B * Ho * Wo
    int H, W, Ho, Wo, S;
    int ldx, G, CiG, CoG, act;
};

__device__ __forceinline__ nk_f32x16_t nk_mfma(uint4 a, uint4 b, nk_f32x16_t c)
{
    return mfma_act_32x32x16(a, b, c);
}

// SPLIT (round 5): a wave's nine taps are nine dependent load round trips (~2 us each: 18-20 us for a level of any size,
// 48 us at level 0 with two waves per SIMD) around 0.4 us of MFMA work per tap.  With SPLIT = 3 a workgroup is three
// waves on ONE 64-pixel tile, each with one kernel row (three taps, fully unrolled: their loads are in flight together);
// the partial accumulators meet in LDS in the fixed order w0 + w1 + w2 and wave 0 runs the epilogue.
template <bool CHUNK4, int SPLIT>  // CHUNK4: in_per_group % 64 == 0, k-steps in groups of four
__global__ void __launch_bounds__(SPLIT == 1 ? kBlock : 64 * SPLIT) __attribute__((amdgpu_waves_per_eu(2, 4)))
conv3x3_mfma_kernel(ConvMfmaArgs p)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = lane & 31, h = lane >> 5;
    const int MT = p.CoG / 32, KC = p.CiG / 16;
    const int cblocks = p.CoG / 64;
    const int g = blockIdx.y / cblocks, mt0 = (blockIdx.y % cblocks) * 2;
    const int64_t p0 = SPLIT == 1 ? ((int64_t)blockIdx.x * 4 + wave) * 64 : (int64_t)blockIdx.x * 64;
    if (p0 >= p.P) return;  // SPLIT == 1: whole wave idle (no barriers); otherwise the whole workgroup
    constexpr int kTaps = 9 / SPLIT;
    const int tap0 = SPLIT == 1 ? 0 : wave * kTaps;

    int oy[2], ox[2];
    int64_t img[2];
    bool live[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int64_t pix = p0 + u * 32 + t;
        live[u] = pix < p.P;
        const int64_t q = live[u] ? pix : 0;
        const int64_t b = q / ((int64_t)p.Ho * p.Wo);
        const int rem = (int)(q - b * (int64_t)p.Ho * p.Wo);
        oy[u] = rem / p.Wo;
        ox[u] = rem - oy[u] * p.Wo;
        img[u] = b * p.H;
    }
    nk_f32x16_t acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][u][r] = 0.f;

    const char *wbase = p.pw + ((int64_t)(g * MT + mt0) * 9 * KC) * 1024 + lane * 16;
    const int64_t mstride = (int64_t)9 * KC * 1024;  // next 32-row tile of output channels
    for (int tap = tap0; tap < tap0 + kTaps; ++tap) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const bf16_t *src[2];
        bool inb[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int iy = oy[u] * p.S + ky - 1, ix = ox[u] * p.S + kx - 1;
            inb[u] = live[u] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            src[u] = p.x + ((img[u] + (inb[u] ? iy : 0)) * p.W + (inb[u] ? ix : 0)) * p.ldx + g * p.CiG + h * 8;
        }
        const char *wt = wbase + (int64_t)tap * KC * 1024;
        if constexpr (CHUNK4) {
            // four k-steps at a time: their 16 loads are in flight together (a k-step per load round trip left the
            // small levels at 36 x ~0.75 us whatever their size)
            for (int c0 = 0; c0 < KC; c0 += 4) {
                uint4 a0[4], a1[4], b0[4], b1[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a0[i] = *reinterpret_cast<const uint4 *>(wt + (int64_t)(c0 + i) * 1024);
                    a1[i] = *reinterpret_cast<const uint4 *>(wt + mstride + (int64_t)(c0 + i) * 1024);
                    b0[i] = make_uint4(0u, 0u, 0u, 0u);
                    b1[i] = make_uint4(0u, 0u, 0u, 0u);
                    if (inb[0]) b0[i] = *reinterpret_cast<const uint4 *>(src[0] + (c0 + i) * 16);
                    if (inb[1]) b1[i] = *reinterpret_cast<const uint4 *>(src[1] + (c0 + i) * 16);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[0][0] = nk_mfma(a0[i], b0[i], acc[0][0]);
                    acc[1][0] = nk_mfma(a1[i], b0[i], acc[1][0]);
                    acc[0][1] = nk_mfma(a0[i], b1[i], acc[0][1]);
                    acc[1][1] = nk_mfma(a1[i], b1[i], acc[1][1]);
                }
            }
        } else {
            for (int c = 0; c < KC; ++c) {
                const uint4 a0 = *reinterpret_cast<const uint4 *>(wt + (int64_t)c * 1024);
                const uint4 a1 = *reinterpret_cast<const uint4 *>(wt + mstride + (int64_t)c * 1024);
                uint4 b0 = make_uint4(0u, 0u, 0u, 0u), b1 = make_uint4(0u, 0u, 0u, 0u);
                if (inb[0]) b0 = *reinterpret_cast<const uint4 *>(src[0] + c * 16);
                if (inb[1]) b1 = *reinterpret_cast<const uint4 *>(src[1] + c * 16);
                acc[0][0] = nk_mfma(a0, b0, acc[0][0]);
                acc[1][0] = nk_mfma(a1, b0, acc[1][0]);
                acc[0][1] = nk_mfma(a0, b1, acc[0][1]);
                acc[1][1] = nk_mfma(a1, b1, acc[1][1]);
            }
        }
    }

    if constexpr (SPLIT > 1) {
        // slots of 16 float4 x 64 lanes (16 KB): a wave's four accumulators, lane-contiguous (conflict-free both ways)
        extern __shared__ __align__(16) unsigned char conv_lds[];
        float4 *slots = reinterpret_cast<float4 *>(conv_lds);
        auto put = [&](int slot) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        slots[(slot * 16 + (m * 2 + u) * 4 + q) * 64 + lane] =
                            make_float4(acc[m][u][4 * q], acc[m][u][4 * q + 1], acc[m][u][4 * q + 2], acc[m][u][4 * q + 3]);
        };
        auto add = [&](int slot) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = slots[(slot * 16 + (m * 2 + u) * 4 + q) * 64 + lane];
                        acc[m][u][4 * q] += v.x; acc[m][u][4 * q + 1] += v.y;
                        acc[m][u][4 * q + 2] += v.z; acc[m][u][4 * q + 3] += v.w;
                    }
        };
        if (wave == 1 || wave == 2) put(wave - 1);
        __syncthreads();
        if (wave != 0) return;
        add(0);
        add(1);
    }
    // lane (t, h), register r of tile m: output channel (mt0 + m) * 32 + 8 * (r / 4) + 4 * h + r % 4 of pixel t
    const int cout = p.G * p.CoG;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (!live[u]) continue;
        bf16_t *orow = p.out + (p0 + u * 32 + t) * cout + g * p.CoG;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = (mt0 + m) * 32 + 8 * q + 4 * h;
                float4 v = make_float4(acc[m][u][4 * q], acc[m][u][4 * q + 1], acc[m][u][4 * q + 2], acc[m][u][4 * q + 3]);
                if (p.bias) {
                    const float4 bb = *reinterpret_cast<const float4 *>(p.bias + g * p.CoG + co);
                    v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
                }
                if (p.act) v = make_float4(silu(v.x), silu(v.y), silu(v.z), silu(v.w));
                Store<bf16_t>::store4(orow + co, v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The neck's own block shape -- stride 1, groups of 64 -> 64 channels -- with BOTH operands in LDS (round 5).  The kernel
// above reads a wave's operands fragment by fragment from global memory: nine (three, SPLIT = 3) dependent round trips per
// 64 pixels and, at level 0, 2100 such wave tiles on 1024 SIMDs -- 43-47 us for 3.9 us of MFMA work, 10-17 us at the small
// levels.  Here a workgroup of 8 waves owns (group, image, 8 rows x 32 columns of output):
//   1. one bulk fill: the group's 72 KB of packed fragments (as they lie in the packed buffer: lane-contiguous 16-byte
//      pieces) and the 10 x 34 pixel input patch x 64 channels, 144 bytes per pixel (128 + 16 of padding: the 16 lanes a
//      ds_read_b128 serves together are 16 different pixels, 36 words apart -> 16 different 4-bank slots), zero outside
//      the image -- ONE round trip, everything in flight together;
//   2. wave w = output row w of the tile: 9 taps x 4 k-steps x 2 channel tiles of v_mfma_f32_32x32x16, the B operand
//      (pixel t shifted by the tap, 8 channels) and the two A fragments by ds_read_b128, no further global loads;
//   3. bias + SiLU + 16-bit stores as above.
constexpr int kLdsConvRows = 8, kLdsConvCols = 32;
constexpr int kLdsConvPatchCols = kLdsConvCols + 2, kLdsConvPatchRows = kLdsConvRows + 2;
constexpr int kLdsConvPixelBytes = 144;
constexpr int kLdsConvWeightBytes = 2 * 9 * 4 * 1024;
constexpr int kLdsConvBytes = kLdsConvWeightBytes + kLdsConvPatchRows * kLdsConvPatchCols * kLdsConvPixelBytes;

__global__ void __launch_bounds__(64 * kLdsConvRows) conv3x3_lds_kernel(ConvMfmaArgs p, int row_tiles, int col_tiles)
{
    extern __shared__ __align__(16) unsigned char conv_lds[];
    unsigned char *wl = conv_lds, *patch = conv_lds + kLdsConvWeightBytes;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = lane & 31, h = lane >> 5;
    const int g = blockIdx.y;
    int tile = blockIdx.x;
    const int ct = tile % col_tiles; tile /= col_tiles;
    const int rt = tile % row_tiles;
    const int b = tile / row_tiles;
    const int y0 = rt * kLdsConvRows, x0 = ct * kLdsConvCols;

    // ---- the fill: weights (4608 pieces) and patch (340 pixels x 8 pieces), all loads issued before the first store ----
    {
        const uint4 *wsrc = reinterpret_cast<const uint4 *>(p.pw + (int64_t)g * kLdsConvWeightBytes);
        uint4 wv[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) wv[i] = wsrc[i * 512 + tid];
        constexpr int kPieces = kLdsConvPatchRows * kLdsConvPatchCols * 8;     // 2720
        constexpr int kPer = (kPieces + 511) / 512;                             // 6
        uint4 pv[kPer];
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const int e = i * 512 + tid;
            const int pix = e >> 3, piece = e & 7;
            const int pr = pix / kLdsConvPatchCols, pc = pix - pr * kLdsConvPatchCols;
            const int iy = y0 - 1 + pr, ix = x0 - 1 + pc;
            pv[i] = make_uint4(0u, 0u, 0u, 0u);
            if (e < kPieces && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                pv[i] = *reinterpret_cast<const uint4 *>(p.x + (((int64_t)b * p.H + iy) * p.W + ix) * p.ldx + g * 64 + piece * 8);
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) reinterpret_cast<uint4 *>(wl)[i * 512 + tid] = wv[i];
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const int e = i * 512 + tid;
            if (e < kPieces) *reinterpret_cast<uint4 *>(patch + (e >> 3) * kLdsConvPixelBytes + (e & 7) * 16) = pv[i];
        }
    }
    __syncthreads();
    const int oy = y0 + wave, ox = x0 + t;
    if (oy >= p.H) return;                       // (whole wave; no barrier follows)

    nk_f32x16_t acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const unsigned char *wlane = wl + lane * 16;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const unsigned char *bp = patch + ((wave + ky) * kLdsConvPatchCols + t + kx) * kLdsConvPixelBytes + h * 16;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint4 bv = *reinterpret_cast<const uint4 *>(bp + c * 32);
            const uint4 a0 = *reinterpret_cast<const uint4 *>(wlane + (tap * 4 + c) * 1024);
            const uint4 a1 = *reinterpret_cast<const uint4 *>(wlane + ((9 + tap) * 4 + c) * 1024);
            acc[0] = nk_mfma(a0, bv, acc[0]);
            acc[1] = nk_mfma(a1, bv, acc[1]);
        }
    }
    if (ox >= p.W) return;
    // lane (t, h), register r of tile m: output channel m * 32 + 8 * (r / 4) + 4 * h + r % 4 of pixel t
    const int cout = p.G * 64;
    bf16_t *orow = p.out + (((int64_t)b * p.H + oy) * p.W + ox) * cout + g * 64;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int co = m * 32 + 8 * q + 4 * h;
            float4 v = make_float4(acc[m][4 * q], acc[m][4 * q + 1], acc[m][4 * q + 2], acc[m][4 * q + 3]);
            if (p.bias) {
                const float4 bb = *reinterpret_cast<const float4 *>(p.bias + g * 64 + co);
                v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
            }
            if (p.act) v = make_float4(silu(v.x), silu(v.y), silu(v.z), silu(v.w));
            Store<bf16_t>::store4(orow + co, v);
        }
    }
}

// fp32 [G][3][3][CiG][CoG] -> bf16 fragments [G][CoG/32][9][CiG/16][lane][8]:
// element j of lane (n = lane % 32, h = lane / 32) = W[g][tap][c16 * 16 + h * 8 + j][mt * 32 + n]
__global__ void __launch_bounds__(kBlock) conv3x3_pack_kernel(const float *w, int G, int CiG, int CoG, bf16_t *out)
{
    const int MT = CoG / 32, KC = CiG / 16;
    const int64_t n = (int64_t)G * MT * 9 * KC * 512;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(e & 7), lane = (int)((e >> 3) & 63);
        int64_t f = e >> 9;
        const int c16 = (int)(f % KC); f /= KC;
        const int tap = (int)(f % 9); f /= 9;
        const int mt = (int)(f % MT);
        const int g = (int)(f / MT);
        const int ci = c16 * 16 + (lane >> 5) * 8 + j, co = mt * 32 + (lane & 31);
        out[e] = (bf16_t)f32_to_act_bits(w[(((int64_t)g * 9 + tap) * CiG + ci) * CoG + co]);
    }
}

// ------------------------------------------------------------------------------------------------
// out[b, y, x, :] = act(a[b, y, x, :] + up[b, ys, xs, :] + bias), (ys, xs) = the pixel torch's "nearest"
// interpolation reads: min(int(floorf(dst * (float)in / out)), in - 1)   (F.interpolate at repnet.py:224-228)
struct CombineArgs {
    const void *a;
    const void *up;    // NULL: no second term
    const float *bias; // NULL: none
    void *out;
    int64_t rows;      // B * H * W
    int H, W, Hs, Ws, C;
    int lda, ldup, ldo;
    int act;
};

template <typename T>
__global__ void __launch_bounds__(kBlock) neck_combine_kernel(CombineArgs p)
{
    const int c4n = p.C / 4;
    const int64_t n = p.rows * c4n;
    const float sh = (float)p.Hs / (float)p.H, sw = (float)p.Ws / (float)p.W;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % c4n) * 4;
        const int64_t row = e / c4n;
        float4 v = Store<T>::load4(reinterpret_cast<const T *>(p.a) + row * p.lda + c);
        if (p.up) {
            const int xq = (int)(row % p.W);
            const int64_t t = row / p.W;
            const int yq = (int)(t % p.H);
            const int64_t b = t / p.H;
            const int ys = min((int)floorf((float)yq * sh), p.Hs - 1);
            const int xs = min((int)floorf((float)xq * sw), p.Ws - 1);
            const float4 u = Store<T>::load4(reinterpret_cast<const T *>(p.up) + ((b * p.Hs + ys) * p.Ws + xs) * p.ldup + c);
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        if (p.bias) {
            const float4 bb = *reinterpret_cast<const float4 *>(p.bias + c);
            v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
        }
        if (p.act) v = make_float4(silu(v.x), silu(v.y), silu(v.z), silu(v.w));
        Store<T>::store4(reinterpret_cast<T *>(p.out) + row * p.ldo + c, v);
    }
}

// The same for 16-bit maps whose channel count and row strides are multiples of 8 and whose element count fits 32 bits
// (every call of the real neck): 16-byte accesses and 32-bit index arithmetic.  The generic kernel above moves 8 bytes per
// thread behind four 64-bit divisions -- 28 us for the 2 x 16700 x 512 map of level 0, 2.7 TB/s.
__global__ void __launch_bounds__(kBlock) neck_combine8_kernel(CombineArgs p)
{
    const unsigned c8n = (unsigned)p.C / 8u;
    const unsigned n = (unsigned)p.rows * c8n;
    const float sh = (float)p.Hs / (float)p.H, sw = (float)p.Ws / (float)p.W;
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const unsigned row = e / c8n;
        const int c = (int)(e - row * c8n) * 8;
        const uint4 av = *reinterpret_cast<const uint4 *>(reinterpret_cast<const bf16_t *>(p.a) + (int64_t)row * p.lda + c);
        float v[8] = {act_lo(av.x), act_hi(av.x), act_lo(av.y), act_hi(av.y), act_lo(av.z), act_hi(av.z), act_lo(av.w), act_hi(av.w)};
        if (p.up) {
            const unsigned t = row / (unsigned)p.W;
            const int xq = (int)(row - t * (unsigned)p.W);
            const unsigned b = t / (unsigned)p.H;
            const int yq = (int)(t - b * (unsigned)p.H);
            const int ys = min((int)floorf((float)yq * sh), p.Hs - 1);
            const int xs = min((int)floorf((float)xq * sw), p.Ws - 1);
            const uint4 uv = *reinterpret_cast<const uint4 *>(reinterpret_cast<const bf16_t *>(p.up) +
                                                             (((int64_t)b * p.Hs + ys) * p.Ws + xs) * p.ldup + c);
            v[0] += act_lo(uv.x); v[1] += act_hi(uv.x); v[2] += act_lo(uv.y); v[3] += act_hi(uv.y);
            v[4] += act_lo(uv.z); v[5] += act_hi(uv.z); v[6] += act_lo(uv.w); v[7] += act_hi(uv.w);
        }
        if (p.bias) {
            const float4 b0 = *reinterpret_cast<const float4 *>(p.bias + c), b1 = *reinterpret_cast<const float4 *>(p.bias + c + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
            v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        }
        if (p.act) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = silu(v[i]);
        }
        *reinterpret_cast<uint4 *>(reinterpret_cast<bf16_t *>(p.out) + (int64_t)row * p.ldo + c) =
            make_uint4(pack_act2(v[0], v[1]), pack_act2(v[2], v[3]), pack_act2(v[4], v[5]), pack_act2(v[6], v[7]));
    }
}

// ------------------------------------------------------------------------------------------------
// Attention pooling of SqueezeAndExcitation (basic.py:43-54): context[c] = sum_p softmax_p(w_mask . y_p) y_p[c].
// (conv_mask's bias is the same for every pixel and drops out of the softmax.)  One workgroup reduces kSePix pixels
// with a running (max, sum, weighted vector) per wave -- lane l owns channels 4l .. 4l+3 (C <= 256) -- and writes one
// partial [C + 2] = (vector, max, sum); se_gate_kernel merges the partials of an image and runs the two tiny layers.
constexpr int kSePix = 128;      // pixels per workgroup at most
// Pixels per workgroup (a multiple of 16: four waves x four pixels in flight).  Round 5: a wave walks its pixels four at
// a time, one load round trip per step -- with 128 pixels per workgroup the coarse levels (273 pixels: 3 workgroups per
// image, 8 dependent round trips each) cost the same 16 us as the finest; fewer pixels per workgroup until ~512
// workgroups exist cut the chain to 1-2 steps (the gate kernel merges the extra partials in passing).
static inline int se_pixels_per_block(int batch_size, int pixels)
{
    int pix = kSePix;
    while (pix > 16 && (int64_t)batch_size * ((pixels + pix - 1) / pix) < 512) pix >>= 1;
    return pix;
}

// The same for 16-bit maps with 256 channels (the real neck): half a wave per pixel, 8 channels (16 bytes) per lane, up to
// 16 pixels of a wave in flight at once and 5-step reductions that serve two pixels each.  The form below (a wave per pixel,
// 8-byte loads, four pixels per round trip, 6-step reductions) ran level 0 at 1.7 TB/s.
__global__ void __launch_bounds__(kBlock) se_context16_kernel(const bf16_t *y_, const float *w_mask, int N, int nblk,
                                                              float *partial, int pix_per_block)
{
    constexpr int C = 256;
    __shared__ float red[8][C + 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, c = (lane & 31) * 8;
    const int b = blockIdx.y, blk = blockIdx.x;
    const bf16_t *y = y_ + (int64_t)b * N * C;
    const float4 wa = *reinterpret_cast<const float4 *>(w_mask + c), wb = *reinterpret_cast<const float4 *>(w_mask + c + 4);
    float M = -INFINITY, Ssum = 0.f;
    float V[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) V[j] = 0.f;
    const int per_wave = pix_per_block / 4;
    const int w0 = blk * pix_per_block + wave * per_wave;
    const int w_end = min(N, w0 + per_wave);
    for (int pix = w0; pix < w_end; pix += 16) {
        uint4 v[8];
        float m[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = pix + 2 * i + half;
            v[i] = make_uint4(0u, 0u, 0u, 0u);
            if (q < w_end) v[i] = *reinterpret_cast<const uint4 *>(y + (int64_t)q * C + c);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
            m[i] = act_lo(v[i].x) * wa.x + act_hi(v[i].x) * wa.y + act_lo(v[i].y) * wa.z + act_hi(v[i].y) * wa.w +
                   act_lo(v[i].z) * wb.x + act_hi(v[i].z) * wb.y + act_lo(v[i].w) * wb.z + act_hi(v[i].w) * wb.w;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1)
#pragma unroll
            for (int i = 0; i < 8; ++i) m[i] += __shfl_xor(m[i], o, 64);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (pix + 2 * i + half < w_end) {  // uniform over a half wave
                const float nM = fmaxf(M, m[i]);
                const float sc = __expf(M - nM), e = __expf(m[i] - nM);  // first pixel: exp(-inf) = 0
                Ssum = Ssum * sc + e;
                V[0] = V[0] * sc + e * act_lo(v[i].x); V[1] = V[1] * sc + e * act_hi(v[i].x);
                V[2] = V[2] * sc + e * act_lo(v[i].y); V[3] = V[3] * sc + e * act_hi(v[i].y);
                V[4] = V[4] * sc + e * act_lo(v[i].z); V[5] = V[5] * sc + e * act_hi(v[i].z);
                V[6] = V[6] * sc + e * act_lo(v[i].w); V[7] = V[7] * sc + e * act_hi(v[i].w);
                M = nM;
            }
        }
    }
    const int slot = wave * 2 + half;
#pragma unroll
    for (int j = 0; j < 8; ++j) red[slot][c + j] = V[j];
    if ((lane & 31) == 0) {
        red[slot][C] = M;
        red[slot][C + 1] = Ssum;
    }
    __syncthreads();
    float gM = red[0][C];
#pragma unroll
    for (int i = 1; i < 8; ++i) gM = fmaxf(gM, red[i][C]);
    float *dst = partial + ((int64_t)b * nblk + blk) * (C + 2);
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = red[i][C] == -INFINITY ? 0.f : __expf(red[i][C] - gM);
    {
        float o = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) o = fmaf(red[i][tid], f[i], o);
        dst[tid] = o;
    }
    if (tid == 0) {
        float o = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) o = fmaf(red[i][C + 1], f[i], o);
        dst[C] = gM;
        dst[C + 1] = o;
    }
}

template <typename T>
__global__ void __launch_bounds__(kBlock) se_context_kernel(const void *y_, const float *w_mask, int N, int C,
                                                            int nblk, float *partial, int pix_per_block)
{
    __shared__ float red[4][256 + 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, blk = blockIdx.x;
    const T *y = reinterpret_cast<const T *>(y_) + (int64_t)b * N * C;
    const int c = lane * 4;
    const bool on = c < C;
    float4 wm = make_float4(0.f, 0.f, 0.f, 0.f);
    if (on) wm = *reinterpret_cast<const float4 *>(w_mask + c);
    float M = -INFINITY, Ssum = 0.f;
    float4 V = make_float4(0.f, 0.f, 0.f, 0.f);
    // a wave owns 32 consecutive pixels, four in flight at a time (one pixel per load round trip made every level
    // cost the same 18 us)
    const int w0 = blk * pix_per_block + wave * (pix_per_block / 4);
    const int w_end = min(N, w0 + pix_per_block / 4);
    for (int pix = w0; pix < w_end; pix += 4) {
        float4 v[4];
        float m[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (on && pix + i < w_end) v[i] = Store<T>::load4(y + (int64_t)(pix + i) * C + c);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) m[i] = v[i].x * wm.x + v[i].y * wm.y + v[i].z * wm.z + v[i].w * wm.w;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int i = 0; i < 4; ++i) m[i] += __shfl_xor(m[i], o, 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (pix + i < w_end) {  // wave-uniform
                const float nM = fmaxf(M, m[i]);
                const float sc = __expf(M - nM), e = __expf(m[i] - nM);  // first pixel: exp(-inf) = 0
                Ssum = Ssum * sc + e;
                V.x = V.x * sc + e * v[i].x; V.y = V.y * sc + e * v[i].y;
                V.z = V.z * sc + e * v[i].z; V.w = V.w * sc + e * v[i].w;
                M = nM;
            }
        }
    }
    if (on) {
        red[wave][c] = V.x; red[wave][c + 1] = V.y; red[wave][c + 2] = V.z; red[wave][c + 3] = V.w;
    }
    if (lane == 0) {
        red[wave][256] = M;
        red[wave][257] = Ssum;
    }
    __syncthreads();
    const float gM = fmaxf(fmaxf(red[0][256], red[1][256]), fmaxf(red[2][256], red[3][256]));
    float *dst = partial + ((int64_t)b * nblk + blk) * (C + 2);
    float f[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = red[i][256] == -INFINITY ? 0.f : __expf(red[i][256] - gM);
    if (tid < C) dst[tid] = red[0][tid] * f[0] + red[1][tid] * f[1] + red[2][tid] * f[2] + red[3][tid] * f[3];
    if (tid == 0) {
        dst[C] = gM;
        dst[C + 1] = red[0][257] * f[0] + red[1][257] * f[1] + red[2][257] * f[2] + red[3][257] * f[3];
    }
}

// gate[b, c] = sigmoid(W2 relu(W1 context[b])), W1 [R, C], W2 [C, R]  (basic.py:34-39, bias-free 1x1 convolutions).
// One 1024-thread workgroup per image: thread (ch = tid % 256, grp = tid / 256) adds every fourth partial of channel ch.
constexpr int kGateThreads = 1024;

__global__ void __launch_bounds__(kGateThreads) se_gate_kernel(const float *partial, int nblk, int C, int R,
                                                               const float *w1, const float *w2, float *gate)
{
    // One workgroup per image merges up to ~260 partials of 258 floats and runs the two tiny layers.  The work is nothing;
    // what the first two forms of this kernel paid for (10-14 us, 18 times a step) was a CHAIN of dependent global-memory
    // phases: partial maxima -> their factors -> the weighted columns -> W1 -> W2, each a round trip to data the context
    // launch had just written from other XCDs.  Round 5: everything a thread will ever read is requested in the first
    // instructions -- its W1 slice, its partial's (max, sum) pair, its 34 column pairs of the first 272 partials -- and
    // only then do the reductions start (wave shuffles + one LDS hop); W2 is requested while the context vector is being
    // reduced.  Partials beyond 272 (other batch sizes) take further passes of the same loop.
    constexpr int kGroups = 8, kWaves = kGateThreads / 64, kRowsPerThread = 34, kPass = kGroups * kRowsPerThread;   // 272
    __shared__ float ctx[256];
    __shared__ float hid[64];
    __shared__ float wf[kPass];
    __shared__ float red[kGroups][256];
    __shared__ float wred[kWaves];
    __shared__ float sred[kWaves];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const int pair = tid & 127, grp = tid >> 7;
    const int r1 = tid >> 4, c1 = tid & 15;
    const int64_t ld = C + 2;
    const float *part = partial + (int64_t)b * nblk * ld;
    const bool has_cols = 2 * pair < C;

    // ---- everything this thread reads, requested at once ----
    float w1v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int c = c1 + 16 * j;
        w1v[j] = (r1 < R && c < C) ? w1[(int64_t)r1 * C + c] : 0.f;
    }
    float2 ms = make_float2(-INFINITY, 0.f);                                   // (max, sum) of partial `tid`
    if (tid < nblk) ms = *reinterpret_cast<const float2 *>(part + tid * ld + C);
    float2 cv[kRowsPerThread];
    const int n0 = min(kPass, nblk);
#pragma unroll
    for (int j = 0; j < kRowsPerThread; ++j) {
        const int k = grp + kGroups * j;
        cv[j] = make_float2(0.f, 0.f);
        if (has_cols && k < n0) cv[j] = *reinterpret_cast<const float2 *>(part + k * ld + 2 * pair);
    }

    // ---- global max ----
    float m = ms.x;
    for (int i = tid + kGateThreads; i < nblk; i += kGateThreads) m = fmaxf(m, part[i * ld + C]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) wred[wave] = m;
    __syncthreads();
    float gM = wred[0];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) gM = fmaxf(gM, wred[w]);

    // ---- factors exp(max_i - max) and weighted channel sums, kPass partials at a time (the first from registers) ----
    float ssum = 0.f, v0 = 0.f, v1 = 0.f;
    if (tid < kPass) {
        const float f = (tid < nblk && ms.x != -INFINITY) ? __expf(ms.x - gM) : 0.f;
        ssum = ms.y * f;
        wf[tid] = f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kRowsPerThread; ++j) {
        const float f = wf[grp + kGroups * j];
        v0 = fmaf(cv[j].x, f, v0);
        v1 = fmaf(cv[j].y, f, v1);
    }
    for (int base = kPass; base < nblk; base += kPass) {
        __syncthreads();                                                       // wf free again
        if (tid < kPass) {
            const int i = base + tid;
            float f = 0.f;
            if (i < nblk) {
                const float2 q = *reinterpret_cast<const float2 *>(part + i * ld + C);
                if (q.x != -INFINITY) {
                    f = __expf(q.x - gM);
                    ssum = fmaf(q.y, f, ssum);
                }
            }
            wf[tid] = f;
        }
        __syncthreads();
        const int n = min(kPass, nblk - base);
        if (has_cols) {
            const float *col = part + (int64_t)base * ld + 2 * pair;
#pragma unroll 8
            for (int k = grp; k < n; k += kGroups) {
                const float2 c = *reinterpret_cast<const float2 *>(col + k * ld);
                const float f = wf[k];
                v0 = fmaf(c.x, f, v0);
                v1 = fmaf(c.y, f, v1);
            }
        }
    }
    // W2's first 16 columns of my row, requested now: they arrive while the context vector is reduced
    float w2v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) w2v[j] = (tid < C && j < R) ? w2[(int64_t)tid * R + j] : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ssum += __shfl_xor(ssum, o);
    if (lane == 0) sred[wave] = ssum;
    red[grp][2 * pair] = v0;
    red[grp][2 * pair + 1] = v1;
    __syncthreads();
    if (tid < C) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) tot += sred[w];
        float c = 0.f;
#pragma unroll
        for (int g = 0; g < kGroups; ++g) c += red[g][tid];
        ctx[tid] = c * (1.0f / tot);
    }
    __syncthreads();
    // hidden layer: 16 lanes per row of W1 (64 rows at once: R <= 64)
    {
        float h = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) h = fmaf(w1v[j], ctx[c1 + 16 * j], h);
        h += __shfl_xor(h, 8, 16);
        h += __shfl_xor(h, 4, 16);
        h += __shfl_xor(h, 2, 16);
        h += __shfl_xor(h, 1, 16);
        if (c1 == 0 && r1 < R) hid[r1] = fmaxf(h, 0.f);
    }
    __syncthreads();
    if (tid < C) {
        float o = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) o = fmaf(w2v[j], j < R ? hid[j] : 0.f, o);
        for (int j = 16; j < R; ++j) o = fmaf(w2[(int64_t)tid * R + j], hid[j], o);
        gate[(int64_t)b * C + tid] = 1.0f / (1.0f + __expf(-o));
    }
}

// out = gate[b, :] * y + x (+ x2): the gated activation plus the block's shortcut (repnet.py:64), and for the last
// block of a CSP layer the layer's second branch as well (repnet.py:121)
struct ApplyArgs {
    const void *y, *x, *x2;
    const float *gate;
    void *out;
    int64_t rows;
    int N, C, ldx, ldx2;
};

template <typename T>
__global__ void __launch_bounds__(kBlock) se_apply_kernel(ApplyArgs p)
{
    const int c4n = p.C / 4;
    const int64_t n = p.rows * c4n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % c4n) * 4;
        const int64_t row = e / c4n;
        const int64_t b = row / p.N;
        const float4 g = *reinterpret_cast<const float4 *>(p.gate + b * p.C + c);
        const float4 y = Store<T>::load4(reinterpret_cast<const T *>(p.y) + row * p.C + c);
        const float4 x = Store<T>::load4(reinterpret_cast<const T *>(p.x) + row * p.ldx + c);
        float4 v = make_float4(fmaf(g.x, y.x, x.x), fmaf(g.y, y.y, x.y), fmaf(g.z, y.z, x.z), fmaf(g.w, y.w, x.w));
        if (p.x2) {
            const float4 s = Store<T>::load4(reinterpret_cast<const T *>(p.x2) + row * p.ldx2 + c);
            v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w;
        }
        Store<T>::store4(reinterpret_cast<T *>(p.out) + row * p.C + c, v);
    }
}

inline unsigned grid_for(int64_t n)
{
    const int64_t blocks = (n + kBlock - 1) / kBlock;
    return (unsigned)(blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks));
}

}  // namespace
}  // namespace sdetr

using namespace sdetr;

extern "C" int sdetr_neck_conv3x3(sdetr_stream_t stream, const void *x, int dtype, int batch_size, int height,
                                  int width, int x_row_stride, const float *weight, const float *bias, int groups,
                                  int in_per_group, int out_per_group, int stride, int activation, void *out)
{
    if (batch_size < 0 || height <= 0 || width <= 0 || groups <= 0 || in_per_group <= 0 || out_per_group <= 0)
        return fail("neck_conv3x3: bad sizes");
    if ((in_per_group % 4) || (out_per_group % 4) || (x_row_stride % 4) || x_row_stride < groups * in_per_group)
        return fail("neck_conv3x3: channels per group and the row stride must be multiples of 4");
    if (stride != 1 && stride != 2) return fail("neck_conv3x3: stride %d (1 or 2)", stride);
    if (dtype != SDETR_F32 && dtype != kActCode) return fail("neck_conv3x3: bad dtype %d", dtype);
    if (activation < 0 || activation > 1) return fail("neck_conv3x3: bad activation %d", activation);
    if (batch_size == 0) return 0;
    if (!x || !weight || !out) return fail("neck_conv3x3: null pointer");
    if (batch_size > 65535) return fail("neck_conv3x3: batch too large");
    ConvArgs a;
    a.x = x; a.w = weight; a.bias = bias; a.out = out;
    a.B = batch_size; a.H = height; a.W = width;
    a.Ho = (height - 1) / stride + 1;  // (H + 2 - 3) / stride + 1
    a.Wo = (width - 1) / stride + 1;
    a.ldx = x_row_stride; a.G = groups; a.CiG = in_per_group; a.CoG = out_per_group; a.act = activation;
    a.tiles_x = (a.Wo + kTileW - 1) / kTileW;
    const int tiles_y = (a.Ho + kTileH - 1) / kTileH;
    const int cblocks = (out_per_group + kCB - 1) / kCB;
    if ((int64_t)groups * cblocks > 65535) return fail("neck_conv3x3: too many channel blocks");
    const dim3 grid((unsigned)(a.tiles_x * tiles_y), (unsigned)(groups * cblocks), (unsigned)batch_size);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SDETR_F32) {
        if (stride == 1) hipLaunchKernelGGL((conv3x3_tokens_kernel<float, 1>), grid, dim3(kBlock), 0, s, a);
        else hipLaunchKernelGGL((conv3x3_tokens_kernel<float, 2>), grid, dim3(kBlock), 0, s, a);
    } else {
        if (stride == 1) hipLaunchKernelGGL((conv3x3_tokens_kernel<bf16_t, 1>), grid, dim3(kBlock), 0, s, a);
        else hipLaunchKernelGGL((conv3x3_tokens_kernel<bf16_t, 2>), grid, dim3(kBlock), 0, s, a);
    }
    return check_launch("neck_conv3x3");
}

extern "C" int64_t sdetr_neck_conv3x3_packed_bytes(int groups, int in_per_group, int out_per_group)
{
    if (groups <= 0 || in_per_group <= 0 || out_per_group <= 0 || (in_per_group % 16) || (out_per_group % 64)) return 0;
    return (int64_t)groups * 9 * in_per_group * out_per_group * 2;
}

extern "C" int sdetr_neck_pack_conv3x3_bf16(sdetr_stream_t stream, const float *weight, int groups, int in_per_group,
                                            int out_per_group, void *packed)
{
    if (sdetr_neck_conv3x3_packed_bytes(groups, in_per_group, out_per_group) == 0)
        return fail("neck_pack_conv3x3: needs in_per_group %% 16 == 0 and out_per_group %% 64 == 0 (got %d, %d)",
                    in_per_group, out_per_group);
    if (!weight || !packed) return fail("neck_pack_conv3x3: null pointer");
    const int64_t n = (int64_t)groups * 9 * in_per_group * out_per_group;
    hipLaunchKernelGGL(conv3x3_pack_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, weight, groups,
                       in_per_group, out_per_group, reinterpret_cast<bf16_t *>(packed));
    return check_launch("neck_pack_conv3x3");
}

// SDETR_CONV_LDS=0 keeps the neck's 64 -> 64 stride-1 blocks on the fragment-streaming kernel (benchmarks/conv_split_ab.py).
static bool conv_lds_form()
{
    static const bool on = [] {
        const char *e = ab_env("SDETR_CONV_LDS");
        return !(e && e[0] == '0');
    }();
    return on;
}

// How many waves share a 64-pixel x 64-channel tile of the MFMA convolution (1 or 3; see the kernel's header).
// SDETR_CONV_SPLIT pins it (benchmarks/conv_split_ab.py).
static int conv_tap_split(int64_t wave_tiles)
{
    static const int pinned = [] {
        const char *e = ab_env("SDETR_CONV_SPLIT");
        const int v = e ? atoi(e) : 0;
        return (v == 1 || v == 3) ? v : 0;
    }();
    if (pinned) return pinned;
    (void)wave_tiles;
    return 3;
}

extern "C" int sdetr_neck_conv3x3_mfma_bf16(sdetr_stream_t stream, const void *x, int batch_size, int height, int width,
                                            int x_row_stride, const void *packed_weight, const float *bias, int groups,
                                            int in_per_group, int out_per_group, int stride, int activation, void *out)
{
    if (batch_size < 0 || height <= 0 || width <= 0) return fail("neck_conv3x3_mfma: bad sizes");
    if (sdetr_neck_conv3x3_packed_bytes(groups, in_per_group, out_per_group) == 0)
        return fail("neck_conv3x3_mfma: needs in_per_group %% 16 == 0 and out_per_group %% 64 == 0 (got %d, %d)",
                    in_per_group, out_per_group);
    if ((x_row_stride % 8) || x_row_stride < groups * in_per_group)
        return fail("neck_conv3x3_mfma: the row stride must be a multiple of 8 elements");
    if (stride != 1 && stride != 2) return fail("neck_conv3x3_mfma: stride %d (1 or 2)", stride);
    if (activation < 0 || activation > 1) return fail("neck_conv3x3_mfma: bad activation %d", activation);
    if (batch_size == 0) return 0;
    if (!x || !packed_weight || !out) return fail("neck_conv3x3_mfma: null pointer");
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(packed_weight) & 15))
        return fail("neck_conv3x3_mfma: x and the packed weights must be 16-byte aligned");
    ConvMfmaArgs a;
    a.x = reinterpret_cast<const bf16_t *>(x);
    a.pw = reinterpret_cast<const char *>(packed_weight);
    a.bias = bias;
    a.out = reinterpret_cast<bf16_t *>(out);
    a.H = height; a.W = width; a.S = stride;
    a.Ho = (height - 1) / stride + 1;
    a.Wo = (width - 1) / stride + 1;
    a.P = (int64_t)batch_size * a.Ho * a.Wo;
    a.ldx = x_row_stride; a.G = groups; a.CiG = in_per_group; a.CoG = out_per_group; a.act = activation;
    if (stride == 1 && in_per_group == 64 && out_per_group == 64 && conv_lds_form()) {
        const int row_tiles = (height + kLdsConvRows - 1) / kLdsConvRows, col_tiles = (width + kLdsConvCols - 1) / kLdsConvCols;
        const int64_t nt = (int64_t)batch_size * row_tiles * col_tiles;
        if (nt > 0x7fffffffLL || groups > 65535) return fail("neck_conv3x3_mfma: grid too large");
        static DeviceOnce lds_once;
        allow_dynamic_lds(conv3x3_lds_kernel, lds_once, kLdsConvBytes);
        hipLaunchKernelGGL(conv3x3_lds_kernel, dim3((unsigned)nt, (unsigned)groups), dim3(64 * kLdsConvRows), kLdsConvBytes,
                           (hipStream_t)stream, a, row_tiles, col_tiles);
        return check_launch("neck_conv3x3_mfma");
    }
    const int64_t gy = (int64_t)groups * (out_per_group / 64);
    const int64_t tiles = (a.P + 63) / 64;
    const int split = conv_tap_split(tiles * gy);
    const int64_t gx = split == 1 ? (a.P + 255) / 256 : tiles;
    if (gx > 0x7fffffffLL || gy > 65535) return fail("neck_conv3x3_mfma: grid too large");
    const dim3 grid((unsigned)gx, (unsigned)gy);
    hipStream_t s = (hipStream_t)stream;
    const bool c4 = in_per_group % 64 == 0;
    if (split == 3) {
        if (c4) hipLaunchKernelGGL((conv3x3_mfma_kernel<true, 3>), grid, dim3(192), 2 * 16384, s, a);
        else hipLaunchKernelGGL((conv3x3_mfma_kernel<false, 3>), grid, dim3(192), 2 * 16384, s, a);
    } else {
        if (c4) hipLaunchKernelGGL((conv3x3_mfma_kernel<true, 1>), grid, dim3(kBlock), 0, s, a);
        else hipLaunchKernelGGL((conv3x3_mfma_kernel<false, 1>), grid, dim3(kBlock), 0, s, a);
    }
    return check_launch("neck_conv3x3_mfma");
}

extern "C" int sdetr_neck_combine(sdetr_stream_t stream, const void *a, int a_row_stride, const void *up,
                                  int up_row_stride, int up_height, int up_width, const float *bias, int dtype,
                                  int batch_size, int height, int width, int channels, int activation, void *out,
                                  int out_row_stride)
{
    if (batch_size < 0 || height <= 0 || width <= 0 || channels <= 0 || (channels % 4))
        return fail("neck_combine: bad sizes");
    if ((a_row_stride % 4) || (out_row_stride % 4) || a_row_stride < channels || out_row_stride < channels)
        return fail("neck_combine: row strides must be multiples of 4 and >= channels");
    if (up && (up_height <= 0 || up_width <= 0 || (up_row_stride % 4) || up_row_stride < channels))
        return fail("neck_combine: bad up-sampling source");
    if (dtype != SDETR_F32 && dtype != kActCode) return fail("neck_combine: bad dtype %d", dtype);
    if (activation < 0 || activation > 1) return fail("neck_combine: bad activation %d", activation);
    if (batch_size == 0) return 0;
    if (!a || !out) return fail("neck_combine: null pointer");
    CombineArgs p;
    p.a = a; p.up = up; p.bias = bias; p.out = out;
    p.rows = (int64_t)batch_size * height * width;
    p.H = height; p.W = width; p.Hs = up ? up_height : 1; p.Ws = up ? up_width : 1; p.C = channels;
    p.lda = a_row_stride; p.ldup = up_row_stride; p.ldo = out_row_stride; p.act = activation;
    hipStream_t s = (hipStream_t)stream;
    const auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (dtype != SDETR_F32 && channels % 8 == 0 && a_row_stride % 8 == 0 && out_row_stride % 8 == 0 &&
        (!up || up_row_stride % 8 == 0) && al16(a) && al16(out) && (!up || al16(up)) && (!bias || al16(bias)) &&
        p.rows * (channels / 8) < 0x7fffffffLL) {
        hipLaunchKernelGGL(neck_combine8_kernel, dim3(grid_for(p.rows * (channels / 8))), dim3(kBlock), 0, s, p);
        return check_launch("neck_combine");
    }
    const unsigned grid = grid_for(p.rows * (channels / 4));
    if (dtype == SDETR_F32) hipLaunchKernelGGL((neck_combine_kernel<float>), dim3(grid), dim3(kBlock), 0, s, p);
    else hipLaunchKernelGGL((neck_combine_kernel<bf16_t>), dim3(grid), dim3(kBlock), 0, s, p);
    return check_launch("neck_combine");
}

extern "C" int64_t sdetr_neck_gate_workspace_bytes(int batch_size, int pixels, int channels)
{
    if (batch_size <= 0 || pixels <= 0 || channels <= 0) return 0;
    const int pix = se_pixels_per_block(batch_size, pixels);
    const int64_t nblk = ((int64_t)pixels + pix - 1) / pix;
    return (int64_t)batch_size * nblk * ((int64_t)channels + 2) * (int64_t)sizeof(float);
}

extern "C" int sdetr_neck_gate_shortcut(sdetr_stream_t stream, const void *y, int dtype, int batch_size, int pixels,
                                        int channels, const float *mask_weight, const float *squeeze_weight,
                                        const float *excite_weight, int hidden, const void *shortcut,
                                        int shortcut_row_stride, const void *shortcut2, int shortcut2_row_stride,
                                        void *workspace, int64_t workspace_bytes, float *gate, void *out)
{
    if (batch_size < 0 || pixels <= 0 || channels <= 0 || (channels % 4) || channels > 256)
        return fail("neck_gate_shortcut: channels must be a multiple of 4, at most 256 (got %d)", channels);
    if (hidden <= 0 || hidden > 64) return fail("neck_gate_shortcut: hidden width %d (1..64)", hidden);
    if (dtype != SDETR_F32 && dtype != kActCode) return fail("neck_gate_shortcut: bad dtype %d", dtype);
    if ((shortcut_row_stride % 4) || shortcut_row_stride < channels) return fail("neck_gate_shortcut: bad shortcut stride");
    if (shortcut2 && ((shortcut2_row_stride % 4) || shortcut2_row_stride < channels))
        return fail("neck_gate_shortcut: bad second shortcut stride");
    if (batch_size == 0) return 0;
    if (!y || !mask_weight || !squeeze_weight || !excite_weight || !shortcut || !gate || !out)
        return fail("neck_gate_shortcut: null pointer");
    if (batch_size > 65535) return fail("neck_gate_shortcut: batch too large");
    const int64_t need = sdetr_neck_gate_workspace_bytes(batch_size, pixels, channels);
    if (!workspace || workspace_bytes < need)
        return fail("neck_gate_shortcut: needs %lld bytes of workspace, got %lld", (long long)need, (long long)workspace_bytes);
    const int pix = se_pixels_per_block(batch_size, pixels);
    const int nblk = (pixels + pix - 1) / pix;
    hipStream_t s = (hipStream_t)stream;
    float *partial = reinterpret_cast<float *>(workspace);
    if (dtype == SDETR_F32)
        hipLaunchKernelGGL((se_context_kernel<float>), dim3((unsigned)nblk, (unsigned)batch_size), dim3(kBlock), 0, s, y,
                           mask_weight, pixels, channels, nblk, partial, pix);
    else if (channels == 256 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 && (reinterpret_cast<uintptr_t>(mask_weight) & 15) == 0)
        hipLaunchKernelGGL(se_context16_kernel, dim3((unsigned)nblk, (unsigned)batch_size), dim3(kBlock), 0, s,
                           reinterpret_cast<const bf16_t *>(y), mask_weight, pixels, nblk, partial, pix);
    else
        hipLaunchKernelGGL((se_context_kernel<bf16_t>), dim3((unsigned)nblk, (unsigned)batch_size), dim3(kBlock), 0, s, y,
                           mask_weight, pixels, channels, nblk, partial, pix);
    int rc = check_launch("neck_gate_shortcut(context)");
    if (rc) return rc;
    hipLaunchKernelGGL(se_gate_kernel, dim3((unsigned)batch_size), dim3(kGateThreads), 0, s, partial, nblk, channels, hidden,
                       squeeze_weight, excite_weight, gate);
    rc = check_launch("neck_gate_shortcut(gate)");
    if (rc) return rc;
    ApplyArgs p;
    p.y = y; p.x = shortcut; p.x2 = shortcut2; p.gate = gate; p.out = out;
    p.rows = (int64_t)batch_size * pixels; p.N = pixels; p.C = channels;
    p.ldx = shortcut_row_stride; p.ldx2 = shortcut2_row_stride;
    const unsigned grid = grid_for(p.rows * (channels / 4));
    if (dtype == SDETR_F32) hipLaunchKernelGGL((se_apply_kernel<float>), dim3(grid), dim3(kBlock), 0, s, p);
    else hipLaunchKernelGGL((se_apply_kernel<bf16_t>), dim3(grid), dim3(kBlock), 0, s, p);
    return check_launch("neck_gate_shortcut(apply)");
}
