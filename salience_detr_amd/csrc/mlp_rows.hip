// The decoder's small MLPs in one launch each (models/bricks/basic.py:6-26 -- Linear + ReLU chains):
//   ref_point_head  512 -> 256 -> 256          (models/bricks/salience_transformer.py:643-644, once per decoder layer)
//   bbox_head[i]    256 -> 256 -> 256 -> 4     (:659-668, on the normed AND the raw queries of a layer: two row sources)
//   encoder_bbox_head (the same shape, :206)
// At 2 x 900 queries these are 1 800-3 600 rows: every library GEMM of the chain is a launch of its own whose run time is
// its latency (6-9 us each, plus a 5 us torch.stack for the two row sources) -- 40 us per decoder layer for 0.8 GFLOP.
//
// One workgroup of 8 waves owns 32 rows; wave w owns output features 32 w .. 32 w + 31 of every hidden layer.
//   * The FIRST instructions request everything the workgroup will ever read from memory: a wave's A-operand fragments
//     of layer 1 AND layer 2 (AND 3), its biases, and its share of the 32 input rows -> one round trip.  The weights are
//     the lane-ordered 1 KB fragments of sdetr_linear_pack_bf16 (include/salience_hip.h (8)): a workgroup pulls ALL of
//     them through its CU's L1 (256-384 KB), and fragment-shaped loads from the row-major matrices -- 32 rows x 32 bytes
//     per instruction -- made that 12-15 us per launch (first form of this kernel); whole 1 KB pieces are 64 bytes/clk.
//   * Y^T = W X^T with v_mfma_f32_32x32x16 (lane = row, registers = features); the B operand comes from LDS: the input
//     rows, then each hidden layer's ReLU output, row-major with a 16-byte pad per row (a ds_read_b128 group = 16 rows,
//     4 words apart mod 64: conflict-free).
//   * The last layer of a 3-layer chain (<= 32 outputs) is one wave's work.
#include "common.h"

namespace sdetr {

typedef float ml_f32x16_t __attribute__((ext_vector_type(16)));

struct MlpArgs {
    const bf16_t *xa, *xb;     // rows [0, rows_a) from xa, [rows_a, rows) from xb (row-major, K1 elements each)
    int rows_a, rows;
    const char *w1;            // packed [K1 / 256][8 tiles][16 k-steps][64 lanes][16 bytes]
    const char *w2;            // packed [8 tiles][16][64][16]
    const char *w3;            // packed, tile 0 used (rows past n3 are zero), or NULL (two layers)
    const float *b1, *b2, *b3; // fp32, zero-padded to the packed tiles
    int n3;
    bf16_t *out;               // [rows, 256] (two layers) or [rows, n3]
    int64_t ldo;
    // SINE form (ref_point_head): the 512 input features of a row are the sine embedding of its reference box, made in
    // the tile fill (models/bricks/salience_transformer.py:642-643, models/bricks/position_encoding.py:105-132)
    const float *ref;          // [rows, 4] boxes (cx, cy, w, h)
    const float *vr;           // [B, L, 2] valid ratios
    int Nq, L;
    float temperature;
    float *ref_in;             // [rows, L, 4] = box * (rw, rh, rw, rh) per level, or NULL
};

__device__ __forceinline__ int ml_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

constexpr int kMlpRows = 32, kMlpHidden = 256, kMlpHPitch = kMlpHidden * 2 + 16;

template <int K1, int NL, bool SINE = false>
__global__ void __launch_bounds__(512) mlp_rows_kernel(MlpArgs p)
{
    static_assert(!SINE || (K1 == 512 && NL == 2), "the sine prologue belongs to ref_point_head");
    constexpr int kXPitch = K1 * 2 + 16;
    extern __shared__ __align__(16) unsigned char mlp_lds[];
    unsigned char *xs = mlp_lds;                              // [32][kXPitch]   input rows
    unsigned char *h1 = xs + kMlpRows * kXPitch;              // [32][kMlpHPitch] layer-1 output
    unsigned char *h2 = h1 + kMlpRows * kMlpHPitch;           // [32][kMlpHPitch] layer-2 output (three layers)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * kMlpRows;

    // ---- every global load of the kernel, requested before anything waits ----
    constexpr int KS1 = K1 / 16;
    uint4 a1[KS1], a2[16], a3[NL == 3 ? 16 : 1];
    constexpr int kHalfBytes = 8 * 16 * 1024;                           // one packed 256 x 256 block
#pragma unroll
    for (int j = 0; j < KS1; ++j)
        a1[j] = *reinterpret_cast<const uint4 *>(p.w1 + (j / 16) * kHalfBytes + ((wave * 16 + (j % 16)) * 64 + lane) * 16);
    constexpr int kPieces = kMlpRows * K1 / 8, kPer = kPieces / 512;      // 16-byte pieces of the input rows: 2 or 4 per thread
    uint4 xv[kPer];
    float coord[kPer];         // (SINE) the box coordinate behind this thread's piece of each of its rows
    if constexpr (SINE) {
        // piece pc = tid % 64 of rows tid / 64 + 8 i: features 8 pc .. 8 pc + 7 = pairs 4 (pc % 16) .. + 3 of block pc / 16;
        // blocks are emitted in (y, x, w, h) order; every coordinate is scaled by the level-0 valid ratio (w, h, w, h)
        const int blk = (tid & 63) >> 4;
        const int c = blk == 0 ? 1 : (blk == 1 ? 0 : blk);
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const int r = min(row0 + (tid >> 6) + 8 * i, p.rows - 1);
            coord[i] = p.ref[(int64_t)r * 4 + c] * p.vr[(int64_t)(r / p.Nq) * p.L * 2 + (c & 1)];
        }
        if (p.ref_in && tid < kMlpRows * p.L) {
            const int r = row0 + tid / p.L, l = tid - (tid / p.L) * p.L;
            if (r < p.rows) {
                const float4 b4 = reinterpret_cast<const float4 *>(p.ref)[r];
                const float rw = p.vr[((int64_t)(r / p.Nq) * p.L + l) * 2], rh = p.vr[((int64_t)(r / p.Nq) * p.L + l) * 2 + 1];
                reinterpret_cast<float4 *>(p.ref_in)[(int64_t)r * p.L + l] = make_float4(b4.x * rw, b4.y * rh, b4.z * rw, b4.w * rh);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const int e = i * 512 + tid;
            const int r = min(row0 + e / (K1 / 8), p.rows - 1), pc = e % (K1 / 8);
            const bf16_t *src = r < p.rows_a ? p.xa + (int64_t)r * K1 : p.xb + (int64_t)(r - p.rows_a) * K1;
            xv[i] = *reinterpret_cast<const uint4 *>(src + pc * 8);
        }
    }
    float bias1[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 bv = *reinterpret_cast<const float4 *>(p.b1 + wave * 32 + 8 * g + 4 * h);
        bias1[4 * g] = bv.x; bias1[4 * g + 1] = bv.y; bias1[4 * g + 2] = bv.z; bias1[4 * g + 3] = bv.w;
    }
    if constexpr (!SINE) {
#pragma unroll
        for (int j = 0; j < 16; ++j) a2[j] = *reinterpret_cast<const uint4 *>(p.w2 + ((wave * 16 + j) * 64 + lane) * 16);
    }
    if constexpr (NL == 3) {
        if (wave == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) a3[j] = *reinterpret_cast<const uint4 *>(p.w3 + (j * 64 + lane) * 16);
        }
    }
    if constexpr (SINE) {
        // (the arithmetic of query_sine_embed_kernel, csrc/decoder_ops.hip: a = c * 2 pi / T^(2p/F), (sin a, cos a))
        float inv_dim[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            inv_dim[q] = powf(p.temperature, (float)(2 * (4 * (tid & 15) + q)) / (float)(K1 / 4));
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            uint32_t w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float sn, cs;
                sincosf(coord[i] * 6.283185307179586f / inv_dim[q], &sn, &cs);
                w[q] = pack_act2(sn, cs);
            }
            xv[i] = make_uint4(w[0], w[1], w[2], w[3]);
        }
        // (layer 2's fragments only now: with them in flight the sine arithmetic spilled 36 registers)
#pragma unroll
        for (int j = 0; j < 16; ++j) a2[j] = *reinterpret_cast<const uint4 *>(p.w2 + ((wave * 16 + j) * 64 + lane) * 16);
    }
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
        const int e = i * 512 + tid;
        *reinterpret_cast<uint4 *>(xs + (e / (K1 / 8)) * kXPitch + (e % (K1 / 8)) * 16) = xv[i];
    }
    __syncthreads();

    // ---- layer 1 ----
    ml_f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bias1[r];
    {
        const unsigned char *bp = xs + t * kXPitch + h * 16;
#pragma unroll
        for (int j = 0; j < KS1; ++j) acc = mfma_act_32x32x16(a1[j], *reinterpret_cast<const uint4 *>(bp + j * 32), acc);
    }
    // second layer's bias: requested now (layer 1's fragments are dead), back by the time the barrier is passed
    float bias2[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 bv = *reinterpret_cast<const float4 *>(p.b2 + wave * 32 + 8 * g + 4 * h);
        bias2[4 * g] = bv.x; bias2[4 * g + 1] = bv.y; bias2[4 * g + 2] = bv.z; bias2[4 * g + 3] = bv.w;
    }
    // lane (t, h), register r: feature 32 w + ml_row(r, h) of row t; four consecutive features per group of registers
#pragma unroll
    for (int g = 0; g < 4; ++g)
        *reinterpret_cast<uint2 *>(h1 + t * kMlpHPitch + (wave * 32 + 8 * g + 4 * h) * 2) =
            make_uint2(pack_act2(fmaxf(acc[4 * g], 0.f), fmaxf(acc[4 * g + 1], 0.f)),
                       pack_act2(fmaxf(acc[4 * g + 2], 0.f), fmaxf(acc[4 * g + 3], 0.f)));
    __syncthreads();

    // ---- layer 2 ----
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bias2[r];
    {
        const unsigned char *bp = h1 + t * kMlpHPitch + h * 16;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc = mfma_act_32x32x16(a2[j], *reinterpret_cast<const uint4 *>(bp + j * 32), acc);
    }
    const int row = row0 + t;
    if constexpr (NL == 2) {
        if (row < p.rows) {
            bf16_t *orow = p.out + (int64_t)row * p.ldo + wave * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<uint2 *>(orow + 8 * g) =
                    make_uint2(pack_act2(acc[4 * g], acc[4 * g + 1]), pack_act2(acc[4 * g + 2], acc[4 * g + 3]));
        }
    } else {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<uint2 *>(h2 + t * kMlpHPitch + (wave * 32 + 8 * g + 4 * h) * 2) =
                make_uint2(pack_act2(fmaxf(acc[4 * g], 0.f), fmaxf(acc[4 * g + 1], 0.f)),
                           pack_act2(fmaxf(acc[4 * g + 2], 0.f), fmaxf(acc[4 * g + 3], 0.f)));
        __syncthreads();
        if (wave != 0) return;
        // ---- layer 3: n3 <= 32 outputs, one tile ----
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const unsigned char *bp = h2 + t * kMlpHPitch + h * 16;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc = mfma_act_32x32x16(a3[j], *reinterpret_cast<const uint4 *>(bp + j * 32), acc);
        if (row < p.rows) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = ml_row(r, h);
                if (f < p.n3) {
                    const float v = acc[r] + p.b3[f];
                    p.out[(int64_t)row * p.ldo + f] = (bf16_t)(pack_act2(v, 0.f) & 0xffffu);
                }
            }
        }
    }
}

// ---- one Linear on rows that may carry a position addend for the first `pos_features` outputs ----------------------
// out[r, f] = (x[r] + (f < pos_features ? pos[r] : 0)) . W[f] + b[f]:  the decoder layer's self-attention in-projection
// (q | k from query + query_pos, v from query: nn.MultiheadAttention with q = k = x + pos, v = x,
// models/bricks/salience_transformer.py:565-570) as ONE launch instead of an elementwise add and two library GEMMs, and the
// cross-attention's offset | weight projection of query + query_pos (models/bricks/ms_deform_attn.py:322-349) likewise.
// Same scheme as above: 32 rows per workgroup, 8 waves, wave w owns the feature tiles w, w + 8, w + 16 (<= 768 features),
// every packed weight fragment requested up front, both B operands (x and x + pos, summed in fp32 and rounded once like
// the elementwise add) from LDS.
struct RowsLinearArgs {
    const bf16_t *x, *pos;     // [rows, 256] each; pos may be NULL
    int rows, n, pos_features; // n output features, the first pos_features (a multiple of 32) see x + pos
    const char *w;             // packed tiles [ceil(n / 128) * 4][16][64][16 bytes]
    const float *b;            // fp32, zero-padded to the packed tiles
    bf16_t *out;
    int64_t ldo;
};

template <int TILES>   // feature tiles per wave: 1, 2 or 3
__global__ void __launch_bounds__(512) rows_linear_kernel(RowsLinearArgs p)
{
    extern __shared__ __align__(16) unsigned char mlp_lds[];
    unsigned char *xs = mlp_lds;                              // [32][kMlpHPitch]  x
    unsigned char *xp = xs + kMlpRows * kMlpHPitch;           // [32][kMlpHPitch]  x + pos
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * kMlpRows;
    const int ntiles = (p.n + 31) / 32;

    uint4 a[TILES][16];
#pragma unroll
    for (int i = 0; i < TILES; ++i) {
        const int tile = wave + 8 * i;
        if (tile < ntiles) {                                  // (wave-uniform)
#pragma unroll
            for (int j = 0; j < 16; ++j) a[i][j] = *reinterpret_cast<const uint4 *>(p.w + ((tile * 16 + j) * 64 + lane) * 16);
        }
    }
    // the input rows: 1024 16-byte pieces, two per thread (and the same of pos)
    uint4 xv[2], pv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = i * 512 + tid;
        const int r = min(row0 + (e >> 5), p.rows - 1), pc = e & 31;
        xv[i] = *reinterpret_cast<const uint4 *>(p.x + (int64_t)r * kMlpHidden + pc * 8);
        pv[i] = make_uint4(0u, 0u, 0u, 0u);
        if (p.pos) pv[i] = *reinterpret_cast<const uint4 *>(p.pos + (int64_t)r * kMlpHidden + pc * 8);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = i * 512 + tid;
        const int off = (e >> 5) * kMlpHPitch + (e & 31) * 16;
        *reinterpret_cast<uint4 *>(xs + off) = xv[i];
        const uint32_t xw[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w}, pw[4] = {pv[i].x, pv[i].y, pv[i].z, pv[i].w};
        uint32_t sw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) sw[q] = pack_act2(act_lo(xw[q]) + act_lo(pw[q]), act_hi(xw[q]) + act_hi(pw[q]));
        *reinterpret_cast<uint4 *>(xp + off) = make_uint4(sw[0], sw[1], sw[2], sw[3]);
    }
    __syncthreads();

    const int row = row0 + t;
#pragma unroll
    for (int i = 0; i < TILES; ++i) {
        const int tile = wave + 8 * i;
        if (tile >= ntiles) break;                            // (wave-uniform)
        // (the tile's bias is requested here and added behind the products: three tiles' worth held from the start of the
        //  kernel pushed the three-tile form over the register file)
        float4 bv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const float4 *>(p.b + tile * 32 + 8 * g + 4 * h);
        ml_f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const unsigned char *bp = (tile * 32 < p.pos_features ? xp : xs) + t * kMlpHPitch + h * 16;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc = mfma_act_32x32x16(a[i][j], *reinterpret_cast<const uint4 *>(bp + j * 32), acc);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            acc[4 * g] += bv[g].x; acc[4 * g + 1] += bv[g].y; acc[4 * g + 2] += bv[g].z; acc[4 * g + 3] += bv[g].w;
        }
        if (row < p.rows) {
            bf16_t *orow = p.out + (int64_t)row * p.ldo + tile * 32 + 4 * h;
            if (tile * 32 + 32 <= p.n && (p.ldo & 3) == 0) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<uint2 *>(orow + 8 * g) =
                        make_uint2(pack_act2(acc[4 * g], acc[4 * g + 1]), pack_act2(acc[4 * g + 2], acc[4 * g + 3]));
            } else {                                          // a ragged last tile or an odd row stride: element stores
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = tile * 32 + ml_row(r, h);
                    if (f < p.n) p.out[(int64_t)row * p.ldo + f] = (bf16_t)(pack_act2(acc[r], 0.f) & 0xffffu);
                }
            }
        }
    }
}

// ---- y = LayerNorm(residual + Linear(x)) for a 256 -> 256 Linear on a few thousand rows ------------------------------
// The tails of the decoder layer's two attention blocks (models/bricks/salience_transformer.py:571-572, 583-585:
// norm2(query + dropout(out_proj(heads))), norm1(query + dropout(output_proj(sampled)))) -- a library GEMM and the fused
// add + LayerNorm launch each.  Wave w owns features 32 w .. 32 w + 31 of the 32 rows; the row statistics (two-pass: mean,
// then centred variance, as csrc/norm.hip) cross the waves through LDS.  The Linear's output is rounded to the activation
// type before the residual is added, as the module-by-module path stores it.
struct RowsLnArgs {
    const bf16_t *x, *res;     // [rows, 256] each
    int rows;
    const char *w;             // packed 8 tiles
    const float *b, *gamma, *beta;
    float eps;
    bf16_t *out;               // [rows, 256]
};

__global__ void __launch_bounds__(512) rows_linear_ln_kernel(RowsLnArgs p)
{
    extern __shared__ __align__(16) unsigned char mlp_lds[];
    unsigned char *xs = mlp_lds;                                       // [32][kMlpHPitch]
    float *part = reinterpret_cast<float *>(xs + kMlpRows * kMlpHPitch);   // [2][8 waves][32 rows]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * kMlpRows;
    const int row = row0 + t, rr = min(row, p.rows - 1);

    uint4 a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = *reinterpret_cast<const uint4 *>(p.w + ((wave * 16 + j) * 64 + lane) * 16);
    uint4 xv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = i * 512 + tid;
        xv[i] = *reinterpret_cast<const uint4 *>(p.x + (int64_t)min(row0 + (e >> 5), p.rows - 1) * kMlpHidden + (e & 31) * 8);
    }
    // my 16 features of row t: 32 w + 8 g + 4 h + 0..3 -- residual, bias and the norm's parameters
    uint2 rv[4];
    float4 bv[4], gv[4], ev[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int f = wave * 32 + 8 * g + 4 * h;
        rv[g] = *reinterpret_cast<const uint2 *>(p.res + (int64_t)rr * kMlpHidden + f);
        bv[g] = *reinterpret_cast<const float4 *>(p.b + f);
        gv[g] = *reinterpret_cast<const float4 *>(p.gamma + f);
        ev[g] = *reinterpret_cast<const float4 *>(p.beta + f);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = i * 512 + tid;
        *reinterpret_cast<uint4 *>(xs + (e >> 5) * kMlpHPitch + (e & 31) * 16) = xv[i];
    }
    __syncthreads();

    ml_f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    {
        const unsigned char *bp = xs + t * kMlpHPitch + h * 16;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc = mfma_act_32x32x16(a[j], *reinterpret_cast<const uint4 *>(bp + j * 32), acc);
    }
    float v[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float bb[4] = {bv[g].x, bv[g].y, bv[g].z, bv[g].w};
        const float rs[4] = {act_lo(rv[g].x), act_hi(rv[g].x), act_lo(rv[g].y), act_hi(rv[g].y)};
#pragma unroll
        for (int k = 0; k < 4; ++k) v[4 * g + k] = act_lo(pack_act2(acc[4 * g + k] + bb[k], 0.f)) + rs[k];
    }
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += v[r];
    sum += __shfl_xor(sum, 32);
    if (h == 0) part[wave * 32 + t] = sum;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) tot += part[w * 32 + t];
    const float mean = tot / (float)kMlpHidden;
    float sq = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sq += (v[r] - mean) * (v[r] - mean);
    sq += __shfl_xor(sq, 32);
    if (h == 0) part[256 + wave * 32 + t] = sq;
    __syncthreads();
    tot = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) tot += part[256 + w * 32 + t];
    const float rstd = rsqrtf(tot / (float)kMlpHidden + p.eps);
    if (row < p.rows) {
        bf16_t *orow = p.out + (int64_t)row * kMlpHidden + wave * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float gg[4] = {gv[g].x, gv[g].y, gv[g].z, gv[g].w}, ee[4] = {ev[g].x, ev[g].y, ev[g].z, ev[g].w};
            float y[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] = (v[4 * g + k] - mean) * rstd * gg[k] + ee[k];
            *reinterpret_cast<uint2 *>(orow + 8 * g) = make_uint2(pack_act2(y[0], y[1]), pack_act2(y[2], y[3]));
        }
    }
}

// ---- the decoder layer's output head in one launch ------------------------------------------------------------------
// models/bricks/salience_transformer.py:655-668 for one layer i:
//     normed  = decoder.norm(query)
//     classes = class_head[i](normed)
//     coords  = sigmoid(bbox_head[i](normed) + inverse_sigmoid(reference_points))
//     next reference_points = sigmoid(bbox_head[i](query) + inverse_sigmoid(reference_points))      (not for the last layer)
// = LayerNorm + library GEMM + the three-layer chain on two row sources + the refinement: four launches after the chain
// became one (mlp_rows above), now one.  32 query rows per workgroup; the LayerNorm is the row-tile fill itself (32 lanes
// per row, two-pass statistics by shuffles -- the arithmetic of csrc/norm.hip); both bbox chains run side by side (one A
// fragment feeds two products); wave 0 finishes the boxes, waves 1.. the class tiles on the normed rows.
struct HeadArgs {
    const bf16_t *q;
    int rows;
    const float *gamma, *beta;
    float eps;
    const char *wc;            // packed class head (tiles of 32 classes), fp32 bias padded
    const float *bc;
    int ncls;
    const char *w1, *w2, *w3;  // packed bbox chain (as for mlp_rows)
    const float *b1, *b2, *b3;
    const float *ref;          // [rows, 4] boxes to refine (cx, cy, w, h in [0, 1])
    float sig_eps;
    bf16_t *logits;            // [rows, ncls], rows ld_logits elements apart
    int64_t ld_logits;
    float *boxes;              // [TWO ? 2 : 1][rows][4]: from the normed rows, then from the raw rows
};

__device__ __forceinline__ float head_inverse_sigmoid(float x, float eps)
{
    x = fminf(fmaxf(x, 0.f), 1.f);
    return logf(fmaxf(x, eps) / fmaxf(1.f - x, eps));
}

template <bool TWO>
__global__ void __launch_bounds__(512) decoder_head_kernel(HeadArgs p)
{
    extern __shared__ __align__(16) unsigned char mlp_lds[];
    constexpr int kTile = kMlpRows * kMlpHPitch;
    unsigned char *xn = mlp_lds, *xq = xn + kTile;            // normed rows, raw rows
    unsigned char *h1n = xq + kTile, *h1r = h1n + kTile, *h2n = h1r + kTile, *h2r = h2n + kTile;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * kMlpRows;
    const int cls_tiles = (p.ncls + 31) / 32;

    // ---- every global load, requested before anything waits ----
    uint4 a1[16], a2[16], a3[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a1[j] = *reinterpret_cast<const uint4 *>(p.w1 + ((wave * 16 + j) * 64 + lane) * 16);
    uint4 xv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = i * 512 + tid;
        const int r = min(row0 + (e >> 5), p.rows - 1);
        xv[i] = *reinterpret_cast<const uint4 *>(p.q + (int64_t)r * kMlpHidden + (e & 31) * 8);
    }
    float gm[8], bt[8];
    {
        const int pc = tid & 31;
        const float4 g0 = *reinterpret_cast<const float4 *>(p.gamma + pc * 8), g1 = *reinterpret_cast<const float4 *>(p.gamma + pc * 8 + 4);
        const float4 e0 = *reinterpret_cast<const float4 *>(p.beta + pc * 8), e1 = *reinterpret_cast<const float4 *>(p.beta + pc * 8 + 4);
        gm[0] = g0.x; gm[1] = g0.y; gm[2] = g0.z; gm[3] = g0.w; gm[4] = g1.x; gm[5] = g1.y; gm[6] = g1.z; gm[7] = g1.w;
        bt[0] = e0.x; bt[1] = e0.y; bt[2] = e0.z; bt[3] = e0.w; bt[4] = e1.x; bt[5] = e1.y; bt[6] = e1.z; bt[7] = e1.w;
    }
    float bias1[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 bv = *reinterpret_cast<const float4 *>(p.b1 + wave * 32 + 8 * g + 4 * h);
        bias1[4 * g] = bv.x; bias1[4 * g + 1] = bv.y; bias1[4 * g + 2] = bv.z; bias1[4 * g + 3] = bv.w;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) a2[j] = *reinterpret_cast<const uint4 *>(p.w2 + ((wave * 16 + j) * 64 + lane) * 16);
    const bool cls_wave = wave >= 1 && wave <= cls_tiles;            // (wave-uniform)

    // ---- the row tiles: raw rows as they are, normed rows through the LayerNorm (32 lanes per row) ----
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = i * 512 + tid;
        const int off = (e >> 5) * kMlpHPitch + (e & 31) * 16;
        *reinterpret_cast<uint4 *>(xq + off) = xv[i];
        float v[8] = {act_lo(xv[i].x), act_hi(xv[i].x), act_lo(xv[i].y), act_hi(xv[i].y),
                      act_lo(xv[i].z), act_hi(xv[i].z), act_lo(xv[i].w), act_hi(xv[i].w)};
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) sum += v[k];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 32);
        const float mean = sum / (float)kMlpHidden;
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) sq += (v[k] - mean) * (v[k] - mean);
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 32);
        const float rstd = rsqrtf(sq / (float)kMlpHidden + p.eps);
        float y[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) y[k] = (v[k] - mean) * rstd * gm[k] + bt[k];
        *reinterpret_cast<uint4 *>(xn + off) =
            make_uint4(pack_act2(y[0], y[1]), pack_act2(y[2], y[3]), pack_act2(y[4], y[5]), pack_act2(y[6], y[7]));
    }
    __syncthreads();

    // ---- bbox chain, layer 1 (both row sources) ----
    ml_f32x16_t an, ar;
#pragma unroll
    for (int r = 0; r < 16; ++r) { an[r] = bias1[r]; ar[r] = bias1[r]; }
    {
        const int lo = t * kMlpHPitch + h * 16;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            an = mfma_act_32x32x16(a1[j], *reinterpret_cast<const uint4 *>(xn + lo + j * 32), an);
            if (TWO) ar = mfma_act_32x32x16(a1[j], *reinterpret_cast<const uint4 *>(xq + lo + j * 32), ar);
        }
    }
    // the last stage's fragments (wave 0: bbox layer 3; waves 1..: a class tile) take layer 1's registers: requested now,
    // back long before layer 2 is through (all three sets at once spilled 65-80 registers)
    if (wave == 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) a3[j] = *reinterpret_cast<const uint4 *>(p.w3 + (j * 64 + lane) * 16);
    } else if (cls_wave) {
#pragma unroll
        for (int j = 0; j < 16; ++j) a3[j] = *reinterpret_cast<const uint4 *>(p.wc + (((wave - 1) * 16 + j) * 64 + lane) * 16);
    }
    float bias2[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 bv = *reinterpret_cast<const float4 *>(p.b2 + wave * 32 + 8 * g + 4 * h);
        bias2[4 * g] = bv.x; bias2[4 * g + 1] = bv.y; bias2[4 * g + 2] = bv.z; bias2[4 * g + 3] = bv.w;
    }
    auto put_relu = [&](unsigned char *dst, const ml_f32x16_t &acc) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<uint2 *>(dst + t * kMlpHPitch + (wave * 32 + 8 * g + 4 * h) * 2) =
                make_uint2(pack_act2(fmaxf(acc[4 * g], 0.f), fmaxf(acc[4 * g + 1], 0.f)),
                           pack_act2(fmaxf(acc[4 * g + 2], 0.f), fmaxf(acc[4 * g + 3], 0.f)));
    };
    put_relu(h1n, an);
    if (TWO) put_relu(h1r, ar);
    __syncthreads();

    // ---- layer 2 ----
#pragma unroll
    for (int r = 0; r < 16; ++r) { an[r] = bias2[r]; ar[r] = bias2[r]; }
    {
        const int lo = t * kMlpHPitch + h * 16;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            an = mfma_act_32x32x16(a2[j], *reinterpret_cast<const uint4 *>(h1n + lo + j * 32), an);
            if (TWO) ar = mfma_act_32x32x16(a2[j], *reinterpret_cast<const uint4 *>(h1r + lo + j * 32), ar);
        }
    }
    put_relu(h2n, an);
    if (TWO) put_relu(h2r, ar);
    __syncthreads();

    const int row = row0 + t;
    if (wave == 0) {
        // ---- layer 3 (4 outputs) and the refinement: lane (t, h = 0) holds the four deltas of row t in registers 0..3 ----
#pragma unroll
        for (int r = 0; r < 16; ++r) { an[r] = 0.f; ar[r] = 0.f; }
        const int lo = t * kMlpHPitch + h * 16;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            an = mfma_act_32x32x16(a3[j], *reinterpret_cast<const uint4 *>(h2n + lo + j * 32), an);
            if (TWO) ar = mfma_act_32x32x16(a3[j], *reinterpret_cast<const uint4 *>(h2r + lo + j * 32), ar);
        }
        if (h == 0 && row < p.rows) {
            const float4 rf = *reinterpret_cast<const float4 *>(p.ref + (int64_t)row * 4);
            const float4 b3 = *reinterpret_cast<const float4 *>(p.b3);
            const float inv[4] = {head_inverse_sigmoid(rf.x, p.sig_eps), head_inverse_sigmoid(rf.y, p.sig_eps),
                                  head_inverse_sigmoid(rf.z, p.sig_eps), head_inverse_sigmoid(rf.w, p.sig_eps)};
            const float bb[4] = {b3.x, b3.y, b3.z, b3.w};
            float o[4];
            // (the chain's output exists in the activation type in the reference's graph: round it before the refinement)
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = 1.f / (1.f + expf(-(act_lo(pack_act2(an[k] + bb[k], 0.f)) + inv[k])));
            reinterpret_cast<float4 *>(p.boxes)[row] = make_float4(o[0], o[1], o[2], o[3]);
            if (TWO) {
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = 1.f / (1.f + expf(-(act_lo(pack_act2(ar[k] + bb[k], 0.f)) + inv[k])));
                reinterpret_cast<float4 *>(p.boxes)[(int64_t)p.rows + row] = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
    } else if (cls_wave) {
        // ---- class logits of the normed rows, tile wave - 1 ----
        const int tile = wave - 1;
#pragma unroll
        for (int r = 0; r < 16; ++r) an[r] = 0.f;
        const int lo = t * kMlpHPitch + h * 16;
#pragma unroll
        for (int j = 0; j < 16; ++j) an = mfma_act_32x32x16(a3[j], *reinterpret_cast<const uint4 *>(xn + lo + j * 32), an);
        if (row < p.rows) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = tile * 32 + ml_row(r, h);
                if (f < p.ncls)
                    p.logits[(int64_t)row * p.ld_logits + f] = (bf16_t)(pack_act2(an[r] + p.bc[f], 0.f) & 0xffffu);
            }
        }
    }
}

template <int K1, int NL, bool SINE = false>
static int mlp_launch(hipStream_t s, const MlpArgs &a)
{
    const size_t lds = (size_t)kMlpRows * (K1 * 2 + 16) + (size_t)(NL == 3 ? 2 : 1) * kMlpRows * kMlpHPitch;
    static DeviceOnce once;
    allow_dynamic_lds((mlp_rows_kernel<K1, NL, SINE>), once, 96 * 1024);
    hipLaunchKernelGGL((mlp_rows_kernel<K1, NL, SINE>), dim3((unsigned)((a.rows + kMlpRows - 1) / kMlpRows)), dim3(512), lds, s, a);
    return check_launch(SINE ? "ref_point_head" : "mlp_rows");
}

}  // namespace sdetr

using namespace sdetr;

extern "C" int sdetr_mlp_rows_bf16(sdetr_stream_t stream, const void *x, const void *x_second, int64_t rows_first,
                                   int64_t rows, int in_features, const void *packed_weight1, const float *bias1,
                                   const void *packed_weight2, const float *bias2, const void *packed_weight3,
                                   const float *bias3, int out_features, void *out, int64_t out_row_stride)
{
    if (rows < 0 || rows_first < 0 || rows_first > rows || rows > 0x7fffffffLL) return fail("mlp_rows: bad row counts");
    if (in_features != 256 && in_features != 512) return fail("mlp_rows: 256 or 512 input features (got %d)", in_features);
    const bool three = packed_weight3 != nullptr;
    if (three ? (out_features < 1 || out_features > 32) : out_features != kMlpHidden)
        return fail("mlp_rows: a two-layer chain ends in 256 features, a three-layer chain in 1..32 (got %d)", out_features);
    if (three && in_features != 256) return fail("mlp_rows: the three-layer chain takes 256 input features");
    if (out_row_stride < out_features || (!three && (out_row_stride % 4))) return fail("mlp_rows: bad output row stride");
    if (rows == 0) return 0;
    if (!x || !packed_weight1 || !bias1 || !packed_weight2 || !bias2 || !out || (three && !bias3) ||
        (rows_first < rows && !x_second))
        return fail("mlp_rows: null pointer");
    const auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!al16(x) || (x_second && !al16(x_second)) || !al16(packed_weight1) || !al16(packed_weight2) ||
        (three && !al16(packed_weight3)) || !al16(bias1) || !al16(bias2) || (!three && (reinterpret_cast<uintptr_t>(out) & 7)))
        return fail("mlp_rows: rows, packed weights and hidden biases must be 16-byte aligned, a 256-wide output 8-byte aligned");
    MlpArgs a{};
    a.xa = (const bf16_t *)x; a.xb = (const bf16_t *)x_second; a.rows_a = (int)rows_first; a.rows = (int)rows;
    a.w1 = (const char *)packed_weight1; a.b1 = bias1; a.w2 = (const char *)packed_weight2; a.b2 = bias2;
    a.w3 = (const char *)packed_weight3; a.b3 = bias3; a.n3 = out_features; a.out = (bf16_t *)out;
    a.ldo = out_row_stride;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (three) return mlp_launch<256, 3>(s, a);
    return in_features == 512 ? mlp_launch<512, 2>(s, a) : mlp_launch<256, 2>(s, a);
}

extern "C" int sdetr_rows_linear_bf16(sdetr_stream_t stream, const void *x, const void *pos, int64_t rows,
                                      int pos_features, const void *packed_weight, const float *bias_padded,
                                      int out_features, void *out, int64_t out_row_stride)
{
    if (rows < 0 || rows > 0x7fffffffLL) return fail("rows_linear: bad row count");
    if (out_features < 1 || out_features > 768) return fail("rows_linear: 1..768 output features (got %d)", out_features);
    if (pos_features < 0 || pos_features > out_features + 31 || (pos_features % 32))
        return fail("rows_linear: pos_features must be a multiple of 32 within the output (got %d)", pos_features);
    if (out_row_stride < out_features) return fail("rows_linear: bad output row stride");
    if (rows == 0) return 0;
    if (!x || !packed_weight || !bias_padded || !out || (pos_features > 0 && !pos)) return fail("rows_linear: null pointer");
    const auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!al16(x) || (pos && !al16(pos)) || !al16(packed_weight) || !al16(bias_padded) || (reinterpret_cast<uintptr_t>(out) & 7))
        return fail("rows_linear: rows, packed weight and bias must be 16-byte aligned, the output 8-byte aligned");
    RowsLinearArgs a{};
    a.x = (const bf16_t *)x; a.pos = pos_features > 0 ? (const bf16_t *)pos : nullptr; a.rows = (int)rows; a.n = out_features;
    a.pos_features = pos_features; a.w = (const char *)packed_weight; a.b = bias_padded; a.out = (bf16_t *)out;
    a.ldo = out_row_stride;
    const int ntiles = (out_features + 31) / 32, per_wave = (ntiles + 7) / 8;
    const dim3 grid((unsigned)((rows + kMlpRows - 1) / kMlpRows));
    const size_t lds = 2 * (size_t)kMlpRows * kMlpHPitch;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (per_wave == 1) hipLaunchKernelGGL(rows_linear_kernel<1>, grid, dim3(512), lds, s, a);
    else if (per_wave == 2) hipLaunchKernelGGL(rows_linear_kernel<2>, grid, dim3(512), lds, s, a);
    else hipLaunchKernelGGL(rows_linear_kernel<3>, grid, dim3(512), lds, s, a);
    return check_launch("rows_linear");
}

extern "C" int sdetr_decoder_head_bf16(sdetr_stream_t stream, const void *query, int64_t rows, const float *norm_weight,
                                       const float *norm_bias, float norm_eps, const void *packed_class_weight,
                                       const float *class_bias_padded, int num_classes, const void *packed_weight1,
                                       const float *bias1, const void *packed_weight2, const float *bias2,
                                       const void *packed_weight3, const float *bias3, const float *reference_points,
                                       float sigmoid_eps, int two_sources, void *logits, int64_t logits_row_stride,
                                       float *boxes)
{
    if (rows < 0 || rows > 0x7fffffffLL) return fail("decoder_head: bad row count");
    if (num_classes < 1 || num_classes > 224) return fail("decoder_head: 1..224 classes (got %d)", num_classes);
    if (logits_row_stride < num_classes) return fail("decoder_head: bad logits row stride");
    if (rows == 0) return 0;
    if (!query || !norm_weight || !norm_bias || !packed_class_weight || !class_bias_padded || !packed_weight1 || !bias1 ||
        !packed_weight2 || !bias2 || !packed_weight3 || !bias3 || !reference_points || !logits || !boxes)
        return fail("decoder_head: null pointer");
    const auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!al16(query) || !al16(norm_weight) || !al16(norm_bias) || !al16(packed_class_weight) || !al16(packed_weight1) ||
        !al16(packed_weight2) || !al16(packed_weight3) || !al16(bias1) || !al16(bias2) || !al16(bias3) ||
        !al16(reference_points) || !al16(boxes))
        return fail("decoder_head: operands must be 16-byte aligned");
    HeadArgs a{};
    a.q = (const bf16_t *)query; a.rows = (int)rows; a.gamma = norm_weight; a.beta = norm_bias; a.eps = norm_eps;
    a.wc = (const char *)packed_class_weight; a.bc = class_bias_padded; a.ncls = num_classes;
    a.w1 = (const char *)packed_weight1; a.w2 = (const char *)packed_weight2; a.w3 = (const char *)packed_weight3;
    a.b1 = bias1; a.b2 = bias2; a.b3 = bias3; a.ref = reference_points; a.sig_eps = sigmoid_eps;
    a.logits = (bf16_t *)logits; a.ld_logits = logits_row_stride; a.boxes = boxes;
    const dim3 grid((unsigned)((rows + kMlpRows - 1) / kMlpRows));
    const size_t lds = 6 * (size_t)kMlpRows * kMlpHPitch;
    hipStream_t s = static_cast<hipStream_t>(stream);
    static DeviceOnce once_a, once_b;
    if (two_sources) {
        allow_dynamic_lds(decoder_head_kernel<true>, once_a, 128 * 1024);
        hipLaunchKernelGGL(decoder_head_kernel<true>, grid, dim3(512), lds, s, a);
    } else {
        allow_dynamic_lds(decoder_head_kernel<false>, once_b, 128 * 1024);
        hipLaunchKernelGGL(decoder_head_kernel<false>, grid, dim3(512), lds, s, a);
    }
    return check_launch("decoder_head");
}

extern "C" int sdetr_ref_point_head_bf16(sdetr_stream_t stream, const float *reference_points, const float *valid_ratios,
                                         int batch_size, int num_queries, int num_levels, float temperature,
                                         const void *packed_weight1, const float *bias1, const void *packed_weight2,
                                         const float *bias2, void *query_pos, float *reference_points_input)
{
    if (batch_size < 0 || num_queries < 0 || num_levels < 1 || num_levels > 8) return fail("ref_point_head: bad sizes");
    const int64_t rows = (int64_t)batch_size * num_queries;
    if (rows > 0x7fffffffLL) return fail("ref_point_head: too many rows");
    if (rows == 0) return 0;
    if (!reference_points || !valid_ratios || !packed_weight1 || !bias1 || !packed_weight2 || !bias2 || !query_pos)
        return fail("ref_point_head: null pointer");
    const auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!al16(reference_points) || !al16(packed_weight1) || !al16(packed_weight2) || !al16(bias1) || !al16(bias2) ||
        (reinterpret_cast<uintptr_t>(query_pos) & 7) || (reference_points_input && !al16(reference_points_input)))
        return fail("ref_point_head: operands must be 16-byte aligned (query_pos 8-byte)");
    MlpArgs a{};
    a.rows = (int)rows; a.rows_a = (int)rows;
    a.w1 = (const char *)packed_weight1; a.b1 = bias1; a.w2 = (const char *)packed_weight2; a.b2 = bias2;
    a.n3 = kMlpHidden; a.out = (bf16_t *)query_pos; a.ldo = kMlpHidden;
    a.ref = reference_points; a.vr = valid_ratios; a.Nq = num_queries; a.L = num_levels; a.temperature = temperature;
    a.ref_in = reference_points_input;
    return mlp_launch<512, 2, true>(static_cast<hipStream_t>(stream), a);
}

extern "C" int sdetr_rows_linear_ln_bf16(sdetr_stream_t stream, const void *x, const void *residual, int64_t rows,
                                         const void *packed_weight, const float *bias_padded, const float *norm_weight,
                                         const float *norm_bias, float norm_eps, void *out)
{
    if (rows < 0 || rows > 0x7fffffffLL) return fail("rows_linear_ln: bad row count");
    if (rows == 0) return 0;
    if (!x || !residual || !packed_weight || !bias_padded || !norm_weight || !norm_bias || !out)
        return fail("rows_linear_ln: null pointer");
    const auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!al16(x) || !al16(packed_weight) || !al16(bias_padded) || !al16(norm_weight) || !al16(norm_bias) ||
        (reinterpret_cast<uintptr_t>(residual) & 7) || (reinterpret_cast<uintptr_t>(out) & 7))
        return fail("rows_linear_ln: operands must be 16-byte aligned (residual and out 8-byte)");
    RowsLnArgs a{};
    a.x = (const bf16_t *)x; a.res = (const bf16_t *)residual; a.rows = (int)rows; a.w = (const char *)packed_weight;
    a.b = bias_padded; a.gamma = norm_weight; a.beta = norm_bias; a.eps = norm_eps; a.out = (bf16_t *)out;
    const size_t lds = (size_t)kMlpRows * kMlpHPitch + 2 * 8 * 32 * sizeof(float);
    hipLaunchKernelGGL(rows_linear_ln_kernel, dim3((unsigned)((rows + kMlpRows - 1) / kMlpRows)), dim3(512), lds,
                       static_cast<hipStream_t>(stream), a);
    return check_launch("rows_linear_ln");
}
