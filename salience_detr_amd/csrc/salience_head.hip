// The salience head (MaskPredictor, models/bricks/salience_transformer.py:16-47) and the projection in front of
// it (enc_output + enc_output_norm, models/bricks/base_transformer.py:110-111) as three launches per level
// instead of ~20 library/elementwise launches:
//
//   stage 1  x -> [enc_output GEMM -> +bias -> enc_output_norm] -> x + x*up*alpha (:143, `up` = bilinear
//            align_corners resize of the coarser level's score, :139-142) -> layer1 LayerNorm -> layer1 GEMM ->
//            GELU;  the "local" half (columns 0..127) is stored, the "global" half (128..255) only contributes
//            its per-block column sums (the token mean of :43-45)
//   const    per-image constant of layer2[0]:  W2[:,128:] @ mean_global + b2
//   stage 2  GELU(z_local @ W2[:, :128]^T + const) -> GELU(. @ W3^T + b3) -> . w4 + b4  = the token's score
//
// Everything is fp32 -- the scores decide WHICH tokens are kept, and the selection is pinned index-for-index to
// the fp32 reference -- so the GEMMs run on the f32-input MFMA (v_mfma_f32_32x32x2_f32, exact fp32 FMA chains,
// 157 TFLOP/s dense peak on MI355X, the same as the vector rate but without per-lane operand broadcasts).  The
// path is bound by that peak: 2*(2*256*256 + 128*128 + 128*64) flops per token.
//
// Tiling: one 256-thread block owns 32 (or 64) tokens, kept in LDS as [32][256+4] fp32 (the +4 makes the 16-byte
// A reads and the row-wise LayerNorm passes bank-conflict free); wave w owns output columns [64w, 64w+64) as 1x2
// (2x2) MFMA tiles.  The contraction index is visited in the order k = 8S + 4h + j (h = lane>>5, j = 0..3) so that a lane
// fetches its four A values with one ds_read_b128 and its four B values with one 16-byte global load from the
// pre-packed weight P[S][n][h][j] = W[n][8S+4h+j] (sdetr_pack_linear_f32) -- consecutive lanes read consecutive
// 16-byte pieces.  Four blocks fit a CU (39 KB LDS each).  Deterministic: no atomics, fixed summation order.
#include <cstdlib>

#include "common.h"

#include "salience_head_core.h"

namespace sdetr {

__global__ void __launch_bounds__(512, 2) salience_head_stage1_x3_kernel(Stage1Args p)
{
    stage1_x3_body(p, (int)blockIdx.x, (int)blockIdx.y);
}


// WAVES = 8 (512 threads, one column tile per wave) halves a block's latency -- two chained 256 x 256 fp32 GEMMs of
// 64-cycle MFMAs, ~7 us each with four waves -- for the coarse levels, whose few blocks leave the chip idle anyway.
template <int RT, int WAVES = 4>
__global__ void __launch_bounds__(64 * WAVES, WAVES == 8 ? 2 : (RT == 2 ? 2 : 4)) salience_head_stage1_kernel(Stage1Args p)
{
    constexpr int TM = 32 * RT, THREADS = 64 * WAVES, CT = 8 / WAVES;   // CT column tiles of 32 per wave
    static_assert(TM * 64 % THREADS == 0 && THREADS % TM == 0, "tile / block shape");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *tile = smem;                           // [TM][kXS]
    float *par = smem + TM * kXS;                 // [kParRows][kC]
    float *srow = par + kParRows * kC;            // [TM] modulation factor up * alpha (0 when absent)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, blk = blockIdx.x;
    const int t0 = blk * TM;
    const int nvalid = min(TM, p.n - t0);
    const int n0 = wave * (32 * CT);
    const bool with_enc = p.w_enc != nullptr;

    WeightStream<CT, 4> ws;
    ws.start(with_enc ? p.w_enc : p.w1, kC, kC / 8, n0, lane);

    // ---- token tile, parameters and row factors -> LDS ----
    {
        const float *xb = p.x + (int64_t)b * p.x_batch_stride + (int64_t)t0 * p.x_row_stride;
        constexpr int NL = TM * 64 / THREADS;   // float4 per thread
        float4 v[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {   // unconditional (clamped) loads: all in flight at once
            const int idx = tid + i * THREADS;
            const int r = idx >> 6, c4 = idx & 63;
            v[i] = *reinterpret_cast<const float4 *>(xb + (int64_t)min(r, nvalid - 1) * p.x_row_stride + c4 * 4);
        }
        if (tid < 256) {
            const int row = tid >> 6, c4 = tid & 63;   // 4 parameter rows per pass
            const float *src0 = row == 0 ? p.b_enc : row == 1 ? p.g_enc : row == 2 ? p.beta_enc : p.g1;
            const float *src1 = row == 0 ? p.beta1 : p.b1;
            if (with_enc || row == 3) *reinterpret_cast<float4 *>(par + row * kC + c4 * 4) =
                                          *reinterpret_cast<const float4 *>(src0 + c4 * 4);
            if (row < 2) *reinterpret_cast<float4 *>(par + (4 + row) * kC + c4 * 4) =
                             *reinterpret_cast<const float4 *>(src1 + c4 * 4);
        }
        if (tid < TM) {
            float s = 0.f;
            const int t = min(t0 + tid, p.n - 1);
            if (p.row_scale) {
                s = p.row_scale[(int64_t)b * p.n + t];
            } else if (p.coarse) {
                // bilinear, align_corners=True (F.interpolate, salience_transformer.py:139-142)
                const int y = t / p.w, x = t - y * p.w;
                const float sh = p.h > 1 ? (float)(p.ch - 1) / (float)(p.h - 1) : 0.f;
                const float sw = p.w > 1 ? (float)(p.cw - 1) / (float)(p.w - 1) : 0.f;
                const float fy = sh * (float)y, fx = sw * (float)x;
                const int y1 = (int)fy, x1 = (int)fx;
                const int yp = y1 < p.ch - 1 ? 1 : 0, xp = x1 < p.cw - 1 ? 1 : 0;
                const float ly = fy - (float)y1, lx = fx - (float)x1;
                const float hy = 1.f - ly, hx = 1.f - lx;
                const float *cm = p.coarse + (int64_t)b * p.ch * p.cw;
                s = hy * (hx * cm[y1 * p.cw + x1] + lx * cm[y1 * p.cw + x1 + xp]) +
                    ly * (hx * cm[(y1 + yp) * p.cw + x1] + lx * cm[(y1 + yp) * p.cw + x1 + xp]);
            }
            // reference order of operations: mem + mem * up * alpha  ==  mem + (mem * up) * alpha
            srow[tid] = s;
            if (tid == 0) srow[TM] = (p.row_scale || p.coarse) ? (p.alpha ? *p.alpha : 1.f) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = tid + i * THREADS;
            const int r = idx >> 6, c4 = idx & 63;
            *reinterpret_cast<float4 *>(tile + r * kXS + c4 * 4) = r < nvalid ? v[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();

    f32x16 acc[RT][CT];
    if (with_enc) {
        zero_acc(acc);
        block_gemm<kC, kXS, RT, CT, 4>(tile, ws, lane, acc);
        ws.start(p.w1, kC, kC / 8, n0, lane);   // layer1's first steps travel during the LayerNorm phase
        __syncthreads();   // every wave is done reading the input tile
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int c = n0 + 32 * ct + (lane & 31);
            const float bias = par[kParBEnc * kC + c];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int i = 0; i < 16; ++i) tile[(32 * rt + acc_row(i, lane)) * kXS + c] = acc[rt][ct][i] + bias;
        }
        __syncthreads();
    }

    // ---- enc_output_norm -> modulation -> layer1 LayerNorm, in place; TPR threads per row, each NV float4 ----
    {
        constexpr int TPR = THREADS / TM, NV = kC / 4 / TPR, CS = 4 * TPR;   // column step between a thread's float4s
        const int r = tid / TPR, q = tid % TPR;
        float *row = tile + r * kXS + 4 * q;
        float4 v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const float4 *>(row + CS * i);
        float mean, rstd;
        if (with_enc) {
            row_stats<NV, TPR>(v, p.eps_enc, mean, rstd);
#pragma unroll
            for (int i = 0; i < NV; ++i)
                v[i] = ln_apply(v[i], mean, rstd, *reinterpret_cast<const float4 *>(par + kParGEnc * kC + CS * i + 4 * q),
                                *reinterpret_cast<const float4 *>(par + kParBetaEnc * kC + CS * i + 4 * q));
            if (p.memory_out && r < nvalid) {
                float *mo = p.memory_out + (int64_t)b * p.mem_batch_stride + (int64_t)(t0 + r) * kC + 4 * q;
#pragma unroll
                for (int i = 0; i < NV; ++i) *reinterpret_cast<float4 *>(mo + CS * i) = v[i];
            }
        }
        const float s = srow[r], a = srow[TM];
        if (a != 0.f) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                v[i] = make_float4(v[i].x + v[i].x * s * a, v[i].y + v[i].y * s * a, v[i].z + v[i].z * s * a,
                                   v[i].w + v[i].w * s * a);
        }
        row_stats<NV, TPR>(v, p.eps1, mean, rstd);
#pragma unroll
        for (int i = 0; i < NV; ++i)
            *reinterpret_cast<float4 *>(row + CS * i) =
                ln_apply(v[i], mean, rstd, *reinterpret_cast<const float4 *>(par + kParG1 * kC + CS * i + 4 * q),
                         *reinterpret_cast<const float4 *>(par + kParBeta1 * kC + CS * i + 4 * q));
    }
    __syncthreads();

    // ---- layer1 Linear + GELU ----
    zero_acc(acc);
    block_gemm<kC, kXS, RT, CT, 4>(tile, ws, lane, acc);
    if (n0 < kHalf) {
        float *zl = p.z_local + ((int64_t)b * p.n + t0) * kHalf;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int c = n0 + 32 * ct + (lane & 31);
            const float bias = par[kParB1 * kC + c];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int r = 32 * rt + acc_row(i, lane);
                    if (r < nvalid) zl[(int64_t)r * kHalf + c] = gelu_erf(acc[rt][ct][i] + bias);
                }
        }
    } else {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int c = n0 + 32 * ct + (lane & 31);
            const float bias = par[kParB1 * kC + c];
            float s = 0.f;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int r = 32 * rt + acc_row(i, lane);
                    s += r < nvalid ? gelu_erf(acc[rt][ct][i] + bias) : 0.f;
                }
            s += __shfl_xor(s, 32);
            if (lane < 32) p.partial[((int64_t)b * p.nblk + blk) * kHalf + (c - kHalf)] = s;
        }
    }
}

// P[S][ctile][plane][lane][j] = plane of W[32 ctile + (lane & 31)][16 S + 8 (lane >> 5) + j]
__global__ void pack_linear_bf16x3_kernel(const float *w, int64_t row_stride, int N, int K, uint16_t *out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * K) return;
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const int64_t rest = i >> 9;
    const int ctile = (int)(rest % (N / 32)), S = (int)(rest / (N / 32));
    const float x = w[(int64_t)(32 * ctile + (lane & 31)) * row_stride + 16 * S + 8 * (lane >> 5) + j];
    const uint32_t h = f32_to_bf16_bits(x);
    const float r1 = x - __uint_as_float(h << 16);
    const uint32_t m = f32_to_bf16_bits(r1);
    const float r2 = r1 - __uint_as_float(m << 16);
    const uint32_t l = f32_to_bf16_bits(r2);
    uint16_t *o = out + ((int64_t)(S * (N / 32) + ctile) * 3) * 512 + lane * 8 + j;
    o[0] = (uint16_t)h;
    o[512] = (uint16_t)m;
    o[1024] = (uint16_t)l;
}

// const[b][j] = b2[j] + sum_c W2[j][128 + c] * mean_c,  mean_c = (sum over blocks of partial) / n.
// 8 groups of 128 threads sum interleaved blocks (fixed order), then a fixed-order tree over the groups.
constexpr int kConstGroups = 8;
__global__ void __launch_bounds__(kHalf * kConstGroups) salience_head_const_kernel(const float *partial, int nblk, int n,
                                                                                   const float *w2, const float *b2,
                                                                                   float *out, float *score_min)
{
    if (score_min && blockIdx.x == 0 && threadIdx.x == 0) *score_min = INFINITY;   // stage 2 takes the min into it
    __shared__ float part[kConstGroups][kHalf];
    __shared__ float mean[kHalf];
    const int b = blockIdx.x, j = threadIdx.x & (kHalf - 1), g = threadIdx.x / kHalf;
    const float *pp = partial + (int64_t)b * nblk * kHalf + j;
    // 16 rows in flight per thread: the loop is a chain of trips to L2 (each ~0.3 us), four at a time it was most of the
    // kernel's 4.6-9 us.  The order of the additions is fixed by the code (same bits every run).
    float s[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) s[u] = 0.f;
    int i = g;
    for (; i + 15 * kConstGroups < nblk; i += 16 * kConstGroups) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = pp[(int64_t)(i + u * kConstGroups) * kHalf];
#pragma unroll
        for (int u = 0; u < 16; ++u) s[u] += v[u];
    }
    {   // tail: up to 15 rows, still requested together
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = (i + u * kConstGroups < nblk) ? pp[(int64_t)(i + u * kConstGroups) * kHalf] : 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) s[u] += v[u];
    }
    part[g][j] = (((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]))) +
                 (((s[8] + s[9]) + (s[10] + s[11])) + ((s[12] + s[13]) + (s[14] + s[15])));
    __syncthreads();
    if (g == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < kConstGroups; ++k) t += part[k][j];
        mean[j] = t / (float)n;
    }
    __syncthreads();
    // 8 threads per output row j2 = tid / 8, each 16 of the 128 columns
    const int j2 = threadIdx.x >> 3, q = threadIdx.x & 7;
    const float *wr = w2 + (int64_t)j2 * kC + kHalf + q * 16;
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < 16; c += 4) {
        const float4 wv = *reinterpret_cast<const float4 *>(wr + c);
        a += (wv.x * mean[q * 16 + c] + wv.y * mean[q * 16 + c + 1]) +
             (wv.z * mean[q * 16 + c + 2] + wv.w * mean[q * 16 + c + 3]);
    }
    a += __shfl_xor(a, 1, 8);
    a += __shfl_xor(a, 2, 8);
    a += __shfl_xor(a, 4, 8);
    if (q == 0) out[(int64_t)b * kHalf + j2] = b2[j2] + a;
}

__global__ void __launch_bounds__(kBlock, 2) salience_head_stage2_kernel(Stage2Args p)
{
    __shared__ __attribute__((aligned(16))) float zt[kStage2TileFloats];
    __shared__ float red[2 * kTM];
    stage2_body(p, (int)blockIdx.x, (int)blockIdx.y, zt, red);
}

// P[S][n][h][j] = W[n][8S + 4h + j]
__global__ void pack_linear_kernel(const float *w, int64_t row_stride, int N, int K, float *out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * K) return;
    const int j = (int)(i & 3), h = (int)((i >> 2) & 1);
    const int64_t rest = i >> 3;
    const int n = (int)(rest % N), S = (int)(rest / N);
    out[i] = w[(int64_t)n * row_stride + 8 * S + 4 * h + j];
}

}  // namespace sdetr

using namespace sdetr;

extern "C" int sdetr_pack_linear_f32(sdetr_stream_t stream, const float *weight, int64_t row_stride, int out_features,
                                     int in_features, float *packed)
{
    if (!weight || !packed) return fail("pack_linear: NULL pointer");
    if (out_features <= 0 || in_features <= 0 || in_features % 8 != 0)
        return fail("pack_linear: in_features must be a positive multiple of 8 (got %d x %d)", out_features, in_features);
    const int64_t total = (int64_t)out_features * in_features;
    hipLaunchKernelGGL(pack_linear_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), weight, row_stride, out_features, in_features, packed);
    return check_launch("pack_linear");
}

// 32-token blocks everywhere: measured on MI355X at 800x1333 (2 x 16 700 tokens on level 0) they take 110 us vs
// 137 us for 64-token blocks -- 1044 blocks spread over 256 CUs x 4 resident blocks with a one-block tail instead
// of 522 over 256 x 2 -- and on the small levels they are what fills the chip at all.  The 64-token variant
// (half the weight traffic per token) stays selectable for experiments: SDETR_HEAD_ROWTILES=2.
static int stage1_lds_bytes(int tm) { return (tm * kXS + kParRows * kC + tm + 4) * (int)sizeof(float); }

static int stage1_block_tokens(int batch_size, int tokens)
{
    static const int forced = [] { const char *e = ab_env("SDETR_HEAD_ROWTILES"); return e ? atoi(e) : 0; }();
    (void)batch_size; (void)tokens;
    return forced == 2 ? 64 : 32;
}

extern "C" int sdetr_salience_head_blocks(int batch_size, int tokens)
{
    if (tokens <= 0 || batch_size <= 0) return 0;
    const int tm = stage1_block_tokens(batch_size, tokens);
    return (tokens + tm - 1) / tm;
}

extern "C" int sdetr_salience_head_stage1(sdetr_stream_t stream, const float *x, int64_t x_batch_stride,
                                          int64_t x_row_stride, int batch_size, int tokens, int channels,
                                          const float *enc_weight_packed, const float *enc_bias,
                                          const float *enc_norm_weight, const float *enc_norm_bias, float enc_norm_eps,
                                          const float *row_scale, const float *coarse_score, int coarse_h, int coarse_w,
                                          int level_h, int level_w, const float *alpha, const float *norm_weight,
                                          const float *norm_bias, float norm_eps, const float *weight_packed,
                                          const float *bias, float *memory_out, int64_t memory_batch_stride,
                                          float *z_local, float *partial_sums)
{
    if (channels != kC) return fail("salience_head_stage1: built for embed_dim = hidden_dim = %d (got %d)", kC, channels);
    if (batch_size < 0 || tokens < 0) return fail("salience_head_stage1: negative size");
    if (batch_size == 0 || tokens == 0) return 0;
    if (!x || !norm_weight || !norm_bias || !weight_packed || !bias || !z_local || !partial_sums)
        return fail("salience_head_stage1: NULL pointer");
    if (enc_weight_packed && (!enc_bias || !enc_norm_weight || !enc_norm_bias))
        return fail("salience_head_stage1: enc_output parameters incomplete");
    if (row_scale && coarse_score) return fail("salience_head_stage1: give row_scale OR coarse_score");
    if (coarse_score && ((int64_t)level_h * level_w != tokens || coarse_h <= 0 || coarse_w <= 0))
        return fail("salience_head_stage1: level %dx%d does not cover %d tokens", level_h, level_w, tokens);
    if ((x_row_stride % 4) || (x_batch_stride % 4)) return fail("salience_head_stage1: rows must be 16-byte aligned");
    Stage1Args a;
    a.x = x; a.x_batch_stride = x_batch_stride; a.x_row_stride = x_row_stride;
    a.w_enc = reinterpret_cast<const float4 *>(enc_weight_packed);
    a.b_enc = enc_bias; a.g_enc = enc_norm_weight; a.beta_enc = enc_norm_bias; a.eps_enc = enc_norm_eps;
    a.row_scale = row_scale; a.coarse = coarse_score; a.ch = coarse_h; a.cw = coarse_w; a.h = level_h; a.w = level_w;
    a.alpha = alpha; a.g1 = norm_weight; a.beta1 = norm_bias; a.eps1 = norm_eps;
    a.w1 = reinterpret_cast<const float4 *>(weight_packed); a.b1 = bias;
    a.memory_out = enc_weight_packed ? memory_out : nullptr; a.mem_batch_stride = memory_batch_stride;
    a.z_local = z_local; a.partial = partial_sums; a.n = tokens; a.nblk = 0;

    hipStream_t s = static_cast<hipStream_t>(stream);
    if (stage1_block_tokens(batch_size, tokens) == 64) {
        static DeviceOnce lds_once;   // > 64 KiB of dynamic LDS has to be requested once per device
        allow_dynamic_lds(salience_head_stage1_kernel<2, 4>, lds_once, (int)stage1_lds_bytes(64));
        a.nblk = (tokens + 63) / 64;
        hipLaunchKernelGGL((salience_head_stage1_kernel<2, 4>), dim3((unsigned)a.nblk, (unsigned)batch_size), dim3(kBlock),
                           (size_t)stage1_lds_bytes(64), s, a);
    } else {
        // eight waves per 32-token block: half the block latency on the coarse levels (whose few blocks leave the chip
        // idle anyway) and a shorter tail on the finest one (1050 blocks on 512 two-per-CU slots)
        a.nblk = (tokens + 31) / 32;
        hipLaunchKernelGGL((salience_head_stage1_kernel<1, 8>), dim3((unsigned)a.nblk, (unsigned)batch_size), dim3(512),
                           (size_t)stage1_lds_bytes(32), s, a);
    }
    return check_launch("salience_head_stage1");
}

extern "C" int sdetr_pack_linear_bf16x3(sdetr_stream_t stream, const float *weight, int64_t row_stride, int out_features,
                                        int in_features, void *packed)
{
    if (!weight || !packed) return fail("pack_linear_bf16x3: NULL pointer");
    if (out_features <= 0 || in_features <= 0 || in_features % 16 != 0 || out_features % 32 != 0)
        return fail("pack_linear_bf16x3: [out,in] must be multiples of (32,16) (got %d x %d)", out_features, in_features);
    const int64_t total = (int64_t)out_features * in_features;
    hipLaunchKernelGGL(pack_linear_bf16x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), weight, row_stride, out_features, in_features,
                       static_cast<uint16_t *>(packed));
    return check_launch("pack_linear_bf16x3");
}

static int stage1_x3_lds_bytes() { return kX3Region + (kParRows * kC + 32 + 4) * (int)sizeof(float); }

extern "C" int sdetr_salience_head_stage1_x3(sdetr_stream_t stream, const float *x, int64_t x_batch_stride,
                                             int64_t x_row_stride, int batch_size, int tokens, int channels,
                                             const void *enc_weight_x3, const float *enc_bias,
                                             const float *enc_norm_weight, const float *enc_norm_bias, float enc_norm_eps,
                                             const float *row_scale, const float *coarse_score, int coarse_h, int coarse_w,
                                             int level_h, int level_w, const float *alpha, const float *norm_weight,
                                             const float *norm_bias, float norm_eps, const void *weight_x3,
                                             const float *bias, float *memory_out, int64_t memory_batch_stride,
                                             float *z_local, float *partial_sums)
{
    if (channels != kC) return fail("salience_head_stage1_x3: built for embed_dim = hidden_dim = %d (got %d)", kC, channels);
    if (batch_size < 0 || tokens < 0) return fail("salience_head_stage1_x3: negative size");
    if (batch_size == 0 || tokens == 0) return 0;
    if (!x || !norm_weight || !norm_bias || !weight_x3 || !bias || !z_local || !partial_sums)
        return fail("salience_head_stage1_x3: NULL pointer");
    if (enc_weight_x3 && (!enc_bias || !enc_norm_weight || !enc_norm_bias))
        return fail("salience_head_stage1_x3: enc_output parameters incomplete");
    if (row_scale && coarse_score) return fail("salience_head_stage1_x3: give row_scale OR coarse_score");
    if (coarse_score && ((int64_t)level_h * level_w != tokens || coarse_h <= 0 || coarse_w <= 0))
        return fail("salience_head_stage1_x3: level %dx%d does not cover %d tokens", level_h, level_w, tokens);
    if ((x_row_stride % 4) || (x_batch_stride % 4)) return fail("salience_head_stage1_x3: rows must be 16-byte aligned");
    if (stage1_block_tokens(batch_size, tokens) != 32)
        return fail("salience_head_stage1_x3: 32-token blocks only (unset SDETR_HEAD_ROWTILES)");
    Stage1Args a;
    a.x = x; a.x_batch_stride = x_batch_stride; a.x_row_stride = x_row_stride;
    a.w_enc = reinterpret_cast<const float4 *>(enc_weight_x3);
    a.b_enc = enc_bias; a.g_enc = enc_norm_weight; a.beta_enc = enc_norm_bias; a.eps_enc = enc_norm_eps;
    a.row_scale = row_scale; a.coarse = coarse_score; a.ch = coarse_h; a.cw = coarse_w; a.h = level_h; a.w = level_w;
    a.alpha = alpha; a.g1 = norm_weight; a.beta1 = norm_bias; a.eps1 = norm_eps;
    a.w1 = reinterpret_cast<const float4 *>(weight_x3); a.b1 = bias;
    a.memory_out = enc_weight_x3 ? memory_out : nullptr; a.mem_batch_stride = memory_batch_stride;
    a.z_local = z_local; a.partial = partial_sums; a.n = tokens; a.nblk = (tokens + 31) / 32;
    hipLaunchKernelGGL(salience_head_stage1_x3_kernel, dim3((unsigned)a.nblk, (unsigned)batch_size), dim3(512),
                       (size_t)stage1_x3_lds_bytes(), static_cast<hipStream_t>(stream), a);
    return check_launch("salience_head_stage1_x3");
}

// The per-image constant of layer2[0] on its own (sdetr_salience_head_stage2 launches it itself; a caller that puts stage
// 2 in a launch of its own -- sdetr_stage2_with_value_proj -- runs it first).
extern "C" int sdetr_salience_head_const(sdetr_stream_t stream, const float *partial_sums, int batch_size, int tokens,
                                         const float *weight2, const float *bias2, float *const_workspace,
                                         float *score_min)
{
    if (batch_size < 0 || tokens < 0) return fail("salience_head_const: negative size");
    if (batch_size == 0 || tokens == 0) return 0;
    if (!partial_sums || !weight2 || !bias2 || !const_workspace) return fail("salience_head_const: NULL pointer");
    hipLaunchKernelGGL(salience_head_const_kernel, dim3((unsigned)batch_size), dim3(kHalf * kConstGroups), 0,
                       static_cast<hipStream_t>(stream), partial_sums, sdetr_salience_head_blocks(batch_size, tokens),
                       tokens, weight2, bias2, const_workspace, score_min);
    return check_launch("salience_head_const");
}

extern "C" int sdetr_salience_head_stage2(sdetr_stream_t stream, const float *z_local, const float *partial_sums,
                                          int batch_size, int tokens, const float *weight2, const float *bias2,
                                          const float *weight2_local_packed, const float *weight3_packed,
                                          const float *bias3, const float *weight4, const float *bias4,
                                          float *const_workspace, float *score, float *score_flat,
                                          int64_t score_flat_stride, float *score_min, const void *weight2_local_x3,
                                          int const_in_block)
{
    if (batch_size < 0 || tokens < 0) return fail("salience_head_stage2: negative size");
    if (batch_size == 0 || tokens == 0) return 0;
    if (!z_local || !partial_sums || !weight2 || !bias2 || (!weight2_local_packed && !weight2_local_x3) || !weight3_packed ||
        !bias3 || !weight4 || !bias4 || (!const_workspace && !const_in_block) || !score)
        return fail("salience_head_stage2: NULL pointer");
    const int nblk = (tokens + kTM - 1) / kTM;
    const int prow = sdetr_salience_head_blocks(batch_size, tokens);
    if (const_in_block && prow > kConstInBlockRows)
        return fail("salience_head_stage2: the constant in the block takes up to %d rows of partial sums (got %d)", kConstInBlockRows, prow);
    hipStream_t s = static_cast<hipStream_t>(stream);
    Stage2Args a;
    if (const_in_block) {
        // (the caller's stage 1 / modulation launch has set *score_min to +inf)
        a.partial = partial_sums; a.partial_rows = prow; a.w2 = weight2; a.b2 = bias2;
    } else {
        hipLaunchKernelGGL(salience_head_const_kernel, dim3((unsigned)batch_size), dim3(kHalf * kConstGroups), 0, s,
                           partial_sums, prow, tokens, weight2, bias2, const_workspace, score_min);
        int rc = check_launch("salience_head_const");
        if (rc) return rc;
    }
    a.z_local = z_local; a.cst = const_workspace;
    a.w2a = reinterpret_cast<const float4 *>(weight2_local_packed);
    a.w2a_x3 = weight2_local_x3;
    a.w3 = reinterpret_cast<const float4 *>(weight3_packed);
    a.b3 = bias3; a.w4 = weight4; a.b4 = bias4; a.score = score; a.score2 = score_flat;
    a.score2_stride = score_flat_stride; a.n = tokens; a.score_min = score_min;
    hipLaunchKernelGGL(salience_head_stage2_kernel, dim3((unsigned)nblk, (unsigned)batch_size), dim3(kBlock), 0, s, a);
    return check_launch("salience_head_stage2");
}
