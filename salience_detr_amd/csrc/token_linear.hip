// Token-resident linear layers for 256-wide bf16 tokens:  y[t][n] = sum_k W[n][k] x[t][k] + b[n],  K = 256.
//
// Every projection of the encoder loop has this shape (value_proj of all six layers: N = 1536;
// sampling_offsets|attention_weights: N = 384; class head: N = 91) with M = 10^4..10^5 tokens.  A library GEMM
// treats them as generic [M,256]x[256,N] problems, writes y row-major, and a second kernel then re-lays / reduces
// it.  Here a wave keeps 32 tokens' activations X^T in 64 VGPRs (the B operands of v_mfma_f32_32x32x16_bf16) for
// the whole kernel and walks the output features in tiles of 32: Y^T[32 x 32] = W[tile] X^T is 16 MFMAs whose
// accumulator STARTS as the bias, and the epilogue consumes the tile straight from registers:
//
//   kStore      bf16 row-major store (lane (t,h) holds features 8g+4h+{0..3} of token t: 8-byte stores);
//               optional prologue x = x + x2 (query + position embedding, salience_transformer.py:380-381)
//   kHeadMajor  value_proj's tail (ms_deform_attn.py:316-321): zero the padded tokens, convert to fp16 / bf16 and
//               store head-major [layer][B][head][pixel][32] -- a 32-feature tile IS one (layer, head), so the
//               re-layout costs nothing and the [tokens, 1536] intermediate never exists
//   kClassMax   mc_score (salience_transformer.py:366): running max over the class logits, times the foreground
//               score -- the [tokens, 91] logits never exist
//
// Weights: pre-packed per tile into 16 lane-ordered 1 KB A-fragments (sdetr_linear_pack_bf16, rows past N are zero) and
// brought global -> registers -> LDS by four loader waves, four tiles (64 KB) at a time, double-buffered and shared by the
// block's compute waves; the 256 x 256 LayerNorm variant copies its whole weight once by LDS-DMA.  Measured
// (benchmarks/token_linear_sweep.py, hipGraph): ~7 us fixed + 0.6-0.7 us per 32-feature tile; MFMA bound 0.22 us.
#include "common.h"

#include "token_linear_core.h"

namespace sdetr {

// ---- y = LayerNorm(residual + W x + b), N = 256, optionally scattered ------------------------------------------
// The attention blocks' tails: output_proj + dropout/residual + norm1 of the deformable attention
// (salience_transformer.py:385-391) and out_proj + residual + pre_norm + scatter of the top-k dense attention
// (:376-379).  Same token-resident scheme; the eight output tiles stay in 128 accumulator registers and the
// epilogue (bias is the accumulator init; + residual, two-pass LayerNorm over the 256 channels a lane pair holds,
// bf16 store) runs in registers like the feed-forward kernel's.
struct TLNArgs {
    const bf16_t *x;           // [T, 256]
    const bf16_t *res;         // residual rows: rows_per_batch per image, images res_batch_stride elements apart
    int64_t res_batch_stride;
    int rows_per_batch;
    const char *pw;            // 8 packed tiles
    const float *bias, *gamma, *beta;
    float eps;
    bf16_t *out;               // [T,256], or [B, out_batch_rows, 256] with scatter_index
    const int64_t *scatter_index;   // [T] destination row inside the image, or NULL
    int64_t out_batch_rows;
    int T;
};

__global__ void __launch_bounds__(kBlock, 1) token_linear_ln_kernel(TLNArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *wbuf = lds;                                                   // 2 step buffers = all 8 tiles
    float *par = reinterpret_cast<float *>(lds + 2 * kTLStepBytes);     // bias | gamma | beta
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = lane & 31, h = lane >> 5;
    const int tok = blockIdx.x * kTLTokBlock + wave * kTLTokWave + t;
    const bool valid = tok < p.T;
    const int tk = valid ? tok : p.T - 1;
    const int img = tk / p.rows_per_batch, ri = tk - img * p.rows_per_batch;

    const uint32_t wbuf_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)wbuf;
    const uint32_t voff = (uint32_t)(wave * 16384 + lane * 16);
    const uint32_t wave_lds = wbuf_lds + wave * 16384;
    tl_issue_step(p.pw, voff, wave_lds);                                   // the whole 256 x 256 weight fits:
    tl_issue_step(p.pw + kTLStepBytes, voff, wave_lds + kTLStepBytes);     // no copy inside the MFMA loop
    par[tid] = p.bias[tid];
    par[kTLK + tid] = p.gamma[tid];
    par[2 * kTLK + tid] = p.beta[tid];

    uint4 xb[16];
    {
        const bf16_t *xr = p.x + (int64_t)tk * kTLK + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) xb[ks] = *reinterpret_cast<const uint4 *>(xr + 16 * ks);
    }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) asm volatile("" : "+v"(xb[ks].x), "+v"(xb[ks].y), "+v"(xb[ks].z), "+v"(xb[ks].w));

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    tl_f32x16_t acc[8];
    constexpr int R = 8;
    const tl_lds_cptr_t cb = (tl_lds_cptr_t)wbuf + lane * 16;   // 128 consecutive fragments: (tile nt, k-step ks) at 16 nt + ks
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
        const tl_lds_cptr_t bb = (tl_lds_cptr_t)(const char *)par + nt * 128 + 16 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint4 bv = tl_lds_read16(bb + 32 * g);
            acc[nt][4 * g] = __uint_as_float(bv.x);
            acc[nt][4 * g + 1] = __uint_as_float(bv.y);
            acc[nt][4 * g + 2] = __uint_as_float(bv.z);
            acc[nt][4 * g + 3] = __uint_as_float(bv.w);
        }
    }
    // the residual rows are requested before the MFMAs and arrive behind them
    const bf16_t *rr = p.res + (int64_t)img * p.res_batch_stride + (int64_t)ri * kTLK;
    uint2 rbuf[32];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) rbuf[nt * 4 + g] = *reinterpret_cast<const uint2 *>(rr + 32 * nt + 8 * g + 4 * h);
    // k-step-major over the eight tiles: two MFMAs into the same accumulator are eight issues apart (tile-major they
    // were chains of 16 dependent MFMAs, each waiting for the full latency of the one before)
    auto frag = [](int f) { return ((f & 7) * 16 + (f >> 3)) * 1024; };
    uint4 ring[R];
#pragma unroll
    for (int f = 0; f < R; ++f) ring[f] = tl_lds_read16(cb + frag(f));
#pragma unroll
    for (int f = 0; f < 128; ++f) {
        acc[f & 7] = tl_mfma(ring[f % R], xb[f >> 3], acc[f & 7]);
        if (f + R < 128) ring[f % R] = tl_lds_read16(cb + frag(f + R));
        __builtin_amdgcn_sched_barrier(0);
    }

    // + residual, LayerNorm over the 256 channels held by lanes (t, 0) and (t, 1), store.  The accumulators are only
    // READ (writing elements of a 16-register tuple makes the allocator copy tuples; with 128 + 64 registers live that
    // spills): pass 1 sums v = y + r and v^2, pass 2 recomputes v and writes (v - mean) * rstd * gamma + beta.
    float sum = 0.f, sq = 0.f;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint2 r = rbuf[nt * 4 + g];
            const float v0 = acc[nt][4 * g] + act_lo(r.x), v1 = acc[nt][4 * g + 1] + act_hi(r.x);
            const float v2 = acc[nt][4 * g + 2] + act_lo(r.y), v3 = acc[nt][4 * g + 3] + act_hi(r.y);
            sum += (v0 + v1) + (v2 + v3);
            sq = fmaf(v0, v0, fmaf(v1, v1, fmaf(v2, v2, fmaf(v3, v3, sq))));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    sum += __shfl_xor(sum, 32);
    sq += __shfl_xor(sq, 32);
    const float mean = sum * (1.f / kTLK);
    const float rstd = rsqrtf(fmaxf(sq * (1.f / kTLK) - mean * mean, 0.f) + p.eps);
    const float shift = -mean * rstd;
    if (valid) {
        const int64_t orow = p.scatter_index ? (int64_t)img * p.out_batch_rows + p.scatter_index[tok] : (int64_t)tok;
        bf16_t *o = p.out + orow * kTLK;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int e0 = 32 * nt + 8 * g + 4 * h;
                const float4 gv = *reinterpret_cast<const float4 *>(par + kTLK + e0);
                const float4 be = *reinterpret_cast<const float4 *>(par + 2 * kTLK + e0);
                const uint2 r = rbuf[nt * 4 + g];
                const float y0 = fmaf(fmaf(acc[nt][4 * g] + act_lo(r.x), rstd, shift), gv.x, be.x);
                const float y1 = fmaf(fmaf(acc[nt][4 * g + 1] + act_hi(r.x), rstd, shift), gv.y, be.y);
                const float y2 = fmaf(fmaf(acc[nt][4 * g + 2] + act_lo(r.y), rstd, shift), gv.z, be.z);
                const float y3 = fmaf(fmaf(acc[nt][4 * g + 3] + act_hi(r.y), rstd, shift), gv.w, be.w);
                *reinterpret_cast<uint2 *>(o + e0) = make_uint2(pack_act2(y0, y1), pack_act2(y2, y3));
            }
            __builtin_amdgcn_sched_barrier(0);   // one tile at a time: all 64 gamma / beta reads up front cost 256 registers
        }
    }
}

// packed[nt][ks][lane = h*32 + j][s] = W[32nt + j][16ks + 8h + s]   (zero for rows >= N)
__global__ void linear_pack_kernel(const bf16_t *w, int64_t row_stride, int N, int ntiles, bf16_t *out)
{
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= (int64_t)ntiles * 8192) return;
    const int nt = (int)(o >> 13), idx = (int)(o & 8191);
    const int s = idx & 7, l = (idx >> 3) & 63, ks = idx >> 9;
    const int n = 32 * nt + (l & 31);
    out[o] = n < N ? w[(int64_t)n * row_stride + 16 * ks + 8 * (l >> 5) + s] : (bf16_t)0;
}

template <int EPI, bool ADD2, int WAVES>
static int tl_launch_one(hipStream_t s, const TLArgs &a)
{
    const int nsteps = (a.ntiles + kTLStepTiles - 1) / kTLStepTiles;
    const size_t lds = 2 * (size_t)kTLStepBytes + (size_t)nsteps * 512;
    static DeviceOnce lds_once;   // (one set of flags per instantiation)
    allow_dynamic_lds(token_linear_kernel<EPI, ADD2, WAVES>, lds_once, 160 * 1024);
    const int tpb = kTLTokWave * WAVES;
    TLArgs b = a;
    b.hm_blocks = (a.T + tpb - 1) / tpb;
    hipLaunchKernelGGL((token_linear_kernel<EPI, ADD2, WAVES>), dim3((unsigned)b.hm_blocks), dim3(64 * (WAVES + 4)), lds, s, b);
    return check_launch("token_linear");
}

static int tl_launch(hipStream_t s, int epi, bool add2, TLArgs &a)
{
    const bool wide = a.T > 256 * kTLTokBlock;   // more 128-token blocks than CUs: two compute waves per SIMD instead
    if (epi == kStore && add2) return tl_launch_one<kStore, true, 4>(s, a);
    if (epi == kStore) return wide ? tl_launch_one<kStore, false, 8>(s, a) : tl_launch_one<kStore, false, 4>(s, a);
    if (epi == kHeadMajor) return wide ? tl_launch_one<kHeadMajor, false, 8>(s, a) : tl_launch_one<kHeadMajor, false, 4>(s, a);
    return tl_launch_one<kClassMax, false, 4>(s, a);
}

static int tl_common(TLArgs &a, const void *x, const void *packed, const float *bias, int tokens, int in_features,
                     int out_features)
{
    if (in_features != kTLK) return fail("token_linear: built for 256 input features (got %d)", in_features);
    if (tokens < 0 || out_features <= 0) return fail("token_linear: bad sizes");
    if (!x || !packed || !bias) return fail("token_linear: null pointer");
    a = TLArgs{};
    a.x = (const bf16_t *)x; a.pw = (const char *)packed; a.bias = bias; a.T = tokens; a.N = out_features;
    a.ntiles = (out_features + 31) / 32; a.rows_per_batch = tokens > 0 ? tokens : 1;
    if ((size_t)a.ntiles * 128 + 2 * kTLStepBytes + 1024 > 160 * 1024) return fail("token_linear: too many output features");
    return 0;
}

}  // namespace sdetr

using namespace sdetr;

extern "C" int64_t sdetr_linear_packed_bytes(int out_features)
{
    return out_features > 0 ? (int64_t)((out_features + 127) / 128) * kTLStepBytes : 0;
}

extern "C" int sdetr_linear_pack_bf16(sdetr_stream_t stream, const void *weight, int64_t row_stride, int out_features,
                                      int in_features, void *packed)
{
    if (in_features != kTLK) return fail("linear_pack: built for 256 input features (got %d)", in_features);
    if (out_features <= 0 || !weight || !packed || row_stride < kTLK) return fail("linear_pack: bad arguments");
    const int ntiles = (out_features + 127) / 128 * kTLStepTiles;   // whole steps (zero rows past out_features)
    const int64_t total = (int64_t)ntiles * 8192;
    hipLaunchKernelGGL(linear_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), (const bf16_t *)weight, row_stride, out_features, ntiles,
                       (bf16_t *)packed);
    return check_launch("linear_pack");
}

extern "C" int sdetr_token_linear_bf16(sdetr_stream_t stream, const void *x, const void *x_add, int64_t x_add_batch_stride,
                                       int rows_per_batch, int tokens, int in_features, const void *packed_weight,
                                       const float *bias_padded, int out_features, void *out, int64_t out_row_stride,
                                       int group_features)
{
    TLArgs a;
    if (int rc = tl_common(a, x, packed_weight, bias_padded, tokens, in_features, out_features)) return rc;
    if (tokens == 0) return 0;
    if (!out || (out_features % 4) || out_row_stride < out_features || (out_row_stride % 4))
        return fail("token_linear: out_features and the output row stride must be multiples of 4");
    if (x_add) {
        if (rows_per_batch <= 0 || tokens % rows_per_batch || x_add_batch_stride < (int64_t)rows_per_batch * kTLK ||
            (x_add_batch_stride % 8))
            return fail("token_linear: bad addend layout");
        a.x2 = (const bf16_t *)x_add; a.x2_batch_stride = x_add_batch_stride; a.rows_per_batch = rows_per_batch;
    }
    if (group_features < 0 || (group_features > 0 && ((group_features % 4) || out_features % group_features ||
                                                      rows_per_batch <= 0 || tokens % rows_per_batch)))
        return fail("token_linear: bad feature grouping");
    if (group_features > 0) a.rows_per_batch = rows_per_batch;
    a.out = (bf16_t *)out; a.out_row_stride = out_row_stride; a.group = group_features;
    return tl_launch(static_cast<hipStream_t>(stream), kStore, x_add != nullptr, a);
}

extern "C" int sdetr_value_proj_head_major(sdetr_stream_t stream, const void *x, const void *packed_weight,
                                           const float *bias_padded, const uint8_t *pad_mask, int batch_size,
                                           int spatial_size, int in_features, int num_heads, int channels,
                                           int num_groups, void *dst, int dst_dtype,
                                           const sdetr_bordered_layout *bordered)
{
    if (channels != 32) return fail("value_proj_head_major: built for 32 channels per head (got %d)", channels);
    if (batch_size < 0 || spatial_size < 0 || num_heads <= 0 || num_groups <= 0) return fail("value_proj_head_major: bad sizes");
    if (dst_dtype != SDETR_F16 && dst_dtype != SDETR_BF16) return fail("value_proj_head_major: dst must be fp16 or bf16");
    TLArgs a;
    const int64_t tokens = (int64_t)batch_size * spatial_size;
    if (tokens > 0x7fffffff) return fail("value_proj_head_major: too many tokens");
    if (int rc = tl_common(a, x, packed_weight, bias_padded, (int)tokens, in_features, num_groups * num_heads * 32)) return rc;
    if (tokens == 0) return 0;
    if (!dst) return fail("value_proj_head_major: null pointer");
    a.rows_per_batch = spatial_size; a.pad = pad_mask; a.hm = dst; a.heads = num_heads; a.batch = batch_size;
    a.hm_f16 = dst_dtype == SDETR_F16;
    if (int rc = tl_set_bordered(a, bordered, spatial_size, 1)) return rc;   // (the launch sets the block count)
    return tl_launch(static_cast<hipStream_t>(stream), kHeadMajor, false, a);
}

extern "C" int sdetr_class_head_max_times(sdetr_stream_t stream, const void *x, const void *packed_weight,
                                          const float *bias_padded, int in_features, int num_classes, const float *scale,
                                          int64_t scale_batch_stride, int batch_size, int rows_per_batch, float *out)
{
    if (batch_size < 0 || rows_per_batch < 0) return fail("class_head_max_times: bad sizes");
    TLArgs a;
    const int64_t tokens = (int64_t)batch_size * rows_per_batch;
    if (tokens > 0x7fffffff) return fail("class_head_max_times: too many tokens");
    if (int rc = tl_common(a, x, packed_weight, bias_padded, (int)tokens, in_features, num_classes)) return rc;
    if (tokens == 0) return 0;
    if (!scale || !out || scale_batch_stride < rows_per_batch) return fail("class_head_max_times: bad scale / out");
    a.rows_per_batch = rows_per_batch; a.scale = scale; a.scale_batch_stride = scale_batch_stride; a.cmax = out;
    return tl_launch(static_cast<hipStream_t>(stream), kClassMax, false, a);
}

extern "C" int sdetr_token_linear_ln_bf16(sdetr_stream_t stream, const void *x, const void *residual,
                                          int64_t residual_batch_stride, int rows_per_batch, int tokens, int in_features,
                                          const void *packed_weight, const float *bias, const float *norm_weight,
                                          const float *norm_bias, float norm_eps, void *out, const int64_t *scatter_index,
                                          int64_t out_batch_rows)
{
    if (in_features != kTLK) return fail("token_linear_ln: built for 256 features (got %d)", in_features);
    if (tokens < 0 || rows_per_batch <= 0) return fail("token_linear_ln: bad sizes");
    if (tokens == 0) return 0;
    if (!x || !residual || !packed_weight || !bias || !norm_weight || !norm_bias || !out)
        return fail("token_linear_ln: null pointer");
    if (tokens % rows_per_batch || residual_batch_stride < (int64_t)rows_per_batch * kTLK || (residual_batch_stride % 4))
        return fail("token_linear_ln: bad residual layout");
    if (scatter_index && out_batch_rows <= 0) return fail("token_linear_ln: scatter needs the rows per image of out");
    TLNArgs a{};
    a.x = (const bf16_t *)x; a.res = (const bf16_t *)residual; a.res_batch_stride = residual_batch_stride;
    a.rows_per_batch = rows_per_batch; a.pw = (const char *)packed_weight; a.bias = bias; a.gamma = norm_weight;
    a.beta = norm_bias; a.eps = norm_eps; a.out = (bf16_t *)out; a.scatter_index = scatter_index;
    a.out_batch_rows = out_batch_rows; a.T = tokens;
    const size_t lds = 2 * (size_t)kTLStepBytes + 3 * kTLK * 4;
    static DeviceOnce lds_once1;
    allow_dynamic_lds(token_linear_ln_kernel, lds_once1, 160 * 1024);
    hipLaunchKernelGGL(token_linear_ln_kernel, dim3((unsigned)((tokens + kTLTokBlock - 1) / kTLTokBlock)), dim3(kBlock),
                       lds, static_cast<hipStream_t>(stream), a);
    return check_launch("token_linear_ln");
}
