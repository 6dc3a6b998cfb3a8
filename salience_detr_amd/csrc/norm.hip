// Fused (residual add | salience modulation) + LayerNorm, and the token-mean of the salience head.
//
//  * layernorm_kernel:   y = LN((x [+ r]) * (1 + s[row] * alpha)) * gamma + beta
//      - encoder layer norms with their residual adds (models/bricks/salience_transformer.py:377-378,
//        390-391, 347-351: norm(query + sublayer(query))),
//      - the level modulation "level_memory + level_memory * upsample_score * alpha[level]" followed by the
//        MaskPredictor's LayerNorm (:143 and :20),
//      - plain LayerNorm when r and s are NULL (enc_output_norm, base_transformer.py:111).
//    With scatter_index the normalised row i of image b goes to row scatter_index[b,i] of out [B, out_batch_rows, C]
//    (the top-k self-attention block's norm + scatter back into the layer's queries, :377-379).
//    HBM-bound: each row is read once and written once.  C/8 lanes own one row (8 channels = one 16/32-byte
//    vector per lane), statistics in fp32 with a two-pass (mean, then centred variance) reduction done with
//    wavefront shuffles -- no LDS, no barrier.  The framework path costs an add kernel plus a LayerNorm kernel
//    that reaches ~0.9 TB/s on these [22 726 x 256] bf16 activations.
//  * column_mean_kernel: mean over the tokens of one level of the "global" half of the salience head's hidden
//    state (salience_transformer.py:43-45); deterministic (fixed summation order, no atomics) so that the
//    salience scores -- and therefore the selected token sets -- are reproducible run to run.
#include "common.h"

namespace sdetr {

struct NormArgs {
    const void *x;          // [B, n, C] with strides
    const void *res;        // same layout as x, or NULL
    const float *row_scale; // [B*n] or NULL
    const float *alpha;     // device scalar or NULL (treated as 1)
    const void *gamma;
    const void *beta;
    void *out;              // [B*n, C] contiguous, or [B, out_batch_rows, C] when scatter_index is given
    const int64_t *scatter_index;  // [B*n] destination row inside the image, or NULL
    int64_t out_batch_rows;
    int gather_x;                  // 1: x row i is read at scatter_index[b,i] as well (in-place update of selected rows)
    int64_t x_batch_stride, x_row_stride, res_batch_stride, res_row_stride;
    int64_t rows;
    int n_per_batch, C;
    float eps;
};

template <typename T>
__device__ __forceinline__ void load8(const T *p, float *v);
template <>
__device__ __forceinline__ void load8<float>(const float *p, float *v)
{
    const float4 a = reinterpret_cast<const float4 *>(p)[0], b = reinterpret_cast<const float4 *>(p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <>
__device__ __forceinline__ void load8<bf16_t>(const bf16_t *p, float *v)
{
    const uint4 a = *reinterpret_cast<const uint4 *>(p);
    v[0] = act_lo(a.x); v[1] = act_hi(a.x); v[2] = act_lo(a.y); v[3] = act_hi(a.y);
    v[4] = act_lo(a.z); v[5] = act_hi(a.z); v[6] = act_lo(a.w); v[7] = act_hi(a.w);
}
template <typename T>
__device__ __forceinline__ void store8(T *p, const float *v);
template <>
__device__ __forceinline__ void store8<float>(float *p, const float *v)
{
    reinterpret_cast<float4 *>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<float4 *>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
}
template <>
__device__ __forceinline__ void store8<bf16_t>(bf16_t *p, const float *v)
{
    *reinterpret_cast<uint4 *>(p) = make_uint4(pack_act2(v[0], v[1]), pack_act2(v[2], v[3]),
                                              pack_act2(v[4], v[5]), pack_act2(v[6], v[7]));
}

// G = lanes per row (C = 8 * G * K with K register rounds; here K = 1: C <= 512)
template <typename XT, typename PT, typename OT, int G>
__global__ void __launch_bounds__(kBlock) layernorm_kernel(NormArgs p)
{
    constexpr int RPB = kBlock / G;
    const int tid = threadIdx.x;
    const int64_t row = (int64_t)blockIdx.x * RPB + tid / G;
    const int l = tid % G;
    const bool live = row < p.rows && l * 8 < p.C;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.f;
    const int64_t rr = row < p.rows ? row : 0;
    const int64_t b = rr / p.n_per_batch, i = rr - b * p.n_per_batch;
    if (live) {
        const int64_t xi = p.gather_x ? p.scatter_index[rr] : i;
        load8<XT>(reinterpret_cast<const XT *>(p.x) + b * p.x_batch_stride + xi * p.x_row_stride + l * 8, v);
        if (p.res) {
            float r[8];
            load8<XT>(reinterpret_cast<const XT *>(p.res) + b * p.res_batch_stride + i * p.res_row_stride + l * 8, r);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += r[k];
        }
        if (p.row_scale) {
            // reference order of operations: mem + mem * up * alpha
            const float s = p.row_scale[rr], a = p.alpha ? *p.alpha : 1.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = v[k] + v[k] * s * a;
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += v[k];
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, G);
    const float mean = sum / (float)p.C;
    float sq = 0.f;
    if (live) {
#pragma unroll
        for (int k = 0; k < 8; ++k) sq += (v[k] - mean) * (v[k] - mean);
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, G);
    const float rstd = rsqrtf(sq / (float)p.C + p.eps);
    if (live) {
        float g[8], be[8], y[8];
        load8<PT>(reinterpret_cast<const PT *>(p.gamma) + l * 8, g);
        load8<PT>(reinterpret_cast<const PT *>(p.beta) + l * 8, be);
#pragma unroll
        for (int k = 0; k < 8; ++k) y[k] = (v[k] - mean) * rstd * g[k] + be[k];
        const int64_t orow = p.scatter_index ? b * p.out_batch_rows + p.scatter_index[row] : row;
        store8<OT>(reinterpret_cast<OT *>(p.out) + orow * p.C + l * 8, y);
    }
}

// x [B, n, C] (strides) -> out [B, C] = mean over n.  grid (C/16, B), 256 threads: 4 lanes x float4 cover 16
// columns, 64 row-groups stride over n; fixed-order tree in LDS.
__global__ void __launch_bounds__(kBlock) column_mean_kernel(const float *x, int64_t batch_stride, int64_t row_stride,
                                                             int n, int C, float *out)
{
    __shared__ float4 part[64][4];
    const int tid = threadIdx.x;
    const int cl = tid & 3, rg = tid >> 2;
    const int c0 = blockIdx.x * 16 + cl * 4;
    const float *xb = x + (int64_t)blockIdx.y * batch_stride + c0;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c0 < C) {
        int i = rg;
        for (; i + 3 * 64 < n; i += 4 * 64) {  // 4 independent loads in flight
            const float4 a0 = *reinterpret_cast<const float4 *>(xb + (int64_t)i * row_stride);
            const float4 a1 = *reinterpret_cast<const float4 *>(xb + (int64_t)(i + 64) * row_stride);
            const float4 a2 = *reinterpret_cast<const float4 *>(xb + (int64_t)(i + 128) * row_stride);
            const float4 a3 = *reinterpret_cast<const float4 *>(xb + (int64_t)(i + 192) * row_stride);
            acc.x += (a0.x + a1.x) + (a2.x + a3.x);
            acc.y += (a0.y + a1.y) + (a2.y + a3.y);
            acc.z += (a0.z + a1.z) + (a2.z + a3.z);
            acc.w += (a0.w + a1.w) + (a2.w + a3.w);
        }
        for (; i < n; i += 64) {
            const float4 a = *reinterpret_cast<const float4 *>(xb + (int64_t)i * row_stride);
            acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
        }
    }
    part[rg][cl] = acc;
    __syncthreads();
    for (int s = 32; s > 0; s >>= 1) {
        if (rg < s) {
            const float4 o = part[rg + s][cl];
            float4 m = part[rg][cl];
            m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w;
            part[rg][cl] = m;
        }
        __syncthreads();
    }
    if (rg == 0 && c0 < C) {
        const float4 m = part[0][cl];
        const float inv = 1.f / (float)n;
        *reinterpret_cast<float4 *>(out + (int64_t)blockIdx.y * C + c0) = make_float4(m.x * inv, m.y * inv, m.z * inv, m.w * inv);
    }
}

template <typename XT, typename PT, typename OT>
static int launch_ln(hipStream_t stream, NormArgs &a)
{
    const int G = a.C <= 64 ? 8 : a.C <= 128 ? 16 : a.C <= 256 ? 32 : 64;
    const int rpb = kBlock / G;
    const dim3 grid((unsigned)((a.rows + rpb - 1) / rpb)), block(kBlock);
    switch (G) {
        case 8: hipLaunchKernelGGL((layernorm_kernel<XT, PT, OT, 8>), grid, block, 0, stream, a); break;
        case 16: hipLaunchKernelGGL((layernorm_kernel<XT, PT, OT, 16>), grid, block, 0, stream, a); break;
        case 32: hipLaunchKernelGGL((layernorm_kernel<XT, PT, OT, 32>), grid, block, 0, stream, a); break;
        default: hipLaunchKernelGGL((layernorm_kernel<XT, PT, OT, 64>), grid, block, 0, stream, a); break;
    }
    return check_launch("layernorm");
}

}  // namespace sdetr

using namespace sdetr;

extern "C" int sdetr_layernorm(sdetr_stream_t stream, const void *x, const void *residual, int x_dtype,
                               int64_t x_batch_stride, int64_t x_row_stride, int64_t res_batch_stride,
                               int64_t res_row_stride, const float *row_scale, const float *alpha, const void *gamma,
                               const void *beta, int param_dtype, float eps, int batch_size, int rows_per_batch,
                               int channels, void *out, int out_dtype, const int64_t *scatter_index,
                               int64_t out_batch_rows, int gather_x)
{
    if (gather_x && !scatter_index) return fail("layernorm: gather_x needs the row index");
    if (batch_size < 0 || rows_per_batch < 0 || channels <= 0) return fail("layernorm: bad dims");
    if (channels % 8 != 0 || channels > 512) return fail("layernorm: channels (%d) must be a multiple of 8, <= 512", channels);
    if ((x_row_stride % 8) || (x_batch_stride % 8) || (residual && ((res_row_stride % 8) || (res_batch_stride % 8))))
        return fail("layernorm: strides must be multiples of 8 elements");
    const int64_t rows = (int64_t)batch_size * rows_per_batch;
    if (rows == 0) return 0;
    if (!x || !gamma || !beta || !out) return fail("layernorm: null pointer");
    NormArgs a{};
    a.x = x; a.res = residual; a.row_scale = row_scale; a.alpha = alpha; a.gamma = gamma; a.beta = beta; a.out = out;
    a.x_batch_stride = x_batch_stride; a.x_row_stride = x_row_stride;
    a.res_batch_stride = res_batch_stride; a.res_row_stride = res_row_stride;
    a.rows = rows; a.n_per_batch = rows_per_batch; a.C = channels; a.eps = eps;
    a.scatter_index = scatter_index; a.out_batch_rows = out_batch_rows; a.gather_x = gather_x ? 1 : 0;
    // dtype codes -> 0 (fp32) / 1 (the library's 16-bit activation type: SDETR_BF16 here, SDETR_F16 in the fp16 flavour)
    auto bit = [](int dt) { return dt == SDETR_F32 ? 0 : (dt == kActCode ? 1 : -64); };
    const int key = bit(x_dtype) * 4 + bit(param_dtype) * 2 + bit(out_dtype);
    switch (key) {
        case 0: return launch_ln<float, float, float>(stream, a);
        case 1: return launch_ln<float, float, bf16_t>(stream, a);
        case 2: return launch_ln<float, bf16_t, float>(stream, a);
        case 3: return launch_ln<float, bf16_t, bf16_t>(stream, a);
        case 4: return launch_ln<bf16_t, float, float>(stream, a);
        case 5: return launch_ln<bf16_t, float, bf16_t>(stream, a);
        case 6: return launch_ln<bf16_t, bf16_t, float>(stream, a);
        case 7: return launch_ln<bf16_t, bf16_t, bf16_t>(stream, a);
        default: return fail("layernorm: bad dtypes");
    }
}

extern "C" int sdetr_column_mean_f32(sdetr_stream_t stream, const float *x, int64_t batch_stride, int64_t row_stride,
                                     int batch_size, int rows, int channels, float *out)
{
    if (batch_size < 0 || rows <= 0 || channels <= 0 || (channels % 4)) return fail("column_mean: bad dims");
    if ((row_stride % 4) || (batch_stride % 4)) return fail("column_mean: strides must be multiples of 4 elements");
    if (batch_size == 0) return 0;
    if (!x || !out) return fail("column_mean: null pointer");
    hipLaunchKernelGGL(column_mean_kernel, dim3((unsigned)((channels + 15) / 16), (unsigned)batch_size), dim3(kBlock), 0,
                       stream, x, batch_stride, row_stride, rows, channels, out);
    return check_launch("column_mean");
}
