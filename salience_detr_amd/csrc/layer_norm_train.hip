// LayerNorm for the TRAINING step (fp32): y = LN(a [+ b]) * gamma + beta with what the backward needs, and its backward in
// one launch (models/bricks/salience_transformer.py:347-351, 377-378, 390-391: norm(query + sublayer(query)) -- three per
// encoder layer, three per decoder layer, plus the salience head's and enc_output_norm).
//
// The framework runs  add -> native_layer_norm  forward and  layer_norm_grad_input + two gamma/beta reduction kernels
// -> add (the residual's second consumer)  backward: ~1.3 ms of the 18.5 ms step over ~60 calls.  Here:
//   forward  : one pass -- reads a (and b), writes z = a + b (only when there is a b: the backward needs the normalised
//              input), y, and the row statistics (mean, 1 / std).  C / 8 lanes own a row, two-pass statistics in
//              registers with wavefront shuffles (as csrc/norm.hip).
//   backward : one pass -- reads dy, z, the statistics, gamma; writes dz (= the gradient of a AND of b) and accumulates
//              d gamma / d beta: a workgroup walks 16 row slices, every thread keeps the sums of its 8 channels, the
//              workgroup's row groups meet in LDS and one fp32 atomic per channel goes out (2 x C per 128-256 rows).
// fp32, C a multiple of 8 up to 512 (C / 8 must divide 64: 64, 128, 256, 512 channels).
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "common.h"

namespace sdetr {

constexpr int kLnThreads = 256;
constexpr int kLnSlices = 16;   // row slices a workgroup of the backward walks

struct LnTrainArgs {
    const float *a, *b;         // [rows, C]; b may be NULL
    const float *gamma, *beta;
    float *z, *y;               // z: [rows, C] (NULL when b is NULL), y: [rows, C]
    float *mean, *rstd;         // [rows]
    int64_t rows;
    int C;
    float eps;
};

struct LnTrainBwdArgs {
    const float *dy, *z, *mean, *rstd, *gamma;
    float *dz;                  // [rows, C]
    float *dgamma, *dbeta;      // [C], accumulated with atomics (zero on entry)
    int64_t rows;
    int C;
};

template <int G>
__device__ __forceinline__ float group_sum(float v)
{
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, G);
    return v;
}

template <int G>
__global__ void __launch_bounds__(kLnThreads) ln_train_fwd_kernel(LnTrainArgs p)
{
    constexpr int RPB = kLnThreads / G;
    const int tid = threadIdx.x, l = tid % G;
    const int64_t row = (int64_t)blockIdx.x * RPB + tid / G;
    if (row >= p.rows) return;   // (whole row groups leave together: G divides the wavefront)
    const float *pa = p.a + row * p.C + 8 * l;
    float v[8];
    {
        const float4 x0 = reinterpret_cast<const float4 *>(pa)[0], x1 = reinterpret_cast<const float4 *>(pa)[1];
        v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
    }
    if (p.b) {
        const float *pb = p.b + row * p.C + 8 * l;
        const float4 r0 = reinterpret_cast<const float4 *>(pb)[0], r1 = reinterpret_cast<const float4 *>(pb)[1];
        v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
        float *pz = p.z + row * p.C + 8 * l;
        reinterpret_cast<float4 *>(pz)[0] = make_float4(v[0], v[1], v[2], v[3]);
        reinterpret_cast<float4 *>(pz)[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
    const float inv_c = 1.0f / (float)p.C;
    const float mean = group_sum<G>(((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) * inv_c;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] -= mean; sq = fmaf(v[i], v[i], sq); }
    const float rstd = rsqrtf(group_sum<G>(sq) * inv_c + p.eps);
    const float4 g0 = reinterpret_cast<const float4 *>(p.gamma + 8 * l)[0], g1 = reinterpret_cast<const float4 *>(p.gamma + 8 * l)[1];
    const float4 b0 = reinterpret_cast<const float4 *>(p.beta + 8 * l)[0], b1 = reinterpret_cast<const float4 *>(p.beta + 8 * l)[1];
    float *py = p.y + row * p.C + 8 * l;
    reinterpret_cast<float4 *>(py)[0] = make_float4(fmaf(v[0] * rstd, g0.x, b0.x), fmaf(v[1] * rstd, g0.y, b0.y),
                                                    fmaf(v[2] * rstd, g0.z, b0.z), fmaf(v[3] * rstd, g0.w, b0.w));
    reinterpret_cast<float4 *>(py)[1] = make_float4(fmaf(v[4] * rstd, g1.x, b1.x), fmaf(v[5] * rstd, g1.y, b1.y),
                                                    fmaf(v[6] * rstd, g1.z, b1.z), fmaf(v[7] * rstd, g1.w, b1.w));
    if (l == 0) { p.mean[row] = mean; p.rstd[row] = rstd; }
}

template <int G>
__global__ void __launch_bounds__(kLnThreads) ln_train_bwd_kernel(LnTrainBwdArgs p)
{
    constexpr int RPB = kLnThreads / G;
    __shared__ float red[2][RPB][8 * G];
    const int tid = threadIdx.x, l = tid % G, rg = tid / G;
    const float4 g0 = reinterpret_cast<const float4 *>(p.gamma + 8 * l)[0], g1 = reinterpret_cast<const float4 *>(p.gamma + 8 * l)[1];
    const float gam[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    const float inv_c = 1.0f / (float)p.C;
    float dg[8], db[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { dg[i] = 0.f; db[i] = 0.f; }
    const int64_t row0 = (int64_t)blockIdx.x * (RPB * kLnSlices) + rg;
#pragma unroll 2
    for (int s = 0; s < kLnSlices; ++s) {
        const int64_t row = row0 + (int64_t)s * RPB;
        if (row >= p.rows) break;   // (uniform per row group; nothing below synchronises the wavefront across groups)
        const float *pd = p.dy + row * p.C + 8 * l, *pz = p.z + row * p.C + 8 * l;
        const float4 d0 = reinterpret_cast<const float4 *>(pd)[0], d1 = reinterpret_cast<const float4 *>(pd)[1];
        const float4 z0 = reinterpret_cast<const float4 *>(pz)[0], z1 = reinterpret_cast<const float4 *>(pz)[1];
        const float mean = p.mean[row], rstd = p.rstd[row];
        const float d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
        const float z[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
        float xh[8], gy[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            xh[i] = (z[i] - mean) * rstd;
            gy[i] = d[i] * gam[i];
            s1 += gy[i];
            s2 = fmaf(gy[i], xh[i], s2);
            dg[i] = fmaf(d[i], xh[i], dg[i]);
            db[i] += d[i];
        }
        s1 = group_sum<G>(s1) * inv_c;
        s2 = group_sum<G>(s2) * inv_c;
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (gy[i] - s1 - xh[i] * s2) * rstd;
        float *po = p.dz + row * p.C + 8 * l;
        reinterpret_cast<float4 *>(po)[0] = make_float4(o[0], o[1], o[2], o[3]);
        reinterpret_cast<float4 *>(po)[1] = make_float4(o[4], o[5], o[6], o[7]);
    }
    // the workgroup's row groups meet in LDS; one thread per channel adds its sums once
#pragma unroll
    for (int i = 0; i < 8; ++i) { red[0][rg][8 * l + i] = dg[i]; red[1][rg][8 * l + i] = db[i]; }
    __syncthreads();
    for (int c = tid; c < p.C; c += kLnThreads) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int r = 0; r < RPB; ++r) { a += red[0][r][c]; b += red[1][r][c]; }
        unsafeAtomicAdd(p.dgamma + c, a);
        unsafeAtomicAdd(p.dbeta + c, b);
    }
}

}  // namespace sdetr

using namespace sdetr;

static int ln_group(int C) { return (C == 64 || C == 128 || C == 256 || C == 512) ? C / 8 : 0; }

extern "C" int sdetr_layer_norm_train_supported(int channels) { return ln_group(channels) ? 1 : 0; }

extern "C" int sdetr_layer_norm_train_forward_f32(sdetr_stream_t stream, const float *a, const float *residual, const float *gamma,
                                                  const float *beta, float eps, int64_t rows, int channels, float *sum_out,
                                                  float *out, float *mean, float *rstd)
{
    const int G = ln_group(channels);
    if (!G) return fail("layer_norm_train: 64 / 128 / 256 / 512 channels (got %d)", channels);
    if (rows < 0) return fail("layer_norm_train: negative row count");
    if (rows == 0) return 0;
    if (!a || !gamma || !beta || !out || !mean || !rstd) return fail("layer_norm_train: null pointer");
    if (residual && !sum_out) return fail("layer_norm_train: a residual needs sum_out (the backward reads a + residual)");
    LnTrainArgs p{};
    p.a = a; p.b = residual; p.gamma = gamma; p.beta = beta; p.z = sum_out; p.y = out; p.mean = mean; p.rstd = rstd;
    p.rows = rows; p.C = channels; p.eps = eps;
    const int rpb = kLnThreads / G;
    const dim3 grid((unsigned)((rows + rpb - 1) / rpb));
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (G) {
        case 8: hipLaunchKernelGGL(ln_train_fwd_kernel<8>, grid, dim3(kLnThreads), 0, s, p); break;
        case 16: hipLaunchKernelGGL(ln_train_fwd_kernel<16>, grid, dim3(kLnThreads), 0, s, p); break;
        case 32: hipLaunchKernelGGL(ln_train_fwd_kernel<32>, grid, dim3(kLnThreads), 0, s, p); break;
        default: hipLaunchKernelGGL(ln_train_fwd_kernel<64>, grid, dim3(kLnThreads), 0, s, p); break;
    }
    return check_launch("layer_norm_train_forward");
}

extern "C" int sdetr_layer_norm_train_backward_f32(sdetr_stream_t stream, const float *grad_out, const float *normalized_input,
                                                   const float *mean, const float *rstd, const float *gamma, int64_t rows,
                                                   int channels, float *grad_input, float *grad_gamma, float *grad_beta)
{
    const int G = ln_group(channels);
    if (!G) return fail("layer_norm_train: 64 / 128 / 256 / 512 channels (got %d)", channels);
    if (rows < 0) return fail("layer_norm_train: negative row count");
    if (rows == 0) return 0;
    if (!grad_out || !normalized_input || !mean || !rstd || !gamma || !grad_input || !grad_gamma || !grad_beta)
        return fail("layer_norm_train: null pointer");
    LnTrainBwdArgs p{};
    p.dy = grad_out; p.z = normalized_input; p.mean = mean; p.rstd = rstd; p.gamma = gamma; p.dz = grad_input;
    p.dgamma = grad_gamma; p.dbeta = grad_beta; p.rows = rows; p.C = channels;
    const int per_block = (kLnThreads / G) * kLnSlices;
    const dim3 grid((unsigned)((rows + per_block - 1) / per_block));
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (G) {
        case 8: hipLaunchKernelGGL(ln_train_bwd_kernel<8>, grid, dim3(kLnThreads), 0, s, p); break;
        case 16: hipLaunchKernelGGL(ln_train_bwd_kernel<16>, grid, dim3(kLnThreads), 0, s, p); break;
        case 32: hipLaunchKernelGGL(ln_train_bwd_kernel<32>, grid, dim3(kLnThreads), 0, s, p); break;
        default: hipLaunchKernelGGL(ln_train_bwd_kernel<64>, grid, dim3(kLnThreads), 0, s, p); break;
    }
    return check_launch("layer_norm_train_backward");
}
