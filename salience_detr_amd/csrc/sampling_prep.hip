// Sampling locations and attention weights of the deformable attention for the TRAINING step, one launch each way
// (models/bricks/ms_deform_attn.py:322-349): from the two Linear outputs (offsets [B,Nq,M*L*P*2], logits [B,Nq,M*L*P]) and
// the reference points,
//     weights   = softmax over the L*P logits of a (query, head)
//     locations = ref_xy + offset / (W_l, H_l)                      (2-d reference points, encoder)
//               = ref_xy + offset / P * ref_wh * 0.5                (4-d reference boxes, decoder)
// and backward the gradients of the two Linear outputs from the op's grad_sampling_loc / grad_attn_weight.  Under autograd
// the framework runs a softmax, a division, a broadcast add (+ views) forward and their three backward kernels on
// [B,Nq,8,4,4,2] tensors per layer: ~0.6 ms of the 18 ms step.  One thread per (query, head): 2 * L * P offsets and
// L * P logits in registers (built for L = P = 4, the reference's configuration).
#include "common.h"

namespace sdetr {

struct SamplingPrepArgs {
    const float *offsets;   // [rows, M, L, P, 2]   rows = B * Nq
    const float *logits;    // [rows, M, L * P]
    const float *ref;       // [rows, L, RD]
    const int64_t *shapes;  // [L, 2] (h, w), device
    float *loc;             // [rows, M, L, P, 2]
    float *weights;         // [rows, M, L, P]
    int64_t rows;
    int M, L, P, RD;
};

struct SamplingPrepBwdArgs {
    const float *grad_loc, *grad_w, *weights, *ref;
    const int64_t *shapes;
    float *grad_offsets, *grad_logits;
    int64_t rows;
    int M, L, P, RD;
};

// (L and P are compile-time: the per-thread arrays stay in registers and every access is a 16-byte vector)
template <int RD, int L, int P>
__global__ void __launch_bounds__(256) sampling_prep_kernel(SamplingPrepArgs p)
{
    constexpr int S = L * P;
    static_assert(S % 4 == 0, "16-byte vectors");
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= p.rows * p.M) return;
    const int64_t row = t / p.M;
    float e[S];
    const float4 *lg = reinterpret_cast<const float4 *>(p.logits + t * S);
#pragma unroll
    for (int i = 0; i < S / 4; ++i) {
        const float4 v = lg[i];
        e[4 * i] = v.x; e[4 * i + 1] = v.y; e[4 * i + 2] = v.z; e[4 * i + 3] = v.w;
    }
    float mx = e[0];
#pragma unroll
    for (int s = 1; s < S; ++s) mx = fmaxf(mx, e[s]);
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) { e[s] = expf(e[s] - mx); sum += e[s]; }
    const float inv = 1.0f / sum;
    float4 *w = reinterpret_cast<float4 *>(p.weights + t * S);
#pragma unroll
    for (int i = 0; i < S / 4; ++i) w[i] = make_float4(e[4 * i] * inv, e[4 * i + 1] * inv, e[4 * i + 2] * inv, e[4 * i + 3] * inv);
    const float4 *off = reinterpret_cast<const float4 *>(p.offsets + t * S * 2);
    float4 *loc = reinterpret_cast<float4 *>(p.loc + t * S * 2);
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const float *r = p.ref + (row * L + l) * RD;
        const float rx = r[0], ry = r[1];
        float dx, dy;   // 2-d: the level's (W, H), divided by as the reference does; 4-d: the box's (w, h)
        if (RD == 2) { dx = (float)p.shapes[2 * l + 1]; dy = (float)p.shapes[2 * l]; }
        else { dx = r[2]; dy = r[3]; }
#pragma unroll
        for (int k = 0; k < P / 2; ++k) {   // two samples (x, y, x, y) per vector
            const float4 o = off[(l * P) / 2 + k];
            float4 v;
            if (RD == 2) {
                v = make_float4(rx + o.x / dx, ry + o.y / dy, rx + o.z / dx, ry + o.w / dy);
            } else {   // ((offset / P) * wh) * 0.5 in the reference's order
                v = make_float4(rx + o.x / (float)P * dx * 0.5f, ry + o.y / (float)P * dy * 0.5f,
                                rx + o.z / (float)P * dx * 0.5f, ry + o.w / (float)P * dy * 0.5f);
            }
            loc[(l * P) / 2 + k] = v;
        }
    }
}

template <int RD, int L, int P>
__global__ void __launch_bounds__(256) sampling_prep_backward_kernel(SamplingPrepBwdArgs p)
{
    constexpr int S = L * P;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= p.rows * p.M) return;
    const int64_t row = t / p.M;
    float w[S], gw[S];
    const float4 *pw = reinterpret_cast<const float4 *>(p.weights + t * S), *pg = reinterpret_cast<const float4 *>(p.grad_w + t * S);
#pragma unroll
    for (int i = 0; i < S / 4; ++i) {
        const float4 a = pw[i], b = pg[i];
        w[4 * i] = a.x; w[4 * i + 1] = a.y; w[4 * i + 2] = a.z; w[4 * i + 3] = a.w;
        gw[4 * i] = b.x; gw[4 * i + 1] = b.y; gw[4 * i + 2] = b.z; gw[4 * i + 3] = b.w;
    }
    float dot = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) dot = fmaf(w[s], gw[s], dot);
    float4 *gl = reinterpret_cast<float4 *>(p.grad_logits + t * S);
#pragma unroll
    for (int i = 0; i < S / 4; ++i)
        gl[i] = make_float4(w[4 * i] * (gw[4 * i] - dot), w[4 * i + 1] * (gw[4 * i + 1] - dot), w[4 * i + 2] * (gw[4 * i + 2] - dot),
                            w[4 * i + 3] * (gw[4 * i + 3] - dot));
    const float4 *g = reinterpret_cast<const float4 *>(p.grad_loc + t * S * 2);
    float4 *go = reinterpret_cast<float4 *>(p.grad_offsets + t * S * 2);
#pragma unroll
    for (int l = 0; l < L; ++l) {
        float dx, dy;
        if (RD == 2) { dx = (float)p.shapes[2 * l + 1]; dy = (float)p.shapes[2 * l]; }
        else {
            const float *r = p.ref + (row * L + l) * RD;
            dx = r[2]; dy = r[3];
        }
#pragma unroll
        for (int k = 0; k < P / 2; ++k) {
            const float4 v = g[(l * P) / 2 + k];
            // the derivative of the forward's expression, evaluated the way autograd evaluates it
            if (RD == 2) go[(l * P) / 2 + k] = make_float4(v.x / dx, v.y / dy, v.z / dx, v.w / dy);
            else go[(l * P) / 2 + k] = make_float4(v.x * 0.5f * dx / (float)P, v.y * 0.5f * dy / (float)P,
                                                   v.z * 0.5f * dx / (float)P, v.w * 0.5f * dy / (float)P);
        }
    }
}

}  // namespace sdetr

using namespace sdetr;

static int sp_check(int64_t rows, int M, int L, int P, int RD)
{
    if (rows < 0 || M <= 0) return fail("sampling_prep: bad sizes");
    if (L != 4 || P != 4) return fail("sampling_prep: built for 4 levels x 4 points (got %d x %d)", L, P);
    if (RD != 2 && RD != 4) return fail("sampling_prep: reference points have 2 or 4 coordinates (got %d)", RD);
    return 0;
}

extern "C" int sdetr_sampling_prep_supported(int num_levels, int num_points) { return num_levels == 4 && num_points == 4 ? 1 : 0; }

extern "C" int sdetr_sampling_prep_f32(sdetr_stream_t stream, const float *offsets, const float *logits, const float *reference_points,
                                       const int64_t *spatial_shapes, int64_t rows, int num_heads, int num_levels, int num_points,
                                       int ref_dim, float *sampling_locations, float *attention_weights)
{
    if (int e = sp_check(rows, num_heads, num_levels, num_points, ref_dim)) return e;
    if (rows == 0) return 0;
    if (!offsets || !logits || !reference_points || !spatial_shapes || !sampling_locations || !attention_weights)
        return fail("sampling_prep: null pointer");
    SamplingPrepArgs p{};
    p.offsets = offsets; p.logits = logits; p.ref = reference_points; p.shapes = spatial_shapes; p.loc = sampling_locations;
    p.weights = attention_weights; p.rows = rows; p.M = num_heads; p.L = num_levels; p.P = num_points; p.RD = ref_dim;
    const int64_t threads = rows * num_heads;
    const dim3 grid((unsigned)((threads + 255) / 256));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (ref_dim == 2) hipLaunchKernelGGL((sampling_prep_kernel<2, 4, 4>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((sampling_prep_kernel<4, 4, 4>), grid, dim3(256), 0, s, p);
    return check_launch("sampling_prep");
}

extern "C" int sdetr_sampling_prep_backward_f32(sdetr_stream_t stream, const float *grad_sampling_locations,
                                                const float *grad_attention_weights, const float *attention_weights,
                                                const float *reference_points, const int64_t *spatial_shapes, int64_t rows,
                                                int num_heads, int num_levels, int num_points, int ref_dim, float *grad_offsets,
                                                float *grad_logits)
{
    if (int e = sp_check(rows, num_heads, num_levels, num_points, ref_dim)) return e;
    if (rows == 0) return 0;
    if (!grad_sampling_locations || !grad_attention_weights || !attention_weights || !reference_points || !spatial_shapes ||
        !grad_offsets || !grad_logits)
        return fail("sampling_prep: null pointer");
    SamplingPrepBwdArgs p{};
    p.grad_loc = grad_sampling_locations; p.grad_w = grad_attention_weights; p.weights = attention_weights; p.ref = reference_points;
    p.shapes = spatial_shapes; p.grad_offsets = grad_offsets; p.grad_logits = grad_logits; p.rows = rows; p.M = num_heads;
    p.L = num_levels; p.P = num_points; p.RD = ref_dim;
    const int64_t threads = rows * num_heads;
    const dim3 grid((unsigned)((threads + 255) / 256));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (ref_dim == 2) hipLaunchKernelGGL((sampling_prep_backward_kernel<2, 4, 4>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((sampling_prep_backward_kernel<4, 4, 4>), grid, dim3(256), 0, s, p);
    return check_launch("sampling_prep_backward");
}
