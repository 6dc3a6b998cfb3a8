// Elementwise stages of the decoder's box-refinement loop (row N2; reference
// models/bricks/salience_transformer.py:641-671), each a chain of 8-15 tiny framework launches in the reference:
//
//  * query_sine_embed_kernel: reference_points_input = ref[:, :, None] * cat(valid_ratios, valid_ratios)[:, None]
//    (:642) and get_sine_pos_embed(reference_points_input[:, :, 0, :]) (:643, position_encoding.py:105-132):
//    per box coordinate c and feature pair p the angle a = c * 2*pi / T^(2p/F); features (2p, 2p+1) = (sin a, cos a);
//    coordinate blocks are emitted in (y, x, w, h) order.
//  * box_refine_kernel: sigmoid(delta + inverse_sigmoid(ref)) (:659-660, :666-668; util/misc.py:31-35) for `groups`
//    delta tensors that share one reference (the layer's output boxes and the next layer's reference boxes).
#include "common.h"

namespace sdetr {

struct SineArgs {
    const float *ref;   // [B, Nq, 4]
    const float *vr;    // [B, L, 2]
    int Nq, L, F;       // F = features per coordinate (even)
    float temperature;
    int64_t rows;       // B * Nq
    void *embed;        // [B, Nq, 4F] f32 | bf16
    int embed_bf16;
    float *ref_in;      // [B, Nq, L, 4]
};

__global__ void __launch_bounds__(256) query_sine_embed_kernel(SineArgs p)
{
    const int half = p.F >> 1;                 // pairs per coordinate
    const int per_row = 4 * half;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= p.rows * per_row) return;
    const int64_t row = gid / per_row;
    const int rem = (int)(gid - row * per_row);
    const int blk = rem / half, pr = rem - blk * half;
    const int b = (int)(row / p.Nq);
    const int c = blk == 0 ? 1 : (blk == 1 ? 0 : blk);      // output block -> box coordinate (y, x, w, h)
    const float *vr = p.vr + (int64_t)b * p.L * 2;
    const float coord = p.ref[row * 4 + c] * vr[c & 1];     // level-0 ratio: (w, h, w, h)
    const float dim_t = powf(p.temperature, (float)(2 * pr) / (float)p.F);
    const float ang = coord * 6.283185307179586f / dim_t;
    float s, co;
    sincosf(ang, &s, &co);
    const int64_t o = row * (4 * (int64_t)p.F) + (int64_t)blk * p.F + 2 * pr;
    if (p.embed_bf16) reinterpret_cast<uint32_t *>(p.embed)[o >> 1] = pack_act2(s, co);
    else reinterpret_cast<float2 *>(p.embed)[o >> 1] = make_float2(s, co);
    if (p.ref_in && rem < p.L) {
        const float4 r = reinterpret_cast<const float4 *>(p.ref)[row];
        const float rw = vr[2 * rem], rh = vr[2 * rem + 1];
        reinterpret_cast<float4 *>(p.ref_in)[row * p.L + rem] = make_float4(r.x * rw, r.y * rh, r.z * rw, r.w * rh);
    }
}

__device__ __forceinline__ float inverse_sigmoid_f(float x, float eps)
{
    x = fminf(fmaxf(x, 0.f), 1.f);
    return logf(fmaxf(x, eps) / fmaxf(1.f - x, eps));
}

template <typename DT>
__global__ void __launch_bounds__(256) box_refine_kernel(const DT *delta, int64_t delta_row_stride, const float *ref,
                                                         int64_t boxes, int groups, float eps, float *out)
{
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= boxes * groups) return;
    const int64_t i = gid % boxes;
    const float4 r = reinterpret_cast<const float4 *>(ref)[i];
    const DT *d = delta + gid * delta_row_stride;
    float v[4];
    if constexpr (sizeof(DT) == 2) {
        v[0] = act_lo((uint32_t)d[0]); v[1] = act_lo((uint32_t)d[1]);
        v[2] = act_lo((uint32_t)d[2]); v[3] = act_lo((uint32_t)d[3]);
    } else {
        v[0] = d[0]; v[1] = d[1]; v[2] = d[2]; v[3] = d[3];
    }
    const float rr[4] = {r.x, r.y, r.z, r.w};
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = 1.f / (1.f + expf(-(v[k] + inverse_sigmoid_f(rr[k], eps))));
    reinterpret_cast<float4 *>(out)[gid] = make_float4(o[0], o[1], o[2], o[3]);
}

}  // namespace sdetr

using namespace sdetr;

extern "C" int sdetr_decoder_query_sine_embed(sdetr_stream_t stream, const float *reference_points,
                                              const float *valid_ratios, int batch_size, int num_queries,
                                              int num_levels, int num_pos_feats, float temperature, void *embed,
                                              int embed_dtype, float *reference_points_input)
{
    if (batch_size < 0 || num_queries < 0 || num_levels <= 0 || num_pos_feats <= 0 || (num_pos_feats & 1))
        return fail("query_sine_embed: bad sizes (num_pos_feats must be even)");
    if (num_levels > 2 * num_pos_feats) return fail("query_sine_embed: more levels than threads per query");
    if (embed_dtype != SDETR_F32 && embed_dtype != kActCode) return fail("query_sine_embed: embed dtype must be f32 or bf16");
    const int64_t rows = (int64_t)batch_size * num_queries;
    if (rows == 0) return 0;
    if (!reference_points || !valid_ratios || !embed) return fail("query_sine_embed: null pointer");
    SineArgs a{reference_points, valid_ratios, num_queries, num_levels, num_pos_feats, temperature, rows, embed,
               embed_dtype == kActCode, reference_points_input};
    const int64_t total = rows * 2 * num_pos_feats;
    hipLaunchKernelGGL(query_sine_embed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, a);
    return check_launch("query_sine_embed");
}

extern "C" int sdetr_box_refine(sdetr_stream_t stream, const void *delta, int delta_dtype, int64_t delta_row_stride,
                                const float *reference_points, int64_t num_boxes, int groups, float eps, float *out)
{
    if (num_boxes < 0 || groups <= 0 || delta_row_stride < 4) return fail("box_refine: bad sizes");
    if (delta_dtype != SDETR_F32 && delta_dtype != kActCode) return fail("box_refine: delta dtype must be f32 or bf16");
    if (num_boxes == 0) return 0;
    if (!delta || !reference_points || !out) return fail("box_refine: null pointer");
    const int64_t total = num_boxes * groups;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (delta_dtype == kActCode)
        hipLaunchKernelGGL(box_refine_kernel<uint16_t>, grid, dim3(256), 0, (hipStream_t)stream,
                           (const uint16_t *)delta, delta_row_stride, reference_points, num_boxes, groups, eps, out);
    else
        hipLaunchKernelGGL(box_refine_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)delta,
                           delta_row_stride, reference_points, num_boxes, groups, eps, out);
    return check_launch("box_refine");
}
