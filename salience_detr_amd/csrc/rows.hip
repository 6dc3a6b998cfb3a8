// Token-row movement and value re-layout kernels of the encoder loop (HBM-bound byte movers).
//
//  * gather_rows / scatter_rows: models/bricks/salience_transformer.py:454-461 (torch.gather of
//    query / pos rows by foreground_inds, which materialises an int64 index TWICE the size of the
//    data it moves) and :474-485 (per-image python loop scattering the first focus_token_nums[b]
//    rows back).  Here one 16-byte lane chunk per thread, index read once per row, no host sync
//    on focus_token_nums (the count is read on the device).
//  * value_to_head_major: tail of value_proj in MultiScaleDeformableAttention.forward
//    (models/bricks/ms_deform_attn.py:316-321): masked_fill(padding, 0) + view as heads, written
//    head-major [B,M,Nv,D] (optionally bf16) so that one head of one pixel is one contiguous
//    64/128-byte segment and one head's map is one contiguous slab (XCD-private in L2).
#include <type_traits>

#include "common.h"

namespace sdetr {

template <typename V>
__global__ void __launch_bounds__(kBlock) gather_rows_kernel(const V *src, const int64_t *idx, int64_t total,
                                                             int src_rows, int n, int vec_per_row, V *dst)
{
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % vec_per_row);
        const int64_t r = t / vec_per_row;  // b*n + i
        const int b = (int)(r / n);
        const int64_t s = idx[r];
        dst[t] = src[((int64_t)b * src_rows + s) * vec_per_row + c];
    }
}

template <typename V>
__global__ void __launch_bounds__(kBlock) scatter_rows_kernel(V *dst, const int64_t *idx, const V *src,
                                                              const int64_t *count, int64_t total, int dst_rows, int n,
                                                              int vec_per_row)
{
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % vec_per_row);
        const int64_t r = t / vec_per_row;
        const int b = (int)(r / n);
        const int i = (int)(r - (int64_t)b * n);
        if (count && i >= count[b]) continue;
        dst[((int64_t)b * dst_rows + idx[r]) * vec_per_row + c] = src[t];
    }
}

// One thread = 8 channels of one (pixel, head).  Lanes are ordered (chunk, pixel, head) so a
// wavefront writes 64/(D/8) consecutive pixels of ONE head: a contiguous run in the destination.
template <typename ST, typename DT>
__global__ void __launch_bounds__(kBlock) head_major_kernel(const ST *src, int64_t src_stride, const uint8_t *pad,
                                                            int64_t total, int B, int Nv, int M_all, int M, int D, DT *dst)
{
    const int cpr = D / 8;           // 8-channel chunks per head row
    const int ppw = kWave / cpr;     // pixels per wavefront
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(t & (kWave - 1));
        const int64_t wave = t >> 6;  // global wave id: (b, pixel block, head)
        const int m = (int)(wave % M_all);  // head index over all groups (group = m / M)
        const int64_t pb = wave / M_all;    // b * nblk + pixel block
        const int nblk = (Nv + ppw - 1) / ppw;
        const int b = (int)(pb / nblk);
        const int pix = (int)(pb - (int64_t)b * nblk) * ppw + lane / cpr;
        const int ch = (lane % cpr) * 8;
        if (pix >= Nv) continue;
        const bool masked = pad && pad[(int64_t)b * Nv + pix];
        float v[8];
        const ST *s = src + ((int64_t)b * Nv + pix) * src_stride + m * D + ch;
        if (masked) {
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = 0.f;
        } else if (sizeof(ST) == 4) {
            const float4 a = reinterpret_cast<const float4 *>(s)[0], c = reinterpret_cast<const float4 *>(s)[1];
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
        } else {
            const uint4 a = *reinterpret_cast<const uint4 *>(s);
            v[0] = act_lo(a.x); v[1] = act_hi(a.x); v[2] = act_lo(a.y); v[3] = act_hi(a.y);
            v[4] = act_lo(a.z); v[5] = act_hi(a.z); v[6] = act_lo(a.w); v[7] = act_hi(a.w);
        }
        // destination [group][B][M][Nv][D]
        DT *d = dst + ((((int64_t)(m / M) * B + b) * M + (m % M)) * Nv + pix) * D + ch;
        if (sizeof(DT) == 4) {
            reinterpret_cast<float4 *>(d)[0] = make_float4(v[0], v[1], v[2], v[3]);
            reinterpret_cast<float4 *>(d)[1] = make_float4(v[4], v[5], v[6], v[7]);
        } else if (std::is_same<DT, half_t>::value) {
            *reinterpret_cast<uint4 *>(d) = make_uint4(pack_f16x2(v[0], v[1]), pack_f16x2(v[2], v[3]),
                                                      pack_f16x2(v[4], v[5]), pack_f16x2(v[6], v[7]));
        } else {
            *reinterpret_cast<uint4 *>(d) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]),
                                                      pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
        }
    }
}

static unsigned grid_for(int64_t total)
{
    int64_t blocks = (total + kBlock - 1) / kBlock;
    const int64_t cap = 256 * 32;  // 256 CUs x 32: enough waves in flight, grid-stride beyond
    return (unsigned)(blocks > cap ? cap : (blocks < 1 ? 1 : blocks));
}

}  // namespace sdetr

using namespace sdetr;

extern "C" int sdetr_gather_rows(sdetr_stream_t stream, const void *src, const int64_t *idx, int B, int src_rows, int n,
                                 int row_bytes, void *dst)
{
    if (B < 0 || src_rows < 0 || n < 0 || row_bytes <= 0 || (row_bytes & 3)) return fail("gather_rows: bad sizes");
    if ((int64_t)B * n == 0) return 0;
    if (!src || !idx || !dst) return fail("gather_rows: null pointer");
    if ((row_bytes & 15) == 0) {
        const int vpr = row_bytes / 16;
        const int64_t total = (int64_t)B * n * vpr;
        hipLaunchKernelGGL(gather_rows_kernel<uint4>, dim3(grid_for(total)), dim3(kBlock), 0, stream,
                           (const uint4 *)src, idx, total, src_rows, n, vpr, (uint4 *)dst);
    } else {
        const int vpr = row_bytes / 4;
        const int64_t total = (int64_t)B * n * vpr;
        hipLaunchKernelGGL(gather_rows_kernel<uint32_t>, dim3(grid_for(total)), dim3(kBlock), 0, stream,
                           (const uint32_t *)src, idx, total, src_rows, n, vpr, (uint32_t *)dst);
    }
    return check_launch("gather_rows");
}

extern "C" int sdetr_scatter_rows(sdetr_stream_t stream, void *dst, const int64_t *idx, const void *src,
                                  const int64_t *count, int B, int dst_rows, int n, int row_bytes)
{
    if (B < 0 || dst_rows < 0 || n < 0 || row_bytes <= 0 || (row_bytes & 3)) return fail("scatter_rows: bad sizes");
    if ((int64_t)B * n == 0) return 0;
    if (!src || !idx || !dst) return fail("scatter_rows: null pointer");
    if ((row_bytes & 15) == 0) {
        const int vpr = row_bytes / 16;
        const int64_t total = (int64_t)B * n * vpr;
        hipLaunchKernelGGL(scatter_rows_kernel<uint4>, dim3(grid_for(total)), dim3(kBlock), 0, stream, (uint4 *)dst, idx,
                           (const uint4 *)src, count, total, dst_rows, n, vpr);
    } else {
        const int vpr = row_bytes / 4;
        const int64_t total = (int64_t)B * n * vpr;
        hipLaunchKernelGGL(scatter_rows_kernel<uint32_t>, dim3(grid_for(total)), dim3(kBlock), 0, stream,
                           (uint32_t *)dst, idx, (const uint32_t *)src, count, total, dst_rows, n, vpr);
    }
    return check_launch("scatter_rows");
}

extern "C" int sdetr_value_to_head_major(sdetr_stream_t stream, const void *src, int src_dtype, int64_t src_row_stride,
                                         const uint8_t *pad_mask, int B, int Nv, int M, int D, int num_groups, void *dst,
                                         int dst_dtype)
{
    if (B < 0 || Nv < 0 || M <= 0 || D <= 0 || num_groups <= 0) return fail("value_to_head_major: bad dims");
    const int M_all = M * num_groups;
    if (D % 8 != 0 || D > 512 || (kWave % (D / 8)) != 0)
        return fail("value_to_head_major: head dim %d must be a multiple of 8 dividing 512", D);
    if (src_row_stride < (int64_t)M_all * D) return fail("value_to_head_major: source row stride too small");
    if ((src_row_stride % 8) != 0) return fail("value_to_head_major: source row stride must be a multiple of 8");
    if ((int64_t)B * Nv == 0) return 0;
    if (!src || !dst) return fail("value_to_head_major: null pointer");
    const int ppw = kWave / (D / 8);
    const int64_t nblk = (Nv + ppw - 1) / ppw;
    const int64_t total = (int64_t)B * nblk * M_all * kWave;
    const dim3 grid(grid_for(total)), block(kBlock);
#define SDETR_HM(ST, DT)                                                                                     \
    hipLaunchKernelGGL((head_major_kernel<ST, DT>), grid, block, 0, stream, (const ST *)src, src_row_stride, \
                       pad_mask, total, B, Nv, M_all, M, D, (DT *)dst)
    if (src_dtype == SDETR_F32 && dst_dtype == SDETR_F32) SDETR_HM(float, float);
    else if (src_dtype == SDETR_F32 && dst_dtype == SDETR_BF16) SDETR_HM(float, bf16_t);
    else if (src_dtype == kActCode && dst_dtype == SDETR_BF16) SDETR_HM(bf16_t, bf16_t);
    else if (src_dtype == kActCode && dst_dtype == SDETR_F32) SDETR_HM(bf16_t, float);
    else if (src_dtype == SDETR_F32 && dst_dtype == SDETR_F16) SDETR_HM(float, half_t);
    else if (src_dtype == kActCode && dst_dtype == SDETR_F16) SDETR_HM(bf16_t, half_t);
    else return fail("value_to_head_major: bad dtypes %d -> %d", src_dtype, dst_dtype);
#undef SDETR_HM
    return check_launch("value_to_head_major");
}
