// Row movers of the encoder loop when every layer's index set is a PREFIX of one globally sorted list
// (models/bricks/salience_transformer.py:156-163 builds the sets exactly like that: selected_inds[:, :k]).
//
// The reference gathers each layer's queries from, and scatters them back into, the full [B,S,C] token buffer
// (:454-485).  With prefix sets the rows a layer reads are simply the first rows of what the previous layer
// wrote, so the loop keeps the tokens in sorted order and never goes back to token space until the end:
//
//   * advance_rows   after layer k: rows i < focus_token_nums[b] of its output are (1) recorded in the sorted
//                    result buffer and (2) handed to layer k+1 (its first c_{k+1} rows); rows past the image's
//                    focus count are never updated by any layer (:474-485), so layer k+1 sees the original token
//                    there -- one pass over the layer output instead of scatter + 4 gathers.
//   * select_stack   the [q+pos ; q] rows of the top-k dense self-attention (:366-376: gather tgt, gather pos,
//                    add, and the stacked in-projection input) in one launch.
//   * encoder_finalize  back to token space + the learnt background embedding of every token that is neither
//                    padding nor in the last layer's set (:487-495), two launches.
//
// All HBM-bound byte movers: one 16-byte chunk per thread, indices read once per row.
#include "common.h"
#include "row_orders_core.h"

namespace sdetr {

static unsigned rows_grid(int64_t total)
{
    int64_t blocks = (total + kBlock - 1) / kBlock;
    const int64_t cap = 256 * 32;
    return (unsigned)(blocks > cap ? cap : (blocks < 1 ? 1 : blocks));
}

__global__ void __launch_bounds__(kBlock) advance_rows_kernel(const uint4 *y, uint4 *result, uint4 *next,
                                                              const uint4 *tokens, const int64_t *sorted_index,
                                                              int64_t index_batch_stride, const int64_t *count,
                                                              int64_t total, int c, int n0, int c_next, int S, int vpr)
{
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int v = (int)(t % vpr);
        const int64_t r = t / vpr;
        const int b = (int)(r / c), i = (int)(r - (int64_t)b * c);
        const bool live = !count || i < count[b];
        uint4 val;
        if (live) {
            val = y[t];
            result[((int64_t)b * n0 + i) * vpr + v] = val;
        }
        if (next && i < c_next) {
            if (!live) val = tokens[((int64_t)b * S + sorted_index[(int64_t)b * index_batch_stride + i]) * vpr + v];
            next[((int64_t)b * c_next + i) * vpr + v] = val;
        }
    }
}

// out[b][i] = q[b][idx] + pos[b][idx],  out[b][N + i] = q[b][idx];  8 channels per thread
template <typename T>
__global__ void __launch_bounds__(kBlock) select_stack_kernel(const T *q, int64_t q_batch_stride, const T *pos,
                                                              int64_t pos_batch_stride, const int64_t *idx,
                                                              int64_t total, int N, int C, T *out)
{
    const int cpr = C / 8;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(t % cpr) * 8;
        const int64_t r = t / cpr;
        const int b = (int)(r / N), i = (int)(r - (int64_t)b * N);
        const int64_t s = idx[r];
        const T *qs = q + b * q_batch_stride + s * C + ch, *ps = pos + b * pos_batch_stride + s * C + ch;
        T *o0 = out + ((int64_t)b * 2 * N + i) * C + ch, *o1 = o0 + (int64_t)N * C;
        if (sizeof(T) == 4) {
            const float4 a0 = reinterpret_cast<const float4 *>(qs)[0], a1 = reinterpret_cast<const float4 *>(qs)[1];
            const float4 p0 = reinterpret_cast<const float4 *>(ps)[0], p1 = reinterpret_cast<const float4 *>(ps)[1];
            reinterpret_cast<float4 *>(o1)[0] = a0;
            reinterpret_cast<float4 *>(o1)[1] = a1;
            reinterpret_cast<float4 *>(o0)[0] = make_float4(a0.x + p0.x, a0.y + p0.y, a0.z + p0.z, a0.w + p0.w);
            reinterpret_cast<float4 *>(o0)[1] = make_float4(a1.x + p1.x, a1.y + p1.y, a1.z + p1.z, a1.w + p1.w);
        } else {
            const uint4 a = *reinterpret_cast<const uint4 *>(qs), pp = *reinterpret_cast<const uint4 *>(ps);
            *reinterpret_cast<uint4 *>(o1) = a;
            *reinterpret_cast<uint4 *>(o0) =
                make_uint4(pack_act2(act_lo(a.x) + act_lo(pp.x), act_hi(a.x) + act_hi(pp.x)),
                           pack_act2(act_lo(a.y) + act_lo(pp.y), act_hi(a.y) + act_hi(pp.y)),
                           pack_act2(act_lo(a.z) + act_lo(pp.z), act_hi(a.z) + act_hi(pp.z)),
                           pack_act2(act_lo(a.w) + act_lo(pp.w), act_hi(a.w) + act_hi(pp.w)));
        }
    }
}

template <typename T>
__device__ __forceinline__ void add8(const T *a, const T *b, bool use_b, T *o)
{
    if (sizeof(T) == 4) {
        float4 a0 = reinterpret_cast<const float4 *>(a)[0], a1 = reinterpret_cast<const float4 *>(a)[1];
        if (use_b) {
            const float4 b0 = reinterpret_cast<const float4 *>(b)[0], b1 = reinterpret_cast<const float4 *>(b)[1];
            a0 = make_float4(a0.x + b0.x, a0.y + b0.y, a0.z + b0.z, a0.w + b0.w);
            a1 = make_float4(a1.x + b1.x, a1.y + b1.y, a1.z + b1.z, a1.w + b1.w);
        }
        reinterpret_cast<float4 *>(o)[0] = a0;
        reinterpret_cast<float4 *>(o)[1] = a1;
    } else {
        uint4 x = *reinterpret_cast<const uint4 *>(a);
        if (use_b) {
            const uint4 y = *reinterpret_cast<const uint4 *>(b);
            x = make_uint4(pack_act2(act_lo(x.x) + act_lo(y.x), act_hi(x.x) + act_hi(y.x)),
                           pack_act2(act_lo(x.y) + act_lo(y.y), act_hi(x.y) + act_hi(y.y)),
                           pack_act2(act_lo(x.z) + act_lo(y.z), act_hi(x.z) + act_hi(y.z)),
                           pack_act2(act_lo(x.w) + act_lo(y.w), act_hi(x.w) + act_hi(y.w)));
        }
        *reinterpret_cast<uint4 *>(o) = x;
    }
}

// every token: out = tokens + (padding ? 0 : background)
template <typename T>
__global__ void __launch_bounds__(kBlock) finalize_all_kernel(const T *tokens, const T *background, const uint8_t *pad,
                                                              int64_t total, int S, int C, T *out)
{
    const int cpr = C / 8;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(t % cpr) * 8;
        const int64_t r = t / cpr;  // b*S + s
        const int s = (int)(r % S);
        add8<T>(tokens + r * C + ch, background + (int64_t)s * C + ch, !(pad && pad[r]), out + r * C + ch);
    }
}

// sorted rows: out[token] = (i < count ? result[i] : tokens[token]) + (i >= c_last && !padding ? background : 0)
template <typename T>
__global__ void __launch_bounds__(kBlock) finalize_sorted_kernel(const T *tokens, const T *result,
                                                                 const int64_t *sorted_index, const int64_t *count,
                                                                 const T *background, const uint8_t *pad,
                                                                 int64_t total, int n0, int c_last, int S, int C, T *out)
{
    const int cpr = C / 8;
    if (total < ((int64_t)1 << 31)) {
        // 32-bit index arithmetic (three 64-bit divisions per 16-byte piece were most of this kernel's instructions)
        const uint32_t ucpr = (uint32_t)cpr, un0 = (uint32_t)n0;
        for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < (uint32_t)total; t += gridDim.x * blockDim.x) {
            const uint32_t r = t / ucpr, ch = (t - r * ucpr) * 8u;  // r = b*n0 + i
            const uint32_t b = r / un0, i = r - b * un0;
            const int64_t tok = (int64_t)b * S + sorted_index[r];
            const bool live = !count || (int64_t)i < count[b];
            const T *base = live ? result + (int64_t)r * C + ch : tokens + tok * C + ch;
            const bool bg = (int)i >= c_last && !(pad && pad[tok]);
            add8<T>(base, background + (tok - (int64_t)b * S) * C + ch, bg, out + tok * C + ch);
        }
        return;
    }
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(t % cpr) * 8;
        const int64_t r = t / cpr;  // b*n0 + i
        const int b = (int)(r / n0), i = (int)(r - (int64_t)b * n0);
        const int64_t tok = (int64_t)b * S + sorted_index[r];
        const bool live = !count || i < count[b];
        const T *base = live ? result + r * C + ch : tokens + tok * C + ch;
        const bool bg = i >= c_last && !(pad && pad[tok]);
        add8<T>(base, background + (tok - (int64_t)b * S) * C + ch, bg, out + tok * C + ch);
    }
}

}  // namespace sdetr

using namespace sdetr;

extern "C" int sdetr_advance_rows(sdetr_stream_t stream, const void *layer_out, void *sorted_result, void *next_query,
                                  const void *tokens, const int64_t *sorted_index, int64_t index_batch_stride,
                                  const int64_t *count, int batch_size, int rows, int sorted_rows, int next_rows,
                                  int spatial_size, int row_bytes)
{
    if (batch_size < 0 || rows < 0 || sorted_rows < rows || next_rows < 0 || next_rows > rows || row_bytes <= 0 ||
        (row_bytes & 15))
        return fail("advance_rows: bad sizes (rows %d of %d, next %d, row bytes %d)", rows, sorted_rows, next_rows, row_bytes);
    if ((int64_t)batch_size * rows == 0) return 0;
    if (!layer_out || !sorted_result || !tokens || !sorted_index) return fail("advance_rows: null pointer");
    if (index_batch_stride < rows) return fail("advance_rows: index batch stride too small");
    const int vpr = row_bytes / 16;
    const int64_t total = (int64_t)batch_size * rows * vpr;
    hipLaunchKernelGGL(advance_rows_kernel, dim3(rows_grid(total)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       (const uint4 *)layer_out, (uint4 *)sorted_result, next_rows > 0 ? (uint4 *)next_query : nullptr,
                       (const uint4 *)tokens, sorted_index, index_batch_stride, count, total, rows, sorted_rows,
                       next_rows, spatial_size, vpr);
    return check_launch("advance_rows");
}

extern "C" int sdetr_select_stack(sdetr_stream_t stream, const void *query, int64_t query_batch_stride, const void *pos,
                                  int64_t pos_batch_stride, const int64_t *index, int batch_size, int num_select,
                                  int channels, int dtype, void *out)
{
    if (batch_size < 0 || num_select < 0 || channels <= 0 || (channels % 8)) return fail("select_stack: bad sizes");
    if ((query_batch_stride % 8) || (pos_batch_stride % 8)) return fail("select_stack: strides must be multiples of 8");
    if ((int64_t)batch_size * num_select == 0) return 0;
    if (!query || !pos || !index || !out) return fail("select_stack: null pointer");
    const int64_t total = (int64_t)batch_size * num_select * (channels / 8);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == SDETR_F32)
        hipLaunchKernelGGL(select_stack_kernel<float>, dim3(rows_grid(total)), dim3(kBlock), 0, s, (const float *)query,
                           query_batch_stride, (const float *)pos, pos_batch_stride, index, total, num_select, channels,
                           (float *)out);
    else if (dtype == kActCode)
        hipLaunchKernelGGL(select_stack_kernel<bf16_t>, dim3(rows_grid(total)), dim3(kBlock), 0, s,
                           (const bf16_t *)query, query_batch_stride, (const bf16_t *)pos, pos_batch_stride, index, total,
                           num_select, channels, (bf16_t *)out);
    else
        return fail("select_stack: bad dtype %d", dtype);
    return check_launch("select_stack");
}

template <typename T>
static int launch_finalize(hipStream_t s, const void *tokens, const void *result, const int64_t *sorted_index,
                           const int64_t *count, const void *background, const uint8_t *pad, int B, int S, int n0,
                           int c_last, int C, void *out, bool all_pass = true)
{
    const int64_t total_all = (int64_t)B * S * (C / 8), total_sorted = (int64_t)B * n0 * (C / 8);
    int rc = 0;
    if (all_pass) {
        hipLaunchKernelGGL(finalize_all_kernel<T>, dim3(rows_grid(total_all)), dim3(kBlock), 0, s, (const T *)tokens,
                           (const T *)background, pad, total_all, S, C, (T *)out);
        rc = check_launch("encoder_finalize");
    }
    if (rc || total_sorted == 0) return rc;
    hipLaunchKernelGGL(finalize_sorted_kernel<T>, dim3(rows_grid(total_sorted)), dim3(kBlock), 0, s, (const T *)tokens,
                       (const T *)result, sorted_index, count, (const T *)background, pad, total_sorted, n0, c_last, S,
                       C, (T *)out);
    return check_launch("encoder_finalize");
}

extern "C" int sdetr_encoder_finalize(sdetr_stream_t stream, const void *tokens, const void *sorted_result,
                                      const int64_t *sorted_index, const int64_t *count, const void *background,
                                      const uint8_t *padding_mask, int batch_size, int spatial_size, int sorted_rows,
                                      int last_rows, int channels, int dtype, void *out)
{
    if (batch_size < 0 || spatial_size < 0 || sorted_rows < 0 || sorted_rows > spatial_size || last_rows < 0 ||
        last_rows > sorted_rows || channels <= 0 || (channels % 8))
        return fail("encoder_finalize: bad sizes");
    if ((int64_t)batch_size * spatial_size == 0) return 0;
    if (!tokens || !background || !out || (sorted_rows && (!sorted_result || !sorted_index)))
        return fail("encoder_finalize: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == SDETR_F32)
        return launch_finalize<float>(s, tokens, sorted_result, sorted_index, count, background, padding_mask,
                                      batch_size, spatial_size, sorted_rows, last_rows, channels, out);
    if (dtype == kActCode)
        return launch_finalize<bf16_t>(s, tokens, sorted_result, sorted_index, count, background, padding_mask,
                                       batch_size, spatial_size, sorted_rows, last_rows, channels, out);
    return fail("encoder_finalize: bad dtype %d", dtype);
}

extern "C" int sdetr_encoder_finalize_sorted(sdetr_stream_t stream, const void *tokens, const void *sorted_result,
                                             const int64_t *sorted_index, const int64_t *count, const void *background,
                                             const uint8_t *padding_mask, int batch_size, int spatial_size,
                                             int sorted_rows, int last_rows, int channels, int dtype, void *out)
{
    if (batch_size < 0 || spatial_size < 0 || sorted_rows < 0 || sorted_rows > spatial_size || last_rows < 0 ||
        last_rows > sorted_rows || channels <= 0 || (channels % 8))
        return fail("encoder_finalize_sorted: bad sizes");
    if ((int64_t)batch_size * spatial_size == 0 || sorted_rows == 0) return 0;
    if (!tokens || !background || !out || !sorted_result || !sorted_index) return fail("encoder_finalize_sorted: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == SDETR_F32)
        return launch_finalize<float>(s, tokens, sorted_result, sorted_index, count, background, padding_mask,
                                      batch_size, spatial_size, sorted_rows, last_rows, channels, out, false);
    if (dtype == kActCode)
        return launch_finalize<bf16_t>(s, tokens, sorted_result, sorted_index, count, background, padding_mask,
                                       batch_size, spatial_size, sorted_rows, last_rows, channels, out, false);
    return fail("encoder_finalize_sorted: bad dtype %d", dtype);
}

// ---- per-layer row orders for the deformable attention (round 4) ------------------------------------------------------
// The encoder keeps its rows sorted by salience score (every layer's set is a prefix of that list), so the rows a
// workgroup of the MSDA kernel works on are scattered over the image and the fine-level records they fetch miss the L1.
// For every layer k this kernel lists the layer's rows 0 .. counts[k]-1 in TILE-MAJOR order of their tokens
// (`tile_pos` [S] int32: position of every token in a static order that keeps the tokens of all levels whose centre
// falls into the same tile of the finest level together): order[k][b][0 .. counts[k]) is a permutation of
// 0 .. counts[k]-1.  One 1024-thread workgroup per image: the rows are scattered into an LDS array indexed by tile
// position (a counting sort with one key per slot), every thread counts the rows of each layer in its run of slots, one
// block scan per layer gives its write offsets.  Pure index work: bit-exact, ~5 us.
__global__ void __launch_bounds__(kOrderThreads) layer_row_orders_kernel(RowOrderArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t order_slot[];
    // block = (part, image, layer)
    const int j = (int)blockIdx.x;
    layer_row_orders_body(a, (j / a.parts) % a.batch, j / (a.parts * a.batch), j % a.parts, order_slot);
}

// order [num_layers][batch][order_batch_stride >= n0] int32; counts_host: rows per layer (<= n0 each).  The counts travel
// as a small device array the caller owns (`counts_dev`, int32 [num_layers]: hipGraph-replayable, no host copy here).
extern "C" int sdetr_layer_row_orders(sdetr_stream_t stream, const int64_t *sorted_index, int64_t index_batch_stride,
                                      const int32_t *tile_pos, int batch_size, int spatial_size, int num_rows,
                                      int num_layers, const int32_t *counts_dev, int32_t *order, int64_t order_batch_stride)
{
    if (batch_size <= 0 || spatial_size <= 0 || num_rows <= 0 || num_layers <= 0) return fail("layer_row_orders: bad sizes");
    if (!sorted_index || !tile_pos || !counts_dev || !order) return fail("layer_row_orders: null pointer");
    if (num_layers > kOrderMaxLayers) return fail("layer_row_orders: at most %d layers", kOrderMaxLayers);
    if (spatial_size > kOrderMaxTokens) return fail("layer_row_orders: at most %d tokens per image (got %d)", kOrderMaxTokens, spatial_size);
    if (num_rows >= 0xffff) return fail("layer_row_orders: at most 65534 rows per image");
    if (index_batch_stride == 0) index_batch_stride = num_rows;
    if (order_batch_stride == 0) order_batch_stride = num_rows;
    if (index_batch_stride < num_rows || order_batch_stride < num_rows) return fail("layer_row_orders: bad strides");
    RowOrderArgs a{};
    a.sorted_index = sorted_index; a.index_batch_stride = index_batch_stride; a.tile_pos = tile_pos; a.S = spatial_size;
    a.n0 = num_rows; a.nl = num_layers; a.batch = batch_size; a.counts_dev = counts_dev; a.order = order;
    a.order_layer_stride = (int64_t)batch_size * order_batch_stride; a.order_batch_stride = order_batch_stride;
    static DeviceOnce once;
    allow_dynamic_lds(layer_row_orders_kernel, once, 160 * 1024 - 1024);
    a.slot_cap = order_slot_cap(spatial_size, 152 * 1024);
    if (a.slot_cap <= 0) return fail("layer_row_orders: %d tokens per image need more than %d parts", spatial_size, kOrderMaxParts);
    a.parts = (spatial_size + a.slot_cap - 1) / a.slot_cap;
    const size_t lds = (((size_t)(a.slot_cap < spatial_size ? a.slot_cap : spatial_size) + 7) & ~(size_t)7) * 2;
    hipLaunchKernelGGL(layer_row_orders_kernel, dim3((unsigned)(batch_size * num_layers * a.parts)), dim3(kOrderThreads), lds,
                       static_cast<hipStream_t>(stream), a);
    return check_launch("layer_row_orders");
}
