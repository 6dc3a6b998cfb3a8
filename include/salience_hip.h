/*
 * include/salience_hip.h -- C ABI of libsalience_hip.so (gfx950 / MI355X).
 *
 * The drop-in boundary for the Salience-DETR encoder hot path (hierarchical
 * salience filtering + multi-scale deformable attention).  Plain pointers and
 * sizes only: every pointer is a DEVICE pointer unless stated otherwise, every
 * launch is enqueued on `stream` and returns without synchronising (same
 * contract as the reference launchers, which enqueue on the current stream).
 *
 * Return value: 0 on success; SDETR_EINVAL (-1) for a rejected argument
 * (nothing was launched); a positive hipError_t if the launch itself failed
 * (the reference only printf()s launch errors --
 * models/bricks/ops/cuda/ms_deform_im2col_cuda.cuh:937-941,1310-1314 -- this
 * library reports them).  sdetr_last_error() returns a thread-local message.
 *
 * Each entry point cites the reference interface it replaces (paths relative
 * to the upstream repository root).
 */
#ifndef SALIENCE_HIP_H_
#define SALIENCE_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t *sdetr_stream_t; /* == hipStream_t */

#define SDETR_ABI_VERSION 1
#define SDETR_EINVAL (-1)

/* element types for the native (non drop-in) entry points */
#define SDETR_F32 0
#define SDETR_BF16 1
#define SDETR_F16 2 /* IEEE half: the head-major value maps (either library); the activations of the fp16 flavour */

/*
 * Two builds of this ABI (round 5).  The token-resident kernels keep their ACTIVATIONS -- token rows, projection slabs,
 * attention / feed-forward operands, 16-bit outputs, the 16-bit weights the *_pack_bf16 entry points permute -- in one
 * 16-bit type per library:
 *   libsalience_hip.so      bfloat16 (activation arguments carry SDETR_BF16): the benchmark mode, BASELINE configs[1];
 *   libsalience_hip_f16.so  IEEE half (activation arguments carry SDETR_F16): the SAME sources built with
 *                           -DSDETR_ACT_F16 (csrc/common.h) -- v_mfma_f32_*_f16 instead of *_bf16, fp32 accumulators /
 *                           LayerNorm / softmax / class scores / sampling locations unchanged, stores saturating at
 *                           +-65504 -- for the reference's `--mixed-precision fp16` (main.py:24-56, BASELINE configs[4]).
 * Same symbols, same argument lists; the `_bf16` suffix of an entry point reads "the library's 16-bit activation type".
 * fp32 / fp64 / integer entry points are identical in both.  The value maps keep the explicit type their own dtype
 * argument names in both.  A host loads the library that matches its activations (the Python layer: _hip.lib(dtype));
 * both can be loaded side by side (dlopen RTLD_LOCAL; each keeps its own error text and last-kernel record).
 */

int sdetr_abi_version(void);
const char *sdetr_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * (1) MSDA forward, reference layout -- replaces
 *     ms_deformable_im2col_cuda<scalar_t>(stream, data_value, data_spatial_shapes,
 *         data_level_start_index, data_sampling_loc, data_attn_weight, batch_size, spatial_size,
 *         num_heads, channels, num_levels, num_query, num_point, data_col)
 *     models/bricks/ops/cuda/ms_deform_im2col_cuda.cuh:912-943 (kernel :226-288),
 *     called from ms_deform_attn_cuda_forward, models/bricks/ops/cuda/ms_deform_attn_cuda.cu:12-72.
 *   value [B,Nv,M,D]  shapes [L,2] int64 (H,W)  level_start_index [L] int64
 *   loc [B,Nq,M,L,P,2] (x,y in [0,1])  aw [B,Nq,M,L,P]  ->  data_col [B,Nq,M*D] (fully overwritten)
 * ------------------------------------------------------------------------------------------- */
int sdetr_msda_im2col_f32(sdetr_stream_t stream, const float *data_value,
                          const int64_t *data_spatial_shapes, const int64_t *data_level_start_index,
                          const float *data_sampling_loc, const float *data_attn_weight,
                          int batch_size, int spatial_size, int num_heads, int channels,
                          int num_levels, int num_query, int num_point, float *data_col);
int sdetr_msda_im2col_f64(sdetr_stream_t stream, const double *data_value,
                          const int64_t *data_spatial_shapes, const int64_t *data_level_start_index,
                          const double *data_sampling_loc, const double *data_attn_weight,
                          int batch_size, int spatial_size, int num_heads, int channels,
                          int num_levels, int num_query, int num_point, double *data_col);

/* ---------------------------------------------------------------------------------------------
 * (2) MSDA backward, reference layout -- replaces
 *     ms_deformable_col2im_cuda<scalar_t>(stream, grad_col, data_value, ..., grad_value,
 *         grad_sampling_loc, grad_attn_weight)
 *     models/bricks/ops/cuda/ms_deform_im2col_cuda.cuh:945-1316 (kernels :290-909),
 *     called from ms_deform_attn_cuda_backward, ms_deform_attn_cuda.cu:75-145.
 *   grad_value must be zero on entry (the op accumulates with float atomics, as the reference
 *   does); grad_sampling_loc / grad_attn_weight are fully overwritten.
 * ------------------------------------------------------------------------------------------- */
int sdetr_msda_col2im_f32(sdetr_stream_t stream, const float *grad_col, const float *data_value,
                          const int64_t *data_spatial_shapes, const int64_t *data_level_start_index,
                          const float *data_sampling_loc, const float *data_attn_weight,
                          int batch_size, int spatial_size, int num_heads, int channels,
                          int num_levels, int num_query, int num_point, float *grad_value,
                          float *grad_sampling_loc, float *grad_attn_weight);
int sdetr_msda_col2im_f64(sdetr_stream_t stream, const double *grad_col, const double *data_value,
                          const int64_t *data_spatial_shapes, const int64_t *data_level_start_index,
                          const double *data_sampling_loc, const double *data_attn_weight,
                          int batch_size, int spatial_size, int num_heads, int channels,
                          int num_levels, int num_query, int num_point, double *grad_value,
                          double *grad_sampling_loc, double *grad_attn_weight);

/* LDS-accumulating variant of sdetr_msda_col2im_f32 (csrc/msda_backward_tiled.hip), same operands, same results up to
 * summation order: grad_value is accumulated in per-workgroup LDS windows in fixed point (32x16-pixel tiles + 5-pixel
 * halo of the large levels, small levels held whole) and flushed once per window as whole-line fp32 atomics; samples
 * outside their window fall back to global atomics.  Shape: channels = 32, num_point = 4, num_levels <= 8, num_heads <= 64, value
 * below 4 GiB per image (sdetr_msda_col2im_lds_supported).  Level geometry is read on the device from
 * data_spatial_shapes / data_level_start_index (no host copy).  `workspace` = device scratch of at least
 * sdetr_msda_col2im_lds_workspace_bytes(batch_size, num_query, num_heads, num_levels) bytes (query bucketing,
 * row bounds, work counter; the entry clears what it needs).
 * sdetr_msda_last_backward_kernel(): SDETR_KERNEL_MSDA_BWD_* of the calling thread's last backward call. */
#define SDETR_KERNEL_MSDA_BWD_DIRECT 1 /* msda_col2im_*: every contribution an fp32 atomic on global memory */
#define SDETR_KERNEL_MSDA_BWD_LDS 2    /* bt_main_kernel: fixed-point accumulation in LDS windows */
size_t sdetr_msda_col2im_lds_workspace_bytes(int batch_size, int num_query, int num_heads, int num_levels);
int sdetr_msda_col2im_lds_supported(int num_heads, int channels, int num_levels, int num_point, int spatial_size);
int sdetr_msda_col2im_lds_f32(sdetr_stream_t stream, const float *grad_col, const float *data_value,
                              const int64_t *data_spatial_shapes, const int64_t *data_level_start_index,
                              const float *data_sampling_loc, const float *data_attn_weight,
                              int batch_size, int spatial_size, int num_heads, int channels,
                              int num_levels, int num_query, int num_point, float *grad_value,
                              float *grad_sampling_loc, float *grad_attn_weight, void *workspace,
                              size_t workspace_bytes);
int sdetr_msda_last_backward_kernel(void);

/* ---------------------------------------------------------------------------------------------
 * (3) Native MI355X path behind MultiScaleDeformableAttention.forward
 *     (models/bricks/ms_deform_attn.py:286-377).
 *
 * sdetr_value_to_head_major: the tail of `value_proj` (ms_deform_attn.py:316-321): zero the
 *   padded tokens (masked_fill) and re-lay the projected value out head-major,
 *   src [B,Nv,G*M*D] (row stride `src_row_stride` elements) -> dst [G,B,M,Nv,D], optional dtype cast;
 *   G = num_groups lets one launch split the batched value projection of all encoder layers
 *   (the six layers sample the same, never-updated feature map).  pad_mask [B,Nv] bytes or NULL.
 *
 * sdetr_msda_fused_forward: softmax over the L*P logits, sampling-location arithmetic
 *   (ms_deform_attn.py:322-355, 2-d or 4-d reference points) and the gather-reduce, in one launch.
 *   value_hm  [B,M,Nv,D] (value_dtype)        ref_points [B,Nq,L,ref_dim] f32, ref_dim in {2,4}; images are
 *             ref_batch_stride floats apart (0 = contiguous), so a row prefix of a longer buffer can be passed
 *   proj      [B,Nq,row_stride] (proj_dtype): row = [ M*L*P*2 offsets | M*L*P logits | ... ]
 *             i.e. the concatenated output of sampling_offsets and attention_weights Linear;
 *             with proj_head_major != 0 instead [B,M,Nq,3*L*P]: per (image, head, query) the 2*L*P offsets then the
 *             L*P logits (D=32, L=P=4, bf16 only) -- each head's slab is then read by one XCD only.
 *   order     [B,Nq] int32 processing order (slot i handles query order[b][i]) or NULL
 *   out       [B,Nq,M*D] (out_dtype)
 * ------------------------------------------------------------------------------------------- */
int sdetr_value_to_head_major(sdetr_stream_t stream, const void *src, int src_dtype,
                              int64_t src_row_stride, const uint8_t *pad_mask, int batch_size,
                              int spatial_size, int num_heads, int channels, int num_groups, void *dst,
                              int dst_dtype);

int sdetr_msda_fused_forward(sdetr_stream_t stream, const void *value_hm, int value_dtype,
                             const int64_t *data_spatial_shapes,
                             const int64_t *data_level_start_index, const float *ref_points,
                             int ref_dim, int64_t ref_batch_stride, const void *proj, int proj_dtype,
                             int64_t proj_row_stride, int proj_head_major,
                             const int32_t *order, int batch_size, int spatial_size, int num_heads,
                             int channels, int num_levels, int num_query, int num_point, void *out,
                             int out_dtype);

/* Coarse-levels-in-LDS variant of sdetr_msda_fused_forward (csrc/msda_resident.hip): one persistent 1024-thread
 * workgroup per CU copies levels 2 and 3 of its (image, head) into LDS once and serves their samples from there, so
 * the vector memory path only carries the level-0 / level-1 samples.  Same arithmetic and results as the direct
 * kernel.  Shape: 4 levels, 4 points, 32 channels per head, fp16 | bf16 head-major value, bf16 head-major projection
 * slab [B,M,Nq,48], levels 2+3 no larger than sdetr_msda_resident_max_pixels() pixels.
 *   level_hw_host: HOST pointer to the 4 (H, W) pairs as int32 (the kernel takes the geometry as launch arguments
 *                  instead of reading data_spatial_shapes / data_level_start_index from device memory)
 *   chunks:        workgroups per (image, head); <= 0 = one workgroup per CU
 * sdetr_msda_last_kernel(): which forward kernel the calling thread's last MSDA forward call dispatched to
 *   (SDETR_KERNEL_*; lets a parity test assert that it exercised the kernel it names). */
#define SDETR_KERNEL_MSDA_GENERIC 1  /* one thread per output element, reference layout, any shape */
#define SDETR_KERNEL_MSDA_GATHER 2   /* msda_gather_kernel: lane groups per row, LDS descriptor table */
#define SDETR_KERNEL_MSDA_L4P4 3     /* msda_gather_l4p4_kernel: the Salience-DETR shape, direct gather */
#define SDETR_KERNEL_MSDA_RESIDENT 4 /* msda_resident_kernel: levels 2+3 resident in LDS */
int sdetr_msda_resident_max_pixels(void);
int sdetr_msda_resident_forward(sdetr_stream_t stream, const void *value_hm, int value_dtype,
                                const int32_t *level_hw_host, const float *ref_points, int ref_dim,
                                int64_t ref_batch_stride, const void *proj_head_major_bf16, int batch_size,
                                int spatial_size, int num_heads, int num_query, void *out, int out_dtype, int chunks);
/* ..._ex: `image_lanes` >= 0 fixes how many images the chip works on at a time (0 = all), -1 = by the maps' size. */
int sdetr_msda_resident_forward_ex(sdetr_stream_t stream, const void *value_hm, int value_dtype,
                                   const int32_t *level_hw_host, const float *ref_points, int ref_dim,
                                   int64_t ref_batch_stride, const void *proj_head_major_bf16, int batch_size,
                                   int spatial_size, int num_heads, int num_query, void *out, int out_dtype, int chunks,
                                   int image_lanes);
int sdetr_msda_last_kernel(void);

/* The same kernel on BORDERED head-major maps (round 4, csrc/msda_resident.hip `msda_bordered_kernel`): every level of
 * the fp16 map [B, M, Np, 32] is stored as (H_l + 2) rows of (W_l + 1) 64-byte records -- row -1 and row H_l are zero
 * records, record -1 of every row is zero and doubles as record W_l of the row above -- the levels follow each other
 * and one more zero record closes the map: Np = sdetr_msda_bordered_records(level_hw, 4).  Pixel (y, x) of level l is
 * record P_l + (y + 1)(W_l + 1) + x + 1.  The producer is sdetr_value_proj_head_major(..., pixel_map, border, ...);
 * with zero borders the kernel needs none of the corner tests of ms_deform_im2col_cuda.cuh:31-66 (a position is clamped
 * onto the border, whose records are zero), which removes ~150 of the ~830 vector instructions per 16 rows.
 *   row_order: optional int32 [B, row_order_batch_stride >= Nq], a permutation of 0..Nq-1 per image: the rows are
 *              processed in this order (neighbours in the image next to each other keep the fine levels' records in
 *              the L1); results do not depend on it.  NULL = 0..Nq-1. */
#define SDETR_KERNEL_MSDA_BORDERED 5 /* msda_bordered_kernel: bordered maps, levels 2+3 resident in LDS, rows in list order */
#define SDETR_KERNEL_MSDA_BORDERED_ORDERED 6 /* the same kernel walking the caller's row_order (tile-major in the encoder) */
int64_t sdetr_msda_bordered_records(const int32_t *level_hw_host, int num_levels);
int sdetr_msda_bordered_max_resident_records(void);
int sdetr_msda_bordered_forward(sdetr_stream_t stream, const void *value_bordered, int value_dtype,
                                const int32_t *level_hw_host, const float *ref_points, int ref_dim,
                                int64_t ref_batch_stride, const void *proj_head_major_bf16, const int32_t *row_order,
                                int64_t row_order_batch_stride, int batch_size, int bordered_records, int num_heads,
                                int num_query, void *out, int out_dtype, int chunks);
/* The same call with its launch choices as explicit arguments (round 6: nothing in the product libraries reads the
 * environment).  sdetr_msda_bordered_forward == ..._ex(..., SDETR_MSDA_ACC_DEFAULT, -1, -1).
 *   accumulate:  how a 16-bit output's corner products are summed (fp32 outputs always take the exact form):
 *                SDETR_MSDA_ACC_EXACT         fp32 products and sums (ms_deform_im2col_cuda.cuh:226-288 rounded once at the end)
 *                SDETR_MSDA_ACC_PACKED_SAMPLE a sample's four corners summed in packed fp16 (v_pk_fma_f16), fp32 across samples
 *                SDETR_MSDA_ACC_PACKED_LEVEL  a level's four samples x four corners summed in packed fp16, fp32 across levels
 *                SDETR_MSDA_ACC_DEFAULT       the library's choice: PACKED_LEVEL for bf16 activations, EXACT for fp16 ones
 *   image_lanes: >= 0 fixes how many images the chip works on at a time (0 = all); -1 = by the maps' total size
 *   l2_warmup:   >= 0 fixes the warm-up mask (bit 0 fine levels, bit 1 projection rows); -1 = by the maps' size */
#define SDETR_MSDA_ACC_DEFAULT (-1)
#define SDETR_MSDA_ACC_EXACT 0
#define SDETR_MSDA_ACC_PACKED_SAMPLE 1
#define SDETR_MSDA_ACC_PACKED_LEVEL 2
int sdetr_msda_bordered_forward_ex(sdetr_stream_t stream, const void *value_bordered, int value_dtype,
                                   const int32_t *level_hw_host, const float *ref_points, int ref_dim,
                                   int64_t ref_batch_stride, const void *proj_head_major_bf16, const int32_t *row_order,
                                   int64_t row_order_batch_stride, int batch_size, int bordered_records, int num_heads,
                                   int num_query, void *out, int out_dtype, int chunks, int accumulate, int image_lanes,
                                   int l2_warmup);

/* Same gather on a head-major value with explicit sampling locations / weights (the reference
 * op's math on the native layout); loc/aw as in (1), fp32. */
int sdetr_msda_forward_head_major(sdetr_stream_t stream, const void *value_hm, int value_dtype,
                                  const int64_t *data_spatial_shapes,
                                  const int64_t *data_level_start_index, const float *data_sampling_loc,
                                  const float *data_attn_weight, int batch_size, int spatial_size,
                                  int num_heads, int channels, int num_levels, int num_query,
                                  int num_point, void *out, int out_dtype);

/* The encoder layer's dense self-attention over its selected rows, bf16, embed_dim 256, 8 heads
 * (models/bricks/salience_transformer.py:366-379: gather of select_tgt / select_pos, nn.MultiheadAttention with
 * q = k = x + pos and v = x, residual + pre_norm, scatter back) in two launches (csrc/topk_attention.hip):
 *   query   [B, num_rows, 256] bf16, images query_batch_stride elements apart: rows selected[b][i] are the inputs AND
 *           receive pre_norm(x + out_proj(attention)) in place
 *   pos     [B, >= num_rows, 256] bf16 position rows in the same row order, images pos_batch_stride elements apart
 *   selected [B, num_selected] int64 distinct row numbers
 *   in_proj_weight [768,256] / in_proj_bias [768], out_proj_weight [256,256] / out_proj_bias [256], norm weight /
 *   bias [256]: bf16 device tensors exactly as nn.MultiheadAttention / nn.LayerNorm hold them
 *   workspace: sdetr_topk_attention_workspace_bytes(B, num_selected) bytes of device scratch (q / k rows, V^T). */
int64_t sdetr_topk_attention_workspace_bytes(int batch_size, int num_selected);
int sdetr_topk_attention_bf16(sdetr_stream_t stream, void *query, int64_t query_batch_stride, const void *pos,
                              int64_t pos_batch_stride, const int64_t *selected, int batch_size, int num_rows,
                              int num_selected, const void *in_proj_weight, const void *in_proj_bias,
                              const void *out_proj_weight, const void *out_proj_bias, const void *norm_weight,
                              const void *norm_bias, float norm_eps, int embed_dim, int num_heads, void *workspace,
                              int64_t workspace_bytes);

/* ---------------------------------------------------------------------------------------------
 * (4) Salience filtering: masked top-k, sorted descending, ties -> lower index first.
 *     Replaces the torch.topk / torch.sort calls of the inline filtering code
 *     models/bricks/salience_transformer.py:146-150 (per-level top-k with
 *     masked_fill(mask, score.min())), :156-158 (global sort + index gather) and :366-367
 *     (per-layer top-300).
 *   score [B,N] f32; mask [B,N] bytes (rows mask_row_stride bytes apart, 0 = N) or NULL.  fill_mode 0: masked
 *   scores are left alone (mask must be NULL); 1: masked scores are replaced by min(score) over the WHOLE
 *   [B,N] array (computed here, masked entries included -- reference :146); 2: by the device scalar
 *   *fill_value (e.g. the minimum sdetr_salience_head_stage2 already produced).
 *   payload [B,N] int64 or NULL: out_index[b][j] = payload[b][pos] if given, else pos+index_offset.
 *   out_score [B,k] f32 (may be NULL), out_index [B,k] int64, rows out_row_stride elements apart (0 = k) so that
 *   several calls can fill column blocks of one buffer (the concatenation of :155).
 *   workspace: device scratch of at least sdetr_topk_workspace_bytes(B,N,k) bytes (may be NULL
 *   when that returns 0, i.e. whenever the problem fits the in-LDS path).
 * ------------------------------------------------------------------------------------------- */
size_t sdetr_topk_workspace_bytes(int batch_size, int n, int k);
/* 1 when sdetr_masked_topk_desc_f32 runs its prefilter launch for this shape (then the call is two launches) */
int sdetr_topk_uses_prefilter(int n, int k);
int sdetr_masked_topk_desc_f32(sdetr_stream_t stream, const float *score, const uint8_t *mask,
                               int64_t mask_row_stride, int fill_mode, const float *fill_value,
                               const int64_t *payload, int batch_size, int n, int k,
                               int64_t index_offset, float *out_score, int64_t *out_index,
                               int64_t out_row_stride, void *workspace, size_t workspace_bytes);

/*   sdetr_masked_topk_sliced_f32 (round 6): the same top-k (fill_mode 2 semantics: masked entries compete with
 *   *fill_value; ties -> lower position first; bit-identical results) for k a sizeable fraction of a long row, as two
 *   chip-wide launches: the row is cut into `slices` (2..8) slices, every slice is sorted completely by rank counting,
 *   the sorted slices are merged and the first k written.  Replaces torch.topk at salience_transformer.py:150 for the
 *   finest level (6680 of 16 800 scores at 800 x 1333; 16 700 of 67 200 on the 5scale pyramid).
 *   workspace: sdetr_topk_sliced_workspace_bytes(B, n) bytes.  score contiguous [B, n]. */
size_t sdetr_topk_sliced_workspace_bytes(int batch_size, int n);
int sdetr_masked_topk_sliced_f32(sdetr_stream_t stream, const float *score, const uint8_t *mask, int64_t mask_row_stride,
                                 const float *fill_value, int batch_size, int n, int k, int slices, int64_t index_offset,
                                 float *out_score, int64_t *out_index, int64_t out_row_stride, void *workspace,
                                 size_t workspace_bytes);

/*   sdetr_merge_sorted_desc: stable descending sort of a [B,n] score array that is the concatenation of
 *   num_segments segments (host array segment_start, first = 0), each already sorted descending with ties in position
 *   order -- exactly what the per-level calls above leave in the column blocks of one buffer -- done as a merge
 *   (one thread per element, num_segments-1 binary searches).  out_index[b][rank] = payload[b][pos]; out_score
 *   optional.  Bit-identical to the stable sort of salience_transformer.py:156-158. */
int sdetr_merge_sorted_desc(sdetr_stream_t stream, const float *score, const int64_t *payload,
                            const int *segment_start, int num_segments, int batch_size, int n, int64_t *out_index,
                            float *out_score);

/* ---------------------------------------------------------------------------------------------
 * (5) Token row movement of the encoder loop (models/bricks/salience_transformer.py:454-461
 *     torch.gather by foreground_inds; :474-485 per-image scatter of the first
 *     focus_token_nums[b] rows).  Rows are `row_bytes` bytes (multiple of 4).
 *   gather : dst[b][i] = src[b][idx[b][i]]  (+ addend[b][idx[b][i]] if addend != NULL, f32 only)
 *   scatter: dst[b][idx[b][i]] = src[b][i] for i < (count ? count[b] : n)    (in place)
 * ------------------------------------------------------------------------------------------- */
int sdetr_gather_rows(sdetr_stream_t stream, const void *src, const int64_t *idx, int batch_size,
                      int src_rows, int n, int row_bytes, void *dst);
int sdetr_scatter_rows(sdetr_stream_t stream, void *dst, const int64_t *idx, const void *src,
                       const int64_t *count, int batch_size, int dst_rows, int n, int row_bytes);

/* Row movers for index sets that are PREFIXES of one sorted list -- what salience_transformer.py:156-163 builds
 * (foreground_inds[k] = sorted_index[:, :rows_k], rows_k non-increasing).  The loop then keeps the tokens in sorted
 * order between layers instead of scattering to / gathering from token space (:454-485).
 *   sdetr_advance_rows: after a layer.  For i < rows:  live = i < count[b] (count NULL = all);
 *       live rows: sorted_result[b,i] = layer_out[b,i];
 *       next_query[b,i] (i < next_rows) = live ? layer_out[b,i] : tokens[b, sorted_index[b,i]]   (rows past the
 *       image's focus count are never updated, :474-485).  sorted_result [batch, sorted_rows, .], tokens
 *       [batch, spatial_size, .], sorted_index rows index_batch_stride apart; next_query may be NULL (next_rows 0).
 *   sdetr_select_stack: out[b,i] = query[b,index[b,i]] + pos[b,index[b,i]], out[b,num_select+i] = query[b,index[b,i]]
 *       (the q/k and v inputs of the top-k dense self-attention, :366-376); query / pos images are *_batch_stride
 *       elements apart; dtype SDETR_F32 | SDETR_BF16; out [batch, 2*num_select, channels].
 *   sdetr_encoder_finalize: token space again + background embedding (:487-495):
 *       out[b,t] = tokens[b,t] + (padding[b,t] ? 0 : background[t])                       for t not in the sorted list
 *       out[b,t] = (i < count[b] ? sorted_result[b,i] : tokens[b,t]) + (i >= last_rows && !padding[b,t] ? background[t] : 0)
 *                                                                                           for t = sorted_index[b,i]. */
int sdetr_advance_rows(sdetr_stream_t stream, const void *layer_out, void *sorted_result, void *next_query,
                       const void *tokens, const int64_t *sorted_index, int64_t index_batch_stride,
                       const int64_t *count, int batch_size, int rows, int sorted_rows, int next_rows,
                       int spatial_size, int row_bytes);
int sdetr_select_stack(sdetr_stream_t stream, const void *query, int64_t query_batch_stride, const void *pos,
                       int64_t pos_batch_stride, const int64_t *index, int batch_size, int num_select, int channels,
                       int dtype, void *out);
/* sdetr_encoder_finalize_sorted: the second launch of sdetr_encoder_finalize alone (same arguments; `out` already holds
 * the token-space pass, e.g. from a sdetr_finalize_job). */
int sdetr_encoder_finalize_sorted(sdetr_stream_t stream, const void *tokens, const void *sorted_result,
                                  const int64_t *sorted_index, const int64_t *count, const void *background,
                                  const uint8_t *padding_mask, int batch_size, int spatial_size, int sorted_rows,
                                  int last_rows, int channels, int dtype, void *out);
/* Row orders for sdetr_msda_bordered_forward (round 4): for every encoder layer k the rows 0 .. counts[k]-1 of the sorted
 * list (sorted_index [batch, num_rows] int64: the token of every row, salience_transformer.py:156-163) in TILE-MAJOR
 * order of their tokens -- tile_pos int32 [spatial_size] is the position of every token in a static order that keeps
 * neighbours in the image together (tiles of the finest level; tokens of all levels by the tile their centre falls
 * into).  order int32 [num_layers][batch][order_batch_stride]: order[k][b][0 .. counts[k]) is a permutation of
 * 0 .. counts[k]-1.  counts_dev: device int32 [num_layers].  Bit-exact index work; since round 5 one workgroup per
 * (image, layer, part of the tile positions): any pyramid size (the reference's 5scale configuration, 89 250 tokens, is
 * six parts), at most 65 534 rows per image.  A list that holds a token twice or tokens outside the pyramid still gets a
 * permutation: such rows follow their part's run / close the order. */
int sdetr_layer_row_orders(sdetr_stream_t stream, const int64_t *sorted_index, int64_t index_batch_stride,
                           const int32_t *tile_pos, int batch_size, int spatial_size, int num_rows, int num_layers,
                           const int32_t *counts_dev, int32_t *order, int64_t order_batch_stride);
/* The same as a job another launch carries: sdetr_masked_topk_desc_with_orders_f32 = sdetr_masked_topk_desc_f32 + the job
 * (when the selection is the one-workgroup-per-row histogram sort, the jobs' workgroups ride in its launch -- the encoder's
 * first top-300 selection runs 2 workgroups on an otherwise empty chip; otherwise the job is launched behind it). */
typedef struct {
    const int64_t *sorted_index; /* [batch, num_rows] rows index_batch_stride apart (0 = num_rows) */
    int64_t index_batch_stride;
    const int32_t *tile_pos;     /* [spatial_size] */
    int batch, spatial_size, num_rows, num_layers;
    const int32_t *counts;       /* device int32 [num_layers] */
    int32_t *order;              /* [num_layers][batch][order_batch_stride] */
    int64_t order_batch_stride;  /* 0 = num_rows */
} sdetr_row_orders_job;
int sdetr_masked_topk_desc_with_orders_f32(sdetr_stream_t stream, const float *score, const uint8_t *mask,
                                           int64_t mask_row_stride, int fill_mode, const float *fill_value,
                                           const int64_t *payload, int B, int n, int k, int64_t index_offset,
                                           float *out_score, int64_t *out_index, int64_t out_row_stride,
                                           void *workspace, size_t workspace_bytes, const sdetr_row_orders_job *job);
int sdetr_encoder_finalize(sdetr_stream_t stream, const void *tokens, const void *sorted_result,
                           const int64_t *sorted_index, const int64_t *count, const void *background,
                           const uint8_t *padding_mask, int batch_size, int spatial_size, int sorted_rows,
                           int last_rows, int channels, int dtype, void *out);

/* ---------------------------------------------------------------------------------------------
 * (6) Plumbing that feeds / surrounds the path.
 *   sdetr_pyramid_flatten_level: one level of flatten_multi_level + get_lvl_pos_embed
 *     (models/bricks/base_transformer.py:22-33) and the token validity of gen_encoder_output_proposals
 *     (:74-112): NCHW feat/pos -> token-major feat_out, pos_out (= pos + level_embed),
 *     sum_out = (feat + pos_out) * keep (the input of enc_output), mask_out (flattened padding mask),
 *     optional bf16 copies; feat_out / pos_out may be NULL when only the bf16 copies are wanted.  mask [B,H,W]
 *     bytes (1 = padding); outputs are [B,S,C] / [B,S], this level
 *     occupying tokens [level_start, level_start + H*W).
 *   sdetr_class_max_times: out[b,i] = max_c score[b,i,c] * scale[b,i]  (mc_score of
 *     models/bricks/salience_transformer.py:366), score [batch,rows_per_batch,num_classes] f32|bf16 contiguous,
 *     scale f32 with images scale_batch_stride floats apart, out f32 [batch,rows_per_batch].
 * ------------------------------------------------------------------------------------------- */
int sdetr_pyramid_flatten_level(sdetr_stream_t stream, const float *feat, const float *pos, const uint8_t *mask,
                                const float *level_embed, int batch_size, int channels, int height, int width,
                                int level, int level_start, int spatial_size, float *feat_out, float *pos_out,
                                float *sum_out, uint8_t *mask_out, void *feat_bf16, void *pos_bf16,
                                float *valid_ratio /* this level's (w,h) of image 0, or NULL */,
                                int valid_ratio_stride /* floats between images */);
/*   sdetr_pyramid_flatten: the whole pyramid in ONE launch -- the outputs of `num_levels` calls of
 *     sdetr_pyramid_flatten_level with level_start = the running pixel count (bit-identical).  feats / pos / masks are
 *     HOST arrays of `num_levels` (<= 8) device pointers, heights / widths host arrays; level_embeds [num_levels,C] and
 *     valid_ratios [batch,num_levels,2] (or NULL) live on the device; spatial_size must be the pyramid's pixel count. */
int sdetr_pyramid_flatten(sdetr_stream_t stream, int num_levels, const float *const *feats, const float *const *pos,
                          const uint8_t *const *masks, const int *heights, const int *widths, const float *level_embeds,
                          int batch_size, int channels, int spatial_size, float *feat_out, float *pos_out,
                          float *sum_out, uint8_t *mask_out, void *feat_bf16, void *pos_bf16, float *valid_ratios);
/*   sdetr_masked_fill_min: out[i] = mask[i] ? min(mins[0..num_mins)) : score[i] -- foreground_score of
 *     models/bricks/salience_transformer.py:164-168 from the per-level minima that are already known.
 *   sdetr_encoder_reference_points: get_reference_points (:418-432) for the tokens index[b][i] (index NULL: token i):
 *     out [batch, rows, num_levels, 2]; valid_ratios [batch, num_levels, 2]; shapes / level_start_index int64. */
int sdetr_masked_fill_min(sdetr_stream_t stream, const float *score, const uint8_t *mask, const float *mins,
                          int num_mins, int64_t total, float *out);
int sdetr_encoder_reference_points(sdetr_stream_t stream, const float *valid_ratios, const int64_t *shapes,
                                   const int64_t *level_start_index, const int64_t *index, int64_t index_batch_stride,
                                   int batch_size, int rows, int num_levels, float *out);
/*   sdetr_encoder_prepare_sorted: entry of the sorted-order encoder loop in one launch -- for the tokens
 *     index[b][i] (i < rows; images index_batch_stride apart): query_out / pos_out [batch, rows, row_bytes] = the
 *     tokens' rows of `tokens` / `pos` [batch, S, row_bytes] (the gathers of salience_transformer.py:454-461),
 *     score_out [batch, rows] = score[b][index] (NULL: skipped) and reference_points_out [batch, rows, num_levels, 2]
 *     as sdetr_encoder_reference_points.  row_bytes = 16 * (a divisor of 256).
 *     score_mask / score_mins (round 6, both or neither): `score` is the UNFILLED flattened salience score and the gather
 *     applies sdetr_masked_fill_min on the fly -- score_out = score_mask[b][index] ? min(score_mins[0..num_mins)) :
 *     score[b][index] (foreground_score of salience_transformer.py:164-168 restricted to the rows the encoder reads). */
int sdetr_encoder_prepare_sorted(sdetr_stream_t stream, const void *tokens, const void *pos, int row_bytes,
                                 const float *score, const int64_t *index, int64_t index_batch_stride, int batch_size,
                                 int spatial_size, int rows, const float *valid_ratios, const int64_t *shapes,
                                 const int64_t *level_start_index, int num_levels, void *query_out, void *pos_out,
                                 float *score_out, float *reference_points_out, const uint8_t *score_mask,
                                 const float *score_mins, int num_mins);
/*   sdetr_encoder_prepare_sorted_scored: the same launch also writes the FIRST layer's selection score of the gathered rows,
 *     class_score_out [batch, rows] = max_c(class_head(query_out[b][i])) * score_out-value[b][i]
 *     (models/bricks/salience_transformer.py:462, 366) -- 512-byte rows of 16-bit activations only; class_packed =
 *     sdetr_class_head_pack_bf16(class head weight [num_classes <= 96, 256]), class_bias_padded fp32 [96] (-inf on the
 *     padded classes); `score` is required (score_out may still be NULL). */
int sdetr_encoder_prepare_sorted_scored(sdetr_stream_t stream, const void *tokens, const void *pos, int row_bytes,
                                        const float *score, const int64_t *index, int64_t index_batch_stride,
                                        int batch_size, int spatial_size, int rows, const float *valid_ratios,
                                        const int64_t *shapes, const int64_t *level_start_index, int num_levels,
                                        void *query_out, void *pos_out, float *score_out, float *reference_points_out,
                                        const uint8_t *score_mask, const float *score_mins, int num_mins,
                                        const void *class_packed, const float *class_bias_padded, float *class_score_out);
int sdetr_class_max_times(sdetr_stream_t stream, const void *score, int score_dtype, const float *scale,
                          int64_t scale_batch_stride, int batch_size, int rows_per_batch, int num_classes, float *out);

/*   sdetr_layernorm: out = LayerNorm((x [+ residual]) * (1 + row_scale[row] * *alpha)) * gamma + beta, eps as
 *     nn.LayerNorm.  Covers the encoder's residual LayerNorms (models/bricks/salience_transformer.py:347-351,
 *     377-378, 390-391), the level modulation + MaskPredictor LayerNorm (:143, :20) and enc_output_norm
 *     (models/bricks/base_transformer.py:111).  x / residual are [batch_size, rows_per_batch, channels] with the
 *     given element strides (last dim contiguous); residual, row_scale ([batch*rows] f32) and alpha (device
 *     scalar) may be NULL; out is contiguous, or -- with scatter_index [batch*rows] -- row (b,i) is written to row
 *     scatter_index[b,i] of out [batch, out_batch_rows, channels] (norm + scatter of :377-379 in one pass); with
 *     gather_x the x row is read at that index too (x = the buffer being updated: gather + norm + scatter).
 *     dtypes: SDETR_F32 | SDETR_BF16 (x and residual share x_dtype).
 *   sdetr_column_mean_f32: out[b,c] = mean_i x[b,i,c]   (global half of the salience head, :43-45). */
int sdetr_layernorm(sdetr_stream_t stream, const void *x, const void *residual, int x_dtype,
                    int64_t x_batch_stride, int64_t x_row_stride, int64_t res_batch_stride, int64_t res_row_stride,
                    const float *row_scale, const float *alpha, const void *gamma, const void *beta,
                    int param_dtype, float eps, int batch_size, int rows_per_batch, int channels, void *out,
                    int out_dtype, const int64_t *scatter_index, int64_t out_batch_rows, int gather_x);
int sdetr_column_mean_f32(sdetr_stream_t stream, const float *x, int64_t batch_stride, int64_t row_stride,
                          int batch_size, int rows, int channels, float *out);

/* ---- (6) the salience head, fused -----------------------------------------------------------------------------
 * MaskPredictor (models/bricks/salience_transformer.py:16-47) for embed_dim = hidden_dim = 256, optionally with
 * enc_output + enc_output_norm (models/bricks/base_transformer.py:110-111) in front, fp32 throughout (f32-input
 * MFMA).  All tensors fp32, contiguous unless strides are given.
 *
 *   sdetr_pack_linear_f32: packed[S][n][h][j] = weight[n][8S+4h+j] -- the operand order the stage kernels stream
 *     an nn.Linear weight ([out_features, in_features], row stride in elements) in; do it once per weight.
 *   sdetr_salience_head_blocks: number of token blocks stage 1 uses for a level (= rows of partial_sums).
 *   sdetr_salience_head_stage1: z = GELU(layer1.1(layer1.0(m + m * up * alpha))) with
 *       m  = enc_output_norm(enc_output(x))   when enc_weight_packed != NULL (memory_out, if given, receives m:
 *            [batch, tokens, 256] with the given batch stride), else m = x;
 *       up = row_scale[b, t] if given, else the bilinear align_corners=True resize of coarse_score
 *            [batch, coarse_h, coarse_w] to level_h x level_w (:139-142), else no modulation;  alpha: device scalar.
 *     Writes z[..., :128] to z_local [batch, tokens, 128] and the per-block column sums of z[..., 128:] to
 *     partial_sums [batch, blocks, 128] (the token mean of :43-45 is finished by stage 2).
 *   sdetr_salience_head_stage2: score = layer2(cat(z_local, mean(z_global))) -> score [batch, tokens] (and, if
 *     score_flat != NULL, also score_flat[b * score_flat_stride + t]).  weight2 is layer2.0.weight [128, 256]
 *     (unpacked, its [:, 128:] half multiplies the mean), weight2_local_packed = pack(weight2[:, :128]),
 *     weight3_packed = pack(layer2.2.weight [64,128]), weight4 = layer2.4.weight [1,64];
 *     const_workspace: [batch, 128] floats of scratch; score_min (optional device scalar) receives the minimum
 *     over all scores of the call -- the masked_fill value of salience_transformer.py:146-147.
 *   sdetr_salience_head_stage1_x3 / sdetr_pack_linear_bf16x3: stage 1 with its two 256 x 256 GEMMs on the bf16 matrix
 *     cores at fp32 accuracy (every fp32 operand split exactly into three bf16 terms, six bf16 MFMAs per product
 *     -- csrc/salience_head.hip); same arguments and results as sdetr_salience_head_stage1, the two weights packed by
 *     sdetr_pack_linear_bf16x3 instead (6 bytes per element: packed[k/16][n/32][plane][lane][8] bf16). */
int sdetr_pack_linear_f32(sdetr_stream_t stream, const float *weight, int64_t row_stride, int out_features,
                          int in_features, float *packed);
int sdetr_salience_head_blocks(int batch_size, int tokens);
int sdetr_salience_head_stage1(sdetr_stream_t stream, const float *x, int64_t x_batch_stride, int64_t x_row_stride,
                               int batch_size, int tokens, int channels, const float *enc_weight_packed,
                               const float *enc_bias, const float *enc_norm_weight, const float *enc_norm_bias,
                               float enc_norm_eps, const float *row_scale, const float *coarse_score, int coarse_h,
                               int coarse_w, int level_h, int level_w, const float *alpha, const float *norm_weight,
                               const float *norm_bias, float norm_eps, const float *weight_packed, const float *bias,
                               float *memory_out, int64_t memory_batch_stride, float *z_local, float *partial_sums);
int sdetr_pack_linear_bf16x3(sdetr_stream_t stream, const float *weight, int64_t row_stride, int out_features,
                             int in_features, void *packed);
int sdetr_salience_head_stage1_x3(sdetr_stream_t stream, const float *x, int64_t x_batch_stride, int64_t x_row_stride,
                                  int batch_size, int tokens, int channels, const void *enc_weight_x3,
                                  const float *enc_bias, const float *enc_norm_weight, const float *enc_norm_bias,
                                  float enc_norm_eps, const float *row_scale, const float *coarse_score, int coarse_h,
                                  int coarse_w, int level_h, int level_w, const float *alpha, const float *norm_weight,
                                  const float *norm_bias, float norm_eps, const void *weight_x3, const float *bias,
                                  float *memory_out, int64_t memory_batch_stride, float *z_local, float *partial_sums);
/* sdetr_salience_head_stage1_x3 and the value projection of `vp_num_groups` stacked layers (the arguments of
 * sdetr_value_proj_head_major; weight / bias / dst already point at the first of those layers) in ONE launch
 * (csrc/fused_head_value.hip): the stage-1 launches of the two coarsest levels leave the chip nearly empty, the value
 * projection depends only on the flattened tokens -- workgroups [0, blocks * batch) run stage 1, the rest the
 * projection. */
/* A masked top-k by rank counting, as a job another launch carries (the arguments of sdetr_masked_topk_desc_f32 with
 * fill_mode 2 -- or no mask --, no payload; rows short enough that no prefilter is involved):
 * sdetr_stage1_x3_with_jobs = sdetr_salience_head_stage1_x3 + an optional value-projection job (vp_x != NULL) + an
 * optional rank job in one launch. */
/* destination layout of a value-projection job (see sdetr_value_proj_head_major below) */
typedef struct {
    const int32_t *pixel_map;    /* device int32 [spatial_size]: record of every token of an image */
    const int32_t *border;       /* device int32 [num_border]: the zero records of a map */
    int num_border;
    int records;                 /* records per (image, head) */
} sdetr_bordered_layout;
typedef struct {
    const float *score;          /* [batch, n] */
    const uint8_t *mask;         /* [batch, n] rows mask_row_stride bytes apart, or NULL */
    int64_t mask_row_stride;
    const float *fill_value;     /* device scalar substituted for masked scores */
    int batch, n, k;
    int64_t index_offset;
    float *out_score;            /* [batch, k] rows out_row_stride apart, or NULL */
    int64_t *out_index;
    int64_t out_row_stride;
} sdetr_rank_job;
/* The token-space pass of sdetr_encoder_finalize (out = tokens + (padding ? 0 : background), bf16 rows of 256) as a
 * job: it depends on nothing the filtering or the encoder compute; sdetr_encoder_finalize_sorted then finishes alone. */
typedef struct {
    const void *tokens;          /* [batch, spatial_size, 256] bf16 */
    const void *background;      /* [spatial_size, 256] bf16 */
    const uint8_t *padding_mask; /* [batch, spatial_size] or NULL */
    int batch, spatial_size;
    void *out;                   /* [batch, spatial_size, 256] bf16 */
} sdetr_finalize_job;
/* sdetr_masked_topk_sliced_f32 whose merge launch (a few workgroups on an otherwise idle chip) carries a pending rank
 * job -- the deferred top-k of the next coarser level -- and / or the finalize pass (NULL: none).  *jobs_carried (may be
 * NULL) = 1 when the launch took them (rows of up to 24 576 keys), else 0 and the caller launches them itself. */
int sdetr_masked_topk_sliced_with_rank_f32(sdetr_stream_t stream, const float *score, const uint8_t *mask,
                                           int64_t mask_row_stride, const float *fill_value, int batch_size, int n, int k,
                                           int slices, int64_t index_offset, float *out_score, int64_t *out_index,
                                           int64_t out_row_stride, void *workspace, size_t workspace_bytes,
                                           const sdetr_rank_job *rank, const sdetr_finalize_job *finalize,
                                           int *jobs_carried);
int sdetr_stage1_x3_with_jobs(
    sdetr_stream_t stream, const float *x, int64_t x_batch_stride, int64_t x_row_stride, int batch_size, int tokens,
    int channels, const void *enc_weight_x3, const float *enc_bias, const float *enc_norm_weight,
    const float *enc_norm_bias, float enc_norm_eps, const float *row_scale, const float *coarse_score, int coarse_h,
    int coarse_w, int level_h, int level_w, const float *alpha, const float *norm_weight, const float *norm_bias,
    float norm_eps, const void *weight_x3, const float *bias, float *memory_out, int64_t memory_batch_stride,
    float *z_local, float *partial_sums, const void *vp_x, const void *vp_packed_weight, const float *vp_bias_padded,
    const uint8_t *vp_pad_mask, int vp_batch_size, int vp_spatial_size, int vp_num_heads, int vp_num_groups,
    void *vp_dst, int vp_dst_dtype, const sdetr_bordered_layout *vp_bordered, const sdetr_rank_job *rank,
    const sdetr_finalize_job *finalize);
/* The salience head with both 256 x 256 products of stage 1 OUT of the coarse-to-fine chain (round 6; reference
 * models/bricks/salience_transformer.py:20-24 layer1 = LN -> Linear -> GELU, :131-146 the modulation,
 * models/bricks/base_transformer.py:104-111 enc_output + norm).  The modulation is a per-token scalar s = 1 + up * alpha,
 * and LayerNorm of a scaled row is the unscaled row's times a scalar: with mu, sigma the statistics of the row x,
 *     layer1.Linear(LN(s x)) = k G + c0,   k = s sigma / sqrt(s^2 sigma^2 + eps),
 *     G = W (gamma (x - mu) / sigma),      c0 = W beta + b,
 * exactly (the scores agree with the unfactored evaluation to ~5e-7, the index sets of the reference's digests are
 * unchanged: tests/test_hotpath_gpu.py).  sdetr_salience_head_hoist_x3: ONE launch for all levels' tokens -- enc_output +
 * enc_output_norm (optional, memory_out optional), the row statistics and G (bf16 x 3 products at fp32 accuracy as in
 * sdetr_salience_head_stage1_x3); g_out [batch, tokens, 256] and sigma_out [batch, tokens] fp32 with batch strides;
 * it can carry a value-projection job and / or the finalize pass like sdetr_stage1_x3_with_jobs.
 * sdetr_salience_head_modulate: what is left per level -- resize of the coarser score (or row_scale) -> s -> k ->
 * GELU(k G + c0) -> z_local / partial_sums exactly as stage 1 leaves them (sdetr_salience_head_const and stage 2 follow);
 * g / sigma point at the level's first row; it can carry the rank job of the level before and the finalize pass. */
int sdetr_salience_head_hoist_x3(
    sdetr_stream_t stream, const float *x, int64_t x_batch_stride, int64_t x_row_stride, int batch_size, int tokens,
    int channels, const void *enc_weight_x3, const float *enc_bias, const float *enc_norm_weight,
    const float *enc_norm_bias, float enc_norm_eps, const float *norm_weight, const void *weight_x3, float *memory_out,
    int64_t memory_batch_stride, float *g_out, int64_t g_batch_stride, float *sigma_out, int64_t sigma_batch_stride,
    const void *vp_x, const void *vp_packed_weight, const float *vp_bias_padded, const uint8_t *vp_pad_mask,
    int vp_batch_size, int vp_spatial_size, int vp_num_heads, int vp_num_groups, void *vp_dst, int vp_dst_dtype,
    const sdetr_bordered_layout *vp_bordered, const sdetr_finalize_job *finalize);
int sdetr_salience_head_modulate(sdetr_stream_t stream, const float *g, int64_t g_batch_stride, const float *sigma,
                                 int64_t sigma_batch_stride, int batch_size, int tokens, const float *row_scale,
                                 const float *coarse_score, int coarse_h, int coarse_w, int level_h, int level_w,
                                 const float *alpha, float norm_eps, const float *c0, float *z_local, float *partial_sums,
                                 const sdetr_rank_job *rank, const sdetr_finalize_job *finalize, float *score_min_init);
/* score_min_init (may be NULL): device scalar set to +inf -- what sdetr_salience_head_const does for stage 2's running
 * minimum, for a stage 2 that takes the per-image constant in its own blocks (const_in_block != 0 of
 * sdetr_salience_head_stage2, partial_sums != NULL of sdetr_stage2_with_value_proj: every block sums the level's
 * partial sums -- up to 160 rows of them, i.e. 5120 tokens; worth it up to ~40 -- and takes the 128 x 128 product itself;
 * the const launch, 4.6-4.9 us on an idle chip per coarse level, does not exist then). */
int sdetr_stage1_x3_with_value_proj(
    sdetr_stream_t stream, const float *x, int64_t x_batch_stride, int64_t x_row_stride, int batch_size, int tokens,
    int channels, const void *enc_weight_x3, const float *enc_bias, const float *enc_norm_weight,
    const float *enc_norm_bias, float enc_norm_eps, const float *row_scale, const float *coarse_score, int coarse_h,
    int coarse_w, int level_h, int level_w, const float *alpha, const float *norm_weight, const float *norm_bias,
    float norm_eps, const void *weight_x3, const float *bias, float *memory_out, int64_t memory_batch_stride,
    float *z_local, float *partial_sums, const void *vp_x, const void *vp_packed_weight, const float *vp_bias_padded,
    const uint8_t *vp_pad_mask, int vp_batch_size, int vp_spatial_size, int vp_num_heads, int vp_num_groups,
    void *vp_dst, int vp_dst_dtype, const sdetr_bordered_layout *vp_bordered);
/* sdetr_salience_head_const: the per-image constant of stage 2 as a launch of its own; sdetr_stage2_with_value_proj:
 * stage 2 proper (arguments as sdetr_salience_head_stage2; const_workspace already filled) + a value-projection job
 * in one launch (csrc/fused_head_value.hip). */
int sdetr_salience_head_const(sdetr_stream_t stream, const float *partial_sums, int batch_size, int tokens,
                              const float *weight2, const float *bias2, float *const_workspace, float *score_min);
int sdetr_stage2_with_value_proj(sdetr_stream_t stream, const float *z_local, int batch_size, int tokens,
                                 const float *weight2_local_packed, const float *weight3_packed, const float *bias3,
                                 const float *weight4, const float *bias4, const float *const_workspace, float *score,
                                 float *score_flat, int64_t score_flat_stride, float *score_min, const void *vp_x,
                                 const void *vp_packed_weight, const float *vp_bias_padded, const uint8_t *vp_pad_mask,
                                 int vp_batch_size, int vp_spatial_size, int vp_num_heads, int vp_num_groups,
                                 void *vp_dst, int vp_dst_dtype, const sdetr_bordered_layout *vp_bordered,
                                 const void *weight2_local_x3,
                                 const float *partial_sums, const float *weight2, const float *bias2);
/* `weight2_local_x3` (both stage-2 entry points; round 6): sdetr_pack_linear_bf16x3 of W2[:, :128] or NULL.  Given, the
 * first product of stage 2 runs on the bf16 matrix cores at fp32 accuracy like stage 1's (exact three-way split, six
 * products) and `weight2_local_packed` is not read: without its f32 MFMAs the launch is a third as long. */
int sdetr_salience_head_stage2(sdetr_stream_t stream, const float *z_local, const float *partial_sums, int batch_size,
                               int tokens, const float *weight2, const float *bias2,
                               const float *weight2_local_packed, const float *weight3_packed, const float *bias3,
                               const float *weight4, const float *bias4, float *const_workspace, float *score,
                               float *score_flat, int64_t score_flat_stride, float *score_min,
                               const void *weight2_local_x3,
                               int const_in_block);

/* ---- (7) the encoder layer's feed-forward block, fused ----------------------------------------------------------
 * out = LayerNorm(x + W2 relu(W1 x + b1) + b2)   (models/bricks/salience_transformer.py:347-351 forward_ffn with the
 * released configuration: ReLU, embed_dim 256), bf16 activations and weights, fp32 accumulation / bias / LayerNorm.
 * The [tokens, hidden] intermediate never leaves the register file.
 *   sdetr_ffn_packed_bytes: size of the packed weight buffer for `hidden` units (hidden/32 chunks of 32 KiB).
 *   sdetr_ffn_pack_bf16: weight1 [hidden, 256] (linear1.weight), weight2 [256, hidden] (linear2.weight), bf16
 *     row-major -> packed: per 32 hidden units 16 W1 MFMA A-fragments (k-steps) then 16 W2 fragments
 *     ((e-tile, k-block), contraction index permuted to the accumulator layout), 1 KiB each, lane-ordered.
 *     Do it once per weight version.
 *   sdetr_ffn_fused_bf16: x, out [tokens, 256] bf16 contiguous (out may not alias x); bias1 [hidden], bias2,
 *     norm_weight, norm_bias [256] fp32.  hidden_splits > 1 cuts the hidden dimension into that many pieces per
 *     128-token block (more workgroups for small token counts) whose fp32 partial products go through `workspace`
 *     (sdetr_ffn_workspace_bytes) and are finished by a second launch; sdetr_ffn_auto_splits picks the count for
 *     the current device. */
int64_t sdetr_ffn_packed_bytes(int hidden);
int sdetr_ffn_auto_splits(int tokens, int hidden);
int64_t sdetr_ffn_workspace_bytes(int tokens, int hidden_splits);
int sdetr_ffn_pack_bf16(sdetr_stream_t stream, const void *weight1, const void *weight2, int embed_dim, int hidden,
                        void *packed);
int sdetr_ffn_fused_bf16(sdetr_stream_t stream, const void *x, const void *packed_weights, const float *bias1,
                         const float *bias2, const float *norm_weight, const float *norm_bias, float norm_eps,
                         int tokens, int embed_dim, int hidden, void *out, int hidden_splits, void *workspace,
                         int64_t workspace_bytes);
/* sdetr_ffn_fused_bf16 on the [batch_size * rows, 256] queries of an encoder layer, followed by that layer's
 * sdetr_advance_rows (same argument meaning) without the layer output ever being handed back: with hidden_splits > 1
 * the reduce + LayerNorm pass stores straight into sorted_result / next_query (one launch less per layer), with
 * hidden_splits == 1 the two existing launches run.  workspace: sdetr_ffn_workspace_bytes(batch_size * rows,
 * hidden_splits) bytes plus, for hidden_splits == 1, batch_size * rows * 512 bytes for the layer output. */
int sdetr_ffn_fused_advance_bf16(sdetr_stream_t stream, const void *x, const void *packed_weights, const float *bias1,
                                 const float *bias2, const float *norm_weight, const float *norm_bias, float norm_eps,
                                 int batch_size, int rows, int embed_dim, int hidden, int hidden_splits,
                                 void *workspace, int64_t workspace_bytes, void *sorted_result, void *next_query,
                                 const void *tokens, const int64_t *sorted_index, int64_t index_batch_stride,
                                 const int64_t *count, int sorted_rows, int next_rows, int spatial_size);

/* ---- (8) token-resident linear layers (256 input features, bf16) -------------------------------------------------
 * y = W x + b with the activations of 32 tokens resident in a wave's registers and the weights streamed through
 * LDS; the output tile (32 features) is consumed from registers by one of three epilogues.  fp32 accumulation.
 *   sdetr_linear_packed_bytes / sdetr_linear_pack_bf16: weight [out_features, 256] bf16 (row stride in elements)
 *     -> per 32 output features 16 lane-ordered 1 KiB MFMA A-fragments, padded with zero rows to whole steps
 *     of 128 features.
 *   bias_padded: fp32 [ceil(out_features/128)*128], zero past out_features.
 *   sdetr_token_linear_bf16: out[t, :out_features] = bf16(W (x[t] (+ x_add[t])) + b), out row stride in elements;
 *     x [tokens,256]; x_add (optional, e.g. the position embedding of salience_transformer.py:380-381) holds
 *     rows_per_batch rows per image, images x_add_batch_stride elements apart.  group_features > 0 stores
 *     feature-group-major instead: out [batch][out_features/group][rows_per_batch][group] (the head-major
 *     projection layout of sdetr_msda_fused_forward when the weight rows are ordered by head).  Replaces the
 *     sampling_offsets / attention_weights Linear of ms_deform_attn.py:322-329 (one 256 -> 384 call).
 *   sdetr_value_proj_head_major: value_proj + masked_fill + head-major re-layout (ms_deform_attn.py:316-321) for
 *     num_groups stacked projections: x [batch*spatial, 256] -> dst [groups][batch][heads][spatial][32] fp16|bf16.
 *   sdetr_class_head_max_times: out[b,i] = bf16(max_c (W x + b)[c]) * scale[b,i] -- mc_score of
 *     salience_transformer.py:366 without materialising the class logits; scale images scale_batch_stride apart. */
int64_t sdetr_linear_packed_bytes(int out_features);
int sdetr_linear_pack_bf16(sdetr_stream_t stream, const void *weight, int64_t row_stride, int out_features,
                           int in_features, void *packed);
int sdetr_token_linear_bf16(sdetr_stream_t stream, const void *x, const void *x_add, int64_t x_add_batch_stride,
                            int rows_per_batch, int tokens, int in_features, const void *packed_weight,
                            const float *bias_padded, int out_features, void *out, int64_t out_row_stride,
                            int group_features);
/*   Bordered destination (round 4; the layout sdetr_msda_bordered_forward reads): with `bordered` != NULL the maps are
 *     dst [groups][batch][heads][records][32], token i of an image goes to record pixel_map[i] and the launch also writes
 *     zeros to the `num_border` records of every map listed in `border` (the level borders and the closing record), so
 *     dst needs no initialisation.  NULL = the plain layout above. */
/* (sdetr_bordered_layout is declared with the salience head's jobs above) */
int sdetr_value_proj_head_major(sdetr_stream_t stream, const void *x, const void *packed_weight,
                                const float *bias_padded, const uint8_t *pad_mask, int batch_size, int spatial_size,
                                int in_features, int num_heads, int channels, int num_groups, void *dst, int dst_dtype,
                                const sdetr_bordered_layout *bordered);
int sdetr_class_head_max_times(sdetr_stream_t stream, const void *x, const void *packed_weight,
                               const float *bias_padded, int in_features, int num_classes, const float *scale,
                               int64_t scale_batch_stride, int batch_size, int rows_per_batch, float *out);
/*   sdetr_token_linear_ln_bf16: out = LayerNorm(residual + W x + b) for a 256 -> 256 Linear: output_proj + residual
 *     + norm1 of the deformable attention (salience_transformer.py:385-391), or out_proj + residual + pre_norm of
 *     the top-k dense attention written straight back to the selected query rows (:376-379, scatter_index [tokens]
 *     = row inside the image, out [batch, out_batch_rows, 256]).  residual: rows_per_batch rows per image, images
 *     residual_batch_stride elements apart; bias / norm_weight / norm_bias fp32 [256]. */
int sdetr_token_linear_ln_bf16(sdetr_stream_t stream, const void *x, const void *residual,
                               int64_t residual_batch_stride, int rows_per_batch, int tokens, int in_features,
                               const void *packed_weight, const float *bias, const float *norm_weight,
                               const float *norm_bias, float norm_eps, void *out, const int64_t *scatter_index,
                               int64_t out_batch_rows);

/* ---- (9) dense self-attention after the in-projection ---------------------------------------------------------------
 * out[b, i, 32 h + :] = softmax(Q_h K_h^T * scale) V_h for 32-channel heads, bf16, no mask: nn.MultiheadAttention's
 * attention proper for the encoder layer's selected queries (models/bricks/salience_transformer.py:371-376) and the
 * decoder layer's object queries (:565-570).  Element [b][i][32 h + d] of q / k / v lies at base + b * batch_stride +
 * i * row_stride + 32 h + d (so the three may be column slices of one in-projection output); out is
 * [batch, num_tokens, num_heads * 32] contiguous.  num_tokens <= 1152. */
int sdetr_attention_heads_bf16(sdetr_stream_t stream, const void *q, int64_t q_batch_stride, int64_t q_row_stride,
                               const void *k, int64_t k_batch_stride, int64_t k_row_stride, const void *v,
                               int64_t v_batch_stride, int64_t v_row_stride, int batch_size, int num_tokens,
                               int num_heads, int head_dim, float scale, void *out);

/* ---- (10) decoder box-refinement loop, elementwise stages (row N2) -----------------------------------------------
 * models/bricks/salience_transformer.py:641-671.
 *   sdetr_decoder_query_sine_embed: reference_points [batch, num_queries, 4] (cx, cy, w, h) and valid_ratios
 *     [batch, num_levels, 2] -> reference_points_input [batch, num_queries, num_levels, 4] = box * (rw, rh, rw, rh)
 *     (:642; may be NULL) and embed [batch, num_queries, 4*num_pos_feats] (f32 | bf16) =
 *     get_sine_pos_embed(reference_points_input[:, :, 0, :]) (:643; models/bricks/position_encoding.py:105-132 with
 *     exchange_xy: blocks in (y, x, w, h) order, feature pair (2p, 2p+1) = (sin, cos) of c * 2*pi / T^(2p/F)).
 *   sdetr_box_refine: out[g, i, :] = sigmoid(delta[g, i, :] + inverse_sigmoid(reference_points[i, :], eps))
 *     (:659-660 and :666-668; util/misc.py:31-35); delta rows (f32 | bf16) delta_row_stride elements apart in one
 *     [groups * num_boxes] sequence, out fp32 [groups, num_boxes, 4]. */
int sdetr_decoder_query_sine_embed(sdetr_stream_t stream, const float *reference_points, const float *valid_ratios,
                                   int batch_size, int num_queries, int num_levels, int num_pos_feats,
                                   float temperature, void *embed, int embed_dtype, float *reference_points_input);
int sdetr_box_refine(sdetr_stream_t stream, const void *delta, int delta_dtype, int64_t delta_row_stride,
                     const float *reference_points, int64_t num_boxes, int groups, float eps, float *out);

/* sdetr_mlp_rows_bf16 (round 5): the decoder's small MLPs (models/bricks/basic.py:6-26: Linear + ReLU chains) in one
 * launch -- ref_point_head 512 -> 256 -> 256 (models/bricks/salience_transformer.py:643-644) with weight3 = NULL and
 * out_features = 256, bbox_head / encoder_bbox_head 256 -> 256 -> 256 -> out_features <= 32 (:206, :659-668).  Rows
 * [0, rows_first) are read from x, [rows_first, rows) from x_second (the normed and the raw queries of a decoder layer
 * share bbox_head: no stacked copy); rows are contiguous, in_features elements each.  Weights as packed by
 * sdetr_linear_pack_bf16 (section (8): 256 input features per block; a 512-feature first layer is two blocks back to back,
 * columns 0-255 then 256-511, each sdetr_linear_packed_bytes(256) long), biases fp32 zero-padded to the packed tiles
 * (256 floats; the last layer's at least 32).  ReLU after every layer but the last; out [rows, out_features] in the
 * activation type, rows out_row_stride elements apart. */
int sdetr_mlp_rows_bf16(sdetr_stream_t stream, const void *x, const void *x_second, int64_t rows_first, int64_t rows,
                        int in_features, const void *packed_weight1, const float *bias1, const void *packed_weight2,
                        const float *bias2, const void *packed_weight3, const float *bias3, int out_features, void *out,
                        int64_t out_row_stride);

/* sdetr_rows_linear_ln_bf16 (round 5): out = LayerNorm(residual + Linear(x)) for a 256 -> 256 Linear on a few thousand
 * rows -- the tails of the decoder layer's attention blocks (models/bricks/salience_transformer.py:571-572: norm2(query +
 * out_proj(heads)); :583-585: norm1(query + output_proj(sampled))), a library GEMM and the add + LayerNorm launch each
 * before.  (sdetr_token_linear_ln_bf16 of section (8) is the same function for tens of thousands of rows.)  x / residual /
 * out [rows, 256] contiguous in the activation type; packed_weight / bias_padded as in section (8); norm_weight /
 * norm_bias fp32 [256].  The Linear's output is rounded to the activation type before the add, as the separate launches
 * store it. */
int sdetr_rows_linear_ln_bf16(sdetr_stream_t stream, const void *x, const void *residual, int64_t rows,
                              const void *packed_weight, const float *bias_padded, const float *norm_weight,
                              const float *norm_bias, float norm_eps, void *out);

/* sdetr_ref_point_head_bf16 (round 5): query_pos = ref_point_head(get_sine_pos_embed(reference_points_input[:, :, 0, :]))
 * (models/bricks/salience_transformer.py:642-644) in one launch: the sine embedding of sdetr_decoder_query_sine_embed
 * (128 features per coordinate) is made in the row-tile fill of the 512 -> 256 -> 256 chain instead of going through
 * memory.  reference_points fp32 [batch, num_queries, 4], valid_ratios fp32 [batch, num_levels, 2]; weights / biases as
 * for sdetr_mlp_rows_bf16 with in_features = 512; query_pos [batch * num_queries, 256] in the activation type;
 * reference_points_input fp32 [batch, num_queries, num_levels, 4] (:642) or NULL. */
int sdetr_ref_point_head_bf16(sdetr_stream_t stream, const float *reference_points, const float *valid_ratios,
                              int batch_size, int num_queries, int num_levels, float temperature,
                              const void *packed_weight1, const float *bias1, const void *packed_weight2,
                              const float *bias2, void *query_pos, float *reference_points_input);

/* sdetr_rows_linear_bf16 (round 5): out[r, f] = (x[r] + (f < pos_features ? pos[r] : 0)) . W[f] + b[f] for a few
 * thousand rows of 256 features in one launch: the decoder layer's self-attention in-projection (nn.MultiheadAttention
 * with q = k = query + query_pos, v = query, models/bricks/salience_transformer.py:565-570: pos_features = 512 of 768)
 * and the cross-attention's sampling_offsets | attention_weights projection of query + query_pos
 * (models/bricks/ms_deform_attn.py:322-349: all 384) -- each an elementwise add and one or two library GEMMs before.
 * x / pos [rows, 256] contiguous (x + pos is formed in fp32 and rounded once, like the elementwise add); packed_weight /
 * bias_padded as for sdetr_token_linear_bf16 (section (8)); out_features <= 768, pos_features a multiple of 32. */
int sdetr_rows_linear_bf16(sdetr_stream_t stream, const void *x, const void *pos, int64_t rows, int pos_features,
                           const void *packed_weight, const float *bias_padded, int out_features, void *out,
                           int64_t out_row_stride);

/* sdetr_decoder_head_bf16 (round 5): a decoder layer's output head, models/bricks/salience_transformer.py:655-668, in
 * one launch:  normed = LayerNorm(query);  logits = class_head(normed);
 * boxes[0] = sigmoid(bbox_head(normed) + inverse_sigmoid(reference_points, sigmoid_eps))  (the layer's output boxes) and,
 * with two_sources, boxes[1] = the same from the raw query rows (the next layer's reference points, :666-668).
 * query [rows, 256] contiguous; norm_weight / norm_bias fp32 [256]; class head and the three bbox layers as packed by
 * sdetr_linear_pack_bf16 with fp32 zero-padded biases (num_classes <= 224); reference_points fp32 [rows, 4];
 * logits [rows, num_classes] in the activation type; boxes fp32 [two_sources ? 2 : 1][rows][4].  The chain's 4 outputs
 * are rounded to the activation type before the refinement, as the module-by-module path stores them. */
int sdetr_decoder_head_bf16(sdetr_stream_t stream, const void *query, int64_t rows, const float *norm_weight,
                            const float *norm_bias, float norm_eps, const void *packed_class_weight,
                            const float *class_bias_padded, int num_classes, const void *packed_weight1,
                            const float *bias1, const void *packed_weight2, const float *bias2,
                            const void *packed_weight3, const float *bias3, const float *reference_points,
                            float sigmoid_eps, int two_sources, void *logits, int64_t logits_row_stride, float *boxes);

/* ---- (11) two-stage proposal selection after the encoder (row N1) ---------------------------------------------------
 * models/bricks/salience_transformer.py:194-212, 249-295; models/bricks/base_transformer.py:74-112.
 * level_shapes_host: HOST array [num_levels][2] of (h, w); the levels are laid out back to back in the token dimension.
 *   sdetr_encoder_output_proposals: gen_encoder_output_proposals minus its Linear + LayerNorm: keep [batch, S] (1 = not
 *     padding and proposal (cx, cy, w, h) inside (0.01, 0.99)) and proposal_logit [batch, S, 4] = log(p / (1 - p)),
 *     +inf where keep is 0.  Either output may be NULL.
 *   sdetr_grid_nms_topk: nms_on_topk_index.  topk_index [batch, num_topk] = token ids in descending score order
 *     (images index_batch_stride elements apart).  The reference's boxes are the 2x2 cells [x-1, y-1, x+1, y+1] with one
 *     NMS category per (image, level), for which greedy NMS is neighbour suppression: neighbourhood = 4 when
 *     2/6 > iou_threshold (fp32), 8 when also 1/7 > iou_threshold, 0 when neither (nothing is suppressed).  Writes the
 *     first max_keep kept ids per image in score order to out_index [batch, max_keep] (padded with id 0 when fewer
 *     survive) and the number kept (unclamped) to out_count [batch] (device int32).  num_topk <= 65534; 2*S + num_topk bytes must fit 150 KB of LDS.
 *   sdetr_proposal_refine: out [batch, num_select, 4] = sigmoid(delta[b, i] + proposal_logit[b, index[b, i]])
 *     (enc_outputs_coord of the selected tokens, :198-199 + :209); delta f32 | bf16 contiguous. */
int sdetr_encoder_output_proposals(sdetr_stream_t stream, const uint8_t *padding_mask, const int64_t *level_shapes_host,
                                   int num_levels, int batch_size, int spatial_size, uint8_t *keep,
                                   float *proposal_logit);
int sdetr_grid_nms_topk(sdetr_stream_t stream, const int64_t *topk_index, int64_t index_batch_stride,
                        const int64_t *level_shapes_host, int num_levels, int batch_size, int num_topk,
                        int spatial_size, int neighbourhood, int max_keep, int64_t *out_index, int *out_count);
int sdetr_proposal_refine(sdetr_stream_t stream, const void *delta, int delta_dtype, const float *proposal_logit,
                          const int64_t *index, int64_t index_batch_stride, int batch_size, int spatial_size,
                          int num_select, float *out);

/* ---- (12) salience supervision (row N4) -------------------------------------------------------------------------------
 * models/detectors/salience_detr.py:13-116 (SalienceCriterion), models/bricks/losses.py:4-13 (sigmoid_focal_loss).
 *   sdetr_salience_targets: mask_targets [batch, S] (S = sum of the levels' h*w, levels back to back): for every pixel
 *     centre the scale-independent salience confidence of get_mask_single_level (:75-116).  boxes_xyxy [sum m, 4] fp32
 *     ground-truth boxes in input-image pixels, image b owning rows box_offset[b] .. box_offset[b+1] (device int32
 *     [batch+1]).  HOST arrays per level: level_shapes (h, w) int64, level_strides (stride_y, stride_x) fp32,
 *     limit_range (lo, hi) fp32.  noise_scale != 0 mixes in `noise` [batch, S] (the reference's rand_like draws).
 *   sdetr_salience_focal_loss: loss_and_num_pos[0] = sum_i focal(logits_i, target_i) / num_pos, [1] = num_pos =
 *     max(#{target > positive_threshold}, 1) -- SalienceCriterion.forward's loss_salience (:48-58; the "/ S ... * S" of
 *     the reference cancels).  Deterministic two-stage reduction through `workspace`
 *     (sdetr_focal_loss_workspace_bytes).
 *   sdetr_salience_focal_loss_backward: grad_logits = *grad_loss * d loss / d logits (the focal weight keeps its
 *     gradient, as in the reference). */
int sdetr_salience_targets(sdetr_stream_t stream, const float *boxes_xyxy, const int *box_offset, int batch_size,
                           const int64_t *level_shapes_host, const float *level_strides_host,
                           const float *limit_range_host, int num_levels, float noise_scale, const float *noise,
                           float *target);
int64_t sdetr_focal_loss_workspace_bytes(int64_t count);
int sdetr_salience_focal_loss(sdetr_stream_t stream, const float *logits, const float *target, int64_t count, float alpha,
                              float gamma, float positive_threshold, void *workspace, int64_t workspace_bytes,
                              float *loss_and_num_pos);
int sdetr_salience_focal_loss_backward(sdetr_stream_t stream, const float *logits, const float *target, int64_t count,
                                       float alpha, float gamma, const float *loss_and_num_pos, const float *grad_loss,
                                       float *grad_logits);

/* ---- (13) RepVGGPluX neck on the encoder memory (row N3), eval mode ----------------------------------------------------
 * models/necks/repnet.py:12-245, models/bricks/basic.py:29-54, call site models/bricks/salience_transformer.py:185-192.
 * Feature maps are TOKEN-MAJOR (= channels-last): [batch, height * width, channels], f32 | bf16, arithmetic in fp32.  The
 * caller folds every BatchNorm (running statistics) into the convolution before it, and conv1 + alpha * conv2 of a
 * RepVggPluXBlock (:61-64) into one 3x3 kernel.
 *   sdetr_neck_conv3x3: 3x3 convolution, padding 1, stride 1 | 2, `groups` groups of in_per_group -> out_per_group
 *     channels (multiples of 4), + bias (+ SiLU when activation = 1).  x rows are x_row_stride elements apart (the
 *     channels read are the first groups * in_per_group of a row); weight fp32 [groups][3][3][in_per_group]
 *     [out_per_group]; bias fp32 [groups * out_per_group] or NULL; out contiguous
 *     [batch, ((height - 1) / stride + 1) * ((width - 1) / stride + 1), groups * out_per_group].
 *   sdetr_neck_conv3x3_mfma_bf16: the same convolution for bf16 maps on the matrix cores (fp32 accumulate), for
 *     in_per_group % 16 == 0 and out_per_group % 64 == 0; packed_weight = the fp32 weight above re-laid as bf16 MFMA
 *     operand fragments by sdetr_neck_pack_conv3x3_bf16 into sdetr_neck_conv3x3_packed_bytes bytes (0: shape not
 *     supported).  x rows 16-byte aligned, x_row_stride % 8 == 0.
 *   sdetr_neck_combine: out = act(a + nearest_upsample(up) + bias) -- the epilogue of the 1x1 convolutions, whose
 *     GEMMs run on the token rows: `a` [batch, height * width, channels] at the output resolution, `up` (NULL: absent)
 *     [batch, up_height * up_width, channels] read at torch's nearest-neighbour source pixel
 *     min(int(floorf(dst * (float)in / out)), in - 1) (F.interpolate(mode="nearest") at repnet.py:224-228; a 1x1
 *     convolution commutes with it, so the coarse half of the concatenated input is convolved at ITS resolution).
 *     Row strides in elements.
 *   sdetr_neck_gate_shortcut: out = SqueezeAndExcitation(y) + shortcut (+ shortcut2): context = sum_p softmax_p(
 *     mask_weight . y_p) y_p, gate = sigmoid(excite_weight [C, hidden] relu(squeeze_weight [hidden, C] context)),
 *     out = gate * y + shortcut (repnet.py:63-64; shortcut2 = the CSP layer's second branch, :121, for its last
 *     block).  y / out contiguous [batch, pixels, channels], channels <= 256; `gate` fp32 [batch, channels] scratch
 *     that also returns the gate; workspace of sdetr_neck_gate_workspace_bytes. */
int sdetr_neck_conv3x3(sdetr_stream_t stream, const void *x, int dtype, int batch_size, int height, int width,
                       int x_row_stride, const float *weight, const float *bias, int groups, int in_per_group,
                       int out_per_group, int stride, int activation, void *out);
int64_t sdetr_neck_conv3x3_packed_bytes(int groups, int in_per_group, int out_per_group);
int sdetr_neck_pack_conv3x3_bf16(sdetr_stream_t stream, const float *weight, int groups, int in_per_group,
                                 int out_per_group, void *packed);
int sdetr_neck_conv3x3_mfma_bf16(sdetr_stream_t stream, const void *x, int batch_size, int height, int width,
                                 int x_row_stride, const void *packed_weight, const float *bias, int groups,
                                 int in_per_group, int out_per_group, int stride, int activation, void *out);
int sdetr_neck_combine(sdetr_stream_t stream, const void *a, int a_row_stride, const void *up, int up_row_stride,
                       int up_height, int up_width, const float *bias, int dtype, int batch_size, int height, int width,
                       int channels, int activation, void *out, int out_row_stride);
int64_t sdetr_neck_gate_workspace_bytes(int batch_size, int pixels, int channels);
int sdetr_neck_gate_shortcut(sdetr_stream_t stream, const void *y, int dtype, int batch_size, int pixels, int channels,
                             const float *mask_weight, const float *squeeze_weight, const float *excite_weight,
                             int hidden, const void *shortcut, int shortcut_row_stride, const void *shortcut2,
                             int shortcut2_row_stride, void *workspace, int64_t workspace_bytes, float *gate, void *out);

/* ---- fp32 GEMM on the bf16 matrix cores at fp32 accuracy (csrc/gemm_x3.hip) -- the Linear layers of the training
 * step (y = x w^T, dx = dy w, dw = dy^T x without transposed copies); replaces the at::linear / at::mm calls autograd
 * makes for nn.Linear (models/bricks/salience_transformer.py:347-351, ms_deform_attn.py:312-331).
 *   C[M,N] = sum_k A(m,k) B(n,k) (+ bias[n]);  a_kmajor: A(m,k) = a[m * lda + k], else a[k * lda + m]; b likewise.
 *   reduction_splits > 1: slices of the reduction accumulate into C with fp32 atomics (C must be zero on entry).
 *   Every operand 16-byte aligned, leading dimensions multiples of 4; a k-major operand needs K % 4 == 0, the other
 *   kind its row count % 4 == 0.  a_row_sum (optional, [M], zero on entry, reduction-major A only) receives
 *   sum_k A(m,k): the bias gradient sum_t dy[t][n] that comes with dw = dy^T x. */
/* sdetr_gemm_x3_presplit: the three bf16 planes of an fp32 matrix [rows, cols] (rows ld apart), optionally transposed,
 * out = 3 * rows * cols bf16 -- the form of a B operand passed with b_kmajor = 2 (planes [3][N][K], rows ldb
 * elements apart, K % 8 == 0): a weight is split once per call instead of in every workgroup that reads it. */
int sdetr_gemm_x3_presplit(sdetr_stream_t stream, const float *w, int64_t ld, int rows, int cols, int transpose,
                           void *out);
/* sdetr_gemm_x3_generation: pins the kernel generation the CALLING THREAD's later gemm_x3 calls take -- 1 = 128 x 128
 * tiles, 2 = 256 x 128 tiles, 0 = the library's shape rule (the default) -- and returns the previous setting.  Both
 * generations take the same six bf16 products per term; the parity tests run every shape on each. */
int sdetr_gemm_x3_generation(int generation);
int sdetr_gemm_x3_f32(sdetr_stream_t stream, const float *a, int64_t lda, int a_kmajor, const float *b, int64_t ldb,
                      int b_kmajor, float *c, int64_t ldc, int M, int N, int K, const float *bias,
                      int reduction_splits, float *a_row_sum);
/* The same product with an epilogue on C (round 4; unsplit reductions only) -- the feed-forward's ReLU
 * (models/bricks/salience_transformer.py:347-351: linear2(dropout(relu(linear1(x))))) folded into the products around it:
 *   SDETR_GEMM_EPI_RELU  C = relu(acc + bias)                         forward of linear1 + activation
 *   SDETR_GEMM_EPI_GATE  C = gate(m, n) <= 0 ? 0 : acc + bias         dh = (dy w2) * (h > 0): linear2's input gradient
 *                        followed by ReLU's backward; gate = the ReLU output h, [M, N] rows ldg apart
 * NaNs behave as in torch (relu(NaN) = NaN; a NaN gate passes the gradient). */
#define SDETR_GEMM_EPI_NONE 0
#define SDETR_GEMM_EPI_RELU 1
#define SDETR_GEMM_EPI_GATE 2
int sdetr_gemm_x3_epilogue_f32(sdetr_stream_t stream, const float *a, int64_t lda, int a_kmajor, const float *b,
                               int64_t ldb, int b_kmajor, float *c, int64_t ldc, int M, int N, int K, const float *bias,
                               int reduction_splits, float *a_row_sum, int epilogue, const float *gate, int64_t ldg);

/* The END of an encoder layer in one operator (csrc/ffn.hip, ffn_fused_kernel<true>): the deformable attention's tail
 *     x = norm1(residual + output_proj(sampled))              models/bricks/salience_transformer.py:390-391,
 *                                                              models/bricks/ms_deform_attn.py:375
 * runs in front of the feed-forward inside the SAME launch -- Wo streams through the feed-forward's weight ring as
 * four extra chunks, x goes from the accumulators into the B-operand registers of the first feed-forward product and
 * never reaches memory -- followed by sdetr_ffn_fused_advance_bf16's feed-forward, second LayerNorm and row
 * bookkeeping.  sampled / residual: [batch, rows, 256] bf16 contiguous (the MSDA kernel's output, the layer's queries).
 *   packed_tail_ffn: sdetr_attn_tail_packed_bytes() bytes written by sdetr_attn_tail_pack_bf16(output_proj.weight
 *   [256, 256] bf16) immediately followed by the sdetr_ffn_pack_bf16 packing of the two feed-forward weights.
 * All other arguments as sdetr_ffn_fused_advance_bf16 (the workspace has its size). */
int64_t sdetr_attn_tail_packed_bytes(void);
int sdetr_attn_tail_pack_bf16(sdetr_stream_t stream, const void *weight_o, int embed_dim, void *packed);
int sdetr_attn_tail_ffn_advance_bf16(
    sdetr_stream_t stream, const void *sampled, const void *residual, const void *packed_tail_ffn, const float *bias_o,
    const float *norm1_weight, const float *norm1_bias, float norm1_eps, const float *bias1, const float *bias2,
    const float *norm_weight, const float *norm_bias, float norm_eps, int batch_size, int rows, int embed_dim, int hidden,
    int hidden_splits, void *workspace, int64_t workspace_bytes, void *sorted_result, void *next_query, const void *tokens,
    const int64_t *sorted_index, int64_t index_batch_stride, const int64_t *count, int sorted_rows, int next_rows,
    int spatial_size, const float *class_bias_padded, const float *foreground, int64_t foreground_batch_stride,
    float *next_class_score);
/* next_class_score (may be NULL) [batch, next_rows] fp32: with ONE hidden piece and next_rows > 0 the same launch also
 * does the row bookkeeping in its epilogue and writes the NEXT layer's selection score of the rows it hands on,
 *     max_c(class_head(next_query[b, i])) * foreground[b, i]        models/bricks/salience_transformer.py:462, 366
 * (fp32 accumulators, no logits in memory).  packed_tail_ffn then ends with sdetr_class_head_packed_bytes() bytes of
 * sdetr_class_head_pack_bf16(class head weight [num_classes <= 96, 256] bf16); class_bias_padded fp32 [96] (-inf on the
 * padded classes); foreground fp32 rows of at least next_rows, images foreground_batch_stride apart.  With more hidden
 * pieces the second pass (partial sums + LayerNorm + row bookkeeping) computes the same score from the rows it has just
 * finished (ffn_reduce_ln_advance_cls_kernel: 32 rows per block through an LDS tile, one 32-class tile per wave). */
int64_t sdetr_class_head_packed_bytes(void);
int sdetr_class_head_pack_bf16(sdetr_stream_t stream, const void *weight, int num_classes, int embed_dim, void *packed);

/* sdetr_topk_attention_bf16 of an encoder layer together with the deformable attention's offset | weight projection of
 * the layer's queries (sdetr_token_linear_bf16 with x_add = pos and group_features = 48: the head-major slab
 * [batch, 8, num_rows, 48] the MSDA kernel reads) -- csrc/fused_head_value.hip: in-projection launch, then ONE launch
 * for the attention (40 workgroups, which also re-project their 300 updated rows) and the projection of all other rows
 * (from the queries as they stand before the attention).
 *   proj_weight: [384, 256] bf16, rows in head-major order (48 per head); proj_packed / proj_bias_padded: its
 *   sdetr_linear_pack_bf16 packing and zero-padded fp32 bias; hint: int32 [batch, hint_batch_stride >= num_rows]
 *   of scratch, contents irrelevant on entry (the in-projection marks the selected rows in it; a mark only counts
 *   if `selected` confirms it, so garbage is harmless) -- one per call in flight.
 *   289 <= num_selected <= 320 only; the layer's queries contiguous [batch, num_rows, 256]. */
int sdetr_topk_attention_with_projection_bf16(
    sdetr_stream_t stream, void *query, int64_t query_batch_stride, const void *pos, int64_t pos_batch_stride,
    const int64_t *selected, int batch_size, int num_rows, int num_selected, const void *in_proj_weight,
    const void *in_proj_bias, const void *out_proj_weight, const void *out_proj_bias, const void *norm_weight,
    const void *norm_bias, float norm_eps, void *workspace, int64_t workspace_bytes, const void *proj_weight,
    const void *proj_packed, const float *proj_bias_padded, void *slab, int32_t *hint, int64_t hint_batch_stride,
    const void *out_proj_frag, const void *proj_frag, int in_projection_done);
/* in_projection_done != 0: `workspace` and `hint` were filled by sdetr_topk_select_inproj_bf16 (below) for this
 * `selected`; the in-projection launch is skipped. */
/* `out_proj_frag` / `proj_frag` (optional, round 6): out_proj_weight / proj_weight once more in the fragment order of the
 * attention workgroups' 16x16x32 products -- element [h][c][j][lane][e] = weight[R h + 16 c + (lane & 15)][32 j + 8 (lane >> 4) + e]
 * with R = 32, c < 2 (out_proj) or R = 48, c < 3 (projection), j < 8, lane < 64, e < 8: a wave's fragment is then one
 * contiguous KB.  NULL: the row-major weights are read as before. */
/* The encoder layer's top-k selection of its rows by class score (models/bricks/salience_transformer.py:366:
 * torch.topk(mc_score, topk_sa, dim=1)[1]; ties: lower position first) TOGETHER with the in-projection of the selected
 * rows (the first launch of sdetr_topk_attention_*: gather + position add + nn.MultiheadAttention's in_proj) in ONE launch
 * (csrc/topk.hip, topk_hsort_inproj_kernel): every workgroup of an image runs the same one-workgroup histogram sort,
 * keeps the list in LDS and its sixteen waves take one 32-row x 32-feature in-projection tile each.
 *   score fp32 [batch, n] contiguous (no mask); out_index int64 [batch, k] (the positions, descending score);
 *   query / pos [batch, >= n, 256] 16-bit activations, images *_batch_stride elements apart; workspace of
 *   sdetr_topk_attention_workspace_bytes(batch, k) bytes and hint as sdetr_topk_attention_with_projection_bf16 takes
 *   them (call it with in_projection_done = 1 next); hint may be NULL.
 *   Shapes: 1024 <= n <= 17 408, 5 k <= 2 n, k <= 384 (longer rows: below).  job (may be NULL): as
 *   sdetr_masked_topk_desc_with_orders_f32. */
int sdetr_topk_select_inproj_bf16(sdetr_stream_t stream, const float *score, int batch_size, int n, int k,
                                  int64_t *out_index, const void *query, int64_t query_batch_stride, const void *pos,
                                  int64_t pos_batch_stride, const void *in_proj_weight, const void *in_proj_bias,
                                  void *workspace, int64_t workspace_bytes, int32_t *hint, int64_t hint_batch_stride,
                                  const sdetr_row_orders_job *job, void *candidate_workspace, int64_t candidate_bytes);
/* Rows of more than 17 408 scores (the reference's 5scale pyramid: up to 45 330 rows per layer) take one launch more: every
 * slice of a row (<= 8192 keys, balanced) keeps its sorted top-k with the positions in the row, then the launch above selects
 * among the slices x k candidates (ties still by position).  candidate_workspace: sdetr_topk_select_candidate_bytes(batch,
 * n, k) bytes (0 = the row needs none, or the sliced form does not cover it: n / slices >= 1024 and >= 2.5 k per slice). */
int64_t sdetr_topk_select_candidate_bytes(int batch_size, int n, int k);
/* (internal: the in-projection launch of sdetr_topk_attention_bf16 for csrc/fused_head_value.hip) */
int sdetr_topk_inproj_launch(sdetr_stream_t stream, const void *tk_in_args);

/* ---------------------------------------------------------------------------------------------
 * LayerNorm of the TRAINING step, fp32 (models/bricks/salience_transformer.py:347-351, 377-378, 390-391:
 * norm(query + sublayer(query)); 64 / 128 / 256 / 512 channels, contiguous [rows, channels] operands).
 *   forward : out = LN(a [+ residual]) * gamma + beta; sum_out = a + residual (required iff residual != NULL: the
 *             tensor the backward normalises again); mean / rstd [rows] = the row statistics.
 *   backward: grad_input = d loss / d (a + residual) (the gradient of a and of the residual alike) from grad_out,
 *             normalized_input (= sum_out, or a without a residual), the saved statistics and gamma;
 *             grad_gamma / grad_beta [channels] are ACCUMULATED with fp32 atomics: zero (or a running sum) on entry.
 * --------------------------------------------------------------------------------------------- */
int sdetr_layer_norm_train_supported(int channels);
int sdetr_layer_norm_train_forward_f32(sdetr_stream_t stream, const float *a, const float *residual, const float *gamma,
                                       const float *beta, float eps, int64_t rows, int channels, float *sum_out,
                                       float *out, float *mean, float *rstd);
int sdetr_layer_norm_train_backward_f32(sdetr_stream_t stream, const float *grad_out, const float *normalized_input,
                                        const float *mean, const float *rstd, const float *gamma, int64_t rows,
                                        int channels, float *grad_input, float *grad_gamma, float *grad_beta);

/* ---------------------------------------------------------------------------------------------
 * Sampling locations + attention weights of the deformable attention for the TRAINING step, fp32, 4 levels x 4 points
 * (models/bricks/ms_deform_attn.py:322-349): offsets [rows, M, L, P, 2] and logits [rows, M, L*P] (the two Linear
 * outputs, rows = B * Nq), reference_points [rows, L, 2 | 4], spatial_shapes [L, 2] int64 (h, w) on the device ->
 * sampling_locations [rows, M, L, P, 2], attention_weights [rows, M, L, P] (softmax over L*P).  backward: the gradients
 * of the two Linear outputs from the op's grad_sampling_locations / grad_attention_weights and the saved weights.
 * --------------------------------------------------------------------------------------------- */
int sdetr_sampling_prep_supported(int num_levels, int num_points);
int sdetr_sampling_prep_f32(sdetr_stream_t stream, const float *offsets, const float *logits, const float *reference_points,
                            const int64_t *spatial_shapes, int64_t rows, int num_heads, int num_levels, int num_points,
                            int ref_dim, float *sampling_locations, float *attention_weights);
int sdetr_sampling_prep_backward_f32(sdetr_stream_t stream, const float *grad_sampling_locations,
                                     const float *grad_attention_weights, const float *attention_weights,
                                     const float *reference_points, const int64_t *spatial_shapes, int64_t rows,
                                     int num_heads, int num_levels, int num_points, int ref_dim, float *grad_offsets,
                                     float *grad_logits);

/* ---------------------------------------------------------------------------------------------
 * Dense multi-head attention over a few hundred rows for the TRAINING step, fp32, 32-channel heads
 * (models/bricks/salience_transformer.py:371-376: the encoder layer's self-attention over its top-300 rows, between the
 * in- and the out-projection of nn.MultiheadAttention).  q / k / v: element (b, n, h, c) at base + b * batch_stride +
 * n * row_stride + h * 32 + c (floats; strides multiples of 4) -- e.g. the q and k halves of one [B, N, 512] projection.
 * out [B, N, H * 32] = softmax(q k^T * scale) v with the heads concatenated; lse [B, H, N] = the rows' log-sum-exp
 * (saved for the backward).  backward: grad_q / grad_k / grad_v in the layouts of q / k / v (every element written).
 * At most sdetr_attention_train_max_rows() rows.
 * --------------------------------------------------------------------------------------------- */
int sdetr_attention_train_max_rows(void);
int sdetr_attention_train_forward_f32(sdetr_stream_t stream, const float *q, int64_t q_batch_stride, int64_t q_row_stride,
                                      const float *k, int64_t k_batch_stride, int64_t k_row_stride, const float *v,
                                      int64_t v_batch_stride, int64_t v_row_stride, int batch_size, int num_heads,
                                      int num_rows, int head_dim, float scale, float *out, float *lse);
int sdetr_attention_train_backward_f32(sdetr_stream_t stream, const float *q, int64_t q_batch_stride, int64_t q_row_stride,
                                       const float *k, int64_t k_batch_stride, int64_t k_row_stride, const float *v,
                                       int64_t v_batch_stride, int64_t v_row_stride, int batch_size, int num_heads,
                                       int num_rows, int head_dim, float scale, const float *out, const float *lse,
                                       const float *grad_out, float *grad_q, float *grad_k, float *grad_v);

#ifdef __cplusplus
}
#endif
#endif /* SALIENCE_HIP_H_ */
