// Launches that carry a second (and third) kernel body: independent work on the CUs a small launch leaves idle.
//
// The hot path is one chain of dependent launches, and many of them are small: the filtering stage walks the levels
// coarse to fine (each level's scores modulate the next finer one), so stage 1 of the salience head has 10 workgroups
// on level 3 and 34 on level 2 of the 800 x 1333 pyramid, each ~20 us of dependent phases; the top-300 attention of
// every encoder layer has 40.  Next to them sits work that does not depend on them: the value projection of the six
// layers (175 workgroups x ~60 us; needs only the flattened tokens), the deformable attention's offset | weight
// projection of the rows the attention does not touch (107 x ~17 us), the top-k INDICES of a level (needed at the
// merge, not by the next level).  A second graph branch costs more than it hides on this stack (a fork / join pair
// ~80 us under hipGraph replay) and one queue cannot express a partial order, so the kernel BODIES
// (device functions in *_core.h) share launches: block ranges take roles, waves a role does not use exit (the hardware
// barrier accounts for them), LDS is the larger of the roles' needs.
//   fused_stage1_value_kernel   stage 1 (level 3 / 2) | value projection of 2 layers | rank of the level before
//   fused_stage1_rank_kernel    stage 1 (level 1 / 0) | rank of the level before | token-space pass of the output
//   fused_stage2_value_kernel   stage 2 (level 3 / 2) | value projection of 1 layer
//   fused_attn_proj_kernel      top-300 attention (+ projection of its own updated rows) | projection of all other rows
// Measured: 1.391 -> 1.29 ms per step, 93 -> 76 launches.
#include "salience_head_core.h"
#include "token_linear_core.h"
#include "topk_attention_core.h"
#include "topk_core.h"
#include "finalize_core.h"

namespace sdetr {

// Roles by block index: [0, n1) stage 1 of the head; then `tl_blocks` workgroups of the value projection; then the
// rank-by-counting top-k of the NEXT COARSER level (its indices are not needed before the merge at the end of the
// filtering, its scores are: the level's stage 2 has run) -- `rk_blocks_x` key blocks x `rk_rows` rows.
__global__ void __launch_bounds__(768, 1) fused_stage1_value_kernel(Stage1Args s1, int s1_blocks, int s1_images, TLArgs tl,
                                                                    int tl_blocks, RankArgs rk, int rk_blocks_x)
{
    extern __shared__ __attribute__((aligned(16))) char fused_lds[];
    const int blk = (int)blockIdx.x;
    const int n1 = s1_blocks * s1_images;   // stage-1 workgroups: (image, token block) folded into blockIdx.x
    if (blk < n1) {
        if (threadIdx.x >= 512) return;
        stage1_x3_body(s1, blk % s1_blocks, blk / s1_blocks);
    } else if (blk < n1 + tl_blocks) {
        token_linear_body<kHeadMajor, false, 8>(tl, blk - n1);
    } else {
        if (threadIdx.x >= kRankThreads) return;
        const int r = blk - n1 - tl_blocks;
        uint32_t *lds = reinterpret_cast<uint32_t *>(fused_lds);
        topk_rank_body(rk, r % rk_blocks_x, r / rk_blocks_x, lds, lds + kRankTile);
    }
}

// Stage 1 with a rank job and / or the finalize pass: the head kernel's own shape (512 threads, two workgroups per CU).
__global__ void __launch_bounds__(512, 2) fused_stage1_rank_kernel(Stage1Args s1, int s1_blocks, int s1_images, RankArgs rk,
                                                                   int rk_blocks_x, int rk_blocks, FinalizeJob fin,
                                                                   int fin_blocks)
{
    extern __shared__ __attribute__((aligned(16))) char fused_lds[];
    const int blk = (int)blockIdx.x;
    const int n1 = s1_blocks * s1_images;
    if (blk < n1) {
        stage1_x3_body(s1, blk % s1_blocks, blk / s1_blocks);
    } else if (blk < n1 + rk_blocks) {
        const int r = blk - n1;
        uint32_t *lds = reinterpret_cast<uint32_t *>(fused_lds);
        topk_rank_body(rk, r % rk_blocks_x, r / rk_blocks_x, lds, lds + kRankTile);
    } else {
        finalize_all_role(fin, blk - n1 - rk_blocks, fin_blocks);
    }
}

// What the hoisted head leaves in the coarse-to-fine chain (modulate_body: the first 256 threads) with the jobs stage 1 used
// to carry: the rank of the level before, the token-space pass of the encoder's output.
// `rk_tile`: words of LDS in front of the rank body's partial counts -- the whole row of keys when it is shorter than a
// full tile (the launch's LDS is what every workgroup reserves, the modulation blocks too: with the full 48 KB tile only
// three of them fit a CU and the finest level's launch took 20 us instead of 9).
__global__ void __launch_bounds__(512, 2) fused_modulate_rank_kernel(ModulateArgs m, int m_blocks, int m_images, RankArgs rk,
                                                                     int rk_blocks_x, int rk_blocks, int rk_tile,
                                                                     FinalizeJob fin, int fin_blocks)
{
    extern __shared__ __attribute__((aligned(16))) char fused_lds[];
    // The jobs' workgroups FIRST: each is a longer chain than a modulation block (a rank workgroup scans the whole row of
    // keys), behind the step's own blocks they would start when those are done -- 22.6 us for the finest level's launch
    // instead of the longer of the two.
    const int blk = (int)blockIdx.x;
    const int n1 = m_blocks * m_images;
    if (blk < rk_blocks) {
        uint32_t *lds = reinterpret_cast<uint32_t *>(fused_lds);
        topk_rank_body(rk, blk % rk_blocks_x, blk / rk_blocks_x, lds, lds + rk_tile);
    } else if (blk < rk_blocks + fin_blocks) {
        finalize_all_role(fin, blk - rk_blocks, fin_blocks);
    } else {
        // blockDim.x / 256 modulation blocks per workgroup (an odd last one: its waves leave, the hardware barrier counts
        // the waves that are left)
        const int per = (int)blockDim.x / kModThreads, half = (int)threadIdx.x / kModThreads;
        const int e = (blk - rk_blocks - fin_blocks) * per + half;
        if (e >= n1) return;
        modulate_body(m, e % m_blocks, e / m_blocks, reinterpret_cast<float *>(fused_lds) + half * kModLdsFloats,
                      (int)threadIdx.x - half * kModThreads);
    }
}

// The same for stage 2 (64-token blocks, first 256 threads; its tile lives behind the projection's LDS buffers).
__global__ void __launch_bounds__(768, 1) fused_stage2_value_kernel(Stage2Args s2, int s2_blocks, int s2_images, TLArgs tl)
{
    extern __shared__ __attribute__((aligned(16))) char fused_lds[];
    const int blk = (int)blockIdx.x;
    const int n1 = s2_blocks * s2_images;
    if (blk < n1) {
        if (threadIdx.x >= kBlock) return;
        float *zt = reinterpret_cast<float *>(fused_lds);
        stage2_body(s2, blk % s2_blocks, blk / s2_blocks, zt, zt + kStage2TileFloats);
    } else {
        token_linear_body<kHeadMajor, false, 8>(tl, blk - n1);
    }
}

// The top-300 attention (40 workgroups of ~15 us) carries the layer's offset | weight projection of ALL rows
// (token_linear<kStore, x + pos>, ~107 workgroups x ~17 us): the projection reads the queries as they are before the
// attention updates 300 of them and skips those rows (the in-projection launch in front marked them in a hint array
// that the projection validates against the selection, so the array needs no initialisation); the attention
// kernel projects its 300 rows itself from the updated values.  One writer per slab row, no ordering needed.
template <int KT>
__global__ void __launch_bounds__(512, 1) fused_attn_proj_kernel(TkOutArgs at, int at_blocks, TLArgs tl)
{
    const int blk = (int)blockIdx.x;
#if defined(FH_KO_ATTN)      // (benchmark builds: which of the launch's two bodies sets its length)
    if (blk < at_blocks) return;
#elif defined(FH_KO_PROJ)
    if (blk >= at_blocks) return;
#endif
    if (blk < at_blocks) topk_attn_out_body<KT>(at, blk);
    else token_linear_body<kStore, true, 4>(tl, blk - at_blocks);
}

}  // namespace sdetr

using namespace sdetr;

extern "C" int sdetr_topk_inproj_launch(sdetr_stream_t stream, const void *tk_in_args);   // topk_attention.hip

static int fill_value_job(TLArgs &t, size_t &lds_tl, int &n2, const void *vp_x, const void *vp_packed_weight,
                          const float *vp_bias_padded, const uint8_t *vp_pad_mask, int vp_batch_size, int vp_spatial_size,
                          int vp_num_heads, int vp_num_groups, void *vp_dst, int vp_dst_dtype,
                          const sdetr_bordered_layout *vp_bordered)
{
    if (vp_batch_size <= 0 || vp_spatial_size <= 0 || vp_num_heads <= 0 || vp_num_groups <= 0)
        return fail("value-projection job: bad sizes");
    if (vp_dst_dtype != SDETR_F16 && vp_dst_dtype != SDETR_BF16) return fail("value-projection job: dst must be fp16 or bf16");
    if (!vp_x || !vp_packed_weight || !vp_bias_padded || !vp_dst) return fail("value-projection job: NULL pointer");
    const int64_t vp_tokens = (int64_t)vp_batch_size * vp_spatial_size;
    if (vp_tokens > 0x7fffffff) return fail("value-projection job: too many tokens");
    t = TLArgs{};
    t.x = (const bf16_t *)vp_x; t.pw = (const char *)vp_packed_weight; t.bias = vp_bias_padded; t.T = (int)vp_tokens;
    t.N = vp_num_groups * vp_num_heads * 32; t.ntiles = t.N / 32; t.rows_per_batch = vp_spatial_size;
    t.pad = vp_pad_mask; t.hm = vp_dst; t.heads = vp_num_heads; t.batch = vp_batch_size; t.hm_f16 = vp_dst_dtype == SDETR_F16;
    const int nsteps = (t.ntiles + kTLStepTiles - 1) / kTLStepTiles;
    if ((size_t)t.ntiles * 128 + 2 * kTLStepBytes + 1024 > 160 * 1024) return fail("value-projection job: too many output features");
    lds_tl = 2 * (size_t)kTLStepBytes + (size_t)nsteps * 512;
    n2 = (int)((vp_tokens + kTLTokWave * 8 - 1) / (kTLTokWave * 8));
    return tl_set_bordered(t, vp_bordered, vp_spatial_size, n2);
}

// The launch of a filled-in Stage1Args with the jobs it carries (sdetr_stage1_x3_with_jobs, sdetr_salience_head_hoist_x3).
static int stage1_launch_with_jobs(sdetr_stream_t stream, Stage1Args &a, int batch_size, const void *vp_x,
                                   const void *vp_packed_weight, const float *vp_bias_padded, const uint8_t *vp_pad_mask,
                                   int vp_batch_size, int vp_spatial_size, int vp_num_heads, int vp_num_groups, void *vp_dst,
                                   int vp_dst_dtype, const sdetr_bordered_layout *vp_bordered, const sdetr_rank_job *rank,
                                   const sdetr_finalize_job *finalize)
{
    const int n1 = a.nblk * batch_size;
    const size_t lds_s1 = (size_t)kX3Region + (kParRows * kC + 32 + 4) * sizeof(float);
    // the rank job
    RankArgs r{};
    int rk_bx = 1, n3 = 0;
    size_t lds_rk = 0;
    if (rank) {
        if (rank->batch <= 0 || rank->n <= 0 || rank->k <= 0 || rank->k > rank->n || !rank->score || !rank->out_index)
            return fail("stage1_x3_with_jobs: bad rank job");
        if (rank->n >= (1 << 30) || rank->batch > 65535) return fail("stage1_x3_with_jobs: rank row too long / too many rows");
        const int64_t ors = rank->out_row_stride ? rank->out_row_stride : rank->k;
        const int64_t mrs = rank->mask && !rank->mask_row_stride ? rank->n : rank->mask_row_stride;
        if (ors < rank->k || (rank->mask && mrs < rank->n)) return fail("stage1_x3_with_jobs: rank job strides too small");
        if (rank->mask && !rank->fill_value) return fail("stage1_x3_with_jobs: a masked rank job needs its fill value");
        r.score = rank->score; r.mask = rank->mask; r.mask_stride = mrs; r.fill = rank->fill_value; r.N = rank->n;
        r.k = rank->k; r.index_offset = rank->index_offset; r.out_score = rank->out_score; r.out_index = rank->out_index;
        r.out_stride = ors;
        rk_bx = (rank->n + 63) / 64;
        n3 = rk_bx * rank->batch;
        lds_rk = (size_t)kRankLdsWords * 4;
    }
    hipStream_t hs = static_cast<hipStream_t>(stream);
    FinalizeJob fj{};
    int n4 = 0;
    if (finalize) {
        if (vp_x) return fail("stage1_x3_with_jobs: the finalize pass rides with launches that carry no value projection");
        if (int rc = fill_finalize_job(fj, finalize, "stage1_x3_with_jobs")) return rc;
        n4 = 128;   // 65 536 threads in a grid-stride loop over the 16-byte pieces
    }
    if (!vp_x) {
        const size_t lds = lds_s1 > lds_rk ? lds_s1 : lds_rk;
        hipLaunchKernelGGL(fused_stage1_rank_kernel, dim3((unsigned)(n1 + n3 + n4)), dim3(512), lds, hs, a, a.nblk, batch_size, r,
                           rk_bx, n3, fj, n4);
        return check_launch("stage1_x3_with_jobs");
    }
    TLArgs t;
    size_t lds_tl = 0;
    int n2 = 0;
    if (int rc = fill_value_job(t, lds_tl, n2, vp_x, vp_packed_weight, vp_bias_padded, vp_pad_mask, vp_batch_size,
                                vp_spatial_size, vp_num_heads, vp_num_groups, vp_dst, vp_dst_dtype, vp_bordered))
        return rc;
    size_t lds = lds_tl > lds_s1 ? lds_tl : lds_s1;
    if (lds_rk > lds) lds = lds_rk;
    static DeviceOnce once;
    allow_dynamic_lds(fused_stage1_value_kernel, once, 160 * 1024);
    hipLaunchKernelGGL(fused_stage1_value_kernel, dim3((unsigned)(n1 + n2 + n3)), dim3(768), lds, hs, a, a.nblk, batch_size, t,
                       n2, r, rk_bx);
    return check_launch("stage1_x3_with_jobs");
}

// sdetr_salience_head_stage1_x3 (same arguments) + up to two jobs carried by the same launch: the value projection of
// `vp_num_groups` stacked layers (arguments of sdetr_value_proj_head_major; `vp_packed_weight`, `vp_bias_padded` and
// `vp_dst` already point at the first of those layers; vp_x == NULL: none) and a plain masked top-k by rank counting
// (`rank`: the arguments of sdetr_masked_topk_desc_f32 with fill_mode 2 and no payload / prefilter; NULL: none).
extern "C" int sdetr_stage1_x3_with_jobs(
    sdetr_stream_t stream, const float *x, int64_t x_batch_stride, int64_t x_row_stride, int batch_size, int tokens,
    int channels, const void *enc_weight_x3, const float *enc_bias, const float *enc_norm_weight,
    const float *enc_norm_bias, float enc_norm_eps, const float *row_scale, const float *coarse_score, int coarse_h,
    int coarse_w, int level_h, int level_w, const float *alpha, const float *norm_weight, const float *norm_bias,
    float norm_eps, const void *weight_x3, const float *bias, float *memory_out, int64_t memory_batch_stride,
    float *z_local, float *partial_sums,
    const void *vp_x, const void *vp_packed_weight, const float *vp_bias_padded, const uint8_t *vp_pad_mask,
    int vp_batch_size, int vp_spatial_size, int vp_num_heads, int vp_num_groups, void *vp_dst, int vp_dst_dtype,
    const sdetr_bordered_layout *vp_bordered, const sdetr_rank_job *rank, const sdetr_finalize_job *finalize)
{
    if (channels != kC) return fail("stage1_x3_with_jobs: built for embed_dim = hidden_dim = %d (got %d)", kC, channels);
    if (batch_size <= 0 || tokens <= 0) return fail("stage1_x3_with_jobs: empty level");
    if (!x || !norm_weight || !norm_bias || !weight_x3 || !bias || !z_local || !partial_sums)
        return fail("stage1_x3_with_jobs: NULL pointer");
    if (enc_weight_x3 && (!enc_bias || !enc_norm_weight || !enc_norm_bias))
        return fail("stage1_x3_with_jobs: enc_output parameters incomplete");
    if (row_scale && coarse_score) return fail("stage1_x3_with_jobs: give row_scale OR coarse_score");
    if (coarse_score && ((int64_t)level_h * level_w != tokens || coarse_h <= 0 || coarse_w <= 0))
        return fail("stage1_x3_with_jobs: level %dx%d does not cover %d tokens", level_h, level_w, tokens);
    if ((x_row_stride % 4) || (x_batch_stride % 4)) return fail("stage1_x3_with_jobs: rows must be 16-byte aligned");
    Stage1Args a;
    a.x = x; a.x_batch_stride = x_batch_stride; a.x_row_stride = x_row_stride;
    a.w_enc = reinterpret_cast<const float4 *>(enc_weight_x3);
    a.b_enc = enc_bias; a.g_enc = enc_norm_weight; a.beta_enc = enc_norm_bias; a.eps_enc = enc_norm_eps;
    a.row_scale = row_scale; a.coarse = coarse_score; a.ch = coarse_h; a.cw = coarse_w; a.h = level_h; a.w = level_w;
    a.alpha = alpha; a.g1 = norm_weight; a.beta1 = norm_bias; a.eps1 = norm_eps;
    a.w1 = reinterpret_cast<const float4 *>(weight_x3); a.b1 = bias;
    a.memory_out = enc_weight_x3 ? memory_out : nullptr; a.mem_batch_stride = memory_batch_stride;
    a.z_local = z_local; a.partial = partial_sums; a.n = tokens; a.nblk = (tokens + 31) / 32;
    return stage1_launch_with_jobs(stream, a, batch_size, vp_x, vp_packed_weight, vp_bias_padded, vp_pad_mask, vp_batch_size,
                                   vp_spatial_size, vp_num_heads, vp_num_groups, vp_dst, vp_dst_dtype, vp_bordered, rank,
                                   finalize);
}

// The hoisted stage 1 (Stage1Args, hoisted form) for ALL levels' tokens: x [batch, tokens, 256] -> g_out [batch, tokens, 256]
// = layer1.Linear's weight times the unit-variance row (no bias) and sigma_out [batch, tokens] = the row's standard
// deviation, after enc_output + enc_output_norm when their parameters are given (memory_out optional, as in stage 1).
// Carries a value-projection job and / or the finalize pass like sdetr_stage1_x3_with_jobs (vp_x / finalize NULL: none).
extern "C" int sdetr_salience_head_hoist_x3(
    sdetr_stream_t stream, const float *x, int64_t x_batch_stride, int64_t x_row_stride, int batch_size, int tokens,
    int channels, const void *enc_weight_x3, const float *enc_bias, const float *enc_norm_weight,
    const float *enc_norm_bias, float enc_norm_eps, const float *norm_weight, const void *weight_x3, float *memory_out,
    int64_t memory_batch_stride, float *g_out, int64_t g_batch_stride, float *sigma_out, int64_t sigma_batch_stride,
    const void *vp_x, const void *vp_packed_weight, const float *vp_bias_padded, const uint8_t *vp_pad_mask,
    int vp_batch_size, int vp_spatial_size, int vp_num_heads, int vp_num_groups, void *vp_dst, int vp_dst_dtype,
    const sdetr_bordered_layout *vp_bordered, const sdetr_finalize_job *finalize)
{
    if (channels != kC) return fail("salience_head_hoist_x3: built for embed_dim = hidden_dim = %d (got %d)", kC, channels);
    if (batch_size <= 0 || tokens <= 0) return fail("salience_head_hoist_x3: empty input");
    if (!x || !norm_weight || !weight_x3 || !g_out || !sigma_out) return fail("salience_head_hoist_x3: NULL pointer");
    if (enc_weight_x3 && (!enc_bias || !enc_norm_weight || !enc_norm_bias))
        return fail("salience_head_hoist_x3: enc_output parameters incomplete");
    if ((x_row_stride % 4) || (x_batch_stride % 4) || (g_batch_stride % 4))
        return fail("salience_head_hoist_x3: rows must be 16-byte aligned");
    if (g_batch_stride < (int64_t)tokens * kC || sigma_batch_stride < tokens)
        return fail("salience_head_hoist_x3: output batch strides too small");
    Stage1Args a;
    a.x = x; a.x_batch_stride = x_batch_stride; a.x_row_stride = x_row_stride;
    a.w_enc = reinterpret_cast<const float4 *>(enc_weight_x3);
    a.b_enc = enc_bias; a.g_enc = enc_norm_weight; a.beta_enc = enc_norm_bias; a.eps_enc = enc_norm_eps;
    a.row_scale = nullptr; a.coarse = nullptr; a.ch = a.cw = a.h = a.w = 0;
    a.alpha = nullptr; a.g1 = norm_weight; a.beta1 = norm_weight; a.eps1 = 0.f;   // (beta1 / b1: staged, not used)
    a.w1 = reinterpret_cast<const float4 *>(weight_x3); a.b1 = norm_weight;
    a.memory_out = enc_weight_x3 ? memory_out : nullptr; a.mem_batch_stride = memory_batch_stride;
    a.z_local = nullptr; a.partial = nullptr; a.n = tokens; a.nblk = (tokens + 31) / 32;
    a.g_out = g_out; a.g_batch_stride = g_batch_stride; a.sigma_out = sigma_out; a.sigma_batch_stride = sigma_batch_stride;
    return stage1_launch_with_jobs(stream, a, batch_size, vp_x, vp_packed_weight, vp_bias_padded, vp_pad_mask, vp_batch_size,
                                   vp_spatial_size, vp_num_heads, vp_num_groups, vp_dst, vp_dst_dtype, vp_bordered, nullptr,
                                   finalize);
}

// One level's step of the hoisted head (modulate_body): g / sigma are the level's rows of sdetr_salience_head_hoist_x3's
// outputs, c0 [256] = layer1.Linear.weight @ layer1.LayerNorm.bias + layer1.Linear.bias; z_local / partial_sums as stage 1
// leaves them (sdetr_salience_head_const / _stage2 follow).  `rank` / `finalize`: jobs carried as by
// sdetr_stage1_x3_with_jobs.
extern "C" int sdetr_salience_head_modulate(sdetr_stream_t stream, const float *g, int64_t g_batch_stride, const float *sigma,
                                            int64_t sigma_batch_stride, int batch_size, int tokens, const float *row_scale,
                                            const float *coarse_score, int coarse_h, int coarse_w, int level_h, int level_w,
                                            const float *alpha, float norm_eps, const float *c0, float *z_local,
                                            float *partial_sums, const sdetr_rank_job *rank,
                                            const sdetr_finalize_job *finalize, float *score_min_init)
{
    if (batch_size <= 0 || tokens <= 0) return fail("salience_head_modulate: empty level");
    if (!g || !sigma || !c0 || !z_local || !partial_sums) return fail("salience_head_modulate: NULL pointer");
    if (row_scale && coarse_score) return fail("salience_head_modulate: give row_scale OR coarse_score");
    if (coarse_score && ((int64_t)level_h * level_w != tokens || coarse_h <= 0 || coarse_w <= 0))
        return fail("salience_head_modulate: level %dx%d does not cover %d tokens", level_h, level_w, tokens);
    if ((g_batch_stride % 4) || (reinterpret_cast<uintptr_t>(g) & 15)) return fail("salience_head_modulate: rows must be 16-byte aligned");
    ModulateArgs m;
    m.g = g; m.g_batch_stride = g_batch_stride; m.sigma = sigma; m.sigma_batch_stride = sigma_batch_stride;
    m.row_scale = row_scale; m.coarse = coarse_score; m.ch = coarse_h; m.cw = coarse_w; m.h = level_h; m.w = level_w;
    m.alpha = alpha; m.eps1 = norm_eps; m.c0 = c0; m.z_local = z_local; m.partial = partial_sums; m.n = tokens;
    m.nblk = (tokens + 31) / 32; m.score_min_init = score_min_init;
    const int n1 = m.nblk * batch_size;
    RankArgs r{};
    int rk_bx = 1, n3 = 0, rk_tile = kRankTile;
    size_t lds = 2 * (size_t)kModLdsFloats * sizeof(float);
    if (rank) {
        if (rank->batch <= 0 || rank->n <= 0 || rank->k <= 0 || rank->k > rank->n || !rank->score || !rank->out_index)
            return fail("salience_head_modulate: bad rank job");
        if (rank->n >= (1 << 30) || rank->batch > 65535) return fail("salience_head_modulate: rank row too long / too many rows");
        const int64_t ors = rank->out_row_stride ? rank->out_row_stride : rank->k;
        const int64_t mrs = rank->mask && !rank->mask_row_stride ? rank->n : rank->mask_row_stride;
        if (ors < rank->k || (rank->mask && mrs < rank->n)) return fail("salience_head_modulate: rank job strides too small");
        if (rank->mask && !rank->fill_value) return fail("salience_head_modulate: a masked rank job needs its fill value");
        r.score = rank->score; r.mask = rank->mask; r.mask_stride = mrs; r.fill = rank->fill_value; r.N = rank->n;
        r.k = rank->k; r.index_offset = rank->index_offset; r.out_score = rank->out_score; r.out_index = rank->out_index;
        r.out_stride = ors;
        rk_bx = (rank->n + 63) / 64;
        n3 = rk_bx * rank->batch;
        rk_tile = rk_bx * 64 < kRankTile ? rk_bx * 64 : kRankTile;
        const size_t lds_rk = (size_t)(rk_tile + kRankWaves * 64) * 4;
        if (lds_rk > lds) lds = lds_rk;
    }
    FinalizeJob fj{};
    int n4 = 0;
    if (finalize) {
        if (int rc = fill_finalize_job(fj, finalize, "salience_head_modulate")) return rc;
        n4 = 512;   // (a short host: the pass's own chain -- pieces per thread -- has to be short as well)
    }
    // (the jobs' bodies need 512 threads: two modulation blocks per workgroup then; alone the step runs in 256)
    const unsigned threads = (n3 || n4) ? 512 : kModThreads;
    const int per = (int)threads / kModThreads;
    hipLaunchKernelGGL(fused_modulate_rank_kernel, dim3((unsigned)((n1 + per - 1) / per + n3 + n4)), dim3(threads), lds,
                       static_cast<hipStream_t>(stream), m, m.nblk, batch_size, r, rk_bx, n3, rk_tile, fj, n4);
    return check_launch("salience_head_modulate");
}

extern "C" int sdetr_stage1_x3_with_value_proj(
    sdetr_stream_t stream, const float *x, int64_t x_batch_stride, int64_t x_row_stride, int batch_size, int tokens,
    int channels, const void *enc_weight_x3, const float *enc_bias, const float *enc_norm_weight,
    const float *enc_norm_bias, float enc_norm_eps, const float *row_scale, const float *coarse_score, int coarse_h,
    int coarse_w, int level_h, int level_w, const float *alpha, const float *norm_weight, const float *norm_bias,
    float norm_eps, const void *weight_x3, const float *bias, float *memory_out, int64_t memory_batch_stride,
    float *z_local, float *partial_sums,
    const void *vp_x, const void *vp_packed_weight, const float *vp_bias_padded, const uint8_t *vp_pad_mask,
    int vp_batch_size, int vp_spatial_size, int vp_num_heads, int vp_num_groups, void *vp_dst, int vp_dst_dtype,
    const sdetr_bordered_layout *vp_bordered)
{
    if (!vp_x) return fail("stage1_x3_with_value_proj: NULL pointer");
    return sdetr_stage1_x3_with_jobs(stream, x, x_batch_stride, x_row_stride, batch_size, tokens, channels, enc_weight_x3,
                                     enc_bias, enc_norm_weight, enc_norm_bias, enc_norm_eps, row_scale, coarse_score,
                                     coarse_h, coarse_w, level_h, level_w, alpha, norm_weight, norm_bias, norm_eps,
                                     weight_x3, bias, memory_out, memory_batch_stride, z_local, partial_sums, vp_x,
                                     vp_packed_weight, vp_bias_padded, vp_pad_mask, vp_batch_size, vp_spatial_size,
                                     vp_num_heads, vp_num_groups, vp_dst, vp_dst_dtype, vp_bordered, nullptr, nullptr);
}

// The stage-2 half of sdetr_salience_head_stage2 (the caller has launched the per-image constant already:
// `const_workspace` holds it) + a value-projection job in one launch.
extern "C" int sdetr_stage2_with_value_proj(sdetr_stream_t stream, const float *z_local, int batch_size, int tokens,
                                            const float *weight2_local_packed, const float *weight3_packed,
                                            const float *bias3, const float *weight4, const float *bias4,
                                            const float *const_workspace, float *score, float *score_flat,
                                            int64_t score_flat_stride, float *score_min, const void *vp_x,
                                            const void *vp_packed_weight, const float *vp_bias_padded,
                                            const uint8_t *vp_pad_mask, int vp_batch_size, int vp_spatial_size,
                                            int vp_num_heads, int vp_num_groups, void *vp_dst, int vp_dst_dtype,
                                            const sdetr_bordered_layout *vp_bordered, const void *weight2_local_x3,
                                            const float *partial_sums, const float *weight2, const float *bias2)
{
    if (batch_size <= 0 || tokens <= 0) return fail("stage2_with_value_proj: empty level");
    if (!z_local || (!weight2_local_packed && !weight2_local_x3) || !weight3_packed || !bias3 || !weight4 || !bias4 ||
        (!const_workspace && !partial_sums) || !score)
        return fail("stage2_with_value_proj: NULL pointer");
    if (partial_sums && (!weight2 || !bias2)) return fail("stage2_with_value_proj: the constant in the block needs layer2[0]");
    if (partial_sums && (tokens + 31) / 32 > kConstInBlockRows)
        return fail("stage2_with_value_proj: the constant in the block takes up to %d rows of partial sums", kConstInBlockRows);
    TLArgs t;
    size_t lds_tl = 0;
    int n2 = 0;
    if (int rc = fill_value_job(t, lds_tl, n2, vp_x, vp_packed_weight, vp_bias_padded, vp_pad_mask, vp_batch_size,
                                vp_spatial_size, vp_num_heads, vp_num_groups, vp_dst, vp_dst_dtype, vp_bordered))
        return rc;
    Stage2Args a;
    a.z_local = z_local; a.cst = const_workspace;
    a.w2a = reinterpret_cast<const float4 *>(weight2_local_packed);
    a.w2a_x3 = weight2_local_x3;
    a.w3 = reinterpret_cast<const float4 *>(weight3_packed);
    a.b3 = bias3; a.w4 = weight4; a.b4 = bias4; a.score = score; a.score2 = score_flat;
    a.score2_stride = score_flat_stride; a.n = tokens; a.score_min = score_min;
    if (partial_sums) {
        a.partial = partial_sums; a.partial_rows = (tokens + 31) / 32; a.w2 = weight2; a.b2 = bias2;
    }
    const int nblk = (tokens + kTM - 1) / kTM;
    const size_t lds_s2 = (size_t)kStage2LdsFloats * sizeof(float);
    const size_t lds = lds_tl > lds_s2 ? lds_tl : lds_s2;
    static DeviceOnce once;
    allow_dynamic_lds(fused_stage2_value_kernel, once, 160 * 1024);
    hipLaunchKernelGGL(fused_stage2_value_kernel, dim3((unsigned)(nblk * batch_size + n2)), dim3(768), lds,
                       static_cast<hipStream_t>(stream), a, nblk, batch_size, t);
    return check_launch("stage2_with_value_proj");
}

// sdetr_topk_attention_bf16 of an encoder layer + the deformable attention's offset | weight projection of the layer's
// queries (sdetr_token_linear_bf16 with x_add = pos, group_features = 48: the head-major slab [B, 8, rows, 48] the MSDA
// kernel reads): in-projection launch (which also marks the selected rows in `hint`), then ONE launch for the attention
// (40 workgroups; re-projects its 300 updated rows) and the projection of all other rows.  `proj_weight` is the plain
// [384, 256] bf16 weight in head-major row order, `proj_packed` / `proj_bias_padded` its token-linear packing; `hint`:
// int32 [batch, hint_batch_stride >= rows] of scratch, contents irrelevant on entry (a mark only counts if the
// selection confirms it); one per call in flight.  Only for 289..320 selected rows (the model's 300).
extern "C" int sdetr_topk_attention_with_projection_bf16(
    sdetr_stream_t stream, void *query, int64_t query_batch_stride, const void *pos, int64_t pos_batch_stride,
    const int64_t *selected, int batch_size, int num_rows, int num_selected, const void *in_proj_weight,
    const void *in_proj_bias, const void *out_proj_weight, const void *out_proj_bias, const void *norm_weight,
    const void *norm_bias, float norm_eps, void *workspace, int64_t workspace_bytes, const void *proj_weight,
    const void *proj_packed, const float *proj_bias_padded, void *slab, int32_t *hint, int64_t hint_batch_stride,
    const void *out_proj_frag, const void *proj_frag, int in_projection_done)
{
    if (batch_size <= 0 || num_rows <= 0 || num_selected <= 0 || num_selected > num_rows)
        return fail("topk_attention_with_projection: bad sizes (rows %d, selected %d)", num_rows, num_selected);
    const int npad = (num_selected + 31) / 32 * 32;
    if (npad != 320) return fail("topk_attention_with_projection: built for 289..320 selected rows (got %d)", num_selected);
    if (!query || !pos || !selected || !in_proj_weight || !in_proj_bias || !out_proj_weight || !out_proj_bias ||
        !norm_weight || !norm_bias || !workspace || !proj_weight || !proj_packed || !proj_bias_padded || !slab || !hint)
        return fail("topk_attention_with_projection: null pointer");
    if (hint_batch_stride < num_rows) return fail("topk_attention_with_projection: hint rows shorter than the layer");
    if (workspace_bytes < (int64_t)batch_size * npad * (512 + 256) * 2)
        return fail("topk_attention_with_projection: workspace too small");
    if (query_batch_stride < (int64_t)num_rows * kTkE || pos_batch_stride < (int64_t)num_rows * kTkE ||
        (query_batch_stride & 7) || (pos_batch_stride & 7))
        return fail("topk_attention_with_projection: bad batch strides");
    if (query_batch_stride != (int64_t)num_rows * kTkE)
        return fail("topk_attention_with_projection: the layer's queries must be contiguous [B, rows, 256]");
    hipStream_t hs = static_cast<hipStream_t>(stream);
    TkInArgs a{};
    a.query = (const bf16_t *)query; a.q_bs = query_batch_stride; a.pos = (const bf16_t *)pos; a.p_bs = pos_batch_stride;
    a.sel = selected; a.w = (const bf16_t *)in_proj_weight; a.bias = (const bf16_t *)in_proj_bias;
    a.qk = (bf16_t *)workspace; a.vt = a.qk + (int64_t)batch_size * npad * 512;
    a.B = batch_size; a.N = num_selected; a.Npad = npad; a.hint = hint; a.hint_bs = hint_batch_stride;
    // (in_projection_done: sdetr_topk_select_inproj_bf16 has filled `workspace` and `hint` in the selection's launch)
    if (!in_projection_done)
        if (int rc = sdetr_topk_inproj_launch(stream, &a)) return rc;
    TkOutArgs o{};
    o.qk = a.qk; o.vt = a.vt; o.sel = selected; o.query = (bf16_t *)query; o.q_bs = query_batch_stride;
    o.wo = (const bf16_t *)out_proj_weight; o.bo = (const bf16_t *)out_proj_bias;
    o.gamma = (const bf16_t *)norm_weight; o.beta = (const bf16_t *)norm_bias; o.eps = norm_eps;
    o.scale = 0.17677669529663687f;   // 1 / sqrt(32)
    o.B = batch_size; o.N = num_selected; o.Npad = npad;
    o.fx_w = (const bf16_t *)proj_weight; o.fx_b = proj_bias_padded; o.fx_pos = (const bf16_t *)pos;
    o.fx_p_bs = pos_batch_stride; o.fx_slab = (bf16_t *)slab; o.fx_rows = num_rows; o.fx_by_selection = 0;
    o.wo_frag = (const bf16_t *)out_proj_frag; o.fx_w_frag = (const bf16_t *)proj_frag;
    TLArgs t{};
    t.x = (const bf16_t *)query; t.pw = (const char *)proj_packed; t.bias = proj_bias_padded;
    t.T = batch_size * num_rows; t.N = 384; t.ntiles = 12; t.rows_per_batch = num_rows;
    t.x2 = (const bf16_t *)pos; t.x2_batch_stride = pos_batch_stride;
    t.out = (bf16_t *)slab; t.out_row_stride = 384; t.group = 48;
    t.skip_hint = hint; t.skip_hint_bs = hint_batch_stride; t.skip_sel = selected; t.skip_n = num_selected;
    const int nsteps = (t.ntiles + kTLStepTiles - 1) / kTLStepTiles;
    const size_t lds = 2 * (size_t)kTLStepBytes + (size_t)nsteps * 512;
    static DeviceOnce once;
    allow_dynamic_lds(fused_attn_proj_kernel<20>, once, 148 * 1024);   // (+ ~10 KB of static LDS in the attention body)
    const int at_blocks = batch_size * ((num_selected + kTkQ - 1) / kTkQ);
    const int tl_blocks = (t.T + kTLTokBlock - 1) / kTLTokBlock;
    hipLaunchKernelGGL((fused_attn_proj_kernel<20>), dim3((unsigned)(at_blocks + tl_blocks)), dim3(512), lds, hs, o, at_blocks, t);
    return check_launch("topk_attention_with_projection");
}
