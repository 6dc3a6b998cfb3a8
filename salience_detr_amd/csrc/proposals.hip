// Two-stage proposal selection after the encoder (row N1; reference models/bricks/salience_transformer.py:194-212,
// 249-295 and models/bricks/base_transformer.py:74-112).
//
//  * proposal_geometry_kernel: gen_encoder_output_proposals without the Linear/LayerNorm: per token the keep flag
//    (not padding AND the proposal box (cx, cy, w, h) inside (0.01, 0.99)) and the proposal in logit space
//    (+inf where the flag is off).  cx = (x + 0.5) / valid_w, cy = (y + 0.5) / valid_h, w = h = 0.05 * 2^level; the
//    valid extents are recounted from row 0 / column 0 of the mask by every block, exactly like the reference.
//  * grid_nms_kernel: nms_on_topk_index.  The reference builds the boxes [x-1, y-1, x+1, y+1] around the grid cell
//    of every selected token and calls torchvision.ops.batched_nms with one category per (image, level).  Those are
//    2x2 boxes on the integer grid: two of them overlap with IoU 1/3 (edge neighbours), 1/7 (diagonal neighbours)
//    or 0, so greedy NMS in descending score order is "drop a token iff an already kept token of higher rank is its
//    neighbour" with the neighbourhood fixed by the threshold (the host evaluates 2/6 > thr and 1/7 > thr in fp32
//    like the torchvision kernel would).  One workgroup per image keeps the rank of every selected token in an LDS
//    map over the flattened pyramid and resolves the greedy order by relaxation: a token whose higher-ranked
//    neighbours are all decided is decided (by induction on the rank this is exactly the sequential result); the
//    number of rounds is the longest chain of strictly descending neighbours, a handful on real score maps.
//    Kept tokens are compacted in rank order (= descending score, ties in list order).
//  * proposal_refine_kernel: enc_outputs_coord of the selected tokens: sigmoid(bbox_head(x)[i] + logit[index[i]]).
#include "common.h"

namespace sdetr {

constexpr int kMaxLvl = 8;

struct GeomArgs {
    const uint8_t *mask;   // [B, S]
    int64_t shapes[kMaxLvl][2];  // (h, w), host copy
    int64_t start[kMaxLvl];
    int B, S, L;
    uint8_t *keep;         // [B, S] or NULL
    float *logit;          // [B, S, 4] or NULL
};

// grid (chunks of 256 tokens of the widest level, L, B)
__global__ void __launch_bounds__(256) proposal_geometry_kernel(GeomArgs p)
{
    __shared__ int valid_hw[2];
    const int lvl = blockIdx.y, b = blockIdx.z;
    const int H = (int)p.shapes[lvl][0], W = (int)p.shapes[lvl][1];
    const int HW = H * W;
    if ((int)blockIdx.x * 256 >= HW) return;
    const uint8_t *mb = p.mask + (int64_t)b * p.S + p.start[lvl];
    const int tid = threadIdx.x;
    if (tid < 128) {
        int cnt = 0;
        if (tid < 64) { for (int i = tid; i < H; i += 64) cnt += mb[(int64_t)i * W] == 0; }
        else          { for (int i = tid - 64; i < W; i += 64) cnt += mb[i] == 0; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        if ((tid & 63) == 0) valid_hw[tid >> 6] = cnt;
    }
    __syncthreads();
    const int t = blockIdx.x * 256 + tid;
    if (t >= HW) return;
    const float vh = (float)valid_hw[0], vw = (float)valid_hw[1];
    const int y = t / W, x = t - y * W;
    const float cx = ((float)x + 0.5f) / vw, cy = ((float)y + 0.5f) / vh;
    const float wh = 0.05f * (float)(1 << lvl);
    const bool ok = mb[t] == 0 && cx > 0.01f && cx < 0.99f && cy > 0.01f && cy < 0.99f && wh > 0.01f && wh < 0.99f;
    const int64_t o = (int64_t)b * p.S + p.start[lvl] + t;
    if (p.keep) p.keep[o] = ok ? 1 : 0;
    if (p.logit) {
        float4 v;
        if (ok) {
            const float lw = logf(wh / (1.f - wh));
            v = make_float4(logf(cx / (1.f - cx)), logf(cy / (1.f - cy)), lw, lw);
        } else {
            v = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
        }
        reinterpret_cast<float4 *>(p.logit)[o] = v;
    }
}

struct NmsArgs {
    const int64_t *index;  // [B, K] token ids in descending score order
    int64_t index_batch_stride;
    int64_t shapes[kMaxLvl][2];
    int64_t start[kMaxLvl];
    int L, K, S, neighbourhood, max_keep;
    int64_t *out_index;    // [B, max_keep]
    int *out_count;        // [B] kept tokens (not clamped to max_keep)
};

constexpr int kNmsThreads = 1024;
constexpr uint16_t kNoRank = 0xffffu;

// LDS: rank map u16[S] | state u8[K] | (CACHE) the ranks of a token's higher-priority selected neighbours u16[K][8]
// CACHE (round 5): the decision rounds used to redo, for every undecided rank and every round, the id load, the level
// search, the division and eight rank-map lookups; the neighbourhood never changes, so it is resolved once into LDS and a
// round is one 16-byte read plus the neighbours' states.  Same fixpoint, same output.
template <bool CACHE>
__global__ void __launch_bounds__(kNmsThreads) grid_nms_kernel(NmsArgs p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint16_t *rank_of = reinterpret_cast<uint16_t *>(smem);
    uint8_t *state = smem + (((size_t)p.S * 2 + 15) & ~(size_t)15);            // 0 undecided, 1 kept, 2 dropped
    uint16_t *nbr = reinterpret_cast<uint16_t *>(state + (((size_t)p.K + 15) & ~(size_t)15));
    __shared__ int wave_sum[kNmsThreads / 64];
    __shared__ int carry;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t *idx = p.index + (int64_t)b * p.index_batch_stride;

    for (int i = tid; i < p.S; i += kNmsThreads) rank_of[i] = kNoRank;
    for (int r = tid; r < p.K; r += kNmsThreads) state[r] = 0;
    __syncthreads();
    for (int r = tid; r < p.K; r += kNmsThreads) {
        const int64_t t = idx[r];
        if (t >= 0 && t < p.S) rank_of[t] = (uint16_t)r;
        else state[r] = 2;                                                    // out-of-range id: never kept
    }
    __syncthreads();

    const int nnb = p.neighbourhood;   // 0, 4 or 8
    // the higher-priority selected neighbours of rank r (kNoRank: none in that direction)
    auto neighbours = [&](int r, uint16_t (&out)[8]) {
        const int t = (int)idx[r];
        int lvl = 0;
        while (lvl + 1 < p.L && t >= p.start[lvl + 1]) ++lvl;
        const int H = (int)p.shapes[lvl][0], W = (int)p.shapes[lvl][1];
        const int sp = t - (int)p.start[lvl];
        const int y = sp / W, x = sp - y * W;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            // k: W, E, N, S, NW, NE, SW, SE
            constexpr int kDx[8] = {-1, 1, 0, 0, -1, 1, -1, 1};
            constexpr int kDy[8] = {0, 0, -1, 1, -1, -1, 1, 1};
            out[k] = kNoRank;
            if (k >= nnb) continue;
            const int xx = x + kDx[k], yy = y + kDy[k];
            if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
            const uint16_t rr = rank_of[p.start[lvl] + yy * W + xx];
            if (rr != kNoRank && (int)rr < r) out[k] = rr;                    // selected and of higher priority
        }
    };
    if constexpr (CACHE) {
        for (int r = tid; r < p.K; r += kNmsThreads) {
            uint16_t nb[8];
            if (state[r] == 0) neighbours(r, nb);
            else
                for (int k = 0; k < 8; ++k) nb[k] = kNoRank;
            uint4 w;
            w.x = nb[0] | ((uint32_t)nb[1] << 16); w.y = nb[2] | ((uint32_t)nb[3] << 16);
            w.z = nb[4] | ((uint32_t)nb[5] << 16); w.w = nb[6] | ((uint32_t)nb[7] << 16);
            reinterpret_cast<uint4 *>(nbr)[r] = w;
        }
        __syncthreads();
    }
    // Decision rounds.  A chain of dependent tokens (a score ridge along a row: kept, dropped, kept, ...) resolves one link
    // per look at the states; a barrier per look made a round ~0.45 us and the benchmark's proposals ~150 rounds.  The
    // states only ever go 0 -> 1 | 2 and a decision reads decided neighbours only, so looking again WITHOUT a barrier is
    // safe (whatever another wave has already written is final): kSweeps looks per barrier, same fixpoint.
    constexpr int kSweeps = 4;     // (1, 4, 8, 16 measured alike in the step: a look costs what the barrier costs)
    volatile uint8_t *vstate = state;
    int pending = 1;
    while (pending) {
        int mine = 0;
#pragma unroll 1
        for (int sweep = 0; sweep < kSweeps; ++sweep) {
            mine = 0;
            for (int r = tid; r < p.K; r += kNmsThreads) {
                if (vstate[r] != 0) continue;
                uint16_t nb[8];
                if constexpr (CACHE) {
                    const uint4 w = reinterpret_cast<const uint4 *>(nbr)[r];
                    nb[0] = (uint16_t)w.x; nb[1] = (uint16_t)(w.x >> 16); nb[2] = (uint16_t)w.y; nb[3] = (uint16_t)(w.y >> 16);
                    nb[4] = (uint16_t)w.z; nb[5] = (uint16_t)(w.z >> 16); nb[6] = (uint16_t)w.w; nb[7] = (uint16_t)(w.w >> 16);
                } else {
                    neighbours(r, nb);
                }
                bool any_kept = false, any_open = false;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (nb[k] == kNoRank) continue;
                    const uint8_t s = vstate[nb[k]];
                    any_kept |= s == 1;
                    any_open |= s == 0;
                }
                if (any_kept) vstate[r] = 2;
                else if (!any_open) vstate[r] = 1;
                else mine = 1;
            }
            if (!__any(mine)) break;                         // (my wave has nothing left to decide this round)
        }
        pending = __syncthreads_or(mine);
    }

    // compact the kept tokens in rank order
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < p.K; base += kNmsThreads) {
        const int r = base + tid;
        const int kept = (r < p.K && state[r] == 1) ? 1 : 0;
        const uint64_t ballot = __ballot(kept);
        const int lane = tid & 63, wave = tid >> 6;
        const int before = __popcll(ballot & ((1ull << lane) - 1ull));
        if (lane == 0) wave_sum[wave] = __popcll(ballot);
        __syncthreads();
        int off = carry;
        for (int w = 0; w < wave; ++w) off += wave_sum[w];
        if (kept && off + before < p.max_keep) p.out_index[(int64_t)b * p.max_keep + off + before] = idx[r];
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < kNmsThreads / 64; ++w) tot += wave_sum[w];
            carry += tot;
        }
        __syncthreads();
    }
    if (tid == 0) p.out_count[b] = carry;
    // rows with fewer than max_keep survivors: pad with token 0 so that a caller that does not read the count back
    // (graph replay) never dereferences an undefined id
    for (int i = carry + tid; i < p.max_keep; i += kNmsThreads) p.out_index[(int64_t)b * p.max_keep + i] = 0;
}

template <typename DT>
__global__ void __launch_bounds__(256) proposal_refine_kernel(const DT *delta, const float *logit, const int64_t *index,
                                                              int64_t index_batch_stride, int S, int n, int64_t total,
                                                              float *out)
{
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int64_t b = gid / n, i = gid - b * n;
    const int64_t t = index[b * index_batch_stride + i];
    const float4 lg = reinterpret_cast<const float4 *>(logit)[b * S + t];
    const DT *d = delta + gid * 4;
    float v[4];
    if constexpr (sizeof(DT) == 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = act_lo((uint32_t)d[k]);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = d[k];
    }
    const float l4[4] = {lg.x, lg.y, lg.z, lg.w};
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = 1.f / (1.f + expf(-(v[k] + l4[k])));
    reinterpret_cast<float4 *>(out)[gid] = make_float4(o[0], o[1], o[2], o[3]);
}

static int copy_levels(const int64_t *shapes_host, int L, int64_t (*shapes)[2], int64_t *start, int64_t *total)
{
    int64_t cur = 0;
    for (int l = 0; l < L; ++l) {
        shapes[l][0] = shapes_host[2 * l];
        shapes[l][1] = shapes_host[2 * l + 1];
        if (shapes[l][0] <= 0 || shapes[l][1] <= 0) return 1;
        start[l] = cur;
        cur += shapes[l][0] * shapes[l][1];
    }
    *total = cur;
    return 0;
}

}  // namespace sdetr

using namespace sdetr;

extern "C" int sdetr_encoder_output_proposals(sdetr_stream_t stream, const uint8_t *padding_mask,
                                              const int64_t *level_shapes_host, int num_levels, int batch_size,
                                              int spatial_size, uint8_t *keep, float *proposal_logit)
{
    if (num_levels <= 0 || num_levels > kMaxLvl || batch_size < 0 || spatial_size < 0)
        return fail("encoder_output_proposals: bad sizes");
    if (!level_shapes_host) return fail("encoder_output_proposals: null level shapes");
    GeomArgs a{};
    int64_t total = 0;
    if (copy_levels(level_shapes_host, num_levels, a.shapes, a.start, &total) || total != spatial_size)
        return fail("encoder_output_proposals: level shapes do not add up to spatial_size");
    if (batch_size == 0 || spatial_size == 0) return 0;
    if (!padding_mask || (!keep && !proposal_logit)) return fail("encoder_output_proposals: null pointer");
    a.mask = padding_mask; a.B = batch_size; a.S = spatial_size; a.L = num_levels; a.keep = keep; a.logit = proposal_logit;
    int64_t widest = 0;
    for (int l = 0; l < num_levels; ++l) widest = a.shapes[l][0] * a.shapes[l][1] > widest ? a.shapes[l][0] * a.shapes[l][1] : widest;
    hipLaunchKernelGGL(proposal_geometry_kernel, dim3((unsigned)((widest + 255) / 256), num_levels, batch_size),
                       dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("encoder_output_proposals");
}

extern "C" int sdetr_grid_nms_topk(sdetr_stream_t stream, const int64_t *topk_index, int64_t index_batch_stride,
                                   const int64_t *level_shapes_host, int num_levels, int batch_size, int num_topk,
                                   int spatial_size, int neighbourhood, int max_keep, int64_t *out_index,
                                   int *out_count)
{
    if (num_levels <= 0 || num_levels > kMaxLvl || batch_size < 0 || num_topk < 0 || max_keep <= 0)
        return fail("grid_nms_topk: bad sizes");
    if (neighbourhood != 0 && neighbourhood != 4 && neighbourhood != 8)
        return fail("grid_nms_topk: neighbourhood must be 0, 4 or 8");
    if (num_topk >= 0xffff) return fail("grid_nms_topk: at most 65534 selected tokens per image");
    if (!level_shapes_host) return fail("grid_nms_topk: null level shapes");
    NmsArgs a{};
    int64_t total = 0;
    if (copy_levels(level_shapes_host, num_levels, a.shapes, a.start, &total) || total != spatial_size)
        return fail("grid_nms_topk: level shapes do not add up to spatial_size");
    const size_t lds_plain = (((size_t)spatial_size * 2 + 15) & ~(size_t)15) + (((size_t)num_topk + 15) & ~(size_t)15);
    if (lds_plain > 150 * 1024) return fail("grid_nms_topk: pyramid of %d tokens + %d selected does not fit the LDS rank map",
                                            spatial_size, num_topk);
    const size_t lds_cached = lds_plain + (size_t)num_topk * 16;
    const bool cached = lds_cached <= 150 * 1024;
    if (batch_size == 0) return 0;
    if (!topk_index || !out_index || !out_count) return fail("grid_nms_topk: null pointer");
    if (index_batch_stride < num_topk) return fail("grid_nms_topk: index batch stride too small");
    a.index = topk_index; a.index_batch_stride = index_batch_stride; a.L = num_levels; a.K = num_topk;
    a.S = spatial_size; a.neighbourhood = neighbourhood; a.max_keep = max_keep; a.out_index = out_index;
    a.out_count = out_count;
    static DeviceOnce lds_once, lds_once_cached;
    if (cached) {
        allow_dynamic_lds(grid_nms_kernel<true>, lds_once_cached, 150 * 1024);
        hipLaunchKernelGGL(grid_nms_kernel<true>, dim3(batch_size), dim3(kNmsThreads), lds_cached, (hipStream_t)stream, a);
    } else {
        allow_dynamic_lds(grid_nms_kernel<false>, lds_once, 150 * 1024);
        hipLaunchKernelGGL(grid_nms_kernel<false>, dim3(batch_size), dim3(kNmsThreads), lds_plain, (hipStream_t)stream, a);
    }
    return check_launch("grid_nms_topk");
}

extern "C" int sdetr_proposal_refine(sdetr_stream_t stream, const void *delta, int delta_dtype,
                                     const float *proposal_logit, const int64_t *index, int64_t index_batch_stride,
                                     int batch_size, int spatial_size, int num_select, float *out)
{
    if (batch_size < 0 || spatial_size < 0 || num_select < 0) return fail("proposal_refine: bad sizes");
    if (delta_dtype != SDETR_F32 && delta_dtype != kActCode) return fail("proposal_refine: delta dtype must be f32 or bf16");
    const int64_t total = (int64_t)batch_size * num_select;
    if (total == 0) return 0;
    if (!delta || !proposal_logit || !index || !out) return fail("proposal_refine: null pointer");
    if (index_batch_stride < num_select) return fail("proposal_refine: index batch stride too small");
    const dim3 grid((unsigned)((total + 255) / 256));
    if (delta_dtype == kActCode)
        hipLaunchKernelGGL(proposal_refine_kernel<uint16_t>, grid, dim3(256), 0, (hipStream_t)stream,
                           (const uint16_t *)delta, proposal_logit, index, index_batch_stride, spatial_size, num_select,
                           total, out);
    else
        hipLaunchKernelGGL(proposal_refine_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)delta,
                           proposal_logit, index, index_batch_stride, spatial_size, num_select, total, out);
    return check_launch("proposal_refine");
}
