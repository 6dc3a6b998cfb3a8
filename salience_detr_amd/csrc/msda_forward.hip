// Multi-scale deformable attention, forward gather-reduce, for gfx950 (MI355X).
//
// Semantics follow the reference op (models/bricks/ops/cuda/ms_deform_im2col_cuda.cuh:22-73,
// 226-288 and models/bricks/ms_deform_attn.py:159-212, 322-355); the kernel structure does not:
//
//  * A "group" of G = D*sizeof(value)/16 adjacent lanes owns one (batch, query, head) output row
//    and every lane moves 16 bytes per corner (global_load_dwordx4): a 64-lane wavefront serves
//    64/G rows, so a head's D channels of one pixel are exactly one 64/128-byte segment.
//  * A workgroup serves ONE head of ONE image: blockIdx % num_heads is the head, and because the
//    dispatcher places block b on XCD b % 8, with 8 heads each XCD's private 4 MiB L2 only ever
//    holds "its" head's slice of the value map (2.9 MB for the 800x1333 pyramid, batch 2, bf16
//    head-major) instead of thrashing on the whole 23-46 MB tensor.
//  * Sampling set-up (softmax over the L*P logits, location arithmetic, bounds tests, the four
//    corner byte offsets and the four bilinear*attention weights) is done ONCE per sample by one
//    lane and broadcast to the group through LDS (two ds_read_b128 per sample), instead of being
//    recomputed by every channel thread as in the reference (32x redundant there).
//  * The fused entry point consumes the raw sampling_offsets / attention_weights projections and
//    the reference points directly, so sampling_locations [B,Nq,M,L,P,2] and the softmaxed weights
//    are never written to HBM.
#include "common.h"

namespace sdetr {

void note_forward_kernel(int which);  // abi.hip

constexpr int kChunk = 16;  // samples staged per LDS round

template <typename VT>
struct ValTraits;
template <>
struct ValTraits<float> {
    static constexpr int kCpl = 4;  // channels per 16-byte lane load
    __device__ static __forceinline__ void fma4(float *acc, const uint4 &v, float w)
    {
        acc[0] = fmaf(w, __uint_as_float(v.x), acc[0]);
        acc[1] = fmaf(w, __uint_as_float(v.y), acc[1]);
        acc[2] = fmaf(w, __uint_as_float(v.z), acc[2]);
        acc[3] = fmaf(w, __uint_as_float(v.w), acc[3]);
    }
};
template <>
struct ValTraits<bf16_t> {
    static constexpr int kCpl = 8;
    __device__ static __forceinline__ void fma4(float *acc, const uint4 &v, float w)
    {
        acc[0] = fmaf(w, bf16_lo(v.x), acc[0]);
        acc[1] = fmaf(w, bf16_hi(v.x), acc[1]);
        acc[2] = fmaf(w, bf16_lo(v.y), acc[2]);
        acc[3] = fmaf(w, bf16_hi(v.y), acc[3]);
        acc[4] = fmaf(w, bf16_lo(v.z), acc[4]);
        acc[5] = fmaf(w, bf16_hi(v.z), acc[5]);
        acc[6] = fmaf(w, bf16_lo(v.w), acc[6]);
        acc[7] = fmaf(w, bf16_hi(v.w), acc[7]);
    }
};

template <>
struct ValTraits<half_t> {
    static constexpr int kCpl = 8;
    __device__ static __forceinline__ void fma4(float *acc, const uint4 &v, float w)
    {
        acc[0] = fma_f16lo(v.x, w, acc[0]);
        acc[1] = fma_f16hi(v.x, w, acc[1]);
        acc[2] = fma_f16lo(v.y, w, acc[2]);
        acc[3] = fma_f16hi(v.y, w, acc[3]);
        acc[4] = fma_f16lo(v.z, w, acc[4]);
        acc[5] = fma_f16hi(v.z, w, acc[5]);
        acc[6] = fma_f16lo(v.w, w, acc[6]);
        acc[7] = fma_f16hi(v.w, w, acc[7]);
    }
};

struct GatherArgs {
    const char *value;
    const int64_t *shapes;
    const int64_t *lsi;
    // explicit mode
    const float *loc;
    const float *aw;
    // fused mode
    const float *ref;
    int64_t ref_batch_stride;   // floats between images (Nq * L * ref_dim when contiguous)
    int ref_dim;
    const void *proj;
    int proj_bf16;
    int proj_hm;          // 1: [B, M, Nq, 3*L*P] (per head: offsets then logits), 0: token rows of proj_stride
    int64_t proj_stride;
    const int32_t *order;
    void *out;
    int out_bf16;
    int B, Nv, M, L, Nq, P;
    int nchunk;  // query chunks per image
};

template <typename PT>
__device__ __forceinline__ float proj_elem(const void *proj, int64_t idx);
template <>
__device__ __forceinline__ float proj_elem<float>(const void *proj, int64_t idx)
{
    return reinterpret_cast<const float *>(proj)[idx];
}
template <>
__device__ __forceinline__ float proj_elem<bf16_t>(const void *proj, int64_t idx)
{
    return act_lo((uint32_t) reinterpret_cast<const bf16_t *>(proj)[idx]);   // (the projection slab is an activation)
}

// One sample's descriptor: 4 corner byte offsets (relative to the block's value base, lane offset excluded)
// and 4 weights = bilinear weight * attention weight (0 for an out-of-range corner, whose offset is clamped
// into the map so the load stays legal).  Written for instruction count -- the set-up is ~40 % of the
// kernel's VALU work: separable validity (row / column) folded into the 1-d weights, clamps as v_med3,
// no division, no branches.
__device__ __forceinline__ void make_descriptor(float x, float y, float a, int H, int W, int level_start,
                                                uint32_t pixel_bytes, uint32_t *d)
{
    const float fH = (float)H, fW = (float)W;
    const float h_im = fmaf(y, fH, -0.5f);
    const float w_im = fmaf(x, fW, -0.5f);
    const bool inside = (h_im > -1.f) & (w_im > -1.f) & (h_im < fH) & (w_im < fW);
    const float fy = floorf(h_im), fx = floorf(w_im);
    const float ly = h_im - fy, lx = w_im - fx;
    const int y0 = (int)fy, x0 = (int)fx;  // (int) saturates for wild values; weights are zero then
    a = inside ? a : 0.f;
    const float wy0 = (y0 >= 0) ? (1.f - ly) * a : 0.f;
    const float wy1 = (y0 + 1 <= H - 1) ? ly * a : 0.f;
    const float wx0 = (x0 >= 0) ? (1.f - lx) : 0.f;
    const float wx1 = (x0 + 1 <= W - 1) ? lx : 0.f;
    const int y0c = min(max(y0, 0), H - 1), y1c = min(max(y0 + 1, 0), H - 1);
    const int x0c = min(max(x0, 0), W - 1), x1c = min(max(x0 + 1, 0), W - 1);
    const uint32_t r0 = (uint32_t)(level_start + y0c * W), r1 = (uint32_t)(level_start + y1c * W);
    d[0] = (r0 + x0c) * pixel_bytes;
    d[1] = (r0 + x1c) * pixel_bytes;
    d[2] = (r1 + x0c) * pixel_bytes;
    d[3] = (r1 + x1c) * pixel_bytes;
    d[4] = __float_as_uint(wy0 * wx0);
    d[5] = __float_as_uint(wy0 * wx1);
    d[6] = __float_as_uint(wy1 * wx0);
    d[7] = __float_as_uint(wy1 * wx1);
}

template <typename VT, int D, bool HEAD_MAJOR, bool FUSED>
__global__ void __launch_bounds__(kBlock) msda_gather_kernel(GatherArgs p)
{
    using T = ValTraits<VT>;
    constexpr int CPL = T::kCpl;
    constexpr int G = D / CPL;            // lanes per (b,q,m) row
    constexpr int GPB = kBlock / G;       // rows per workgroup
    constexpr int DSTRIDE = kChunk * 8 + 4;  // descriptor dwords per row (+4: bank spread)
    static_assert(D % CPL == 0 && (G & (G - 1)) == 0 && G <= kWave, "unsupported head dim");

    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *desc = smem;                                  // [GPB][DSTRIDE]
    int *lvl_tab = reinterpret_cast<int *>(smem + GPB * DSTRIDE);  // [L][3] = H, W, start
    float *lvl_inv = reinterpret_cast<float *>(lvl_tab + kMaxLevels * 3);  // [L][2] = 1/W, 1/H
    uint8_t *lvl_of = reinterpret_cast<uint8_t *>(lvl_inv + kMaxLevels * 2);  // [kChunk] level of a round's slot

    const int tid = threadIdx.x;
    const int m = blockIdx.x % p.M;
    const int chunk_global = blockIdx.x / p.M;
    const int b = chunk_global / p.nchunk;
    const int chunk = chunk_global - b * p.nchunk;
    const int g = tid / G;        // row within the block
    const int j = tid - g * G;    // lane within the row
    const int slot = chunk * GPB + g;
    const bool active = slot < p.Nq;
    int q = 0;
    if (active) q = p.order ? p.order[(int64_t)b * p.Nq + slot] : slot;

    if (tid < p.L) {  // the only divisions of the kernel: once per level per block
        const int H = (int)p.shapes[2 * tid], W = (int)p.shapes[2 * tid + 1];
        lvl_tab[tid * 3 + 0] = H;
        lvl_tab[tid * 3 + 1] = W;
        lvl_tab[tid * 3 + 2] = (int)p.lsi[tid];
        lvl_inv[tid * 2 + 0] = 1.0f / (float)W;
        lvl_inv[tid * 2 + 1] = 1.0f / (float)H;
    }
    __syncthreads();
    const float inv_P = 0.5f / (float)p.P;  // offsets / num_points * wh * 0.5 (4-d reference boxes)

    const int LP = p.L * p.P;
    const int64_t row = ((int64_t)b * p.Nq + q) * p.M + m;  // (b,q,m) row index
    constexpr uint32_t kPixelBytesHM = D * sizeof(VT);
    const uint32_t pixel_bytes = HEAD_MAJOR ? kPixelBytesHM : (uint32_t)(p.M * D * sizeof(VT));
    // block-uniform base of this image (and head, when head-major)
    const char *base = p.value + (HEAD_MAJOR ? ((int64_t)b * p.M + m) * p.Nv * (int64_t)kPixelBytesHM
                                             : (int64_t)b * p.Nv * (int64_t)pixel_bytes);
    const uint32_t lane_off = (HEAD_MAJOR ? 0u : (uint32_t)(m * D * sizeof(VT))) + (uint32_t)(j * 16);
    const __amdgpu_buffer_rsrc_t rsrc = make_uniform_rsrc(base, (uint32_t)((int64_t)p.Nv * pixel_bytes));

    // ---- fused mode: softmax statistics over this row's L*P logits, spread over the G lanes ----
    // All of a lane's logits are loaded in ONE batch (clamped indices, no per-sample branch) so the set-up
    // costs one memory round trip instead of one per sample.
    constexpr int TMAX = (kChunk + G - 1) / G;  // samples a lane owns per LDS round
    float sm_max = 0.f, sm_inv = 1.f;
    const int64_t proj_row = FUSED ? ((int64_t)b * p.Nq + q) * p.proj_stride : 0;
    const int64_t logit0 = proj_row + (int64_t)p.M * LP * 2 + (int64_t)m * LP;
    if (FUSED) {
        float mx = -INFINITY, sum = 0.f;
        for (int s0 = 0; s0 < LP; s0 += kChunk) {  // one trip when L*P <= 16
            float lg[TMAX];
            if (p.proj_bf16) {  // dtype branch OUTSIDE the batch so the loads issue back to back
#pragma unroll
                for (int t = 0; t < TMAX; ++t) lg[t] = proj_elem<bf16_t>(p.proj, logit0 + min(s0 + j + t * G, LP - 1));
            } else {
#pragma unroll
                for (int t = 0; t < TMAX; ++t) lg[t] = proj_elem<float>(p.proj, logit0 + min(s0 + j + t * G, LP - 1));
            }
            float cmx = -INFINITY;
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
                if (s0 + j + t * G < min(LP, s0 + kChunk)) cmx = fmaxf(cmx, lg[t]);
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) cmx = fmaxf(cmx, __shfl_xor(cmx, o, G));
            const float nmx = fmaxf(mx, cmx);
            float csum = 0.f;
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
                if (s0 + j + t * G < min(LP, s0 + kChunk)) csum += __expf(lg[t] - nmx);
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) csum += __shfl_xor(csum, o, G);
            sum = sum * __expf(mx - nmx) + csum;  // online softmax across rounds (mx = -inf first: factor 0)
            mx = nmx;
        }
        sm_max = mx;
        sm_inv = 1.f / sum;
    }

    float acc[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) acc[c] = 0.f;

    uint32_t *my_desc = desc + g * DSTRIDE;
    for (int c0 = 0; c0 < LP; c0 += kChunk) {
        const int ns = min(kChunk, LP - c0);
        if (c0 > 0) __syncthreads();  // previous chunk fully consumed
        if (tid < kChunk) lvl_of[tid] = (uint8_t)(min(c0 + tid, LP - 1) / p.P);  // one division per slot per block
        __syncthreads();
        {
            // raw per-sample inputs of this lane, all loads first
            float rx[TMAX], ry[TMAX], ra[TMAX], rr[TMAX][4];
            int lv[TMAX];
#pragma unroll
            for (int t = 0; t < TMAX; ++t) lv[t] = lvl_of[min(j + t * G, kChunk - 1)];
            if (FUSED) {
                if (p.proj_bf16) {
#pragma unroll
                    for (int t = 0; t < TMAX; ++t) {
                        const int s = min(c0 + j + t * G, LP - 1);
                        const uint32_t xy = *reinterpret_cast<const uint32_t *>(
                            reinterpret_cast<const bf16_t *>(p.proj) + proj_row + ((int64_t)m * LP + s) * 2);
                        rx[t] = act_lo(xy);
                        ry[t] = act_hi(xy);
                        ra[t] = proj_elem<bf16_t>(p.proj, logit0 + s);
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < TMAX; ++t) {
                        const int s = min(c0 + j + t * G, LP - 1);
                        const float2 xy = *reinterpret_cast<const float2 *>(
                            reinterpret_cast<const float *>(p.proj) + proj_row + ((int64_t)m * LP + s) * 2);
                        rx[t] = xy.x;
                        ry[t] = xy.y;
                        ra[t] = proj_elem<float>(p.proj, logit0 + s);
                    }
                }
                if (p.ref_dim == 4) {
#pragma unroll
                    for (int t = 0; t < TMAX; ++t) {
                        const float4 r = *reinterpret_cast<const float4 *>(p.ref + (int64_t)b * p.ref_batch_stride + ((int64_t)q * p.L + lv[t]) * 4);
                        rr[t][0] = r.x; rr[t][1] = r.y; rr[t][2] = r.z; rr[t][3] = r.w;
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < TMAX; ++t) {
                        const float2 r = *reinterpret_cast<const float2 *>(p.ref + (int64_t)b * p.ref_batch_stride + ((int64_t)q * p.L + lv[t]) * 2);
                        rr[t][0] = r.x; rr[t][1] = r.y; rr[t][2] = 0.f; rr[t][3] = 0.f;
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < TMAX; ++t) {
                    const int s = min(c0 + j + t * G, LP - 1);
                    const float2 xy = reinterpret_cast<const float2 *>(p.loc)[row * LP + s];
                    rx[t] = xy.x;
                    ry[t] = xy.y;
                    ra[t] = p.aw[row * LP + s];
                }
            }
#pragma unroll
            for (int t = 0; t < TMAX; ++t) {
                const int ts = j + t * G;  // slot within this round
                if (ts < ns && active) {
                    const int l = lv[t];
                    const int H = lvl_tab[l * 3], W = lvl_tab[l * 3 + 1], start = lvl_tab[l * 3 + 2];
                    float x, y, a;
                    if (FUSED) {
                        a = __expf(ra[t] - sm_max) * sm_inv;
                        if (p.ref_dim == 2) {  // ref + offset / (W, H): reciprocal + fma (<= 1 ulp from a true division)
                            x = fmaf(rx[t], lvl_inv[l * 2], rr[t][0]);
                            y = fmaf(ry[t], lvl_inv[l * 2 + 1], rr[t][1]);
                        } else {
                            x = fmaf(rx[t] * inv_P, rr[t][2], rr[t][0]);
                            y = fmaf(ry[t] * inv_P, rr[t][3], rr[t][1]);
                        }
                    } else {
                        x = rx[t];
                        y = ry[t];
                        a = ra[t];
                    }
                    uint32_t d[8];
                    make_descriptor(x, y, a, H, W, start, pixel_bytes, d);
                    *reinterpret_cast<uint4 *>(my_desc + ts * 8) = make_uint4(d[0], d[1], d[2], d[3]);
                    *reinterpret_cast<uint4 *>(my_desc + ts * 8 + 4) = make_uint4(d[4], d[5], d[6], d[7]);
                }
            }
        }
        __syncthreads();
        if (active) {
            // Software-pipelined gather: batches of 2 samples (8 x 16-byte loads), two register buffers; the
            // loads of batch k+1 are issued BEFORE batch k is accumulated, so the memory pipe never drains
            // while the wave does its unpack/FMA work (the loop was latency-bound with load -> wait -> math).
            auto issue = [&](uint4 (&v)[2][4], int t) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int tt = min(t + u, ns - 1);  // clamped: a tail slot re-reads a valid sample
                    const uint4 o = *reinterpret_cast<const uint4 *>(my_desc + tt * 8);
                    v[u][0] = buffer_load16(rsrc, o.x + lane_off);
                    v[u][1] = buffer_load16(rsrc, o.y + lane_off);
                    v[u][2] = buffer_load16(rsrc, o.z + lane_off);
                    v[u][3] = buffer_load16(rsrc, o.w + lane_off);
                }
            };
            auto accumulate = [&](const uint4 (&v)[2][4], int t) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    uint4 w = *reinterpret_cast<const uint4 *>(my_desc + min(t + u, ns - 1) * 8 + 4);
                    if (t + u >= ns) w = make_uint4(0u, 0u, 0u, 0u);  // tail slot contributes nothing
                    T::fma4(acc, v[u][0], __uint_as_float(w.x));
                    T::fma4(acc, v[u][1], __uint_as_float(w.y));
                    T::fma4(acc, v[u][2], __uint_as_float(w.z));
                    T::fma4(acc, v[u][3], __uint_as_float(w.w));
                }
            };
            uint4 va[2][4], vb[2][4];
            issue(va, 0);
            for (int t = 0; t < ns; t += 4) {
                if (t + 2 < ns) issue(vb, t + 2);
                accumulate(va, t);
                if (t + 4 < ns) issue(va, t + 4);
                if (t + 2 < ns) accumulate(vb, t + 2);
            }
        }
    }

    if (active) {
        const int64_t o = row * D + j * CPL;
        if (p.out_bf16) {
            bf16_t *out = reinterpret_cast<bf16_t *>(p.out) + o;
            if (CPL == 8) {
                *reinterpret_cast<uint4 *>(out) =
                    make_uint4(pack_act2(acc[0], acc[1]), pack_act2(acc[2], acc[3]),
                               pack_act2(acc[4 % CPL], acc[5 % CPL]), pack_act2(acc[6 % CPL], acc[7 % CPL]));
            } else {
                *reinterpret_cast<uint2 *>(out) = make_uint2(pack_act2(acc[0], acc[1]), pack_act2(acc[2], acc[3]));
            }
        } else {
            float *out = reinterpret_cast<float *>(p.out) + o;
#pragma unroll
            for (int c = 0; c < CPL; c += 4)
                *reinterpret_cast<float4 *>(out + c) = make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The Salience-DETR shape -- head-major 16-bit value, D = 32 (4 lanes per row), L = 4, P = 4, fused
// projections -- with everything that shape makes constant folded in: lane j of a row's quad owns level j
// (its four points are contiguous in the projection row: ONE 16-byte load of offsets, one 8-byte load of
// logits, one reference point), softmax across the quad by two shuffles, no clamps, no integer division,
// no per-slot level table.  Same descriptors / gather loop as the general kernel.
template <typename VT>
__global__ void __launch_bounds__(kBlock) msda_gather_l4p4_kernel(GatherArgs p)
{
    using T = ValTraits<VT>;
    constexpr int D = 32, G = 4, GPB = kBlock / G, LP = 16, P = 4, L = 4;
    constexpr int DSTRIDE = LP * 8 + 4;
    constexpr uint32_t kPixBytes = D * sizeof(VT);
    static_assert(sizeof(VT) == 2, "16-bit value maps");
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *desc = smem;

    const int tid = threadIdx.x;
    const int m = blockIdx.x % p.M;
    const int chunk_global = blockIdx.x / p.M;
    const int b = chunk_global / p.nchunk;
    const int chunk = chunk_global - b * p.nchunk;
    const int g = tid >> 2, j = tid & 3;
    const int slot = chunk * GPB + g;
    const bool active = slot < p.Nq;
    const int q = active ? (p.order ? p.order[(int64_t)b * p.Nq + slot] : slot) : 0;

    // this lane's level (uniform across quads): scalar loads, no LDS table
    const int H = (int)p.shapes[2 * j], W = (int)p.shapes[2 * j + 1], start = (int)p.lsi[j];
    const float invW = 1.0f / (float)W, invH = 1.0f / (float)H;

    const int64_t bq = (int64_t)b * p.Nq + q;
    const int64_t row = bq * p.M + m;
    const char *base = p.value + ((int64_t)b * p.M + m) * p.Nv * (int64_t)kPixBytes;
    const __amdgpu_buffer_rsrc_t rsrc = make_uniform_rsrc(base, (uint32_t)((int64_t)p.Nv * kPixBytes));
    const uint32_t lane_off = (uint32_t)(j * 16);

    // ---- raw inputs: offsets of my 4 points (x,y), their logits, my level's reference point ----
    float ox[4], oy[4], lg[4];
    {
        // head-major projection: this head's 32 offsets + 16 logits of a query are one 96-byte piece of a slab that
        // only this head (= this XCD) reads; in token rows the 128-byte lines are shared by 2 (offsets) or 4 (logits)
        // heads and every XCD's L2 fetches them again (measured 1.65x the algorithmic bytes at layer 0)
        const int64_t hq = p.proj_hm ? (((int64_t)b * p.M + m) * p.Nq + q) * (LP * 3) : 0;
        const int64_t o_idx = p.proj_hm ? hq + j * (P * 2) : bq * p.proj_stride + (m * LP + j * P) * 2;
        const int64_t l_idx = p.proj_hm ? hq + LP * 2 + j * P : bq * p.proj_stride + p.M * LP * 2 + m * LP + j * P;
        if (p.proj_bf16) {
            const bf16_t *pp = reinterpret_cast<const bf16_t *>(p.proj);
            const uint4 o = *reinterpret_cast<const uint4 *>(pp + o_idx);
            const uint2 gg = *reinterpret_cast<const uint2 *>(pp + l_idx);
            ox[0] = act_lo(o.x); oy[0] = act_hi(o.x); ox[1] = act_lo(o.y); oy[1] = act_hi(o.y);
            ox[2] = act_lo(o.z); oy[2] = act_hi(o.z); ox[3] = act_lo(o.w); oy[3] = act_hi(o.w);
            lg[0] = act_lo(gg.x); lg[1] = act_hi(gg.x); lg[2] = act_lo(gg.y); lg[3] = act_hi(gg.y);
        } else {
            const float *pp = reinterpret_cast<const float *>(p.proj);
            const float4 o0 = *reinterpret_cast<const float4 *>(pp + o_idx);
            const float4 o1 = *reinterpret_cast<const float4 *>(pp + o_idx + 4);
            const float4 gg = *reinterpret_cast<const float4 *>(pp + l_idx);
            ox[0] = o0.x; oy[0] = o0.y; ox[1] = o0.z; oy[1] = o0.w;
            ox[2] = o1.x; oy[2] = o1.y; ox[3] = o1.z; oy[3] = o1.w;
            lg[0] = gg.x; lg[1] = gg.y; lg[2] = gg.z; lg[3] = gg.w;
        }
    }
    float rx, ry, rw = 0.f, rh = 0.f;
    if (p.ref_dim == 4) {
        const float4 r = *reinterpret_cast<const float4 *>(p.ref + (int64_t)b * p.ref_batch_stride + ((int64_t)q * L + j) * 4);
        rx = r.x; ry = r.y; rw = r.z; rh = r.w;
    } else {
        const float2 r = *reinterpret_cast<const float2 *>(p.ref + (int64_t)b * p.ref_batch_stride + ((int64_t)q * L + j) * 2);
        rx = r.x; ry = r.y;
    }
    // softmax over the quad's 16 logits
    float mx = fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3]));
    mx = fmaxf(mx, __shfl_xor(mx, 1, 4));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 4));
    float e[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) e[t] = __expf(lg[t] - mx);
    float sum = (e[0] + e[1]) + (e[2] + e[3]);
    sum += __shfl_xor(sum, 1, 4);
    sum += __shfl_xor(sum, 2, 4);
    const float inv = active ? __builtin_amdgcn_rcpf(sum) : 0.f;  // inactive rows: all weights zero

    uint32_t *my_desc = desc + g * DSTRIDE;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float x, y;
        if (p.ref_dim == 2) {
            x = fmaf(ox[t], invW, rx);
            y = fmaf(oy[t], invH, ry);
        } else {
            x = fmaf(ox[t] * 0.125f, rw, rx);  // offset / num_points * w * 0.5, num_points = 4
            y = fmaf(oy[t] * 0.125f, rh, ry);
        }
        uint32_t d[8];
        make_descriptor(x, y, e[t] * inv, H, W, start, kPixBytes, d);
        *reinterpret_cast<uint4 *>(my_desc + (j * 4 + t) * 8) = make_uint4(d[0], d[1], d[2], d[3]);
        *reinterpret_cast<uint4 *>(my_desc + (j * 4 + t) * 8 + 4) = make_uint4(d[4], d[5], d[6], d[7]);
    }
    // the four lanes of a quad are in one wavefront: LDS writes above are visible to the reads below once the
    // wave's own LDS queue drains -- no workgroup barrier
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();

    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    auto issue = [&](uint4 (&v)[2][4], int t) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint4 o = *reinterpret_cast<const uint4 *>(my_desc + (t + u) * 8);
            v[u][0] = buffer_load16(rsrc, o.x + lane_off);
            v[u][1] = buffer_load16(rsrc, o.y + lane_off);
            v[u][2] = buffer_load16(rsrc, o.z + lane_off);
            v[u][3] = buffer_load16(rsrc, o.w + lane_off);
        }
    };
    auto accumulate = [&](const uint4 (&v)[2][4], int t) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint4 w = *reinterpret_cast<const uint4 *>(my_desc + (t + u) * 8 + 4);
            T::fma4(acc, v[u][0], __uint_as_float(w.x));
            T::fma4(acc, v[u][1], __uint_as_float(w.y));
            T::fma4(acc, v[u][2], __uint_as_float(w.z));
            T::fma4(acc, v[u][3], __uint_as_float(w.w));
        }
    };
    uint4 va[2][4], vb[2][4];
    issue(va, 0);
#pragma unroll
    for (int t = 0; t < LP; t += 4) {
        issue(vb, t + 2);
        accumulate(va, t);
        if (t + 4 < LP) issue(va, t + 4);
        accumulate(vb, t + 2);
    }
    if (active) {
        const int64_t o = row * D + j * 8;
        if (p.out_bf16) {
            *reinterpret_cast<uint4 *>(reinterpret_cast<bf16_t *>(p.out) + o) =
                make_uint4(pack_act2(acc[0], acc[1]), pack_act2(acc[2], acc[3]), pack_act2(acc[4], acc[5]),
                           pack_act2(acc[6], acc[7]));
        } else {
            float *out = reinterpret_cast<float *>(p.out) + o;
            *reinterpret_cast<float4 *>(out) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            *reinterpret_cast<float4 *>(out + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
        }
    }
}

template <typename VT>
static int launch_gather_l4p4(hipStream_t stream, GatherArgs &a)
{
    constexpr int GPB = kBlock / 4;
    a.nchunk = (a.Nq + GPB - 1) / GPB;
    const int64_t blocks = (int64_t)a.B * a.nchunk * a.M;
    if (blocks > 0x7fffffffLL) return fail("msda: grid too large");
    const size_t lds = (size_t)GPB * (16 * 8 + 4) * 4;
    hipLaunchKernelGGL((msda_gather_l4p4_kernel<VT>), dim3((unsigned)blocks), dim3(kBlock), lds, stream, a);
    note_forward_kernel(SDETR_KERNEL_MSDA_L4P4);
    return check_launch("msda_gather_l4p4");
}

// Generic fallback: any head dim, fp32/fp64, reference layout.  One thread per (b,q,m,c).
template <typename S>
__global__ void __launch_bounds__(kBlock) msda_generic_kernel(int64_t n, const S *value, const int64_t *shapes,
                                                              const int64_t *lsi, const S *loc, const S *aw, int Nv,
                                                              int M, int D, int L, int Nq, int P, S *out)
{
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % D);
        const int64_t row = idx / D;
        const int m = (int)(row % M);
        const int64_t bq = row / M;
        const int b = (int)(bq / Nq);
        S col = 0;
        const int LP = L * P;
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            const S *vl = value + ((int64_t)b * Nv + lsi[l]) * M * D + (int64_t)m * D + c;
            for (int pp = 0; pp < P; ++pp) {
                const int s = l * P + pp;
                const S x = loc[(row * LP + s) * 2] * (S)W - (S)0.5;
                const S y = loc[(row * LP + s) * 2 + 1] * (S)H - (S)0.5;
                const S a = aw[row * LP + s];
                if (!(y > (S)-1 && x > (S)-1 && y < (S)H && x < (S)W)) continue;
                const S fy = floor(y), fx = floor(x);
                const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
                const S ly = y - fy, lx = x - fx, hy = (S)1 - ly, hx = (S)1 - lx;
                const int64_t sp = (int64_t)M * D;
                S v = 0;
                if (y0 >= 0 && x0 >= 0) v += hy * hx * vl[((int64_t)y0 * W + x0) * sp];
                if (y0 >= 0 && x1 <= W - 1) v += hy * lx * vl[((int64_t)y0 * W + x1) * sp];
                if (y1 <= H - 1 && x0 >= 0) v += ly * hx * vl[((int64_t)y1 * W + x0) * sp];
                if (y1 <= H - 1 && x1 <= W - 1) v += ly * lx * vl[((int64_t)y1 * W + x1) * sp];
                col += v * a;
            }
        }
        out[idx] = col;
    }
}

template <typename VT, int D, bool HM, bool FUSED>
static int launch_gather(hipStream_t stream, GatherArgs &a)
{
    constexpr int G = D / ValTraits<VT>::kCpl;
    constexpr int GPB = kBlock / G;
    a.nchunk = (a.Nq + GPB - 1) / GPB;
    const int64_t blocks = (int64_t)a.B * a.nchunk * a.M;
    if (blocks == 0) return 0;
    if (blocks > 0x7fffffffLL) return fail("msda: grid too large");
    const size_t lds = (size_t)(GPB * (kChunk * 8 + 4) + kMaxLevels * 5) * 4 + kChunk;
    hipLaunchKernelGGL((msda_gather_kernel<VT, D, HM, FUSED>), dim3((unsigned)blocks), dim3(kBlock), lds, stream, a);
    note_forward_kernel(SDETR_KERNEL_MSDA_GATHER);
    return check_launch("msda_gather");
}

template <typename VT, bool HM, bool FUSED>
static int dispatch_d(hipStream_t stream, GatherArgs &a, int D)
{
    constexpr int CPL = ValTraits<VT>::kCpl;
    switch (D) {
        case 4: if (CPL == 4) return launch_gather<VT, (CPL == 4 ? 4 : 8), HM, FUSED>(stream, a); break;
        case 8: return launch_gather<VT, 8, HM, FUSED>(stream, a);
        case 16: return launch_gather<VT, 16, HM, FUSED>(stream, a);
        case 32: return launch_gather<VT, 32, HM, FUSED>(stream, a);
        case 64: return launch_gather<VT, 64, HM, FUSED>(stream, a);
        case 128: return launch_gather<VT, 128, HM, FUSED>(stream, a);
        default: break;
    }
    return fail("msda: head dim %d not supported by the tiled kernel for this value type", D);
}

static bool offsets_fit_32bit(int64_t Nv, int64_t pixel_bytes) { return Nv * pixel_bytes < 0xffffffffLL; }

}  // namespace sdetr

using namespace sdetr;

static int check_dims(int B, int Nv, int M, int D, int L, int Nq, int P)
{
    if (B < 0 || Nv < 0 || M <= 0 || D <= 0 || L <= 0 || Nq < 0 || P <= 0)
        return fail("msda: bad dims B=%d Nv=%d M=%d D=%d L=%d Nq=%d P=%d", B, Nv, M, D, L, Nq, P);
    return 0;
}

extern "C" int sdetr_msda_im2col_f32(sdetr_stream_t stream, const float *value, const int64_t *shapes,
                                     const int64_t *lsi, const float *loc, const float *aw, int B, int Nv, int M,
                                     int D, int L, int Nq, int P, float *out)
{
    if (int e = check_dims(B, Nv, M, D, L, Nq, P)) return e;
    if (!value || !shapes || !lsi || !loc || !aw || !out) return fail("msda_im2col_f32: null pointer");
    if ((int64_t)B * Nq == 0) return 0;
    const bool tiled = (D == 4 || D == 8 || D == 16 || D == 32 || D == 64 || D == 128) && L <= kMaxLevels &&
                       offsets_fit_32bit(Nv, (int64_t)M * D * 4);
    if (tiled) {
        GatherArgs a{};
        a.value = reinterpret_cast<const char *>(value);
        a.shapes = shapes; a.lsi = lsi; a.loc = loc; a.aw = aw; a.out = out; a.out_bf16 = 0;
        a.B = B; a.Nv = Nv; a.M = M; a.L = L; a.Nq = Nq; a.P = P;
        return dispatch_d<float, false, false>(stream, a, D);
    }
    const int64_t n = (int64_t)B * Nq * M * D;
    const int64_t blocks = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(msda_generic_kernel<float>, dim3((unsigned)(blocks > 65535 * 16 ? 65535 * 16 : blocks)),
                       dim3(kBlock), 0, stream, n, value, shapes, lsi, loc, aw, Nv, M, D, L, Nq, P, out);
    note_forward_kernel(SDETR_KERNEL_MSDA_GENERIC);
    return check_launch("msda_generic_f32");
}

extern "C" int sdetr_msda_im2col_f64(sdetr_stream_t stream, const double *value, const int64_t *shapes,
                                     const int64_t *lsi, const double *loc, const double *aw, int B, int Nv, int M,
                                     int D, int L, int Nq, int P, double *out)
{
    if (int e = check_dims(B, Nv, M, D, L, Nq, P)) return e;
    if (!value || !shapes || !lsi || !loc || !aw || !out) return fail("msda_im2col_f64: null pointer");
    const int64_t n = (int64_t)B * Nq * M * D;
    if (n == 0) return 0;
    const int64_t blocks = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(msda_generic_kernel<double>, dim3((unsigned)(blocks > 65535 * 16 ? 65535 * 16 : blocks)),
                       dim3(kBlock), 0, stream, n, value, shapes, lsi, loc, aw, Nv, M, D, L, Nq, P, out);
    note_forward_kernel(SDETR_KERNEL_MSDA_GENERIC);
    return check_launch("msda_generic_f64");
}

extern "C" int sdetr_msda_forward_head_major(sdetr_stream_t stream, const void *value_hm, int value_dtype,
                                             const int64_t *shapes, const int64_t *lsi, const float *loc,
                                             const float *aw, int B, int Nv, int M, int D, int L, int Nq, int P,
                                             void *out, int out_dtype)
{
    if (int e = check_dims(B, Nv, M, D, L, Nq, P)) return e;
    if (!value_hm || !shapes || !lsi || !loc || !aw || !out) return fail("msda_forward_head_major: null pointer");
    if (L > kMaxLevels) return fail("msda_forward_head_major: at most %d levels", kMaxLevels);
    if ((int64_t)B * Nq == 0) return 0;
    GatherArgs a{};
    a.value = reinterpret_cast<const char *>(value_hm);
    a.shapes = shapes; a.lsi = lsi; a.loc = loc; a.aw = aw; a.out = out; a.out_bf16 = (out_dtype == kActCode);
    a.B = B; a.Nv = Nv; a.M = M; a.L = L; a.Nq = Nq; a.P = P;
    if (value_dtype == SDETR_F32) return dispatch_d<float, true, false>(stream, a, D);
    if (value_dtype == SDETR_BF16) return dispatch_d<bf16_t, true, false>(stream, a, D);
    if (value_dtype == SDETR_F16) return dispatch_d<half_t, true, false>(stream, a, D);
    return fail("msda_forward_head_major: bad value dtype %d", value_dtype);
}

extern "C" int sdetr_msda_fused_forward(sdetr_stream_t stream, const void *value_hm, int value_dtype,
                                        const int64_t *shapes, const int64_t *lsi, const float *ref, int ref_dim,
                                        int64_t ref_batch_stride, const void *proj, int proj_dtype,
                                        int64_t proj_row_stride, int proj_head_major, const int32_t *order, int B,
                                        int Nv, int M, int D, int L, int Nq, int P, void *out, int out_dtype)
{
    if (int e = check_dims(B, Nv, M, D, L, Nq, P)) return e;
    if (ref_batch_stride == 0) ref_batch_stride = (int64_t)Nq * L * ref_dim;
    if (ref_batch_stride < (int64_t)Nq * L * ref_dim || (ref_batch_stride % ref_dim))
        return fail("msda_fused_forward: bad reference point batch stride");
    if (!value_hm || !shapes || !lsi || !ref || !proj || !out) return fail("msda_fused_forward: null pointer");
    if (ref_dim != 2 && ref_dim != 4)
        return fail("Last dim of reference_points must be 2 or 4, but get %d instead.", ref_dim);
    if (L > kMaxLevels) return fail("msda_fused_forward: at most %d levels", kMaxLevels);
    if (!proj_head_major && proj_row_stride < (int64_t)M * L * P * 3)
        return fail("msda_fused_forward: proj row stride too small");
    if (proj_head_major && !(D == 32 && L == 4 && P == 4 && proj_dtype == kActCode &&
                             (value_dtype == SDETR_BF16 || value_dtype == SDETR_F16)))
        return fail("msda_fused_forward: the head-major projection layout is built for D=32, L=P=4, bf16 projections "
                    "and 16-bit value maps");
    if ((int64_t)B * Nq == 0) return 0;
    GatherArgs a{};
    a.value = reinterpret_cast<const char *>(value_hm);
    a.shapes = shapes; a.lsi = lsi; a.ref = ref; a.ref_dim = ref_dim; a.ref_batch_stride = ref_batch_stride;
    a.proj = proj; a.proj_bf16 = (proj_dtype == kActCode); a.proj_stride = proj_row_stride; a.order = order;
    a.proj_hm = proj_head_major ? 1 : 0;
    a.out = out; a.out_bf16 = (out_dtype == kActCode);
    a.B = B; a.Nv = Nv; a.M = M; a.L = L; a.Nq = Nq; a.P = P;
    const bool l4p4 = D == 32 && L == 4 && P == 4 && (proj_head_major || (proj_row_stride % 8) == 0) &&
                      (reinterpret_cast<uintptr_t>(proj) % 16) == 0 && (reinterpret_cast<uintptr_t>(ref) % 16) == 0;
    if (l4p4 && value_dtype == SDETR_BF16) return launch_gather_l4p4<bf16_t>(stream, a);
    if (l4p4 && value_dtype == SDETR_F16) return launch_gather_l4p4<half_t>(stream, a);
    if (value_dtype == SDETR_F32) return dispatch_d<float, true, true>(stream, a, D);
    if (value_dtype == SDETR_BF16) return dispatch_d<bf16_t, true, true>(stream, a, D);
    if (value_dtype == SDETR_F16) return dispatch_d<half_t, true, true>(stream, a, D);
    return fail("msda_fused_forward: bad value dtype %d", value_dtype);
}
