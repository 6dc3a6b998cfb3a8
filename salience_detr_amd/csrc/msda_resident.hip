// Multi-scale deformable attention forward with the two COARSE levels of the value pyramid resident in LDS
// (gfx950, the Salience-DETR shape: head-major 16-bit value, 32 channels per head, 4 levels, 4 points).
//
// Semantics: models/bricks/ops/cuda/ms_deform_im2col_cuda.cuh:22-73, 226-288 (bilinear gather-reduce) fused with
// models/bricks/ms_deform_attn.py:322-355 (softmax over the 16 logits, sampling locations from the reference
// points and the projected offsets), exactly as msda_gather_l4p4_kernel (msda_forward.hip) computes them.
//
// Why this shape.  The direct gather moves 16 samples x 4 corners x 64 bytes per (query, head) through the vector
// memory path: 745 MB per launch at encoder layer 0 (2 x 11 363 queries), 19 us at the L1's 64 B/clk/CU before any
// arithmetic -- for 53 MB of algorithmic bytes.  Half of those samples fall on levels 2 and 3, whose maps are small:
// (25x42 + 13x21) pixels x 64 B = 85 KB per (image, head).  So:
//
//  * one 1024-thread workgroup per CU, persistent over a chunk of one (image, head)'s queries.  It copies that
//    head's level-2 and level-3 maps into LDS ONCE (coalesced 16-byte loads; 85 KB of the CU's 160 KB), then
//    every sample of those levels is four ds_read_b128 instead of four L1 line fetches: the vector memory path
//    carries only the level-0 / level-1 samples (half the bytes), the LDS (128-256 B/clk/CU) the other half, and
//    the two pipes run side by side;
//  * no spatial bucketing of the queries, no windows, no fallback path: the coarse maps are resident as a whole,
//    so any sampling location is served (the LDS-windowed kernel of round 1 needed all three);
//  * blockIdx % 8 = head, and block b runs on XCD b % 8: an XCD's L2 only ever sees its own head's slabs;
//  * a quad of lanes owns one (query, head) row, lane j of the quad owns level j's four points for the set-up
//    (softmax across the quad by two DPP shuffles) and 8 of the 32 channels for the gather.  The four bilinear x
//    attention weights of a sample travel through a wave-private LDS table (one ds_read_b128 per sample, all 16
//    quads in one conflict-free 256-byte row); the two row offsets of a sample stay in the owner lane's registers
//    and reach the other three lanes through the DPP operand of the address add they need anyway;
//  * a sample's two x-neighbours are ONE 128-byte span (pixel xa and xa+1 of a head-major row), addressed by the
//    instruction's immediate offset: two address computations per sample instead of four.  At the left border the
//    span starts at pixel 0 and the weights shift over; at the right border the second pixel has weight zero (it
//    is the next row's first pixel, or the 64-byte zero pad after the resident slab).
#include "common.h"

namespace sdetr {

void note_forward_kernel(int which);  // abi.hip

constexpr int kRWaves = 16;                      // waves per workgroup, one workgroup per CU
constexpr int kRThreads = kRWaves * kWave;       // 1024
constexpr int kRWeightBytes = 16 * 16 * 16;      // per wave: [sample 16][row 16] float4
constexpr int kRPad = 64;                        // zeroed pixel before and after the resident slab
constexpr int kRLdsBudget = 160 * 1024;
constexpr int kRMaxResidentPx = (kRLdsBudget - kRWaves * kRWeightBytes - 2 * kRPad) / 64;  // 1534 pixels

struct ResidentArgs {
    const char *value;        // [B, M, Nv, 32] fp16 | bf16
    const float *ref;         // [B, Nq, 4, ref_dim]
    int64_t ref_batch_stride; // floats between images
    int ref_dim;
    const bf16_t *proj;       // [B, M, Nq, 48] bf16: 32 offsets (x,y per level, point) then 16 logits
    void *out;                // [B, Nq, M*32]
    int out_bf16;
    int B, Nv, M, Nq;
    int H0, W0, H1, W1, H2, W2, H3, W3;   // host copy of the level shapes
    int S1, S2, S3;                       // level start pixels (S0 = 0)
    int image_serial;                     // G > 0: a workgroup = (image lane g < G, head, chunk) and walks the images g, g + G, ...;
                                          // 0: a workgroup = (image, head, chunk)
    int res_start;                        // first resident pixel: S2 (levels 2 + 3 resident) or S3 (level 3 only)
    int res_px;                           // Nv - res_start: pixels resident in LDS
    int chunks;                           // workgroups per (image, head)
};

template <typename VT>
struct ResFma;
template <>
struct ResFma<half_t> {
    __device__ static __forceinline__ void fma8(float *acc, const uint4 &v, float w)
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
template <>
struct ResFma<bf16_t> {
    __device__ static __forceinline__ void fma8(float *acc, const uint4 &v, float w)
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

// value of lane J of this lane's quad (DPP quad_perm broadcast; folds into the consuming VALU instruction)
template <int J>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, J * 0x55, 0xf, 0xf, false);
}

// clamp as ONE instruction (hipcc emits v_max + v_min: it cannot prove lo <= hi for run-time bounds)
__device__ __forceinline__ int med3_i32(int v, int lo, int hi)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi));
    return r;
}

__device__ __forceinline__ float mul_f32(float a, float b)
{
    float r;
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float quad_xor1(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));  // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float quad_xor2(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));  // quad_perm [2,3,0,1]
}

// RES = number of resident levels: 2 (levels 2 + 3, the benchmark pyramid: 8 of a row's 16 samples come from LDS) or 1
// (level 3 only, for pyramids whose two coarse levels together exceed the LDS -- the reference's 5scale configuration,
// configs/salience_detr/salience_detr_resnet50_5scale_800_1333.py:33-36: 4 of the 16 samples).
template <typename VT, bool REF4, int RES = 2, bool SERIAL = false>
__global__ void __launch_bounds__(kRThreads) msda_resident_kernel(ResidentArgs p)
{
    using F = ResFma<VT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // [64 B zeros][resident slab: levels 2, 3][64 B zeros][weights: kRWaves x 4 KB]
    const int slab_bytes = p.res_px * 64;
    float4 *wts = reinterpret_cast<float4 *>(lds + 2 * kRPad + slab_bytes);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = blockIdx.x % p.M;
    const int rest = blockIdx.x / p.M;
    // one image per workgroup (the workgroups of all images side by side), or -- large batches, `image_serial` -- every
    // workgroup walks ALL images for its (head, chunk of the queries): the chip then works on one image at a time and
    // that image's maps (23 MB) stay in the L2s / the Infinity Cache while they are gathered from; with sixteen images
    // in flight at once their 366 MB of maps evict each other from the 256 MB Infinity Cache (0.14-0.15 of the roofline)
    // (SERIAL is a template parameter: the one-image form keeps its register allocation -- exactly 128, no spills)
    const int b_first = rest / p.chunks;                       // SERIAL: the workgroup's image lane (0 .. image_serial - 1)
    const int b_end = SERIAL ? p.B : b_first + 1;
    const int b_step = SERIAL ? p.image_serial : 1;
    const int chunk = rest - b_first * p.chunks;
    const int rows_per_chunk = (p.Nq + p.chunks - 1) / p.chunks;
    const int q_lo = chunk * rows_per_chunk;
    const int q_hi = min(p.Nq, q_lo + rows_per_chunk);
    if (q_lo >= q_hi) return;  // workgroup-uniform

    for (int b = b_first; b < b_end; b += b_step) {
    const char *base = p.value + ((int64_t)b * p.M + m) * p.Nv * 64;

    // ---- stage levels 2 and 3 of this (image, head) with LDS-DMA (global_load_lds_dwordx4: no registers, no LDS store
    // instructions): wave w copies the 1-KB pieces w, w + 16, ... of the slab; the copies fly while the first row
    // group's inputs are loaded, its locations and weights are set up and its first level-0 / level-1 samples are
    // requested -- the wait for them sits in front of the first LDS read of a map (first iteration of the loop below).
    // The 64-byte zero pads in front of and behind the slab are plain stores. ----
    {
        const char *src = base + (int64_t)p.res_start * 64;
        const uint32_t lds0 = (uint32_t)(uint64_t)(lds + kRPad);
        const int pieces = (slab_bytes + 1023) >> 10;
        for (int piece = wave; piece < pieces; piece += kRWaves) {
            const uint32_t goff = (uint32_t)piece * 1024u + (uint32_t)lane * 16u;
            if (goff < (uint32_t)slab_bytes) {   // (the last piece is partial: its idle lanes copy nothing)
                const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)piece * 1024u);
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %0, %2"
                             :
                             : "v"(goff), "s"(m0v), "s"(src)
                             : "memory", "m0");
            }
        }
        if (tid < 4) *reinterpret_cast<uint4 *>(lds + tid * 16) = make_uint4(0u, 0u, 0u, 0u);
        else if (tid < 8) *reinterpret_cast<uint4 *>(lds + kRPad + slab_bytes + (tid - 4) * 16) = make_uint4(0u, 0u, 0u, 0u);
    }

    // ---- rows: a quad per (query, head); lane j of the quad owns level j ----
    const int g = lane >> 2, j = lane & 3;
    const int H = j == 0 ? p.H0 : j == 1 ? p.H1 : j == 2 ? p.H2 : p.H3;
    const int W = j == 0 ? p.W0 : j == 1 ? p.W1 : j == 2 ? p.W2 : p.W3;
    const int S = j == 0 ? 0 : j == 1 ? p.S1 : j == 2 ? p.S2 : p.S3;
    const float fH = (float)H, fW = (float)W;
    const float invW = 1.0f / fW, invH = 1.0f / fH;
    // byte offset of this lane's level: into the head's global slab (levels 0, 1) or into `lds` (levels 2, 3)
    const uint32_t lvl_base = j < 4 - RES ? (uint32_t)S * 64u : (uint32_t)(kRPad + (S - p.res_start) * 64);
    const uint32_t lane_off = (uint32_t)(j * 16);
    const __amdgpu_buffer_rsrc_t rsrc = make_uniform_rsrc(base, (uint32_t)((int64_t)p.Nv * 64));
    // the head's block of the projection slab and the image's reference points through buffer resources too:
    // 32-bit lane offsets instead of 64-bit VALU address arithmetic
    constexpr int RD = REF4 ? 4 : 2;
    const __amdgpu_buffer_rsrc_t proj_rsrc = make_uniform_rsrc(
        reinterpret_cast<const char *>(p.proj + ((int64_t)b * p.M + m) * p.Nq * 48), (uint32_t)((int64_t)p.Nq * 96));
    const __amdgpu_buffer_rsrc_t ref_rsrc = make_uniform_rsrc(
        reinterpret_cast<const char *>(p.ref + (int64_t)b * p.ref_batch_stride), (uint32_t)((int64_t)p.Nq * 4 * RD * 4));
    float4 *myW = wts + wave * 256;

    // raw inputs of a row: my level's 4 offsets (x,y), 4 logits, reference point -- loaded one row group ahead
    struct RowIn {
        uint4 o;
        uint2 gg;
        float r[RD];
    };
    auto load_row = [&](int rg) {
        const uint32_t slot = (uint32_t)min(q_lo + rg * 16 + g, q_hi - 1);
        RowIn in;
        in.o = buffer_load16(proj_rsrc, slot * 96u + (uint32_t)j * 16u);
        in.gg = buffer_load8(proj_rsrc, slot * 96u + 64u + (uint32_t)j * 8u);
        if (REF4) {
            const uint4 r = buffer_load16(ref_rsrc, (slot * 4u + (uint32_t)j) * 16u);
            in.r[0] = __uint_as_float(r.x); in.r[1] = __uint_as_float(r.y);
            in.r[RD - 2] = __uint_as_float(r.z); in.r[RD - 1] = __uint_as_float(r.w);
        } else {
            const uint2 r = buffer_load8(ref_rsrc, (slot * 4u + (uint32_t)j) * 8u);
            in.r[0] = __uint_as_float(r.x); in.r[1] = __uint_as_float(r.y);
        }
        return in;
    };
    const int ngroups = (q_hi - q_lo + 15) >> 4;
    RowIn nxt = load_row(min(wave, ngroups - 1));
    bool maps_pending = true;   // the staged maps have not been waited for yet (wave-uniform)
    if (wave >= ngroups) {
        // no row group for this wave: its pieces of the maps still have to land before the others read them
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    for (int rg = wave; rg < ngroups; rg += kRWaves) {
        const int slot = q_lo + rg * 16 + g;
        const bool active = slot < q_hi;
        const int q = active ? slot : q_hi - 1;
        const RowIn cur = nxt;
        nxt = load_row(min(rg + kRWaves, ngroups - 1));
        float ox[4], oy[4], lg[4];
        ox[0] = act_lo(cur.o.x); oy[0] = act_hi(cur.o.x); ox[1] = act_lo(cur.o.y); oy[1] = act_hi(cur.o.y);
        ox[2] = act_lo(cur.o.z); oy[2] = act_hi(cur.o.z); ox[3] = act_lo(cur.o.w); oy[3] = act_hi(cur.o.w);
        lg[0] = act_lo(cur.gg.x); lg[1] = act_hi(cur.gg.x); lg[2] = act_lo(cur.gg.y); lg[3] = act_hi(cur.gg.y);
        // softmax over the quad's 16 logits (quad_perm DPP: lane ^ 1, lane ^ 2 -- no LDS round trip)
        float mx = fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3]));
        mx = fmaxf(mx, quad_xor1(mx));
        mx = fmaxf(mx, quad_xor2(mx));
        float e[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) e[t] = __expf(lg[t] - mx);
        float sum = (e[0] + e[1]) + (e[2] + e[3]);
        sum += quad_xor1(sum);
        sum += quad_xor2(sum);
        const float inv = active ? __builtin_amdgcn_rcpf(sum) : 0.f;  // inactive rows: all weights zero

        // ---- my four samples: two row offsets (kept in registers) + four weights (to the wave's LDS table) ----
        uint32_t r0[4], r1[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float x, y;
            if (!REF4) {
                x = fmaf(ox[t], invW, cur.r[0]);
                y = fmaf(oy[t], invH, cur.r[1]);
            } else {
                x = fmaf(ox[t] * 0.125f, cur.r[RD - 2], cur.r[0]);  // offset / num_points * w * 0.5, num_points = 4
                y = fmaf(oy[t] * 0.125f, cur.r[RD - 1], cur.r[1]);
            }
            // Pixel coordinates, clamped to [-2, size + 1] by one v_med3_f32 each: every position the reference's
            // `h_im > -1 && w_im > -1 && h_im < H && w_im < W` test rejects lands on rows / columns that fail the
            // corner tests below anyway (-2 -> corners -2 and -1; size + 1 -> corners size + 1, size + 2), a NaN
            // becomes -2 (v_med3 returns the minimum when an operand is NaN), and the int conversion never saturates --
            // four compares, three scalar ANDs and a select per sample less than the explicit test.
            const float h_im = __builtin_amdgcn_fmed3f(fmaf(y, fH, -0.5f), -2.f, fH + 1.f);
            const float w_im = __builtin_amdgcn_fmed3f(fmaf(x, fW, -0.5f), -2.f, fW + 1.f);
            const float a = e[t] * inv;
            const float fy = floorf(h_im), fx = floorf(w_im);
            const float ly = h_im - fy, lx = w_im - fx;
            const int y0 = (int)fy, x0 = (int)fx;
            const bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)(y0 + 1) < (unsigned)H;
            const bool in0 = (unsigned)x0 < (unsigned)W, in1 = (unsigned)(x0 + 1) < (unsigned)W;
            const float wy0 = vy0 ? (1.f - ly) * a : 0.f;
            const float wy1 = vy1 ? ly * a : 0.f;
            // the span starts at pixel xa = clamp(x0): when x0 = -1 that pixel is the RIGHT corner
            const float wa = in0 ? (1.f - lx) : (in1 ? lx : 0.f);
            const float wb = (in0 & in1) ? lx : 0.f;
            const int y0c = med3_i32(y0, 0, H - 1), y1c = med3_i32(y0 + 1, 0, H - 1);
            const int xa = med3_i32(x0, 0, W - 1);
            r0[t] = lvl_base + (__umul24((uint32_t)y0c, (uint32_t)W) + (uint32_t)xa) * 64u;  // 24-bit operands: full rate
            r1[t] = lvl_base + (__umul24((uint32_t)y1c, (uint32_t)W) + (uint32_t)xa) * 64u;
            // (plain v_mul_f32 each: hipcc otherwise forms v_pk_mul_f32 pairs and pays for them in register shuffles)
            myW[(j * 4 + t) * 16 + g] = make_float4(mul_f32(wy0, wa), mul_f32(wy0, wb), mul_f32(wy1, wa), mul_f32(wy1, wb));
        }
        // the table is wave-private: visible to this wave's reads once its own LDS queue drains
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();

        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = 0.f;

        // Rolling schedule over the 8 global samples (levels 0, 1) and the 8 resident samples (levels 2, 3): four
        // global samples (16 loads) are in flight from the start; step k accumulates resident sample k (four
        // ds_read_b128), then global sample k, then issues global sample k + 4 into the registers just freed --
        // 12 to 16 loads stay in flight for the whole row group.  The scheduling fences keep hipcc from sinking
        // every load down to its first use (it then has two in flight).
#define SDETR_RES_ISSUE(SLOT, JL, T)                                                                           \
    {                                                                                                          \
        const uint32_t o0 = quad_bcast<JL>(r0[T]) + lane_off, o1 = quad_bcast<JL>(r1[T]) + lane_off;           \
        va[SLOT][0] = buffer_load16(rsrc, o0);                                                                 \
        va[SLOT][1] = buffer_load16(rsrc, o0 + 64u);                                                           \
        va[SLOT][2] = buffer_load16(rsrc, o1);                                                                 \
        va[SLOT][3] = buffer_load16(rsrc, o1 + 64u);                                                           \
    }
#define SDETR_RES_ACC(SLOT, JL, T)                                                                             \
    {                                                                                                          \
        const float4 w = myW[(JL * 4 + T) * 16 + g];                                                           \
        F::fma8(acc, va[SLOT][0], w.x);                                                                        \
        F::fma8(acc, va[SLOT][1], w.y);                                                                        \
        F::fma8(acc, va[SLOT][2], w.z);                                                                        \
        F::fma8(acc, va[SLOT][3], w.w);                                                                        \
    }
#define SDETR_RES_LDS(JL, T)                                                                                   \
    {                                                                                                          \
        const uint32_t o0 = quad_bcast<JL>(r0[T]) + lane_off, o1 = quad_bcast<JL>(r1[T]) + lane_off;           \
        const uint4 v0 = *reinterpret_cast<const uint4 *>(lds + o0);                                           \
        const uint4 v1 = *reinterpret_cast<const uint4 *>(lds + o0 + 64);                                      \
        const uint4 v2 = *reinterpret_cast<const uint4 *>(lds + o1);                                           \
        const uint4 v3 = *reinterpret_cast<const uint4 *>(lds + o1 + 64);                                      \
        const float4 w = myW[(JL * 4 + T) * 16 + g];                                                           \
        F::fma8(acc, v0, w.x);                                                                                 \
        F::fma8(acc, v1, w.y);                                                                                 \
        F::fma8(acc, v2, w.z);                                                                                 \
        F::fma8(acc, v3, w.w);                                                                                 \
    }
#define SDETR_FENCE __builtin_amdgcn_sched_barrier(0);
        uint4 va[4][4];
        SDETR_RES_ISSUE(0, 0, 0) SDETR_RES_ISSUE(1, 0, 1) SDETR_RES_ISSUE(2, 0, 2) SDETR_RES_ISSUE(3, 0, 3) SDETR_FENCE
        if (maps_pending) {
            // first row group of this wave: everything it could do without the resident maps is under way.  Loads
            // complete in order, so an empty queue means this wave's copies have landed; the barrier does the same for
            // the other waves' pieces (waves without a row group have exited: the barrier does not count them).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            maps_pending = false;
        }
        SDETR_FENCE
        if (RES == 2) {
            SDETR_RES_LDS(2, 0) SDETR_FENCE SDETR_RES_ACC(0, 0, 0) SDETR_FENCE SDETR_RES_ISSUE(0, 1, 0) SDETR_FENCE
            SDETR_RES_LDS(2, 1) SDETR_FENCE SDETR_RES_ACC(1, 0, 1) SDETR_FENCE SDETR_RES_ISSUE(1, 1, 1) SDETR_FENCE
            SDETR_RES_LDS(2, 2) SDETR_FENCE SDETR_RES_ACC(2, 0, 2) SDETR_FENCE SDETR_RES_ISSUE(2, 1, 2) SDETR_FENCE
            SDETR_RES_LDS(2, 3) SDETR_FENCE SDETR_RES_ACC(3, 0, 3) SDETR_FENCE SDETR_RES_ISSUE(3, 1, 3) SDETR_FENCE
            SDETR_RES_LDS(3, 0) SDETR_FENCE SDETR_RES_ACC(0, 1, 0) SDETR_FENCE
            SDETR_RES_LDS(3, 1) SDETR_FENCE SDETR_RES_ACC(1, 1, 1) SDETR_FENCE
            SDETR_RES_LDS(3, 2) SDETR_FENCE SDETR_RES_ACC(2, 1, 2) SDETR_FENCE
            SDETR_RES_LDS(3, 3) SDETR_FENCE SDETR_RES_ACC(3, 1, 3) SDETR_FENCE
        } else {
            // level 3 from LDS, levels 0-2 through the L1: twelve global samples rolling through the four slots
            SDETR_RES_LDS(3, 0) SDETR_FENCE SDETR_RES_ACC(0, 0, 0) SDETR_FENCE SDETR_RES_ISSUE(0, 1, 0) SDETR_FENCE
            SDETR_RES_LDS(3, 1) SDETR_FENCE SDETR_RES_ACC(1, 0, 1) SDETR_FENCE SDETR_RES_ISSUE(1, 1, 1) SDETR_FENCE
            SDETR_RES_LDS(3, 2) SDETR_FENCE SDETR_RES_ACC(2, 0, 2) SDETR_FENCE SDETR_RES_ISSUE(2, 1, 2) SDETR_FENCE
            SDETR_RES_LDS(3, 3) SDETR_FENCE SDETR_RES_ACC(3, 0, 3) SDETR_FENCE SDETR_RES_ISSUE(3, 1, 3) SDETR_FENCE
            SDETR_RES_ACC(0, 1, 0) SDETR_FENCE SDETR_RES_ISSUE(0, 2, 0) SDETR_FENCE
            SDETR_RES_ACC(1, 1, 1) SDETR_FENCE SDETR_RES_ISSUE(1, 2, 1) SDETR_FENCE
            SDETR_RES_ACC(2, 1, 2) SDETR_FENCE SDETR_RES_ISSUE(2, 2, 2) SDETR_FENCE
            SDETR_RES_ACC(3, 1, 3) SDETR_FENCE SDETR_RES_ISSUE(3, 2, 3) SDETR_FENCE
            SDETR_RES_ACC(0, 2, 0) SDETR_FENCE SDETR_RES_ACC(1, 2, 1) SDETR_FENCE
            SDETR_RES_ACC(2, 2, 2) SDETR_FENCE SDETR_RES_ACC(3, 2, 3) SDETR_FENCE
        }
#undef SDETR_FENCE
#undef SDETR_RES_ISSUE
#undef SDETR_RES_ACC
#undef SDETR_RES_LDS

        if (active) {
            const int64_t o = (((int64_t)b * p.Nq + q) * p.M + m) * 32 + j * 8;
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
        // the next row group rewrites the weight table: this wave's reads of it have all returned (their values
        // were consumed above), and LDS operations of one wave complete in order
    }
    if (SERIAL && b + b_step < b_end) __syncthreads();   // every wave is done with this image's maps: the next image's take their place
    }   // images
}


// ---------------------------------------------------------------------------------------------------------------------
// Round 4: the same gather on BORDERED maps, with an optional per-launch ROW ORDER.
//
// Where the round-3 kernel spends its vector-ALU issue slots (rocprofv3 SQ_INSTS_VALU: 829 wave instructions per
// 16-row group): 512 v_fma_mix_f32 are the gather itself, ~150 are the per-sample bounds logic -- four corner tests,
// the selects that zero a weight or shift it over at the left border, three index clamps, two multiply-adds per
// sample for the row offsets.  None of it is needed when every level of the head-major map carries a BORDER OF ZERO
// RECORDS: level l is stored as (H_l + 2) rows of (W_l + 1) records -- row -1 and row H_l are zero, record -1 of every
// row is zero and doubles as record W_l of the row above (one shared column) -- the levels follow each other and one
// more zero record closes the map (Np = sum (H_l + 2)(W_l + 1) + 1 records per (image, head), +3.7 % at the benchmark
// pyramid).  A sampling position is clamped to [-1, size) by ONE v_med3_f32 per coordinate (a NaN lands on -1) and
// then every corner a lane can address exists and outside corners are zeros: floor / fraction / index are
// v_cvt_u32_f32, v_fract_f32 and one v_mad_u32_u24, the four weights come from the fractions by three multiplies and
// three subtractions.  (Positions in [size - 2^-17, size) read weight 2^-17 x the border pixel instead of nothing: the
// reference's own value there differs from zero by more, ms_deform_im2col_cuda.cuh:22-73 is continuous at the border.)
// The second row of a sample is the first one's address plus the level's row pitch: for the two fine levels that is the
// SCALAR offset operand of the buffer load -- no vector instruction.  With the packed-f32 weight split, the loop
// unrolled over two input sets (no register copies between row groups) and the first product of a row starting the
// accumulators, the loop holds 651 vector instructions per row group (784 in the round-3 kernel; counted from the
// device assembly by benchmarks/tools/asm_loop_count.py).
//
// What the measurements of this round say about the kernel (benchmarks/msda_real_operands.py on the step's own operands,
// ablation builds `ABL`, per-workgroup phase stamps):
//  * with the rows in list (score) order the loop is bound by the fine levels' misses in the 32 KB L1 -- fewer vector
//    instructions bought nothing there (round 3's packed-fp16 experiment, this kernel in list order: 29.4 vs 29.7 us);
//  * with the rows in tile order it is bound by vector-ALU ISSUE: +128 dummy instructions per group cost +2.7 us, removing
//    the fine-level loads only 3.8, the LDS reads 0.7; the 512 v_fma_mix are 12 of the 17.5 us the loop takes at layer 0;
//  * four fine-level samples (16 loads) in flight per wave were too many: 16 waves x 16 KB = 8 x the L1 -- ONE is faster;
//  * ~6 us of every launch pass before the first resident-level sample can be read: the four waves of a SIMD run their
//    first row group's set-up one after the other (vector-ALU bound as well); LDS-DMA staging (25 GB/s per CU whatever
//    the number of issuing waves) would add to it, so the records travel through registers.
//
// Row order (PERM): the encoder hands its rows over sorted by salience score, so the 256 rows a workgroup has in flight
// are scattered over the image and the fine-level records they fetch miss the 32 KB L1 (the L2 -> L1 leg: 64-byte
// records out of 128-byte lines at ~31 B/clk).  `perm` [B, Nq] lists the rows in an order that
// keeps neighbours in the image together (tile-major, built once per step next to the merge of the level results): the
// kernel walks perm instead of 0..Nq-1 -- one extra 4-byte load per row, two row groups ahead -- reads the
// projection / reference point of row perm[i] and writes its output row; nothing else changes, any permutation gives
// the same result.
struct BorderedArgs {
    const char *value;        // [B, M, Np, 32] fp16, bordered layout
    const float *ref;         // [B, Nq, 4, ref_dim]
    int64_t ref_batch_stride; // floats between images
    int ref_dim;
    const bf16_t *proj;       // [B, M, Nq, 48] bf16: 32 offsets (x,y per level, point) then 16 logits
    const int32_t *perm;      // [B, Nq] row order, or NULL
    int64_t perm_batch_stride;
    void *out;                // [B, Nq, M*32]
    int out_bf16;
    int B, Np, M, Nq;
    int H0, W0, H1, W1, H2, W2, H3, W3;   // host copy of the level shapes
    int P1, P2, P3;                       // first record of levels 1..3 (level 0 starts at record 0)
    int image_serial;
    int res_start;                        // first resident record: P2 (levels 2 + 3 resident) or P3 (level 3 only)
    int res_px;                           // Np - res_start: records resident in LDS
    int chunks;                           // workgroups per (image, head)
    int prefetch_fine;                    // touch this workgroup's share of the fine levels' lines first (L2 warm-up)
    unsigned long long *stamps;           // benchmarks only (ABL & 32): [workgroup][8] wall-clock stamps (100 MHz)
};

constexpr int kBMaxResidentPx = (kRLdsBudget - kRWaves * kRWeightBytes) / 64;   // 1536 records
// corner accumulation of the bordered kernel for 16-bit outputs: 0 exact fp32 products, 1 / 2 packed fp16 (round 5, see
// the PK helpers in front of the kernel); SDETR_MSDA_PK overrides it for A/B runs
// (the fp16-activation flavour keeps the exact products: its outputs carry three more mantissa bits than bf16 ones and
// the packed form's roundings would be the larger part of their error)
// bf16 outputs: PK = 2 -- measured in the step (rocprofv3, six launches): 122.2 us exact, 118.5 PK = 1, 116.3 PK = 2, with
// the end-to-end distance from the reference's fp32 run unchanged (tests/test_encoder_timed_mode_gpu.py: non-flipped
// mean 0.0084 / 0.0084 / 0.0081 against the reference's own bf16 autocast at 0.0077; profiles/r05_msda_pk_ab.json)
constexpr int kDefaultPackedAccumulate = SDETR_ACT_IS_F16 ? 0 : 2;
constexpr int kBPieces = (kBMaxResidentPx * 64 + kRWaves * 1024 - 1) / (kRWaves * 1024);   // copy instructions per wave: 6

__device__ __forceinline__ uint4 buffer_load16_s(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, uint32_t soff)
{
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, (int)soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
template <int AUX>
__device__ __forceinline__ uint4 buffer_load16_aux(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, uint32_t soff)
{
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, (int)soff, AUX);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t buffer_load4(__amdgpu_buffer_rsrc_t r, uint32_t byte_off)
{
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0);
}
__device__ __forceinline__ float sub_f32(float a, float b)
{
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint4 lds_read16(uint32_t lds_addr)
{
    const u32x4_t v = *reinterpret_cast<const __attribute__((address_space(3))) u32x4_t *>(lds_addr);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float max3_f32(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float max_f32(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float max_dpp_xor1(float v)
{
    float r;
    asm("v_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ float max_dpp_xor2(float v)
{
    float r;
    asm("v_max_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    return r;
}
// (y * pitch_bytes + base) and ((x << 6) + t) as one instruction each (hipcc picks mul + add_lshl + add)
__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t lshl6_add(uint32_t a, uint32_t c)
{
    uint32_t r;
    asm("v_lshl_add_u32 %0, %1, 6, %2" : "=v"(r) : "v"(a), "v"(c));
    return r;
}
// the first product of a row's accumulators: acc = f32(half of `packed`) * w (no zero fill in front of the chain)
__device__ __forceinline__ float mul_f16lo(uint32_t packed, float w)
{
    float acc;
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(acc) : "v"(packed), "v"(w));
    return acc;
}
__device__ __forceinline__ float mul_f16hi(uint32_t packed, float w)
{
    float acc;
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(acc) : "v"(packed), "v"(w));
    return acc;
}
__device__ __forceinline__ void mul8_f16(float *acc, const uint4 &v, float w)
{
    acc[0] = mul_f16lo(v.x, w); acc[1] = mul_f16hi(v.x, w); acc[2] = mul_f16lo(v.y, w); acc[3] = mul_f16hi(v.y, w);
    acc[4] = mul_f16lo(v.z, w); acc[5] = mul_f16hi(v.z, w); acc[6] = mul_f16lo(v.w, w); acc[7] = mul_f16hi(v.w, w);
}
__device__ __forceinline__ void buffer_store16(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, const uint4 &v)
{
    const u32x4_t d = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(d, r, (int)byte_off, 0, 0);
}

// ---- round 5: packed-fp16 corner accumulation (template parameter PK of msda_bordered_kernel) ---------------------------
// The loop is bound by vector-ALU issue (section "What the measurements ... say" above): 512 of its ~650 instructions per
// 16-row group are one v_fma_mix_f32 per (corner, channel).  v_pk_fma_f16 issues at the same rate and does TWO channels:
//   PK = 1: the four corners of a sample are combined in packed fp16 with the plain bilinear weights (in [0, 1], rounded to
//           fp16), 16 instructions for a lane's 8 channels, and the sample enters the fp32 accumulators through 8
//           v_fma_mix_f32 with the fp32 attention weight: 24 instructions per sample instead of 32.  What is added to the
//           fp16 rounding the maps carry anyway: four fp16 roundings of a sample's interpolated value (2^-12 relative each).
//   PK = 2: the attention weight is folded into the fp16 corner weights and the four POINTS of a level are summed in
//           packed fp16 as well (16 instructions per sample), one conversion-add per level and channel into the fp32
//           accumulators: 18 instructions per sample.  Sixteen fp16 roundings per level sum.
// Both are for 16-bit outputs only (the fp32-output form keeps the exact fp32 products: it is the parity path).
typedef _Float16 f16x2_vec_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk_f16(float lo, float hi)
{
    const f32x2_vec_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_vec_t));   // round to nearest even
}
// r = v * (w.lo, w.lo) [+ c]  /  r = v * (w.hi, w.hi) + c: v_pk_mul_f16 / v_pk_fma_f16 with the broadcast in op_sel.  Vector
// builtins, not inline asm: hipcc pads every inline-asm result that the next inline asm consumes with an s_nop (217 in the
// loop of the PK = 2 form) and cannot interleave the four channel pairs' chains.
__device__ __forceinline__ uint32_t pk_mul_f16_blo(uint32_t v, uint32_t w)
{
    const f16x2_vec_t hv = __builtin_bit_cast(f16x2_vec_t, v), hw = __builtin_bit_cast(f16x2_vec_t, w);
    return __builtin_bit_cast(uint32_t, hv * __builtin_shufflevector(hw, hw, 0, 0));
}
__device__ __forceinline__ uint32_t pk_fma_f16_blo(uint32_t v, uint32_t w, uint32_t c)
{
    const f16x2_vec_t hv = __builtin_bit_cast(f16x2_vec_t, v), hw = __builtin_bit_cast(f16x2_vec_t, w);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(hv, __builtin_shufflevector(hw, hw, 0, 0),
                                                                  __builtin_bit_cast(f16x2_vec_t, c)));
}
__device__ __forceinline__ uint32_t pk_fma_f16_bhi(uint32_t v, uint32_t w, uint32_t c)
{
    const f16x2_vec_t hv = __builtin_bit_cast(f16x2_vec_t, v), hw = __builtin_bit_cast(f16x2_vec_t, w);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(hv, __builtin_shufflevector(hw, hw, 1, 1),
                                                                  __builtin_bit_cast(f16x2_vec_t, c)));
}
// f32(half of `packed`) + acc, and the plain conversion (the first level sum starts the accumulators)
__device__ __forceinline__ float add_f16lo(uint32_t packed, float acc)
{
    asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(packed));
    return acc;
}
__device__ __forceinline__ float add_f16hi(uint32_t packed, float acc)
{
    asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(packed));
    return acc;
}
__device__ __forceinline__ float cvt_f16lo(uint32_t packed)
{
    float r;
    asm("v_cvt_f32_f16 %0, %1" : "=v"(r) : "v"(packed));
    return r;
}
__device__ __forceinline__ float cvt_f16hi(uint32_t packed)
{
    float r;
    asm("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(r) : "v"(packed));
    return r;
}
// One sample's bilinear value for a lane's 8 channels (four registers of two): corners (y0,x0), (y0,x1), (y1,x0), (y1,x1) with
// the packed weights wa = (w(y0,x0), w(y1,x0)), wb = (w(y0,x1), w(y1,x1)).  START: the first product starts the chains, else it
// adds to h.  Written corner-major over the four registers: four INDEPENDENT chains side by side -- a packed instruction that
// consumes the result of the one right in front of it costs a wait state on gfx950 (hipcc pads it with an s_nop).
template <bool START>
__device__ __forceinline__ void pk_corners4(uint32_t *h, const uint4 &v00, const uint4 &v01, const uint4 &v10, const uint4 &v11,
                                            uint32_t wa, uint32_t wb)
{
    uint32_t t0, t1, t2, t3;
    if (START) {
        t0 = pk_mul_f16_blo(v00.x, wa); t1 = pk_mul_f16_blo(v00.y, wa); t2 = pk_mul_f16_blo(v00.z, wa); t3 = pk_mul_f16_blo(v00.w, wa);
    } else {
        t0 = pk_fma_f16_blo(v00.x, wa, h[0]); t1 = pk_fma_f16_blo(v00.y, wa, h[1]);
        t2 = pk_fma_f16_blo(v00.z, wa, h[2]); t3 = pk_fma_f16_blo(v00.w, wa, h[3]);
    }
    t0 = pk_fma_f16_blo(v01.x, wb, t0); t1 = pk_fma_f16_blo(v01.y, wb, t1); t2 = pk_fma_f16_blo(v01.z, wb, t2); t3 = pk_fma_f16_blo(v01.w, wb, t3);
    t0 = pk_fma_f16_bhi(v10.x, wa, t0); t1 = pk_fma_f16_bhi(v10.y, wa, t1); t2 = pk_fma_f16_bhi(v10.z, wa, t2); t3 = pk_fma_f16_bhi(v10.w, wa, t3);
    h[0] = pk_fma_f16_bhi(v11.x, wb, t0); h[1] = pk_fma_f16_bhi(v11.y, wb, t1);
    h[2] = pk_fma_f16_bhi(v11.z, wb, t2); h[3] = pk_fma_f16_bhi(v11.w, wb, t3);
}
// PK = 1: acc (+)= a * bilinear(sample)
template <bool FIRST>
__device__ __forceinline__ void pk1_sample(float *acc, const uint4 &v00, const uint4 &v01, const uint4 &v10, const uint4 &v11,
                                           uint32_t wa, uint32_t wb, float a)
{
    uint32_t t[4];
    pk_corners4<true>(t, v00, v01, v10, v11, wa, wb);
    if (FIRST) {
        acc[0] = mul_f16lo(t[0], a); acc[1] = mul_f16hi(t[0], a); acc[2] = mul_f16lo(t[1], a); acc[3] = mul_f16hi(t[1], a);
        acc[4] = mul_f16lo(t[2], a); acc[5] = mul_f16hi(t[2], a); acc[6] = mul_f16lo(t[3], a); acc[7] = mul_f16hi(t[3], a);
    } else {
        acc[0] = fma_f16lo(t[0], a, acc[0]); acc[1] = fma_f16hi(t[0], a, acc[1]); acc[2] = fma_f16lo(t[1], a, acc[2]);
        acc[3] = fma_f16hi(t[1], a, acc[3]); acc[4] = fma_f16lo(t[2], a, acc[4]); acc[5] = fma_f16hi(t[2], a, acc[5]);
        acc[6] = fma_f16lo(t[3], a, acc[6]); acc[7] = fma_f16hi(t[3], a, acc[7]);
    }
}
// PK = 2: h (+)= (a * bilinear weights) . corners, the level's four points into the same packed sums; then acc (+)= h
template <bool START>
__device__ __forceinline__ void pk2_sample(uint32_t *h, const uint4 &v00, const uint4 &v01, const uint4 &v10, const uint4 &v11,
                                           uint32_t wa, uint32_t wb)
{
    pk_corners4<START>(h, v00, v01, v10, v11, wa, wb);
}
template <bool FIRST>
__device__ __forceinline__ void pk2_flush(float *acc, const uint32_t *h)
{
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (FIRST) {
            acc[2 * r] = cvt_f16lo(h[r]);
            acc[2 * r + 1] = cvt_f16hi(h[r]);
        } else {
            acc[2 * r] = add_f16lo(h[r], acc[2 * r]);
            acc[2 * r + 1] = add_f16hi(h[r], acc[2 * r + 1]);
        }
    }
}

template <bool REF4, int RES, bool SERIAL, bool PERM, int ABL = 0, int PK = 0>
__global__ void __launch_bounds__(kRThreads) msda_bordered_kernel(BorderedArgs p)
{
    static_assert(PK == 0 || (RES == 2 && ABL == 0), "packed-fp16 accumulation: the two-resident-level schedule only");
    using F = ResFma<half_t>;
    constexpr int kAux = ((ABL & 128) ? 2 : 0) | ((ABL & 256) ? 16 : 0);   // experiments: nt / sc1 on the fine-level loads
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // [resident records: levels 2, 3 with their borders + the closing zero record][weights: kRWaves x 4 KB]
    const int slab_bytes = p.res_px * 64;
    float4 *wts = reinterpret_cast<float4 *>(lds + slab_bytes);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = blockIdx.x % p.M;
    const int rest = blockIdx.x / p.M;
    const int b_first = rest / p.chunks;
    const int b_end = SERIAL ? p.B : b_first + 1;
    const int b_step = SERIAL ? p.image_serial : 1;
    const int chunk = rest - b_first * p.chunks;
    const int rows_per_chunk = (p.Nq + p.chunks - 1) / p.chunks;
    const int q_lo = chunk * rows_per_chunk;
    const int q_hi = min(p.Nq, q_lo + rows_per_chunk);
    if (q_lo >= q_hi) return;  // workgroup-uniform
    if ((ABL & 32) && tid == 0) p.stamps[blockIdx.x * 8 + 0] = wall_clock64();
    if ((ABL & 32) && blockIdx.x == 9 && lane == 0) p.stamps[2048 + wave * 4 + 0] = wall_clock64();

    for (int b = b_first; b < b_end; b += b_step) {
    const char *base = p.value + ((int64_t)b * p.M + m) * p.Np * 64;

    // ---- rows: a quad per (query, head); lane j of the quad owns level j ----
    const int g = lane >> 2, j = lane & 3;
    const int H = j == 0 ? p.H0 : j == 1 ? p.H1 : j == 2 ? p.H2 : p.H3;
    const int W = j == 0 ? p.W0 : j == 1 ? p.W1 : j == 2 ? p.W2 : p.W3;
    const int P = j == 0 ? 0 : j == 1 ? p.P1 : j == 2 ? p.P2 : p.P3;
    const float fH = (float)H, fW = (float)W;
    // largest float below size + 1 (bordered coordinates: pixel x lives at x + 1)
    const float xmax = __uint_as_float(__float_as_uint(fW + 1.f) - 1u);
    const float ymax = __uint_as_float(__float_as_uint(fH + 1.f) - 1u);
    const uint32_t pitch64 = ((uint32_t)W + 1u) * 64u;   // row pitch in bytes (24-bit operand of the index arithmetic)
    // byte offset of this lane's level: into the head's global map (fine levels) or into `lds` (resident levels)
    const uint32_t lvl_base = j < 4 - RES ? (uint32_t)P * 64u : (uint32_t)(uint64_t)lds + (uint32_t)(P - p.res_start) * 64u;
    const uint32_t lane_off = (uint32_t)(j * 16);
    // row pitch in bytes of each level, wave-uniform (the buffer loads take it as their scalar offset)
    const uint32_t pb0 = (uint32_t)(p.W0 + 1) * 64u, pb1 = (uint32_t)(p.W1 + 1) * 64u, pb2 = (uint32_t)(p.W2 + 1) * 64u,
                   pb3 = (uint32_t)(p.W3 + 1) * 64u;
    const __amdgpu_buffer_rsrc_t rsrc = make_uniform_rsrc(base, (uint32_t)((int64_t)p.Np * 64));
    constexpr int RD = REF4 ? 4 : 2;
    const __amdgpu_buffer_rsrc_t proj_rsrc = make_uniform_rsrc(
        reinterpret_cast<const char *>(p.proj + ((int64_t)b * p.M + m) * p.Nq * 48), (uint32_t)((int64_t)p.Nq * 96));
    const __amdgpu_buffer_rsrc_t ref_rsrc = make_uniform_rsrc(
        reinterpret_cast<const char *>(p.ref + (int64_t)b * p.ref_batch_stride), (uint32_t)((int64_t)p.Nq * 4 * RD * 4));
    const __amdgpu_buffer_rsrc_t perm_rsrc = make_uniform_rsrc(
        reinterpret_cast<const char *>(PERM ? p.perm + (int64_t)b * p.perm_batch_stride : nullptr),
        PERM ? (uint32_t)((int64_t)p.Nq * 4) : 0u);
    const int64_t out_image_bytes = (int64_t)p.Nq * p.M * 32 * (p.out_bf16 ? 2 : 4);
    const __amdgpu_buffer_rsrc_t out_rsrc = make_uniform_rsrc(reinterpret_cast<const char *>(p.out) + b * out_image_bytes,
                                                              (uint32_t)out_image_bytes);
    float4 *myW = wts + wave * 256;

    struct RowIn {
        uint4 o;
        uint2 gg;
        float r[RD];
    };
    const int ngroups = (q_hi - q_lo + 15) >> 4;
    // the row this quad works on in row group rg: position q_lo + 16 rg + g of the row order
    auto row_slot = [&](int rg) -> uint32_t {
        const uint32_t pos = (uint32_t)min(q_lo + min(rg, ngroups - 1) * 16 + g, q_hi - 1);
        if (PERM) return buffer_load4(perm_rsrc, pos * 4u);
        return pos;
    };
    auto load_row = [&](uint32_t slot) {
        RowIn in;
        const uint32_t prow = __umul24(slot, 96u);
        in.o = buffer_load16(proj_rsrc, prow + (uint32_t)j * 16u);
        in.gg = buffer_load8(proj_rsrc, prow + 64u + (uint32_t)j * 8u);
        if (REF4) {
            const uint4 r = buffer_load16(ref_rsrc, (slot * 4u + (uint32_t)j) * 16u);
            in.r[0] = __uint_as_float(r.x); in.r[1] = __uint_as_float(r.y);
            in.r[RD - 2] = __uint_as_float(r.z); in.r[RD - 1] = __uint_as_float(r.w);
        } else {
            const uint2 r = buffer_load8(ref_rsrc, (slot * 4u + (uint32_t)j) * 8u);
            in.r[0] = __uint_as_float(r.x); in.r[1] = __uint_as_float(r.y);
        }
        return in;
    };
    // ---- prologue.  The resident records reach the LDS through REGISTERS (16-byte loads + ds_write_b128), not by
    // LDS-DMA: a CU's LDS-DMA fills land at ~25 GB/s whatever the number of waves that issue them (MI355X_MICROARCH.md,
    // "ldsdma-fill"; measured here: 95 KB usable 6.1 us after the kernel's start, the workgroup idle for 3 of them), the
    // load path delivers the same bytes in ~1 us.  Vector memory reads return in order, so the request order is: the
    // first two row groups' indices, the wave's six pieces of the slab (into the registers the sample loads use later),
    // the first row's inputs; the pieces are written to LDS while those inputs are on their way.
    RowIn nxt;
    uint32_t slot_cur = row_slot(wave);
    uint32_t slot_nxt = row_slot(wave + kRWaves);
    {
        const uint32_t src0 = (uint32_t)p.res_start * 64u;
        const int pieces = (slab_bytes + 1023) >> 10;
        uint4 stage[kBPieces];
#pragma unroll
        for (int k = 0; k < kBPieces; ++k) {
            // (beyond the slab's last piece: that piece once more; lanes past the slab's end read its last 16 bytes)
            const int piece = min(wave + k * kRWaves, pieces - 1);
            const uint32_t off = min((uint32_t)piece * 1024u + (uint32_t)lane * 16u, (uint32_t)slab_bytes - 16u);
            stage[k] = buffer_load16(rsrc, src0 + off);
        }
        __builtin_amdgcn_sched_barrier(0);
        nxt = load_row(slot_cur);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < kBPieces; ++k) {
            const int piece = min(wave + k * kRWaves, pieces - 1);
            const uint32_t off = min((uint32_t)piece * 1024u + (uint32_t)lane * 16u, (uint32_t)slab_bytes - 16u);
            *reinterpret_cast<uint4 *>(lds + off) = stage[k];
        }
    }
    bool maps_pending = true;
    uint32_t pf_sink = 0;   // destination of the discarded warm-up loads
    if (wave >= ngroups) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // One row group: `cur` = its inputs (loaded one group ahead), `nxt` receives the next group's.  The loop below calls it
    // twice per iteration with the two input sets swapped: no register copies between iterations.
    auto row_group = [&](const int rg, const RowIn &cur, RowIn &nxt) {
        const bool active = q_lo + rg * 16 + g < q_hi;
        const uint32_t q = PERM ? min(slot_cur, (uint32_t)p.Nq - 1u) : slot_cur;   // (a bad order cannot write out of range)
        nxt = load_row(slot_nxt);
        slot_cur = slot_nxt;
        slot_nxt = row_slot(rg + 2 * kRWaves);
        float ox[4], oy[4], lg[4];
        ox[0] = act_lo(cur.o.x); oy[0] = act_hi(cur.o.x); ox[1] = act_lo(cur.o.y); oy[1] = act_hi(cur.o.y);
        ox[2] = act_lo(cur.o.z); oy[2] = act_hi(cur.o.z); ox[3] = act_lo(cur.o.w); oy[3] = act_hi(cur.o.w);
        lg[0] = act_lo(cur.gg.x); lg[1] = act_hi(cur.gg.x); lg[2] = act_lo(cur.gg.y); lg[3] = act_hi(cur.gg.y);
        // (plain v_max3 / v_max with a DPP operand: fmaxf() adds a canonicalising v_max x, x per operand)
        float mx = max3_f32(lg[0], lg[1], lg[2]);
        mx = max_f32(mx, lg[3]);
        mx = max_dpp_xor1(mx);
        mx = max_dpp_xor2(mx);
        const float mxs = mx * -1.4426950408889634f;
        float e[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) e[t] = __builtin_amdgcn_exp2f(fmaf(lg[t], 1.4426950408889634f, mxs));   // exp(lg - mx)
        float sum = (e[0] + e[1]) + (e[2] + e[3]);
        sum += quad_xor1(sum);
        sum += quad_xor2(sum);
        const float inv = active ? __builtin_amdgcn_rcpf(sum) : 0.f;  // inactive rows: all weights zero

        // bordered pixel coordinates of the reference point: w_im + 1 = ref_x * W - 0.5 + 1
        const float bx = fmaf(cur.r[0], fW, 0.5f), by = fmaf(cur.r[1], fH, 0.5f);
        float sx = 1.f, sy = 1.f;
        if (REF4) {   // offset / num_points * box size * 0.5 (num_points = 4), in pixels of this level
            sx = 0.125f * cur.r[RD - 2] * fW;
            sy = 0.125f * cur.r[RD - 1] * fH;
        }
        uint32_t r0[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float wx, wy;
            if (REF4) {
                wx = fmaf(ox[t], sx, bx);
                wy = fmaf(oy[t], sy, by);
            } else {
                wx = bx + ox[t];
                wy = by + oy[t];
            }
            wx = __builtin_amdgcn_fmed3f(wx, 0.f, xmax);   // NaN -> 0: the zero border
            wy = __builtin_amdgcn_fmed3f(wy, 0.f, ymax);
            const float lx = __builtin_amdgcn_fractf(wx), ly = __builtin_amdgcn_fractf(wy);
            const uint32_t x0 = (uint32_t)wx, y0 = (uint32_t)wy;      // truncation = floor (wx, wy >= 0)
            r0[t] = (ABL & 16) ? lvl_base + (uint32_t)((t * 4 + j) * 64) : lshl6_add(x0, mad_u24(y0, pitch64, lvl_base));
            // (w00, w10 | w01, w11) = (row weights) x (1 - lx | lx): the two row weights as one register pair, so the
            // column split is one packed multiply and one packed subtract
            const float a = e[t] * inv;
            // (PK = 1: plain bilinear weights, the attention weight stays fp32 beside them)
            const float wy1 = PK == 1 ? ly : mul_f32(ly, a);
            const f32x2_vec_t rw = {sub_f32(PK == 1 ? 1.f : a, wy1), wy1}, fr = {lx, ly};
            f32x2_vec_t wr, wl;
            asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(wr) : "v"(rw), "v"(fr));   // both halves x lx (fr's low half)
            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(wl) : "v"(rw), "v"(wr));
            // Table layout [point t][row g][level j] (round 5; [level][point][row] before): a wave's store of one point is
            // 64 consecutive entries -- the four levels of a row used to land 1 KB apart, on the same banks (4-way conflict
            // on every store) -- and a sample's read (16 rows, 64 bytes apart, a quad sharing an address) stays conflict free.
            if (PK == 0) myW[(t * 16 + g) * 4 + j] = make_float4(wl.x, wl.y, wr.x, wr.y);   // (y0,x0), (y1,x0), (y0,x1), (y1,x1)
            else if (PK == 1)
                myW[(t * 16 + g) * 4 + j] = make_float4(__uint_as_float(cvt_pk_f16(wl.x, wl.y)),
                                                        __uint_as_float(cvt_pk_f16(wr.x, wr.y)), a, 0.f);
            else
                reinterpret_cast<uint2 *>(myW)[(t * 16 + g) * 4 + j] = make_uint2(cvt_pk_f16(wl.x, wl.y), cvt_pk_f16(wr.x, wr.y));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();

        float acc[8];

#define SDETR_B_PB(JL) (JL == 0 ? pb0 : JL == 1 ? pb1 : JL == 2 ? pb2 : pb3)
#define SDETR_B_ISSUE(SLOT, JL, T)                                                                             \
    {                                                                                                          \
        const uint32_t o0 = quad_bcast<JL>(r0[T]) + lane_off;                                                  \
        if (!(ABL & 1)) {                                                                                      \
        va[SLOT][0] = buffer_load16_aux<kAux>(rsrc, o0, 0u);                                                   \
        va[SLOT][1] = buffer_load16_aux<kAux>(rsrc, o0 + 64u, 0u);                                             \
        va[SLOT][2] = buffer_load16_aux<kAux>(rsrc, o0, SDETR_B_PB(JL));                                       \
        va[SLOT][3] = buffer_load16_aux<kAux>(rsrc, o0 + 64u, SDETR_B_PB(JL));                                 \
        } else { va[SLOT][0] = va[SLOT][1] = va[SLOT][2] = va[SLOT][3] = make_uint4(o0, o0, o0, o0); }        \
    }
#define SDETR_B_WENTRY(JL, T) ((T * 16 + g) * 4 + JL)
    /* one sample's four corner records v0 (y0,x0), v1 (y0,x1), v2 (y1,x0), v3 (y1,x1) into the accumulators; KIND: 0 the
       row's first sample, 1 any other; PK = 2 keeps a packed level sum per memory path (hl: resident levels, hf: fine) */ \
#define SDETR_B_CONSUME(V0, V1, V2, V3, JL, T, KIND, HS)                                                       \
    {                                                                                                          \
        if (PK == 0) {                                                                                         \
            const float4 w = myW[SDETR_B_WENTRY(JL, T)];                                                       \
            if (KIND == 0) mul8_f16(acc, V0, w.x); else F::fma8(acc, V0, w.x);                                 \
            F::fma8(acc, V1, w.z);                                                                             \
            F::fma8(acc, V2, w.y);                                                                             \
            F::fma8(acc, V3, w.w);                                                                             \
        } else if (PK == 1) {                                                                                  \
            const float4 w = myW[SDETR_B_WENTRY(JL, T)];                                                       \
            pk1_sample<KIND == 0>(acc, V0, V1, V2, V3, __float_as_uint(w.x), __float_as_uint(w.y), w.z);       \
        } else {                                                                                               \
            const uint2 w = reinterpret_cast<const uint2 *>(myW)[SDETR_B_WENTRY(JL, T)];                       \
            pk2_sample<T == 0>(HS, V0, V1, V2, V3, w.x, w.y);                                                  \
            if (T == 3) pk2_flush<KIND == 0 || (JL == 2 && RES == 2)>(acc, HS);                                \
        }                                                                                                      \
    }
#define SDETR_B_ACC(SLOT, JL, T)                                                                               \
    {                                                                                                          \
        if (!(ABL & 4)) {                                                                                      \
            SDETR_B_CONSUME(va[SLOT][0], va[SLOT][1], va[SLOT][2], va[SLOT][3], JL, T, 1, hf)                  \
        } else {                                                                                               \
            const float4 w = myW[SDETR_B_WENTRY(JL, T)];                                                       \
            acc[0] += w.x + __uint_as_float(va[SLOT][0].x ^ va[SLOT][1].y ^ va[SLOT][2].z ^ va[SLOT][3].w);   \
        }                                                                                                      \
    }
#define SDETR_B_LDS(JL, T)                                                                                     \
    {                                                                                                          \
        const uint32_t o0 = quad_bcast<JL>(r0[T]) + lane_off, o1 = o0 + SDETR_B_PB(JL);                        \
        uint4 v0, v1, v2, v3;                                                                                  \
        if (!(ABL & 2)) {                                                                                      \
            v0 = lds_read16(o0); v1 = lds_read16(o0 + 64u); v2 = lds_read16(o1); v3 = lds_read16(o1 + 64u);    \
        } else { v0 = v1 = v2 = v3 = make_uint4(o0, o1, o0, o1); }                                             \
        if (!(ABL & 8)) {                                                                                      \
            SDETR_B_CONSUME(v0, v1, v2, v3, JL, T, 1, hl)                                                      \
        } else {                                                                                               \
            const float4 w = myW[SDETR_B_WENTRY(JL, T)];                                                       \
            acc[0] += w.x + __uint_as_float(v0.x ^ v1.y ^ v2.z ^ v3.w);                                        \
        }                                                                                                      \
    }
    /* the row's first sample: its first product starts the accumulators */                                  \
#define SDETR_B_LDS_FIRST(JL, T)                                                                               \
    {                                                                                                          \
        const uint32_t o0 = quad_bcast<JL>(r0[T]) + lane_off, o1 = o0 + SDETR_B_PB(JL);                        \
        uint4 v0, v1, v2, v3;                                                                                  \
        if (!(ABL & 2)) {                                                                                      \
            v0 = lds_read16(o0); v1 = lds_read16(o0 + 64u); v2 = lds_read16(o1); v3 = lds_read16(o1 + 64u);    \
        } else { v0 = v1 = v2 = v3 = make_uint4(o0, o1, o0, o1); }                                             \
        SDETR_B_CONSUME(v0, v1, v2, v3, JL, T, 0, hl)                                                          \
    }
#define SDETR_FENCE __builtin_amdgcn_sched_barrier(0);
        // Fine-level samples in flight per wave.  Round 3 kept four (16 loads) rolling; measured on the step's own operands
        // (benchmarks/msda_real_operands.py) one is FASTER: 25.4 vs 26.7 us at layer 0, 16.2 vs 17.5 at layer 2 -- the 16
        // waves of a CU request 256 KB at a time with four slots, eight times the 32 KB L1 they allocate in, and with the
        // rows in spatial order the loop is bound by vector-ALU issue, not by the loads' latency (+128 dummy instructions
        // per group cost +2.7 us, -12 loads in flight nothing).  The level-3-only variant (twelve fine-level samples) keeps two.
        uint4 va[4][4];
        uint32_t hl[4], hf[4];   // PK = 2: packed fp16 sums of the current resident / fine level's points
        SDETR_B_ISSUE(0, 0, 0)
        if ((ABL & (64 | 2048)) || RES != 2) { SDETR_B_ISSUE(1, 0, 1) }
        if ((ABL & 2048) && RES == 2) { SDETR_B_ISSUE(2, 0, 2) SDETR_B_ISSUE(3, 0, 3) }
        SDETR_FENCE
        if (maps_pending) {
            if ((ABL & 32) && tid == 0) p.stamps[blockIdx.x * 8 + 1] = wall_clock64();
            if ((ABL & 32) && blockIdx.x == 9 && lane == 0) p.stamps[2048 + wave * 4 + 1] = wall_clock64();
            // First row group of this wave: everything it can do without the resident records is under way; its own
            // pieces of them were written above (the barrier's release covers the LDS stores), the sample loads and the
            // next row's inputs stay in flight across the barrier (a bare s_barrier behind the LDS wait: __syncthreads()
            // would drain the vector memory queue too).
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if ((ABL & 32) && tid == 0) p.stamps[blockIdx.x * 8 + 2] = wall_clock64();
            if ((ABL & 32) && blockIdx.x == 9 && lane == 0) p.stamps[2048 + wave * 4 + 2] = wall_clock64();
            if (p.prefetch_fine & 2) {
                // L2 warm-up of this workgroup's rows of the projection slab (written by the launch in front on other XCDs:
                // every row group's first action is a round trip to them)
                for (int pos = q_lo + tid; pos < q_hi; pos += kRThreads) {
                    uint32_t row = (uint32_t)pos;
                    if (PERM) row = buffer_load4(perm_rsrc, (uint32_t)pos * 4u);
                    const uint32_t off = __umul24(row, 96u);
                    pf_sink ^= buffer_load4(proj_rsrc, off);
                }
            }
            if (p.prefetch_fine & 1) {
                // L2 warm-up (in the step the maps were written ~0.5 ms earlier and come from HBM / the Infinity Cache):
                // one dword of every 128-byte line of this workgroup's share of the fine levels -- the head's 1.3 MB reach
                // the XCD's L2 as one bulk read instead of as the gather's demand misses.  The values are folded into a
                // dummy that is consumed after the row loop.
                const uint32_t lines = ((uint32_t)p.res_start * 64u + 127u) >> 7;
                const uint32_t share = (lines + (uint32_t)p.chunks - 1u) / (uint32_t)p.chunks;
                const uint32_t l0 = (uint32_t)chunk * share, l1 = min(lines, l0 + share);
                for (uint32_t ln = l0 + (uint32_t)wave * 64u + (uint32_t)lane; ln < l1; ln += kRWaves * 64u) {
                    const uint32_t off = ln * 128u;
                    pf_sink ^= buffer_load4(rsrc, off);     // (compiler-tracked: its wait counters know the load)
                }
            }
            maps_pending = false;
        }
        SDETR_FENCE
        if (RES == 2 && !(ABL & (64 | 2048))) {
            // ONE fine-level sample (four loads) in flight per wave: see the note at `va` above
            SDETR_B_LDS_FIRST(2, 0) SDETR_FENCE SDETR_B_ACC(0, 0, 0) SDETR_FENCE SDETR_B_ISSUE(0, 0, 1) SDETR_FENCE
            SDETR_B_LDS(2, 1) SDETR_FENCE SDETR_B_ACC(0, 0, 1) SDETR_FENCE SDETR_B_ISSUE(0, 0, 2) SDETR_FENCE
            SDETR_B_LDS(2, 2) SDETR_FENCE SDETR_B_ACC(0, 0, 2) SDETR_FENCE SDETR_B_ISSUE(0, 0, 3) SDETR_FENCE
            SDETR_B_LDS(2, 3) SDETR_FENCE SDETR_B_ACC(0, 0, 3) SDETR_FENCE SDETR_B_ISSUE(0, 1, 0) SDETR_FENCE
            SDETR_B_LDS(3, 0) SDETR_FENCE SDETR_B_ACC(0, 1, 0) SDETR_FENCE SDETR_B_ISSUE(0, 1, 1) SDETR_FENCE
            SDETR_B_LDS(3, 1) SDETR_FENCE SDETR_B_ACC(0, 1, 1) SDETR_FENCE SDETR_B_ISSUE(0, 1, 2) SDETR_FENCE
            SDETR_B_LDS(3, 2) SDETR_FENCE SDETR_B_ACC(0, 1, 2) SDETR_FENCE SDETR_B_ISSUE(0, 1, 3) SDETR_FENCE
            SDETR_B_LDS(3, 3) SDETR_FENCE SDETR_B_ACC(0, 1, 3) SDETR_FENCE
        } else if (RES == 2 && (ABL & 64)) {
            // experiment: two sample slots in flight instead of four
            SDETR_B_LDS_FIRST(2, 0) SDETR_FENCE SDETR_B_ACC(0, 0, 0) SDETR_FENCE SDETR_B_ISSUE(0, 0, 2) SDETR_FENCE
            SDETR_B_LDS(2, 1) SDETR_FENCE SDETR_B_ACC(1, 0, 1) SDETR_FENCE SDETR_B_ISSUE(1, 0, 3) SDETR_FENCE
            SDETR_B_LDS(2, 2) SDETR_FENCE SDETR_B_ACC(0, 0, 2) SDETR_FENCE SDETR_B_ISSUE(0, 1, 0) SDETR_FENCE
            SDETR_B_LDS(2, 3) SDETR_FENCE SDETR_B_ACC(1, 0, 3) SDETR_FENCE SDETR_B_ISSUE(1, 1, 1) SDETR_FENCE
            SDETR_B_LDS(3, 0) SDETR_FENCE SDETR_B_ACC(0, 1, 0) SDETR_FENCE SDETR_B_ISSUE(0, 1, 2) SDETR_FENCE
            SDETR_B_LDS(3, 1) SDETR_FENCE SDETR_B_ACC(1, 1, 1) SDETR_FENCE SDETR_B_ISSUE(1, 1, 3) SDETR_FENCE
            SDETR_B_LDS(3, 2) SDETR_FENCE SDETR_B_ACC(0, 1, 2) SDETR_FENCE
            SDETR_B_LDS(3, 3) SDETR_FENCE SDETR_B_ACC(1, 1, 3) SDETR_FENCE
        } else if (RES == 2) {
            SDETR_B_LDS_FIRST(2, 0) SDETR_FENCE SDETR_B_ACC(0, 0, 0) SDETR_FENCE SDETR_B_ISSUE(0, 1, 0) SDETR_FENCE
            SDETR_B_LDS(2, 1) SDETR_FENCE SDETR_B_ACC(1, 0, 1) SDETR_FENCE SDETR_B_ISSUE(1, 1, 1) SDETR_FENCE
            SDETR_B_LDS(2, 2) SDETR_FENCE SDETR_B_ACC(2, 0, 2) SDETR_FENCE SDETR_B_ISSUE(2, 1, 2) SDETR_FENCE
            SDETR_B_LDS(2, 3) SDETR_FENCE SDETR_B_ACC(3, 0, 3) SDETR_FENCE SDETR_B_ISSUE(3, 1, 3) SDETR_FENCE
            SDETR_B_LDS(3, 0) SDETR_FENCE SDETR_B_ACC(0, 1, 0) SDETR_FENCE
            SDETR_B_LDS(3, 1) SDETR_FENCE SDETR_B_ACC(1, 1, 1) SDETR_FENCE
            SDETR_B_LDS(3, 2) SDETR_FENCE SDETR_B_ACC(2, 1, 2) SDETR_FENCE
            SDETR_B_LDS(3, 3) SDETR_FENCE SDETR_B_ACC(3, 1, 3) SDETR_FENCE
        } else {
            // level 3 alone resident: twelve fine-level samples through TWO slots (two in flight from the start)
            SDETR_B_LDS_FIRST(3, 0) SDETR_FENCE SDETR_B_ACC(0, 0, 0) SDETR_FENCE SDETR_B_ISSUE(0, 0, 2) SDETR_FENCE
            SDETR_B_LDS(3, 1) SDETR_FENCE SDETR_B_ACC(1, 0, 1) SDETR_FENCE SDETR_B_ISSUE(1, 0, 3) SDETR_FENCE
            SDETR_B_LDS(3, 2) SDETR_FENCE SDETR_B_ACC(0, 0, 2) SDETR_FENCE SDETR_B_ISSUE(0, 1, 0) SDETR_FENCE
            SDETR_B_LDS(3, 3) SDETR_FENCE SDETR_B_ACC(1, 0, 3) SDETR_FENCE SDETR_B_ISSUE(1, 1, 1) SDETR_FENCE
            SDETR_B_ACC(0, 1, 0) SDETR_FENCE SDETR_B_ISSUE(0, 1, 2) SDETR_FENCE
            SDETR_B_ACC(1, 1, 1) SDETR_FENCE SDETR_B_ISSUE(1, 1, 3) SDETR_FENCE
            SDETR_B_ACC(0, 1, 2) SDETR_FENCE SDETR_B_ISSUE(0, 2, 0) SDETR_FENCE
            SDETR_B_ACC(1, 1, 3) SDETR_FENCE SDETR_B_ISSUE(1, 2, 1) SDETR_FENCE
            SDETR_B_ACC(0, 2, 0) SDETR_FENCE SDETR_B_ISSUE(0, 2, 2) SDETR_FENCE
            SDETR_B_ACC(1, 2, 1) SDETR_FENCE SDETR_B_ISSUE(1, 2, 3) SDETR_FENCE
            SDETR_B_ACC(0, 2, 2) SDETR_FENCE SDETR_B_ACC(1, 2, 3) SDETR_FENCE
        }
#undef SDETR_FENCE
#undef SDETR_B_ISSUE
#undef SDETR_B_ACC
#undef SDETR_B_LDS
#undef SDETR_B_LDS_FIRST
#undef SDETR_B_CONSUME
#undef SDETR_B_WENTRY
#undef SDETR_B_PB

        if (ABL & 1024) {
            // experiment: 128 extra vector instructions per row group (is the loop bound by vector-ALU issue?)
            float d = acc[0];
#pragma unroll
            for (int i = 0; i < 128; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(d) : "v"(acc[1]));
            if (__float_as_uint(d) == 0x7fc12345u) acc[0] = d;
        }
        if (active) {
            // this image's output rows through a buffer resource: (row * M + head) * 32 channels, 32-bit offsets
            const uint32_t oe = (__umul24(q, (uint32_t)p.M) + (uint32_t)m) * 32u + (uint32_t)j * 8u;
            if (p.out_bf16) {
                buffer_store16(out_rsrc, oe * 2u, make_uint4(pack_act2(acc[0], acc[1]), pack_act2(acc[2], acc[3]),
                                                             pack_act2(acc[4], acc[5]), pack_act2(acc[6], acc[7])));
            } else {
                buffer_store16(out_rsrc, oe * 4u, make_uint4(__float_as_uint(acc[0]), __float_as_uint(acc[1]),
                                                             __float_as_uint(acc[2]), __float_as_uint(acc[3])));
                buffer_store16(out_rsrc, oe * 4u + 16u, make_uint4(__float_as_uint(acc[4]), __float_as_uint(acc[5]),
                                                                   __float_as_uint(acc[6]), __float_as_uint(acc[7])));
            }
        }
    };
    {
        // (one copy of the body with the input set moved between iterations -- 12.5 KB of code instead of 22.6 -- runs the
        // same time in the step, 121.8 / 122.2 against 122.3 / 121.4 us over the six launches: the cold-launch penalty is not
        // instruction fetch)
        RowIn in_b;
        for (int rg = wave; rg < ngroups; rg += 2 * kRWaves) {
            row_group(rg, nxt, in_b);
            if (rg + kRWaves < ngroups) row_group(rg + kRWaves, in_b, nxt);
        }
    }
    asm volatile("" ::"v"(pf_sink));   // (the warm-up loads' values are consumed here, after the loop: nothing waits for them earlier)
    if (SERIAL && b + b_step < b_end) __syncthreads();
    }   // images
    if (ABL & 32) {
        __syncthreads();
        if (tid == 0) p.stamps[blockIdx.x * 8 + 3] = wall_clock64();
    }
}

static int device_cu_count()
{
    static thread_local int cached_dev = -1, cached_cus = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev != cached_dev) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cached_dev = dev;
        cached_cus = cus;
    }
    return cached_cus;
}

}  // namespace sdetr

using namespace sdetr;

extern "C" int sdetr_msda_resident_max_pixels(void) { return kRMaxResidentPx; }

extern "C" int sdetr_msda_resident_forward(sdetr_stream_t stream, const void *value_hm, int value_dtype,
                                           const int32_t *level_hw_host, const float *ref, int ref_dim,
                                           int64_t ref_batch_stride, const void *proj_hm_bf16, int B, int Nv, int M,
                                           int Nq, void *out, int out_dtype, int chunks)
{
    return sdetr_msda_resident_forward_ex(stream, value_hm, value_dtype, level_hw_host, ref, ref_dim, ref_batch_stride,
                                          proj_hm_bf16, B, Nv, M, Nq, out, out_dtype, chunks, -1);
}

extern "C" int sdetr_msda_resident_forward_ex(sdetr_stream_t stream, const void *value_hm, int value_dtype,
                                              const int32_t *level_hw_host, const float *ref, int ref_dim,
                                              int64_t ref_batch_stride, const void *proj_hm_bf16, int B, int Nv, int M,
                                              int Nq, void *out, int out_dtype, int chunks, int image_lanes)
{
    if (B < 0 || Nv <= 0 || M <= 0 || Nq < 0) return fail("msda_resident_forward: bad dims B=%d Nv=%d M=%d Nq=%d", B, Nv, M, Nq);
    if (!value_hm || !level_hw_host || !ref || !proj_hm_bf16 || !out) return fail("msda_resident_forward: null pointer");
    if (ref_dim != 2 && ref_dim != 4)
        return fail("Last dim of reference_points must be 2 or 4, but get %d instead.", ref_dim);
    // (a bf16-map instantiation exists in the template but needs the unpack registers the 16 loads in flight
    // occupy: it spills at the 128-register budget of a 16-wave workgroup, so bf16 maps stay on the direct kernel)
    if (value_dtype != SDETR_F16)
        return fail("msda_resident_forward: fp16 head-major value maps only (dtype %d)", value_dtype);
    if (out_dtype != kActCode && out_dtype != SDETR_F32) return fail("msda_resident_forward: out must be bf16 or f32");
    if (ref_batch_stride == 0) ref_batch_stride = (int64_t)Nq * 4 * ref_dim;
    if (ref_batch_stride < (int64_t)Nq * 4 * ref_dim || (ref_batch_stride % ref_dim))
        return fail("msda_resident_forward: bad reference point batch stride");
    if ((reinterpret_cast<uintptr_t>(proj_hm_bf16) % 16) || (reinterpret_cast<uintptr_t>(ref) % 16) ||
        (reinterpret_cast<uintptr_t>(value_hm) % 16) || (reinterpret_cast<uintptr_t>(out) % 16))
        return fail("msda_resident_forward: operands must be 16-byte aligned");
    ResidentArgs a{};
    const int32_t *hw = level_hw_host;
    for (int l = 0; l < 4; ++l)
        if (hw[2 * l] <= 0 || hw[2 * l + 1] <= 0) return fail("msda_resident_forward: bad level shape");
    a.H0 = hw[0]; a.W0 = hw[1]; a.H1 = hw[2]; a.W1 = hw[3]; a.H2 = hw[4]; a.W2 = hw[5]; a.H3 = hw[6]; a.W3 = hw[7];
    const int64_t s1 = (int64_t)a.H0 * a.W0, s2 = s1 + (int64_t)a.H1 * a.W1, s3 = s2 + (int64_t)a.H2 * a.W2;
    if (s3 + (int64_t)a.H3 * a.W3 != Nv) return fail("msda_resident_forward: level shapes do not add up to %d pixels", Nv);
    if ((int64_t)Nv * 64 >= 0xffffffffLL) return fail("msda_resident_forward: value map too large for 32-bit offsets");
    a.S1 = (int)s1; a.S2 = (int)s2; a.S3 = (int)s3;
    // levels 2 + 3 resident when they fit together, else level 3 alone
    int res_levels = 2;
    a.res_start = a.S2;
    if (Nv - a.S2 > kRMaxResidentPx) {
        res_levels = 1;
        a.res_start = a.S3;
    }
    a.res_px = Nv - a.res_start;
    if (a.res_px > kRMaxResidentPx)
        return fail("msda_resident_forward: level 3 alone holds %d pixels, more than the %d that fit in LDS", a.res_px,
                    kRMaxResidentPx);
    if ((int64_t)B * Nq == 0) return 0;
    a.value = reinterpret_cast<const char *>(value_hm);
    a.ref = ref; a.ref_batch_stride = ref_batch_stride; a.ref_dim = ref_dim;
    a.proj = reinterpret_cast<const bf16_t *>(proj_hm_bf16);
    a.out = out; a.out_bf16 = (out_dtype == kActCode);
    a.B = B; a.Nv = Nv; a.M = M; a.Nq = Nq;
    // Maps beyond what the Infinity Cache holds next to everything else (256 MiB; the benchmark's two images: 46 MB): the
    // chip works on FOUR images at a time (every workgroup walks the images of its lane), so that the maps being gathered
    // from stay cached: at batch 16 (366 MB of maps) 299 / 187 / 78 us at 11 363 / 6817 / 2272 queries against 380 / 231 /
    // 79 with all sixteen in flight and 336 / 249 / 129 one at a time (direct kernel: 350 / 234 / 68).
    // `image_lanes` >= 0 fixes the number of lanes (0 = every image in flight); -1 = the rule above.
    const int cus = device_cu_count();
    int lanes = 0;
    if ((int64_t)B * M * Nv * 64 > ((int64_t)160 << 20) && B > 4) lanes = 4;
    if (image_lanes >= 0) lanes = image_lanes > B ? B : image_lanes;
    a.image_serial = lanes;
    const int groups = lanes ? lanes : B;   // (image, head) slots the workgroups are spread over
    if (chunks <= 0) {
        // one workgroup per CU: the (image, head) pairs share the CUs evenly; at least four row groups per workgroup
        // (measured at 900 / 2272 queries: spreading a small layer over all CUs beats filling fewer CUs' waves)
        chunks = cus / (groups * M);
        const int max_chunks = (Nq + 63) / 64;
        if (chunks > max_chunks) chunks = max_chunks;
        if (chunks < 1) chunks = 1;
    }
    a.chunks = chunks;
    const int64_t blocks = (int64_t)groups * M * chunks;
    if (blocks > 0x7fffffffLL) return fail("msda_resident_forward: grid too large");
    const int lds_bytes = 2 * kRPad + a.res_px * 64 + kRWaves * kRWeightBytes;
    // the attribute is per device and the call is cheap: set before every launch (no process-wide flag)
#define SDETR_RES_LAUNCH(VT, REF4, RES, SER)                                                                    \
    do {                                                                                                            \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(msda_resident_kernel<VT, REF4, RES, SER>),     \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kRLdsBudget);                         \
        hipLaunchKernelGGL((msda_resident_kernel<VT, REF4, RES, SER>), dim3((unsigned)blocks), dim3(kRThreads), \
                           lds_bytes, stream, a);                                                                   \
    } while (0)
#define SDETR_RES_PICK(REF4, RES)                                                                               \
    do {                                                                                                            \
        if (a.image_serial) SDETR_RES_LAUNCH(half_t, REF4, RES, true);                                          \
        else SDETR_RES_LAUNCH(half_t, REF4, RES, false);                                                        \
    } while (0)
    if (res_levels == 2) {
        if (ref_dim == 4) SDETR_RES_PICK(true, 2);
        else SDETR_RES_PICK(false, 2);
    } else {
        if (ref_dim == 4) SDETR_RES_PICK(true, 1);
        else SDETR_RES_PICK(false, 1);
    }
#undef SDETR_RES_PICK
#undef SDETR_RES_LAUNCH
    note_forward_kernel(SDETR_KERNEL_MSDA_RESIDENT);
    return check_launch("msda_resident");
}

// ---- bordered maps (round 4) ----------------------------------------------------------------------------------------

// Records per (image, head) of the bordered layout of a pyramid: sum (H_l + 2)(W_l + 1) + 1 (see msda_bordered_kernel).
extern "C" int64_t sdetr_msda_bordered_records(const int32_t *level_hw_host, int num_levels)
{
    if (!level_hw_host || num_levels <= 0) return -1;
    int64_t n = 1;
    for (int l = 0; l < num_levels; ++l) {
        const int64_t h = level_hw_host[2 * l], w = level_hw_host[2 * l + 1];
        if (h <= 0 || w <= 0) return -1;
        n += (h + 2) * (w + 1);
    }
    return n;
}

extern "C" int sdetr_msda_bordered_max_resident_records(void) { return kBMaxResidentPx; }

// The ablated instantiations of the kernel (wrong results by construction) and the phase stamps exist only in the
// benchmark build of this file (-DSDETR_MSDA_ABLATIONS: `python salience_detr_amd/csrc/build.py --ablations` ->
// libsalience_hip_ablate.so, loaded by benchmarks/msda_bordered_ab.py / msda_real_operands.py); the product library ignores
// SDETR_MSDA_ABLATE and does not export sdetr_msda_debug_stamps.
#ifdef SDETR_MSDA_ABLATIONS
// device buffer of [workgroups][4] uint64 for the SDETR_MSDA_ABLATE=32 phase stamps
static unsigned long long *g_stamps = nullptr;
extern "C" void sdetr_msda_debug_stamps(void *device_buffer) { g_stamps = static_cast<unsigned long long *>(device_buffer); }
#endif

extern "C" int sdetr_msda_bordered_forward(sdetr_stream_t stream, const void *value_bordered, int value_dtype,
                                           const int32_t *level_hw_host, const float *ref, int ref_dim,
                                           int64_t ref_batch_stride, const void *proj_hm_bf16, const int32_t *row_order,
                                           int64_t row_order_batch_stride, int B, int Np, int M, int Nq, void *out,
                                           int out_dtype, int chunks)
{
    return sdetr_msda_bordered_forward_ex(stream, value_bordered, value_dtype, level_hw_host, ref, ref_dim, ref_batch_stride,
                                          proj_hm_bf16, row_order, row_order_batch_stride, B, Np, M, Nq, out, out_dtype,
                                          chunks, SDETR_MSDA_ACC_DEFAULT, -1, -1);
}

extern "C" int sdetr_msda_bordered_forward_ex(sdetr_stream_t stream, const void *value_bordered, int value_dtype,
                                              const int32_t *level_hw_host, const float *ref, int ref_dim,
                                              int64_t ref_batch_stride, const void *proj_hm_bf16, const int32_t *row_order,
                                              int64_t row_order_batch_stride, int B, int Np, int M, int Nq, void *out,
                                              int out_dtype, int chunks, int accumulate, int image_lanes, int l2_warmup)
{
    if (B < 0 || Np <= 0 || M <= 0 || Nq < 0) return fail("msda_bordered_forward: bad dims B=%d Np=%d M=%d Nq=%d", B, Np, M, Nq);
    if (!value_bordered || !level_hw_host || !ref || !proj_hm_bf16 || !out) return fail("msda_bordered_forward: null pointer");
    if (ref_dim != 2 && ref_dim != 4)
        return fail("Last dim of reference_points must be 2 or 4, but get %d instead.", ref_dim);
    if (value_dtype != SDETR_F16)
        return fail("msda_bordered_forward: fp16 bordered head-major value maps only (dtype %d)", value_dtype);
    if (out_dtype != kActCode && out_dtype != SDETR_F32) return fail("msda_bordered_forward: out must be bf16 or f32");
    if (ref_batch_stride == 0) ref_batch_stride = (int64_t)Nq * 4 * ref_dim;
    if (ref_batch_stride < (int64_t)Nq * 4 * ref_dim || (ref_batch_stride % ref_dim))
        return fail("msda_bordered_forward: bad reference point batch stride");
    if (row_order && row_order_batch_stride == 0) row_order_batch_stride = Nq;
    if (row_order && row_order_batch_stride < Nq) return fail("msda_bordered_forward: bad row order batch stride");
    if ((reinterpret_cast<uintptr_t>(proj_hm_bf16) % 16) || (reinterpret_cast<uintptr_t>(ref) % 16) ||
        (reinterpret_cast<uintptr_t>(value_bordered) % 16) || (reinterpret_cast<uintptr_t>(out) % 16) ||
        (reinterpret_cast<uintptr_t>(row_order) % 4))
        return fail("msda_bordered_forward: operands must be 16-byte aligned");
    BorderedArgs a{};
    const int32_t *hw = level_hw_host;
    if (sdetr_msda_bordered_records(hw, 4) != Np)
        return fail("msda_bordered_forward: the level shapes give %lld bordered records, the map holds %d",
                    (long long)sdetr_msda_bordered_records(hw, 4), Np);
    a.H0 = hw[0]; a.W0 = hw[1]; a.H1 = hw[2]; a.W1 = hw[3]; a.H2 = hw[4]; a.W2 = hw[5]; a.H3 = hw[6]; a.W3 = hw[7];
    const int64_t p1 = (int64_t)(a.H0 + 2) * (a.W0 + 1), p2 = p1 + (int64_t)(a.H1 + 2) * (a.W1 + 1),
                  p3 = p2 + (int64_t)(a.H2 + 2) * (a.W2 + 1);
    if ((int64_t)Np * 64 >= 0xffffffffLL) return fail("msda_bordered_forward: value map too large for 32-bit offsets");
    if ((int64_t)Nq * M * 32 * 4 >= 0xffffffffLL || Nq >= (1 << 24) || M >= (1 << 24))
        return fail("msda_bordered_forward: too many rows per image for 32-bit offsets (%d x %d heads)", Nq, M);
    // 24-bit operands in the record index arithmetic
    if (a.H0 + 2 >= (1 << 24) || a.W0 + 1 >= (1 << 24)) return fail("msda_bordered_forward: level too large");
    a.P1 = (int)p1; a.P2 = (int)p2; a.P3 = (int)p3;
    int res_levels = 2;
    a.res_start = a.P2;
    if (Np - a.P2 > kBMaxResidentPx) {
        res_levels = 1;
        a.res_start = a.P3;
    }
    a.res_px = Np - a.res_start;
    if (a.res_px > kBMaxResidentPx)
        return fail("msda_bordered_forward: level 3 alone holds %d records, more than the %d that fit in LDS", a.res_px,
                    kBMaxResidentPx);
    if ((int64_t)B * Nq == 0) return 0;
    a.value = reinterpret_cast<const char *>(value_bordered);
    a.ref = ref; a.ref_batch_stride = ref_batch_stride; a.ref_dim = ref_dim;
    a.proj = reinterpret_cast<const bf16_t *>(proj_hm_bf16);
    a.perm = row_order; a.perm_batch_stride = row_order_batch_stride;
    a.out = out; a.out_bf16 = (out_dtype == kActCode);
    a.B = B; a.Np = Np; a.M = M; a.Nq = Nq;
    const int cus = device_cu_count();
    int lanes = 0;
    if ((int64_t)B * M * Np * 64 > ((int64_t)160 << 20) && B > 4) lanes = 4;
    if (image_lanes >= 0) lanes = image_lanes > B ? B : image_lanes;
    a.image_serial = lanes;
    const int groups = lanes ? lanes : B;
    if (chunks <= 0) {
        chunks = cus / (groups * M);
        const int max_chunks = (Nq + 63) / 64;
        if (chunks > max_chunks) chunks = max_chunks;
        if (chunks < 1) chunks = 1;
    }
    a.chunks = chunks;
    // bit 0: L2 warm-up of the fine levels (on: in the step -132.7 -> 124.3 us over the six launches by rocprofv3; nothing
    // to gain in a warm replay), bit 1: of the workgroup's projection rows (measured: no gain).  `l2_warmup` >= 0 fixes it.
    // Only while the fine levels of the images an XCD serves at a time fit its 4 MB L2 beside the rest: on the 5scale
    // pyramid (5.7 MB per head) the warm-up costs 4 us per launch (profiles/r04_msda_ab_5scale_bordered.json).
    // ... and only when a wave walks more than one row group: measured in the step per layer (benchmarks/msda_warmup_ab.sh,
    // rocprofv3, with / without): 28.5 / 32.3, 23.3 / 27.5, 19.4 / 22.2, 18.5 / 20.1, 15.4 / 17.1 us at 11 363 ... 4545 rows
    // per image, but 12.4 / 11.6 at 2272 -- there every wave has at most one group and the warm-up's loads queue in
    // front of its only samples.
    a.prefetch_fine = ((int64_t)groups * a.res_start * 64 <= ((int64_t)7 << 19) && (int64_t)Nq > (int64_t)chunks * kRWaves * 16) ? 1 : 0;
    if (l2_warmup >= 0) a.prefetch_fine = l2_warmup & 3;
    const int64_t blocks = (int64_t)groups * M * chunks;
    if (blocks > 0x7fffffffLL) return fail("msda_bordered_forward: grid too large");
    // corner accumulation: exact fp32 products for fp32 outputs (the parity path); for 16-bit outputs the packed-fp16 form
    // `accumulate` asks for (SDETR_MSDA_ACC_*; SDETR_MSDA_ACC_DEFAULT = the library's choice for the output type)
    if (accumulate < SDETR_MSDA_ACC_DEFAULT || accumulate > SDETR_MSDA_ACC_PACKED_LEVEL)
        return fail("msda_bordered_forward: accumulate must be one of SDETR_MSDA_ACC_* (got %d)", accumulate);
    int pk = 0;
    if (a.out_bf16) pk = accumulate == SDETR_MSDA_ACC_DEFAULT ? kDefaultPackedAccumulate : accumulate;
    const int lds_bytes = a.res_px * 64 + kRWaves * kRWeightBytes;
#define SDETR_B_LAUNCH(REF4, RES, SER, PERM)                                                                        \
    do {                                                                                                            \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(msda_bordered_kernel<REF4, RES, SER, PERM>),       \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kRLdsBudget);                         \
        hipLaunchKernelGGL((msda_bordered_kernel<REF4, RES, SER, PERM>), dim3((unsigned)blocks), dim3(kRThreads),   \
                           lds_bytes, static_cast<hipStream_t>(stream), a);                                         \
    } while (0)
#define SDETR_B_PICK2(REF4, RES, SER)                                                                               \
    do {                                                                                                            \
        if (a.perm) SDETR_B_LAUNCH(REF4, RES, SER, true);                                                           \
        else SDETR_B_LAUNCH(REF4, RES, SER, false);                                                                 \
    } while (0)
    // packed-fp16 corner accumulation (PK, see the helpers above the kernel): 16-bit outputs, two resident levels, one
    // image per workgroup
#define SDETR_B_LAUNCH_PK(REF4, PERM, PKV)                                                                          \
    do {                                                                                                            \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(msda_bordered_kernel<REF4, 2, false, PERM, 0, PKV>), \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kRLdsBudget);                         \
        hipLaunchKernelGGL((msda_bordered_kernel<REF4, 2, false, PERM, 0, PKV>), dim3((unsigned)blocks),            \
                           dim3(kRThreads), lds_bytes, static_cast<hipStream_t>(stream), a);                        \
    } while (0)
#define SDETR_B_PICK_PK(REF4, PKV)                                                                                  \
    do {                                                                                                            \
        if (a.perm) SDETR_B_LAUNCH_PK(REF4, true, PKV);                                                             \
        else SDETR_B_LAUNCH_PK(REF4, false, PKV);                                                                   \
    } while (0)
#ifdef SDETR_MSDA_ABLATIONS
#define SDETR_B_LAUNCH_ABL(ABL)                                                                                     \
    do {                                                                                                            \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(msda_bordered_kernel<false, 2, false, true, ABL>), \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kRLdsBudget);                         \
        hipLaunchKernelGGL((msda_bordered_kernel<false, 2, false, true, ABL>), dim3((unsigned)blocks),              \
                           dim3(kRThreads), lds_bytes, static_cast<hipStream_t>(stream), a);                        \
    } while (0)
    // Ablations for benchmarks/msda_bordered_ab.py (wrong results by construction): SDETR_MSDA_ABLATE = bit mask of
    // 1 no fine-level loads, 2 no LDS map reads, 4 no fine-level products, 8 no resident-level products, 16 every row
    // samples the same records
    if (const char *e = ab_env("SDETR_MSDA_ABLATE")) {
        const int abl = atoi(e);
        if (abl && a.perm && res_levels == 2 && ref_dim == 2 && !a.image_serial) {
            switch (abl) {
            case 1: SDETR_B_LAUNCH_ABL(1); break;
            case 2: SDETR_B_LAUNCH_ABL(2); break;
            case 3: SDETR_B_LAUNCH_ABL(3); break;
            case 5: SDETR_B_LAUNCH_ABL(5); break;
            case 10: SDETR_B_LAUNCH_ABL(10); break;
            case 12: SDETR_B_LAUNCH_ABL(12); break;
            case 15: SDETR_B_LAUNCH_ABL(15); break;
            case 16: SDETR_B_LAUNCH_ABL(16); break;
            case 64: SDETR_B_LAUNCH_ABL(64); break;
            case 2048: SDETR_B_LAUNCH_ABL(2048); break;
            case 1024: SDETR_B_LAUNCH_ABL(1024); break;
            case 3072: SDETR_B_LAUNCH_ABL(3072); break;
            case 128: SDETR_B_LAUNCH_ABL(128); break;
            case 256: SDETR_B_LAUNCH_ABL(256); break;
            case 384: SDETR_B_LAUNCH_ABL(384); break;
            case 192: SDETR_B_LAUNCH_ABL(192); break;
            case 32: a.stamps = g_stamps; if (!a.stamps) return fail("msda_bordered_forward: no stamp buffer"); SDETR_B_LAUNCH_ABL(32); break;
            default: return fail("msda_bordered_forward: no such ablation %d", abl);
            }
            return check_launch("msda_bordered (ablated)");
        }
    }
#endif   // SDETR_MSDA_ABLATIONS
#define SDETR_B_PICK(REF4, RES)                                                                                     \
    do {                                                                                                            \
        if (a.image_serial) SDETR_B_PICK2(REF4, RES, true);                                                         \
        else SDETR_B_PICK2(REF4, RES, false);                                                                       \
    } while (0)
    if (pk && res_levels == 2 && !a.image_serial) {
        if (pk == 1) {
            if (ref_dim == 4) SDETR_B_PICK_PK(true, 1);
            else SDETR_B_PICK_PK(false, 1);
        } else {
            if (ref_dim == 4) SDETR_B_PICK_PK(true, 2);
            else SDETR_B_PICK_PK(false, 2);
        }
    } else if (res_levels == 2) {
        if (ref_dim == 4) SDETR_B_PICK(true, 2);
        else SDETR_B_PICK(false, 2);
    } else {
        if (ref_dim == 4) SDETR_B_PICK(true, 1);
        else SDETR_B_PICK(false, 1);
    }
#undef SDETR_B_PICK
#undef SDETR_B_PICK2
#undef SDETR_B_PICK_PK
#undef SDETR_B_LAUNCH_PK
#undef SDETR_B_LAUNCH_ABL
#undef SDETR_B_LAUNCH
    note_forward_kernel(a.perm ? SDETR_KERNEL_MSDA_BORDERED_ORDERED : SDETR_KERNEL_MSDA_BORDERED);
    return check_launch("msda_bordered");
}
