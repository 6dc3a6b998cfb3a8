// Multi-scale deformable attention backward with grad_value accumulated in LDS (gfx950), D = 32, P = 4, fp32,
// reference layout [B,Nv,M,D].  Arithmetic as msda_backward.hip / ms_deform_im2col_cuda.cuh:76-148,290-392.
//
// Why.  The direct kernel (msda_col2im_chan_kernel) sends every one of the 2048 per-row contributions to L2 as an
// fp32 atomic: 372 M adds at encoder layer 0 (B = 2), and the L2 retires them per 128-byte line request
// (~0.3-0.7 T adds/s measured): 0.6-0.9 ms per launch, 3 % of the HBM roofline.  Accumulating in LDS first was
// tried in round 1 and dropped because `ds_add_f32` is slower still -- benchmarks/micro/lds_atomic_rate.hip puts
// numbers on it: ds_add_f32 0.2 T adds/s (one lane every ~2.6 clocks, whatever the address pattern), but
// ds_add_u32 >= 5.4 T adds/s.  So the accumulation here is FIXED POINT:
//
//  * a workgroup owns (image, head, level, part) where part = a 32x16-pixel tile of the level (levels too large for
//    LDS; its rows = the queries whose sampling centroid on that level falls in the tile, bucketed by two small
//    launches) or a chunk of the queries (levels of <= 1092 pixels, held whole);
//  * its window -- tile + 5 pixels all round, or the whole level -- is 32 int32 accumulators per pixel in LDS
//    (139 776 bytes); every contribution w_corner * aw * grad_out[c] is scaled by a power of two, rounded to nearest
//    (v_cvt_rpi_i32_f32) and added with ds_add_u32.  The scale comes from a bound no accumulator of the item can
//    exceed: (rows of the item) x (largest max_c|g| * sum_p|aw| over the queries of the (level, image, head), found
//    by the bucketing launch), with 2 bits of headroom -- overflow is impossible, integer adds commute (the
//    accumulation inside a window is exact and order-independent), and each add carries <= 2^-30 of that bound as
//    rounding error (the fp32 atomics it replaces: 2^-24 of the running sum).  A NaN or an infinity in grad_out or
//    the weights wins the maximum (compared as bit patterns) and sends the whole (level, image, head) to the
//    floating-point path, so non-finite gradients still propagate;
//  * the window is flushed once, as whole 128-byte lines of fp32 atomics (overlapping halos of neighbouring
//    tiles meet there): ~10x fewer L2 atomics than the direct kernel at layer 0;
//  * a sample whose 2x2 footprint leaves the window (offset beyond the halo) falls back to fp32 atomics on
//    global memory for that sample only, so the result never depends on the window; levels with more than 1024
//    tiles, or an item whose bound is not a normal number, take that path for every sample.
//
// Waves and lanes (round 6; until then eight waves did both halves of the backward, every lane of a row repeating the
// per-sample set-up).  A workgroup is 16 waves of two kinds that work on the SAME item at the same time.  Waves 0-7 (two
// per SIMD) are the SCATTER waves: they own the window, draw the work list, accumulate grad_value and flush.  Waves 8-15
// are the GATHER waves: grad_sampling_loc and grad_attn_weight of the item's rows.  The two kinds meet at one s_barrier
// per item -- in front of the flush, where the next item is published -- after which the scatter waves flush (and
// synchronise among themselves through an LDS counter: the window must be clean before their next add) while the gather
// waves are already in the next item's rows.  Either kind works on 16 rows at a time: the 64 lanes set up the 64 samples
// ONCE (lane = (row, sample): pixel, validity, weights, addresses -- these ~75 vector instructions used to run in all
// eight lanes of a row), the result goes through LDS as a 32-byte record per sample and the eight lanes of a row read it
// back by broadcast.  Gather: lane k of a row holds channels 4k..4k+3 of a corner (one 16-byte buffer load,
// out-of-image corners read as zero through the buffer's bounds check); the three outputs of a sample are linear in its
// four corner dot products <grad_out, v_c>, so they are combined per lane and THEN summed over the 8 lanes (12 fused DPP
// adds per step instead of 16 sums that compiled to ~100 instructions).  Scatter: lane k owns channels {8j + k}: in
// instruction i row r adds octet j = (r & 3) ^ i, so the 32 lanes the LDS serves per clock (4 rows x 8 lanes) always
// cover 32 different banks whatever pixels the rows hit -- no bank conflicts by construction, and no two lanes of an
// instruction ever share an address.
//
// Measured (benchmarks/msda_backward_ab.py, B = 2; profiles/r06_msda_backward_ab.json): 241 us at 11 363 queries (341
// before round 6; the direct kernel: 1050-1070).  Of these ~46 are the other launches (grad_value's zero fill 12,
// bucketing 22 + 10, header clear 2); the main launch is ~195.  The steps of round 6, same box each: set-up once per
// sample through LDS records, fused DPP sums, row-wise flush, long items first 341 -> 270; items pipelined one deep
// (next item's row range, bound and first rows under the flush) 267; two kinds of wave (first walking the rows
// independently: 250, but twice the HBM fetches -- every row operand came from memory once per kind of wave and level)
// 241 with the gather waves on their workgroup's items (HBM fetch 483 -> 216 MB per launch).  Cycle stamps
// (benchmarks/bt_stamps.py on a bt_variant.sh build): a scatter wave's time is 31 % flush, 23 % LDS adds, 16 % record
// round trips (behind its queued adds), 21 % waiting for other waves; vector ALU 40 % busy, LDS 30 %, L1 20 %: every
// wave is a chain of dependent latencies, and 16 waves of 128 registers are what the window leaves room for.  The floor
// under the flush is the memory side: float atomics are executed behind the L2 (TCC_EA0_ATOMIC = all 1.77 M requests of
// a launch), and the chip retires 10 G whole-line fp32 atomics per second when it does nothing else
// (benchmarks/micro/global_atomic_lines.hip: 1.68 M lines in 165 us; plain stores of the same lines: 31 us).  That is
// why this design stops short of 0.15 of the roofline: it would take tiles that own their pixels (no halo, samples
// instead of rows in the buckets, plain stores).
// Tried in round 6 without gain: 64-bit LDS adds on channel pairs (the LDS does retire them at the 32-bit instruction
// rate -- lds_atomic_rate.hip, 18-20 channel adds per clock and CU against 10 -- but the launch did not move: 251.8
// against 248.8 us), the two waves of a SIMD running the LDS-bound and the ALU-bound part in opposite order (+3 us),
// 8 or 21 LDS reads ahead in the flush (0 / +56 us), 8 whole-level items per (level, image, head) instead of 16 (-3 %,
// but the coarser fixed-point quantum crosses the tests' 1e-4), gather waves sweeping contiguous row ranges or all
// levels of a row in one turn (no change).  Earlier: two 256-thread workgroups per CU on half-size windows (438 us:
// 1.5x the flush atomics), whole-pass phases instead of the per-sample pipeline (same time), an exact per-item bound
// from a pre-pass over the rows (an extra round trip per item for 2-3 bits of scale).
//
// Everything that depends on the level shapes is decided on the device from the shape tensors (the reference
// interface hands them over as device tensors; no host copy, no synchronisation): persistent workgroups draw
// (level, image, head, part) items from a counter, one item ahead of their need (the next item's row range, bound and
// first rows travel under the current item's rows and flush).
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "common.h"

namespace sdetr {

void note_backward_kernel(int which);  // abi.hip

constexpr int kBtD = 32, kBtP = 4;
constexpr int kBtTileW = 32, kBtTileH = 16, kBtHalo = 5;
constexpr int kBtWinW = kBtTileW + 2 * kBtHalo;      // 42
constexpr int kBtWinH = kBtTileH + 2 * kBtHalo;      // 26
constexpr int kBtWinPx = kBtWinW * kBtWinH;          // 1092
constexpr int kBtWinBytes = kBtWinPx * kBtD * 4;     // 139 776
constexpr int kBtMaxTiles = 1024;
constexpr int kBtMaxL = 8, kBtMaxHeads = 64;
#ifndef BT_MAX_CHUNKS
// whole-level items per (level, image, head).  Every one flushes the whole level, so fewer would flush less (8: -3 % of the
// launch at 11 363 queries) -- but an item's fixed-point quantum is 2^-29 of (its rows) x (the largest row bound), and
// at 8 the accumulated rounding of level 2 (1421 rows per item) crosses the 1e-4 the tests hold this kernel to
#define BT_MAX_CHUNKS 16
#endif
constexpr int kBtChunkRows = 512, kBtMaxChunks = BT_MAX_CHUNKS;
constexpr int kBtScWaves = 8, kBtGaWaves = 8;        // scatter waves (window, work list, flush) | gather waves
constexpr int kBtThreads = 64 * (kBtScWaves + kBtGaWaves);   // 16 waves, four per SIMD: 128 VGPRs each
constexpr int kBtPassRows = 16;                      // rows of one wave's pass: its 64 lanes set up 16 rows x 4 samples
constexpr int kBtRecRow = 4 * 32 + 16;               // bytes between the records of two rows (4 samples x 32 B + a bank step)
constexpr int kBtRecBytes = 8 * kBtRecRow;           // one wave's records: an 8-row half of a pass
constexpr int kBtLdsBytes = kBtWinBytes + 1024 + (kBtScWaves + kBtGaWaves) * kBtRecBytes;
constexpr uint32_t kBtNoCorner = 0xffffff00u;        // + 16 * lane stays beyond any value slab (sdetr_msda_col2im_lds_supported)
enum { kBtTile = 0, kBtResident = 1, kBtDirect = 2 };

struct BtLevel {
    int H, W, start, mode, tx, ty, parts, items;
};

__device__ __forceinline__ BtLevel bt_level(const int64_t *shapes, const int64_t *lsi, int l, int B, int M, int Nq)
{
    BtLevel v;
    v.H = (int)shapes[2 * l];
    v.W = (int)shapes[2 * l + 1];
    v.start = (int)lsi[l];
    v.tx = (v.W + kBtTileW - 1) / kBtTileW;
    v.ty = (v.H + kBtTileH - 1) / kBtTileH;
    int chunks = (Nq + kBtChunkRows - 1) / kBtChunkRows;
    chunks = chunks < 1 ? 1 : (chunks > kBtMaxChunks ? kBtMaxChunks : chunks);
    if (v.H * v.W <= kBtWinPx) {
        v.mode = kBtResident;
        v.parts = chunks;
    } else if (v.tx * v.ty <= kBtMaxTiles) {
        v.mode = kBtTile;
        v.parts = v.tx * v.ty;
    } else {
        v.mode = kBtDirect;
        v.parts = chunks;
    }
    v.items = B * M * v.parts;
    return v;
}

struct BtArgs {
    const float *grad_out;   // [B,Nq,M,32]
    const float *value;      // [B,Nv,M,32]
    const int64_t *shapes;   // [L,2]
    const int64_t *lsi;      // [L]
    const float *loc;        // [B,Nq,M,L,4,2]
    const float *aw;         // [B,Nq,M,L,4]
    float *grad_value, *grad_loc, *grad_aw;
    int B, Nv, M, L, Nq;
    int *counter;            // work counter (zeroed by the tile-id launch)
    int32_t *start;          // [L,B,kBtMaxTiles+1]
    int32_t *order;          // [L,B,Nq]
    uint16_t *tile_id;       // [L,B,Nq]
    uint32_t *row_bound;     // [L,B,M] bits of max over the queries of max_c|g| * sum_p|aw| (zeroed before the tile-id launch)
};

// ------------------------------------------------------------------------------------------------
// bucketing of the queries by tile, per tiled level: (1) tile of each (image, query) from the centroid of its
// sampling locations on the level (mean over heads and points: the reference point up to the learned offsets);
// (2) per (level, image) one workgroup: LDS histogram, scan, scatter.
__global__ void __launch_bounds__(256) bt_tile_id_kernel(BtArgs p)
{
    __shared__ uint32_t sh_b[kBtMaxL * kBtMaxHeads];   // [l * M + m] bits of the block's largest row bound
    const int blocks_per_image = (p.Nq + 31) / 32;
    const int b = blockIdx.x / blocks_per_image;
    const int q = (blockIdx.x - b * blocks_per_image) * 32 + (threadIdx.x >> 3), k = threadIdx.x & 7;
    const bool act = q < p.Nq;
    const int64_t gq = (int64_t)b * p.Nq + (act ? q : p.Nq - 1);
    for (int i = threadIdx.x; i < p.L * p.M; i += 256) sh_b[i] = 0u;
    __syncthreads();
    // row bounds max_c|g| * sum_p|aw| per level: compared as bit patterns, so a NaN or an infinity anywhere wins the
    // maximum and sends the (level, image, head) to the floating-point path instead of being rounded away
    for (int m = k; m < p.M; m += 8) {
        const int64_t row = gq * p.M + m;
        const uint4 *g4 = reinterpret_cast<const uint4 *>(p.grad_out + row * kBtD);
        uint32_t mg = 0u;
#pragma unroll
        for (int i = 0; i < kBtD / 4; ++i) {
            const uint4 g = g4[i];
            mg = max(max(mg, g.x & 0x7fffffffu), max(max(g.y & 0x7fffffffu, g.z & 0x7fffffffu), g.w & 0x7fffffffu));
        }
        for (int l = 0; l < p.L; ++l) {
            const float4 a = *reinterpret_cast<const float4 *>(p.aw + (row * p.L + l) * kBtP);
            const float rb = __uint_as_float(mg) * ((fabsf(a.x) + fabsf(a.y)) + (fabsf(a.z) + fabsf(a.w)));
            if (act) atomicMax(&sh_b[l * p.M + m], __float_as_uint(rb) & 0x7fffffffu);
        }
    }
    for (int l = 0; l < p.L; ++l) {
        const BtLevel lv = bt_level(p.shapes, p.lsi, l, p.B, p.M, p.Nq);
        if (lv.mode != kBtTile) continue;
        float sx = 0.f, sy = 0.f;
        for (int m = k; m < p.M; m += 8) {
            const float4 *q4 = reinterpret_cast<const float4 *>(p.loc + ((gq * p.M + m) * p.L + l) * (kBtP * 2));
            const float4 a = q4[0], c = q4[1];
            sx += (a.x + a.z) + (c.x + c.z);
            sy += (a.y + a.w) + (c.y + c.w);
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            sx += __shfl_xor(sx, o);
            sy += __shfl_xor(sy, o);
        }
        if (act && k == 0) {
            const float inv = 1.f / (float)(p.M * kBtP);
            const float cx = sx * inv * (float)lv.W, cy = sy * inv * (float)lv.H;
            int tx = cx > 0.f ? (int)fminf(cx, 1e9f) / kBtTileW : 0;   // NaN -> 0
            int ty = cy > 0.f ? (int)fminf(cy, 1e9f) / kBtTileH : 0;
            tx = tx >= lv.tx ? lv.tx - 1 : tx;
            ty = ty >= lv.ty ? lv.ty - 1 : ty;
            p.tile_id[((int64_t)l * p.B + b) * p.Nq + q] = (uint16_t)(ty * lv.tx + tx);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < p.L * p.M; i += 256) {
        const int l = i / p.M, m = i - l * p.M;
        if (sh_b[i]) atomicMax(&p.row_bound[((int64_t)l * p.B + b) * p.M + m], sh_b[i]);
    }
}

__global__ void __launch_bounds__(1024) bt_order_kernel(BtArgs p)
{
    __shared__ int cnt[kBtMaxTiles];
    __shared__ int part[16];
    const int l = blockIdx.x / p.B, b = blockIdx.x - l * p.B, tid = threadIdx.x;
    const BtLevel lv = bt_level(p.shapes, p.lsi, l, p.B, p.M, p.Nq);
    if (lv.mode != kBtTile) return;
    const int T = lv.parts;
    const uint16_t *id = p.tile_id + ((int64_t)l * p.B + b) * p.Nq;
    int32_t *start = p.start + ((int64_t)l * p.B + b) * (kBtMaxTiles + 1);
    int32_t *order = p.order + ((int64_t)l * p.B + b) * p.Nq;
    cnt[tid] = 0;
    __syncthreads();
    for (int q = tid; q < p.Nq; q += 1024) atomicAdd(&cnt[id[q]], 1);
    __syncthreads();
    const int c = cnt[tid];
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int n = __shfl_up(incl, o, 64);
        if ((tid & 63) >= o) incl += n;
    }
    if ((tid & 63) == 63) part[tid >> 6] = incl;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < (tid >> 6); ++w) before += part[w];
    const int excl = before + incl - c;
    if (tid < T) start[tid] = excl;
    if (tid == 0) start[T] = p.Nq;
    cnt[tid] = excl;
    __syncthreads();
    for (int q = tid; q < p.Nq; q += 1024) order[atomicAdd(&cnt[id[q]], 1)] = q;
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bt_xor1(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float bt_xor2(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
}
__device__ __forceinline__ int bt_round(float v)   // floor(v + 0.5) in one instruction
{
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
typedef float bt_f32x2_t __attribute__((ext_vector_type(2)));
// accumulate into an LDS word by byte address (no return value, nothing to wait for until the flush's barrier)
__device__ __forceinline__ void bt_lds_add(uint32_t byte_addr, int v)
{
    // (no "memory" clobber: the record reads of the next sample may pass the adds; the flush is fenced by its barrier)
    asm volatile("ds_add_u32 %0, %1" : : "v"(byte_addr), "v"(v));
}
__device__ __forceinline__ int bt_clamp(int v, int lo, int hi)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float4 as_f4(uint4 v)
{
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ int bt_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// One bilinear sample of a row, as the scatter and the gather halves need it.
struct BtSample {
    float lx, ly, a;
    int pix;            // lstart + y0 * W + x0 (may be off the level; only valid corners are dereferenced)
    uint32_t flags;     // bit c: corner c inside the image; bit 4: footprint inside the window; bit 5: inside the level
};

__device__ __forceinline__ BtSample bt_setup(float lxn, float lyn, float a, int H, int W, float fH, float fW, int lstart,
                                             int ox, int oy, int ww, int wh, bool use_window, int &x0, int &y0)
{
    BtSample s;
    const float w_im = lxn * fW - 0.5f, h_im = lyn * fH - 0.5f;
    const bool inside = h_im > -1.f && w_im > -1.f && h_im < fH && w_im < fW;
    const float fx = floorf(w_im), fy = floorf(h_im);
    s.lx = w_im - fx;
    s.ly = h_im - fy;
    s.a = a;
    x0 = inside ? (int)fx : 0;
    y0 = inside ? (int)fy : 0;
    const bool vx0 = inside && x0 >= 0, vx1 = inside && x0 + 1 <= W - 1;
    const bool vy0 = inside && y0 >= 0, vy1 = inside && y0 + 1 <= H - 1;
    s.pix = lstart + y0 * W + x0;
    const int x0v = max(x0, 0), x1v = min(x0 + 1, W - 1), y0v = max(y0, 0), y1v = min(y0 + 1, H - 1);
    const bool in_window = use_window && ox <= x0v && x1v < ox + ww && oy <= y0v && y1v < oy + wh;
    s.flags = (uint32_t)(vy0 && vx0) | ((uint32_t)(vy0 && vx1) << 1) | ((uint32_t)(vy1 && vx0) << 2) |
              ((uint32_t)(vy1 && vx1) << 3) | ((uint32_t)in_window << 4) | ((uint32_t)inside << 5);
    return s;
}

// sum over the 8 lanes of a row group of twelve values at once, complete in lanes 0-3 of the group.  Written out as
// three blocks of twelve DPP adds: left to the compiler the 36 steps became ~100 instructions (a v_mov_b32_dpp and the
// hazard no-ops in front of most adds); inside a block an add reads a register written twelve instructions earlier.
#define BT_DPP12(ctrl)                                                                                              \
    asm("s_nop 1\n"                                                                                                 \
        "v_add_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"                               \
        "v_add_f32_dpp %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"                               \
        "v_add_f32_dpp %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"                               \
        "v_add_f32_dpp %3, %3, %3 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"                               \
        "v_add_f32_dpp %4, %4, %4 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"                               \
        "v_add_f32_dpp %5, %5, %5 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"                               \
        "v_add_f32_dpp %6, %6, %6 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"                               \
        "v_add_f32_dpp %7, %7, %7 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"                               \
        "v_add_f32_dpp %8, %8, %8 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"                               \
        "v_add_f32_dpp %9, %9, %9 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"                               \
        "v_add_f32_dpp %10, %10, %10 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"                            \
        "v_add_f32_dpp %11, %11, %11 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"                            \
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), \
          "+v"(x[9]), "+v"(x[10]), "+v"(x[11]))
__device__ __forceinline__ void bt_sum8_x12(float (&x)[12])
{
    BT_DPP12("quad_perm:[1,0,3,2]");
    BT_DPP12("quad_perm:[2,3,0,1]");
    BT_DPP12("row_shl:4");
}
// LDS byte address of a window pixel from its 16-bit index: idx * 128 + base in one instruction
__device__ __forceinline__ uint32_t bt_pix_lo(uint32_t packed, uint32_t pitch, uint32_t base)
{
    uint32_t r;
    asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(r) : "v"(packed), "s"(pitch), "v"(base));
    return r;
}
__device__ __forceinline__ uint32_t bt_pix_hi(uint32_t packed, uint32_t pitch, uint32_t base)
{
    uint32_t r;
    asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(r) : "v"(packed), "s"(pitch), "v"(base));
    return r;
}

// The records the set-up lanes leave in LDS for the eight lanes of a row (32 bytes per sample, rows 144 bytes apart: the
// eight rows of a broadcast read fall on different banks; first rows 0-7 of a pass, then -- from the registers of lanes
// 32-63 -- rows 8-15):
//   scatter record: [0] corner weights x attention weight x fixed-point scale (zero: corner off the image / sample
//                       outside the window / row on the floating-point path)
//                   [1] window pixel of each corner (4 x u16) | dword 2 bit 0: the sample takes the global-atomic path
//   gather record:  [0] byte offsets of the four corners in the (image, head) value slab, kBtNoCorner where the corner
//                       is off the image (the buffer load returns zeros)
//                   [1] lx, ly, W * aw, H * aw

// An item of the work list.
struct BtItem {
    int l, H, W, lstart, mode, m, b, n, begin, ox, oy, ww, wh;
    const int32_t *ord;
    float scale, inv_scale, small_row;
    bool use_window;
};
// item number -> (level, part, head, image).  Items are numbered from the LAST level down: the whole-level items of the
// coarse levels are the longest (all rows of a query chunk), the tiles of level 0 the shortest -- long first keeps the
// tail of the launch short.  `sh_i` = the level table in LDS.
__device__ __forceinline__ void bt_item_place(int r, const int *sh_i, int L, int M, int &l, int &part, int &m, int &b)
{
    l = L - 1;
    while (r >= sh_i[8 * l + 6]) {
        r -= sh_i[8 * l + 6];
        --l;
    }
    const int parts = sh_i[8 * l + 5];
    part = r % parts;
    const int bm = r / parts;
    m = bm % M;
    b = bm / M;
}

// every wave: the item of a published slot (`sh_i` [64 + 4 slot ...]: item number, first row in the tile order, rows, bits of
// the (level, image, head) row bound); n = -1 beyond the end of the list
__device__ __forceinline__ BtItem bt_item_of(const BtArgs &p, const int *sh_i, int total, int slot)
{
    BtItem it{};
    const int *t = sh_i + 64 + 4 * slot;
    const int r = bt_uniform(t[0]);
    it.n = -1;
    if (r >= total) return it;
    int l, part, m, b;
    bt_item_place(r, sh_i, p.L, p.M, l, part, m, b);
    it.l = bt_uniform(l);
    part = bt_uniform(part);
    it.m = bt_uniform(m);
    it.b = bt_uniform(b);
    it.H = bt_uniform(sh_i[8 * it.l]);
    it.W = bt_uniform(sh_i[8 * it.l + 1]);
    it.lstart = bt_uniform(sh_i[8 * it.l + 2]);
    it.mode = bt_uniform(sh_i[8 * it.l + 3]);
    const int tiles_x = bt_uniform(sh_i[8 * it.l + 4]), parts = bt_uniform(sh_i[8 * it.l + 5]);
    if (it.mode == kBtTile) {
        it.n = bt_uniform(t[2]);
        it.ord = p.order + ((int64_t)it.l * p.B + it.b) * p.Nq + bt_uniform(t[1]);
        it.ox = kBtTileW * (part % tiles_x) - kBtHalo;
        it.oy = kBtTileH * (part / tiles_x) - kBtHalo;
        it.ww = kBtWinW;
        it.wh = kBtWinH;
    } else {
        const int per = (p.Nq + parts - 1) / parts;
        it.begin = part * per;
        it.n = min(p.Nq, it.begin + per) - it.begin;
        if (it.mode == kBtResident) {
            it.ww = it.W;
            it.wh = it.H;
        }
    }
    it.n = max(it.n, 0);
    // window and fixed-point scale: no accumulator of the item can exceed n * (largest row bound of the
    // (level, image, head)), bt_tile_id_kernel's max of max_c|g| * sum_p|aw|.  bound < 2^x -> scale 2^(29 - x).
    // A row whose own bound max_c|g| * sum_p|aw| lies more than 2^-20 below the item's accumulator bound would have
    // its contributions rounded at a quantum (2^-30 of the bound) that is coarse for THEM: one outlier query, or a
    // heavy-tailed loss scale, must not cost the small gradients their bits.  Such rows take the fp32
    // path (direct atomics), like samples outside the window.  Rows above the threshold keep a relative rounding
    // error <= 2^-11 per add before the exact integer accumulation; typical gradients are nowhere near it.
    if (it.mode != kBtDirect && it.n > 0) {
        const float bound = (float)it.n * __uint_as_float((uint32_t)bt_uniform(t[3]));
        it.small_row = bound * 0x1p-20f;
        if (bound > 0x1p-90f && bound < 0x1p+90f) {
            const int x = (int)((__float_as_uint(bound) >> 23) & 0xffu) - 126;
            it.scale = __uint_as_float((uint32_t)(29 - x + 127) << 23);
            it.inv_scale = __uint_as_float((uint32_t)(127 - (29 - x)) << 23);
            it.use_window = true;
        } else if (bound == 0.f) {
            it.use_window = true;   // nothing to scatter: every contribution is an exact zero
        }
    }
    return it;
}

// (benchmark builds only: cycle stamps per phase, summed over the waves into the workspace header, words 8...)
#ifdef BT_STAMPS
#define BT_STAMP(i)                                      \
    {                                                    \
        const long long t_now = clock64();               \
        st_acc[i] += (unsigned long long)(t_now - st_t); \
        st_t = t_now;                                    \
    }
#else
#define BT_STAMP(i)
#endif

// barrier among the scatter waves: an ever-growing LDS counter (the gather waves never meet it)
__device__ __forceinline__ void bt_scatter_barrier(int *bar, int &epoch, int lane)
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my LDS adds / stores have landed (the inline-asm adds are invisible to the compiler's counters)
    epoch += kBtScWaves;
    if (lane == 0) __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < epoch) __builtin_amdgcn_s_sleep(1);
}

struct BtCtx {   // what both kinds of wave know
    int lane, k, r8, rho, su_row, su_s;
    char *rec;
    const char *rec_rd;
    char *rec_wr;
    int64_t pix_floats;
    uint32_t pix_bytes;
};

// ------------------------------------------------------------------------------------------------
// gather waves: grad_sampling_loc, grad_attn_weight.  They take the SAME items as their workgroup's scatter waves, at
// the same time (the rows' gradient, locations and weights come from memory once for both kinds -- walking the rows on
// their own the gather waves doubled the launch's HBM fetches), 16 rows per turn.  One s_barrier per item, where the
// scatter waves are done with the item's rows and thread 0 has published the next item: then the scatter waves flush and
// the gather waves go straight on to the next item's rows.
__device__ __forceinline__ void bt_gather_role(const BtArgs &p, const int *sh_i, const BtCtx &c, int total, int rwave)
{
    const int lane = c.lane, k = c.k, r8 = c.r8;
    struct SuIn {
        float2 xy;
        float a;
    };
    struct CoIn {
        float4 g;
        int64_t row;
    };
    auto query_of = [&](const BtItem &it, int i) {
        const int ii = max(min(i, it.n - 1), 0);
        return it.ord ? it.ord[ii] : it.begin + ii;
    };
    auto load_su = [&](const BtItem &it, int q) {
        SuIn r;
        const int64_t row = ((int64_t)it.b * p.Nq + q) * p.M + it.m;
        r.xy = *reinterpret_cast<const float2 *>(p.loc + ((row * p.L + it.l) * kBtP + c.su_s) * 2);
        r.a = p.aw[(row * p.L + it.l) * kBtP + c.su_s];
        return r;
    };
    auto load_co = [&](const BtItem &it, int q) {
        CoIn r;
        r.row = ((int64_t)it.b * p.Nq + q) * p.M + it.m;
        r.g = *reinterpret_cast<const float4 *>(p.grad_out + r.row * kBtD + 4 * k);
        return r;
    };
    constexpr int kStride = kBtGaWaves * kBtPassRows;
    const int b0 = rwave * kBtPassRows;

    BtItem it = bt_item_of(p, sh_i, total, 0);
    // row numbers two turns ahead, operands one turn ahead (a row number is a trip to memory in front of its operands)
    SuIn su_n{};
    CoIn co_n0{}, co_n1{};
    int q_su = 0, q_a = 0, q_b = 0;
    auto first_requests = [&](const BtItem &t) {
        if (b0 >= t.n) return;
        su_n = load_su(t, query_of(t, b0 + c.su_row));
        co_n0 = load_co(t, query_of(t, b0 + r8));
        co_n1 = load_co(t, query_of(t, b0 + 8 + r8));
        q_su = query_of(t, b0 + kStride + c.su_row);
        q_a = query_of(t, b0 + kStride + r8);
        q_b = query_of(t, b0 + kStride + 8 + r8);
    };
    first_requests(it);
    for (int iter = 0; it.n >= 0; ++iter) {
        const int l = it.l, H = it.H, W = it.W, lstart = it.lstart, n = it.n;
        const float fW = (float)W, fH = (float)H;
        const __amdgpu_buffer_rsrc_t vrsrc = make_uniform_rsrc(
            reinterpret_cast<const char *>(p.value + ((int64_t)it.b * p.Nv * p.M + it.m) * kBtD),
            (uint32_t)(((int64_t)p.Nv * p.M - it.m) * kBtD * 4));
        auto half = [&](const CoIn &co, int hbase) {
            const bool act = hbase + r8 < n;
            uint4 q0[kBtP];
#pragma unroll
            for (int s = 0; s < kBtP; ++s) q0[s] = *reinterpret_cast<const uint4 *>(c.rec_rd + s * 32);
            float4 v[kBtP][4];
#pragma unroll
            for (int s = 0; s < kBtP; ++s) {
                v[s][0] = as_f4(buffer_load16(vrsrc, q0[s].x + 16u * k));
                v[s][1] = as_f4(buffer_load16(vrsrc, q0[s].y + 16u * k));
                v[s][2] = as_f4(buffer_load16(vrsrc, q0[s].z + 16u * k));
                v[s][3] = as_f4(buffer_load16(vrsrc, q0[s].w + 16u * k));
            }
            // the three outputs of a sample are linear in its four corner dot products: combined per lane (on the
            // lane's four channels), THEN summed over the eight lanes -- 12 sums per row instead of 16
            float o[12];
            const float4 g = co.g;
#pragma unroll
            for (int s = 0; s < kBtP; ++s) {
                const float4 q1 = *reinterpret_cast<const float4 *>(c.rec_rd + s * 32 + 16);
                float d[4];
#pragma unroll
                for (int cc = 0; cc < 4; ++cc)
                    d[cc] = g.x * v[s][cc].x + g.y * v[s][cc].y + g.z * v[s][cc].z + g.w * v[s][cc].w;
                const float lx = q1.x, ly = q1.y, hx = 1.f - lx, hy = 1.f - ly;
                o[3 * s] = hy * hx * d[0] + hy * lx * d[1] + ly * hx * d[2] + ly * lx * d[3];
                o[3 * s + 1] = q1.z * (hy * (d[1] - d[0]) + ly * (d[3] - d[2]));
                o[3 * s + 2] = q1.w * (hx * (d[2] - d[0]) + lx * (d[3] - d[1]));
            }
            bt_sum8_x12(o);
            if (act && k == 0) {
                *reinterpret_cast<float4 *>(p.grad_aw + (co.row * p.L + l) * kBtP) = make_float4(o[0], o[3], o[6], o[9]);
                float4 *gl = reinterpret_cast<float4 *>(p.grad_loc + (co.row * p.L + l) * (kBtP * 2));
                gl[0] = make_float4(o[1], o[2], o[4], o[5]);
                gl[1] = make_float4(o[7], o[8], o[10], o[11]);
            }
        };
        for (int base = b0; base < n; base += kStride) {
            const SuIn su = su_n;
            const CoIn co0 = co_n0;
            CoIn co1 = co_n1;
            // ---- set-up of the turn's 64 samples, one per lane ----
            uint4 r0;
            float4 r1;
            {
                int x0, y0;
                const BtSample sm = bt_setup(su.xy.x, su.xy.y, su.a, H, W, fH, fW, lstart, 0, 0, 0, 0, false, x0, y0);
                const uint32_t f = sm.flags;
                const uint32_t o00 = (uint32_t)sm.pix * c.pix_bytes;
                r0 = make_uint4((f & 1u) ? o00 : kBtNoCorner, (f & 2u) ? o00 + c.pix_bytes : kBtNoCorner,
                                (f & 4u) ? o00 + (uint32_t)W * c.pix_bytes : kBtNoCorner,
                                (f & 8u) ? o00 + (uint32_t)(W + 1) * c.pix_bytes : kBtNoCorner);
                r1 = make_float4(sm.lx, sm.ly, fW * sm.a, fH * sm.a);
            }
            if (lane < 32) {
                *reinterpret_cast<uint4 *>(c.rec_wr) = r0;
                *reinterpret_cast<float4 *>(c.rec_wr + 16) = r1;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            half(co0, base);
            // the next turn's operands are requested between this turn's halves, behind an explicit use of everything the
            // previous request brought: requested at the top of the turn they sat in front of the set-up's first wait, which
            // (one counter, in order) then waited for THEM -- a trip to memory per turn in plain view
            asm volatile("" : "+v"(co1.g.x), "+v"(co1.g.y), "+v"(co1.g.z), "+v"(co1.g.w), "+v"(q_su), "+v"(q_a), "+v"(q_b));
            if (base + kStride < n) {
                su_n = load_su(it, q_su);
                co_n0 = load_co(it, q_a);
                co_n1 = load_co(it, q_b);
                q_su = query_of(it, base + 2 * kStride + c.su_row);
                q_a = query_of(it, base + 2 * kStride + r8);
                q_b = query_of(it, base + 2 * kStride + 8 + r8);
            }
            if (base + 8 < n) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (lane >= 32) {
                    *reinterpret_cast<uint4 *>(c.rec_wr) = r0;
                    *reinterpret_cast<float4 *>(c.rec_wr + 16) = r1;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                half(co1, base + 8);
            }
        }
        __syncthreads();   // the item's barrier (bt_main_kernel): the next item is published
        it = bt_item_of(p, sh_i, total, (iter + 1) & 1);
        first_requests(it);
    }
}

// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBtThreads) bt_main_kernel(BtArgs p)
{
#ifdef BT_STAMPS
    unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long st_t = clock64();
    const long long st_t0 = st_t;
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t *win = reinterpret_cast<uint32_t *>(smem);
    // [8 l + f] level table; [64 + 4 s + f], s = parity of the item's turn: item number, first row in the tile order,
    // rows, bits of the (level, image, head) row bound -- written by thread 0 one item ahead (see below); [80] the scatter
    // waves' barrier counter
    int *sh_i = reinterpret_cast<int *>(smem + kBtWinBytes);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = bt_uniform(tid >> 6);
    BtCtx c;
    c.lane = lane;
    // the lane as one of the eight of a row
    c.k = lane & 7;
    c.r8 = lane >> 3;
    c.rho = c.r8 & 3;
    // the lane as the set-up of one sample
    c.su_row = lane >> 2;
    c.su_s = lane & 3;
    c.rec = smem + kBtWinBytes + 1024 + wave * kBtRecBytes;
    c.rec_rd = c.rec + c.r8 * kBtRecRow;
    c.rec_wr = c.rec + (c.su_row & 7) * kBtRecRow + c.su_s * 32;
    c.pix_floats = (int64_t)p.M * kBtD;
    c.pix_bytes = (uint32_t)(p.M * kBtD * 4);
    const int k = c.k, r8 = c.r8, rho = c.rho, su_row = c.su_row, su_s = c.su_s;
    const int64_t pix_floats = c.pix_floats;
    // LDS byte address of my accumulator in pixel 0: lane k of a row owns channels {8j + k}; instruction i of a corner
    // takes octet j = (row & 3) ^ i, so that the 32 lanes the LDS serves per clock (4 rows x 8 lanes) always cover 32
    // different banks whatever pixels the rows hit, and no two lanes of an instruction ever share an address
    const uint32_t lane_base = (uint32_t)(uint64_t)win + (uint32_t)(32 * rho + 4 * k);

    if (tid < p.L) {
        const BtLevel v = bt_level(p.shapes, p.lsi, tid, p.B, p.M, p.Nq);
        int *t = sh_i + 8 * tid;
        t[0] = v.H; t[1] = v.W; t[2] = v.start; t[3] = v.mode; t[4] = v.tx; t[5] = v.parts; t[6] = v.items;
    }
    if (tid == 0) sh_i[80] = 0;
    for (int i = tid; i < kBtWinBytes / 16; i += kBtThreads) reinterpret_cast<uint4 *>(win)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    int total = 0;
    for (int l = 0; l < p.L; ++l) total += sh_i[8 * l + 6];

    // (thread 0 and its seven fellow scatter waves run the work list; the gather waves branch off below)
    int *bar = sh_i + 80;
    int epoch = 0;

    // ---- the work list, pipelined one item deep.  An item used to begin with a chain of dependent trips to memory
    // during which the waves idled: counter -> tile's row range and bound -> row order -> row operands.  Thread 0 draws
    // item numbers TWO turns ahead and fetches the next item's row range and bound while the current item's rows are
    // processed; after the barrier in front of the flush every wave knows the next item, requests the numbers of its
    // first rows before the flush and their operands after it ----
    // thread 0: the three scalars of an item that live in memory
    auto fetch = [&](int r, int &s0, int &n, uint32_t &bound) {
        s0 = 0;
        n = 0;
        bound = 0u;
        if (r >= total) return;
        int l, part, m, b;
        bt_item_place(r, sh_i, p.L, p.M, l, part, m, b);
        if (sh_i[8 * l + 3] == kBtTile) {
            const int32_t *st = p.start + ((int64_t)l * p.B + b) * (kBtMaxTiles + 1);
            s0 = st[part];
            n = st[part + 1];   // (minus s0: by the caller, after the loads)
        }
        bound = p.row_bound[((int64_t)l * p.B + b) * p.M + m];
    };
    auto publish = [&](int slot, int r, int s0, int s1, uint32_t bound) {
        int *t = sh_i + 64 + 4 * slot;
        t[0] = r;
        t[1] = s0;
        t[2] = s1 - s0;
        t[3] = (int)bound;
    };
    auto item_of = [&](int slot) { return bt_item_of(p, sh_i, total, slot); };

    // ---- operands of a pass, loaded one pass ahead, their row numbers two (a wave's passes are kBtScWaves * 16 rows apart) ----
    struct SuIn {     // of the lane as a set-up lane: its sample's location and weight, 8 channels of its row's gradient
        float2 xy;
        float a;
        float4 g0, g1;
    };
    struct CoIn {     // of the lane as one of the eight of a row: its four channels of the row's gradient (rotated octets)
        float gs[4];
        int64_t row;
    };
    auto query_of = [&](const BtItem &it, int i) {
        const int ii = max(min(i, it.n - 1), 0);
        return it.ord ? it.ord[ii] : it.begin + ii;
    };
    auto load_su = [&](const BtItem &it, int q) {
        SuIn r;
        const int64_t row = ((int64_t)it.b * p.Nq + q) * p.M + it.m;
        r.xy = *reinterpret_cast<const float2 *>(p.loc + ((row * p.L + it.l) * kBtP + su_s) * 2);
        r.a = p.aw[(row * p.L + it.l) * kBtP + su_s];
        const float4 *gp = reinterpret_cast<const float4 *>(p.grad_out + row * kBtD + 8 * su_s);
        r.g0 = gp[0];
        r.g1 = gp[1];
        return r;
    };
    auto load_co = [&](const BtItem &it, int q) {
        CoIn r;
        r.row = ((int64_t)it.b * p.Nq + q) * p.M + it.m;
#pragma unroll
        for (int j = 0; j < 4; ++j) r.gs[j] = p.grad_out[r.row * kBtD + 8 * (rho ^ j) + k];
        return r;
    };

    // prologue: the first two item numbers in one draw, the first item fetched and published
    int r_next = 0;   // (thread 0) the item after the current one
    if (tid == 0) {
        const int r0 = atomicAdd(p.counter, 2);
        int s0, s1;
        uint32_t bd;
        fetch(r0, s0, s1, bd);
        publish(0, r0, s0, s1, bd);
        r_next = r0 + 1;
    }
    __syncthreads();   // item 0 is published: from here on the two kinds of wave meet once per item (in front of the flush)
    if (wave >= kBtScWaves) {
#ifdef BT_KO_GATHER_WAVES
        for (int iter = 0; bt_item_of(p, sh_i, total, iter & 1).n >= 0; ++iter) __syncthreads();
#else
        bt_gather_role(p, sh_i, c, total, wave - kBtScWaves);
#endif
#ifdef BT_STAMPS
        if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long *>(p.counter) + 4 + 4, (unsigned long long)(clock64() - st_t0));
#endif
        return;
    }
    BtItem it = item_of(0);
    constexpr int kStride = kBtScWaves * kBtPassRows;
    SuIn su_n{};
    CoIn co_n0{}, co_n1{};
    int q_su = 0, q_a = 0, q_b = 0;   // row numbers of the pass after the loaded one
    {
        const int b0 = wave * kBtPassRows;
        if (b0 < it.n) {
            su_n = load_su(it, query_of(it, b0 + su_row));
            co_n0 = load_co(it, query_of(it, b0 + r8));
            co_n1 = load_co(it, query_of(it, b0 + 8 + r8));
            q_su = query_of(it, b0 + kStride + su_row);
            q_a = query_of(it, b0 + kStride + r8);
            q_b = query_of(it, b0 + kStride + 8 + r8);
        }
    }

    BT_STAMP(0);   // prologue
    for (int iter = 0; it.n >= 0; ++iter) {
        // thread 0: draw the item after the next, start the next item's three scalars on their way
        int r_next2 = 0, nx_s0 = 0, nx_s1 = 0;
        uint32_t nx_bd = 0u;
        if (tid == 0) {
            r_next2 = atomicAdd(p.counter, 1);
            fetch(r_next, nx_s0, nx_s1, nx_bd);
        }
        const int l = it.l, H = it.H, W = it.W, lstart = it.lstart, n = it.n;
        const int ox = it.ox, oy = it.oy, ww = it.ww, wh = it.wh;
        const float scale = it.scale, inv_scale = it.inv_scale, small_row = it.small_row;
        const bool use_window = it.use_window;
        float *gv_base = p.grad_value + ((int64_t)it.b * p.Nv * p.M + it.m) * kBtD;
        const float fW = (float)W, fH = (float)H;

        // ---- one 8-row half: the 64 window adds of every lane; samples outside the window by global atomics ----
        auto half = [&](const CoIn &co, int hbase) {
            const bool act = hbase + r8 < n;
            float4 q1[kBtP];
            uint4 q2[kBtP];
#pragma unroll
            for (int s = 0; s < kBtP; ++s) {
                q1[s] = *reinterpret_cast<const float4 *>(c.rec_rd + s * 32);
                q2[s] = *reinterpret_cast<const uint4 *>(c.rec_rd + s * 32 + 16);
            }
            uint32_t fallback = 0;
#pragma unroll
            for (int s = 0; s < kBtP; ++s) fallback |= (q2[s].z & 1u) << s;
            BT_STAMP(2);   // records written and read back
#ifndef BT_KO_SCATTER
            if (use_window) {
                const bt_f32x2_t gs01 = {co.gs[0], co.gs[1]}, gs23 = {co.gs[2], co.gs[3]};
#pragma unroll
                for (int s = 0; s < kBtP; ++s) {
                    const uint32_t ad[4] = {bt_pix_lo(q2[s].x, 128u, lane_base), bt_pix_hi(q2[s].x, 128u, lane_base),
                                            bt_pix_lo(q2[s].y, 128u, lane_base), bt_pix_hi(q2[s].y, 128u, lane_base)};
                    const float wa[4] = {q1[s].x, q1[s].y, q1[s].z, q1[s].w};
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const bt_f32x2_t w2 = {wa[cc], wa[cc]};
                        const bt_f32x2_t p01 = gs01 * w2, p23 = gs23 * w2;   // v_pk_mul_f32
                        bt_lds_add(ad[cc], bt_round(p01.x));
                        bt_lds_add(ad[cc] ^ 32u, bt_round(p01.y));
                        bt_lds_add(ad[cc] ^ 64u, bt_round(p23.x));
                        bt_lds_add(ad[cc] ^ 96u, bt_round(p23.y));
                    }
                }
            }
#endif
            BT_STAMP(3);   // scatter
            // ---- samples that left their window (or items / rows without one): fp32 atomics on global memory, set up
            // again from the row's own location (rare) ----
            if (act && fallback) {
#pragma unroll
                for (int s = 0; s < kBtP; ++s) {
                    if (!(fallback & (1u << s))) continue;
                    const float2 xy = *reinterpret_cast<const float2 *>(p.loc + ((co.row * p.L + l) * kBtP + s) * 2);
                    const float a = p.aw[(co.row * p.L + l) * kBtP + s];
                    int x0, y0;
                    const BtSample sm = bt_setup(xy.x, xy.y, a, H, W, fH, fW, lstart, 0, 0, 0, 0, false, x0, y0);
                    const float lx = sm.lx, ly = sm.ly, hx = 1.f - lx, hy = 1.f - ly;
                    float *gp = gv_base + (int64_t)sm.pix * pix_floats + k;
                    const float wf[4] = {hy * hx * a, hy * lx * a, ly * hx * a, ly * lx * a};
                    const int64_t co_[4] = {0, pix_floats, (int64_t)W * pix_floats, (int64_t)(W + 1) * pix_floats};
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc)
                        if (sm.flags & (1u << cc)) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) unsafeAtomicAdd(gp + co_[cc] + 8 * (rho ^ j), wf[cc] * co.gs[j]);
                        }
                }
            }
            BT_STAMP(5);   // global-atomic samples
        };

        int base = wave * kBtPassRows;
#ifdef BT_KO_PASS
        base = n;
#endif
        for (; base < n; base += kStride) {
            const SuIn su = su_n;
            const CoIn co0 = co_n0;
            CoIn co1 = co_n1;
            // ---- set-up of the pass's 64 samples, one per lane ----
            float4 r0;
            uint4 r1;
            {
                const bool act_su = base + su_row < n;
                // the row's own bound (the four lanes of the row agree on it)
                float mg = fmaxf(fmaxf(fmaxf(fabsf(su.g0.x), fabsf(su.g0.y)), fmaxf(fabsf(su.g0.z), fabsf(su.g0.w))),
                                 fmaxf(fmaxf(fabsf(su.g1.x), fabsf(su.g1.y)), fmaxf(fabsf(su.g1.z), fabsf(su.g1.w))));
                mg = fmaxf(mg, bt_xor1(mg));
                mg = fmaxf(mg, bt_xor2(mg));
                float sa = fabsf(su.a);
                sa += bt_xor1(sa);
                sa += bt_xor2(sa);
                const float row_mag = mg * sa;
                const bool row_in_window = use_window && act_su && !(row_mag < small_row);   // (a NaN stays with the window's NaN handling)
                int x0, y0;
                const BtSample sm = bt_setup(su.xy.x, su.xy.y, su.a, H, W, fH, fW, lstart, ox, oy, ww, wh, row_in_window, x0, y0);
                const uint32_t f = sm.flags;
                // scaled corner weights (zero for corners off the image and for samples that leave the window: their
                // adds are exact zeros on a clamped address) and the window pixel of each corner
                const float hx = 1.f - sm.lx, hy = 1.f - sm.ly;
                const float as = (f & 16u) ? sm.a * scale : 0.f;
                r0 = make_float4((f & 1u) ? hy * hx * as : 0.f, (f & 2u) ? hy * sm.lx * as : 0.f,
                                 (f & 4u) ? sm.ly * hx * as : 0.f, (f & 8u) ? sm.ly * sm.lx * as : 0.f);
                const int cx0 = bt_clamp(x0 - ox, 0, ww - 1), cx1 = bt_clamp(x0 + 1 - ox, 0, ww - 1);
                const int cy0 = bt_clamp(y0 - oy, 0, wh - 1), cy1 = bt_clamp(y0 + 1 - oy, 0, wh - 1);
                r1 = make_uint4((uint32_t)(cy0 * ww + cx0) | ((uint32_t)(cy0 * ww + cx1) << 16),
                                (uint32_t)(cy1 * ww + cx0) | ((uint32_t)(cy1 * ww + cx1) << 16),
                                (act_su && (f & 0x30u) == 0x20u) ? 1u : 0u,   // inside the level, not in the window
                                0u);
            }
            BT_STAMP(1);   // set-up (+ the requests of the next pass)
            if (lane < 32) {
                *reinterpret_cast<float4 *>(c.rec_wr) = r0;
                *reinterpret_cast<uint4 *>(c.rec_wr + 16) = r1;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            half(co0, base);
            // the next pass's operands: requested here, behind an explicit use of everything the previous request brought
            // (see bt_gather_waves)
            asm volatile("" : "+v"(co1.gs[0]), "+v"(co1.gs[1]), "+v"(co1.gs[2]), "+v"(co1.gs[3]), "+v"(q_su), "+v"(q_a), "+v"(q_b));
            if (base + kStride < n) {
                su_n = load_su(it, q_su);
                co_n0 = load_co(it, q_a);
                co_n1 = load_co(it, q_b);
                q_su = query_of(it, base + 2 * kStride + su_row);
                q_a = query_of(it, base + 2 * kStride + r8);
                q_b = query_of(it, base + 2 * kStride + 8 + r8);
            }
            if (base + 8 < n) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (lane >= 32) {
                    *reinterpret_cast<float4 *>(c.rec_wr) = r0;
                    *reinterpret_cast<uint4 *>(c.rec_wr + 16) = r1;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                half(co1, base + 8);
            }
        }
        // thread 0: the next item's scalars (on their way since the item began) and the number of the one after it
        if (tid == 0) {
            publish((iter + 1) & 1, r_next, nx_s0, nx_s1, nx_bd);
            r_next = r_next2;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my window adds have landed (the inline-asm adds are invisible to the compiler's counters)
        __syncthreads();   // every wave of either kind is done with the item's rows, the next item is published
        BT_STAMP(6);   // waiting for the item's other waves
        // the next item; the numbers of its first rows are requested before the flush ...
        const BtItem nx = item_of((iter + 1) & 1);
        const int nb0 = wave * kBtPassRows;
        const bool nx_rows = nb0 < nx.n;
        int nq_su = 0, nq_a = 0, nq_b = 0;
        if (nx_rows) {
            nq_su = query_of(nx, nb0 + su_row);
            nq_a = query_of(nx, nb0 + r8);
            nq_b = query_of(nx, nb0 + 8 + r8);
            q_su = query_of(nx, nb0 + kStride + su_row);
            q_a = query_of(nx, nb0 + kStride + r8);
            q_b = query_of(nx, nb0 + kStride + 8 + r8);
        }
        // ---- flush: whole 128-byte lines of fp32 atomics, accumulators left at zero for the next item.  A wave takes
        // window rows wave, wave + 8, ...: the row's image coordinates and base address are scalars, a lane is (pixel
        // parity, channel) and steps two pixels at a time ----
        if (use_window && scale != 0.f && n > 0) {
            const int ch = lane & 31, h = lane >> 5;
            constexpr int kAhead = 4;
            for (int wy = wave; wy < wh; wy += kBtScWaves) {
                const int Y = oy + wy;
                if (Y < 0 || Y >= H) continue;   // (rows of the halo beyond the image hold zeros: nothing was added there)
                float *rowp = gv_base + (int64_t)(lstart + Y * W + ox) * pix_floats + ch;
                uint32_t *wrow = win + wy * ww * 32 + ch;
                for (int wx0 = h; wx0 < ww; wx0 += 2 * kAhead) {
                    int v[kAhead];
#pragma unroll
                    for (int u = 0; u < kAhead; ++u) v[u] = wx0 + 2 * u < ww ? (int)wrow[(wx0 + 2 * u) * 32] : 0;
#pragma unroll
                    for (int u = 0; u < kAhead; ++u) {
                        const int wx = wx0 + 2 * u;
                        if (v[u] != 0) {
                            wrow[wx * 32] = 0u;
#ifdef BT_KO_FLUSH
                            if (wx == -12345)
#else
                            if (ox + wx >= 0 && ox + wx < W)
#endif
                                unsafeAtomicAdd(rowp + (int64_t)wx * pix_floats, (float)v[u] * inv_scale);
                        }
                    }
                }
            }
        }
        // ... and their operands after it
        if (nx_rows) {
            su_n = load_su(nx, nq_su);
            co_n0 = load_co(nx, nq_a);
            co_n1 = load_co(nx, nq_b);
        }
        it = nx;
        BT_STAMP(7);   // flush, next item's first requests
        // the window is clean again: among the scatter waves only -- the gather waves are already in the next item's rows
        bt_scatter_barrier(bar, epoch, lane);
        BT_STAMP(8);   // waiting for the flush of the other waves
    }
#ifdef BT_STAMPS
    st_acc[9] = (unsigned long long)(clock64() - st_t0);
    if (lane == 0) {
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(p.counter) + 4;
        for (int i = 0; i < 10; ++i)
            if (i != 4) atomicAdd(dst + i, st_acc[i]);
    }
#endif
}

__global__ void __launch_bounds__(256) bt_clear_kernel(uint32_t *words, int n)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < n) words[i] = 0u;
}

}  // namespace sdetr

using namespace sdetr;

static size_t bt_align(size_t v) { return (v + 255) & ~(size_t)255; }
static size_t bt_header_bytes(int B, int M, int L) { return 256 + bt_align((size_t)L * B * M * 4); }

extern "C" size_t sdetr_msda_col2im_lds_workspace_bytes(int B, int Nq, int M, int L)
{
    if (B <= 0 || Nq <= 0 || L <= 0 || M <= 0) return 256;
    return bt_header_bytes(B, M, L) + bt_align((size_t)L * B * (kBtMaxTiles + 1) * 4) + bt_align((size_t)L * B * Nq * 4) +
           bt_align((size_t)L * B * Nq * 2);
}

extern "C" int sdetr_msda_col2im_lds_supported(int M, int D, int L, int P, int Nv)
{
    return D == kBtD && P == kBtP && L >= 1 && L <= kBtMaxL && M >= 1 && M <= kBtMaxHeads &&
           (int64_t)Nv * M * D * 4 < (int64_t)kBtNoCorner;
}

extern "C" int sdetr_msda_col2im_lds_f32(sdetr_stream_t stream, const float *grad_col, const float *value,
                                         const int64_t *shapes, const int64_t *lsi, const float *loc, const float *aw,
                                         int B, int Nv, int M, int D, int L, int Nq, int P, float *grad_value,
                                         float *grad_loc, float *grad_aw, void *workspace, size_t workspace_bytes)
{
    if (B < 0 || Nv < 0 || M <= 0 || D <= 0 || L <= 0 || Nq < 0 || P <= 0) return fail("msda_col2im_lds_f32: bad dims");
    if (!sdetr_msda_col2im_lds_supported(M, D, L, P, Nv))
        return fail("msda_col2im_lds_f32: needs D = 32, P = 4, L <= 8, M <= 64 and a value tensor below 4 GiB per image "
                    "(got D=%d P=%d L=%d M=%d)", D, P, L, M);
    if (B == 0 || Nq == 0) return 0;
    if (!grad_col || !value || !shapes || !lsi || !loc || !aw || !grad_value || !grad_loc || !grad_aw)
        return fail("msda_col2im_lds_f32: null pointer");
    if (!workspace || workspace_bytes < sdetr_msda_col2im_lds_workspace_bytes(B, Nq, M, L))
        return fail("msda_col2im_lds_f32: workspace too small");
    if ((int64_t)B * ((Nq + 31) / 32) > 0x7fffffffLL) return fail("msda_col2im_lds_f32: too many rows");
    BtArgs a{};
    a.grad_out = grad_col; a.value = value; a.shapes = shapes; a.lsi = lsi; a.loc = loc; a.aw = aw;
    a.grad_value = grad_value; a.grad_loc = grad_loc; a.grad_aw = grad_aw;
    a.B = B; a.Nv = Nv; a.M = M; a.L = L; a.Nq = Nq;
    char *ws = static_cast<char *>(workspace);
    a.counter = reinterpret_cast<int *>(ws);
    a.row_bound = reinterpret_cast<uint32_t *>(ws + 256);
    ws += bt_header_bytes(B, M, L);
    a.start = reinterpret_cast<int32_t *>(ws);
    ws += bt_align((size_t)L * B * (kBtMaxTiles + 1) * 4);
    a.order = reinterpret_cast<int32_t *>(ws);
    ws += bt_align((size_t)L * B * Nq * 4);
    a.tile_id = reinterpret_cast<uint16_t *>(ws);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // The header (work counter, per-item bounds) is cleared by a KERNEL: the hipMemsetAsync that stood here until round 4 was
    // not replayed with a captured hipGraph on this stack (ROCm 7.2 / torch 2.10: the training step's replays then ran
    // bt_main_kernel against an exhausted work counter -- no item was processed and grad_loc / grad_aw kept whatever the
    // pool memory held; found by comparing the replayed step's gradients with the eager step's, bench.py loss trace).
    {
        const int words = (int)(bt_header_bytes(B, M, L) / 4);
        hipLaunchKernelGGL(bt_clear_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s,
                           reinterpret_cast<uint32_t *>(workspace), words);
        if (int rc = check_launch("msda_col2im_lds clear")) return rc;
    }
    hipLaunchKernelGGL(bt_tile_id_kernel, dim3((unsigned)(B * ((Nq + 31) / 32))), dim3(256), 0, s, a);
    if (int rc = check_launch("msda_col2im_lds tile ids")) return rc;
    hipLaunchKernelGGL(bt_order_kernel, dim3((unsigned)(L * B)), dim3(1024), 0, s, a);
    if (int rc = check_launch("msda_col2im_lds order")) return rc;
    static DeviceOnce lds_slots;
    allow_dynamic_lds(bt_main_kernel, lds_slots, kBtLdsBytes);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    hipLaunchKernelGGL(bt_main_kernel, dim3((unsigned)cus), dim3(kBtThreads), kBtLdsBytes, s, a);
    note_backward_kernel(SDETR_KERNEL_MSDA_BWD_LDS);
    return check_launch("msda_col2im_lds");
}
