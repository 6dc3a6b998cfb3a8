// Multi-scale deformable attention forward with the value maps staged through LDS (gfx950).
//
// Why: the direct gather (msda_forward.hip) is bound by the L2 -> L1 line-fill path.  At the
// benchmark shape every (pixel, head) of the value pyramid is read ~32 times per launch (11 363
// queries x 16 samples x 4 corners over 22 323 pixels), yet the 32 KiB L1 keeps almost none of it
// (measured: identical time for spatially sorted and random query order, ~15 TB/s of line fills
// for 0.08 GB of algorithmic bytes).  Here the reuse is made explicit:
//
//  * queries are bucketed by the TY x TX level-0 region their reference point falls in
//    (region_bucket_kernel: LDS histogram + scan + scatter, one workgroup per image);
//  * one workgroup = (image, head, region).  It derives, per level, the bounding box of its
//    queries' reference points, pads it by HALO pixels, and loads that window of the head-major
//    bf16 value map into LDS ONCE with coalesced 16-byte loads (out-of-image pixels are stored as
//    zeros, which IS the reference's zero padding -- no per-corner validity tests remain);
//  * the bilinear gather then runs out of LDS: a quad of 4 lanes owns one (query, head) row, each
//    lane 8 channels = one ds_read_b128 per corner; descriptors (two packed LDS row offsets + four
//    weights) are computed once per sample by one lane of the quad (lane j owns level j: L = P = 4)
//    and broadcast through a wave-private LDS slab -- no workgroup barrier in the sample loop;
//  * a sample that falls outside its level's window (offset larger than HALO) takes a per-sample
//    global-memory fallback inside the same kernel, so the result never depends on the window.
//
// blockIdx % num_heads is the head, so (block b -> XCD b % 8) every XCD's L2 serves one head's slab.
// Supported shape: D = 32 bf16 head-major value, L = 4, P = 4 (every Salience-DETR config); anything
// else is routed to the direct kernel by the host wrapper.
#include "common.h"

namespace sdetr {

constexpr int kTX = 16, kTY = 8, kHalo = 4;  // level-0 region and window padding (pixels of each level)
constexpr int kTL = 4, kTP = 4, kTD = 32;    // levels, points, head dim
constexpr int kPixBytes = kTD * 2;           // bf16

__host__ __device__ constexpr int tile_w_cap(int l) { return (kTX >> l) + 2 * kHalo + 2; }
__host__ __device__ constexpr int tile_h_cap(int l) { return ((kTY >> l) > 0 ? (kTY >> l) : 1) + 2 * kHalo + 2; }
__host__ __device__ constexpr int tile_base_px(int l)
{
    int s = 0;
    for (int i = 0; i < l; ++i) s += tile_w_cap(i) * tile_h_cap(i);
    return s;
}
constexpr int kTilePx = tile_base_px(kTL);
constexpr int kTileBytes = kTilePx * kPixBytes;          // 65 280
constexpr int kDescW = 4 * 8 * 16 * 16;                  // [wave][sample][row] float4
constexpr int kDescO = 4 * 8 * 16 * 4;                   // [wave][sample][row] u32
constexpr int kMiscInts = 64;
constexpr int kTiledLds = kTileBytes + kDescW + kDescO + kMiscInts * 4;
static_assert(kTiledLds <= 80 * 1024, "two workgroups per CU");
static_assert(kTiledLds / 16 < 65536, "16-bit LDS offsets");

// ------------------------------------------------------------------------------------------------
struct BucketArgs {
    const float *ref;  // [B,Nq,L,ref_dim]
    int ref_dim, B, Nq, L, H0, W0, RX, RY;
    int32_t *order;         // [B,Nq]
    int32_t *region_start;  // [B,RX*RY+1]
};

__device__ __forceinline__ int region_of(const BucketArgs &p, int b, int q)
{
    const float *r = p.ref + ((int64_t)b * p.Nq + q) * p.L * p.ref_dim;
    int rx = (int)floorf(r[0] * (float)p.W0) / kTX;
    int ry = (int)floorf(r[1] * (float)p.H0) / kTY;
    rx = rx < 0 ? 0 : (rx >= p.RX ? p.RX - 1 : rx);
    ry = ry < 0 ? 0 : (ry >= p.RY ? p.RY - 1 : ry);
    return ry * p.RX + rx;
}

__global__ void __launch_bounds__(1024) region_bucket_kernel(BucketArgs p)
{
    extern __shared__ int bsm[];  // [R] counts -> cursors, [16] scan partials
    const int R = p.RX * p.RY;
    int *cnt = bsm;
    int *part = bsm + R;
    const int tid = threadIdx.x, b = blockIdx.x;
    for (int r = tid; r < R; r += 1024) cnt[r] = 0;
    __syncthreads();
    for (int q = tid; q < p.Nq; q += 1024) atomicAdd(&cnt[region_of(p, b, q)], 1);
    __syncthreads();
    // exclusive scan of cnt[0..R): each thread owns a contiguous run
    const int per = (R + 1023) / 1024;
    const int lo = tid * per, hi = min(R, lo + per);
    int local = 0;
    for (int r = lo; r < hi; ++r) local += cnt[r];
    int incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int n = __shfl_up(incl, o, 64);
        if ((tid & 63) >= o) incl += n;
    }
    if ((tid & 63) == 63) part[tid >> 6] = incl;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < (tid >> 6); ++w) before += part[w];
    int run = before + incl - local;
    int32_t *rs = p.region_start + (int64_t)b * (R + 1);
    for (int r = lo; r < hi; ++r) {
        const int c = cnt[r];
        rs[r] = run;
        cnt[r] = run;  // becomes the scatter cursor
        run += c;
    }
    if (tid == 0) rs[R] = p.Nq;
    __syncthreads();
    for (int q = tid; q < p.Nq; q += 1024) {
        const int pos = atomicAdd(&cnt[region_of(p, b, q)], 1);
        p.order[(int64_t)b * p.Nq + pos] = q;
    }
}

// ------------------------------------------------------------------------------------------------
struct TiledArgs {
    const char *value;  // [B,M,Nv,32] bf16
    const int64_t *shapes;
    const int64_t *lsi;
    const float *ref;
    int ref_dim;
    const void *proj;
    int proj_bf16;
    int64_t proj_stride;
    const int32_t *order;
    const int32_t *region_start;
    int R;
    void *out;
    int out_bf16;
    int B, Nv, M, Nq;
};

__device__ __forceinline__ void fma8(float *acc, const uint4 &v, float w)
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

// misc ints layout
enum { kLvH = 0, kLvW = 4, kLvS = 8, kBbMinX = 12, kBbMaxX = 16, kBbMinY = 20, kBbMaxY = 24, kOx = 28, kOy = 32,
       kTw = 36, kTh = 40 };

__global__ void __launch_bounds__(kBlock, 2) msda_tiled_kernel(TiledArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *tile = lds;
    float4 *dW = reinterpret_cast<float4 *>(lds + kTileBytes);
    uint32_t *dO = reinterpret_cast<uint32_t *>(lds + kTileBytes + kDescW);
    int *misc = reinterpret_cast<int *>(lds + kTileBytes + kDescW + kDescO);

    const int tid = threadIdx.x;
    const int m = blockIdx.x % p.M;
    const int br = blockIdx.x / p.M;
    const int b = br / p.R;
    const int region = br - b * p.R;
    const int rs = p.region_start[(int64_t)b * (p.R + 1) + region];
    const int nq = p.region_start[(int64_t)b * (p.R + 1) + region + 1] - rs;
    if (nq <= 0) return;  // uniform
    const int32_t *ord = p.order + (int64_t)b * p.Nq + rs;

    if (tid < kTL) {
        misc[kLvH + tid] = (int)p.shapes[2 * tid];
        misc[kLvW + tid] = (int)p.shapes[2 * tid + 1];
        misc[kLvS + tid] = (int)p.lsi[tid];
        misc[kBbMinX + tid] = 1 << 30;
        misc[kBbMinY + tid] = 1 << 30;
        misc[kBbMaxX + tid] = -(1 << 30);
        misc[kBbMaxY + tid] = -(1 << 30);
    }
    __syncthreads();

    // ---- per-level bounding box of the region's reference points (in that level's pixels) ----
    for (int i = tid; i < nq * kTL; i += kBlock) {
        const int q = ord[i >> 2];
        const int l = i & 3;
        const float *r = p.ref + (((int64_t)b * p.Nq + q) * kTL + l) * p.ref_dim;
        const int ix = (int)floorf(r[0] * (float)misc[kLvW + l] - 0.5f);
        const int iy = (int)floorf(r[1] * (float)misc[kLvH + l] - 0.5f);
        atomicMin(&misc[kBbMinX + l], ix);
        atomicMax(&misc[kBbMaxX + l], ix);
        atomicMin(&misc[kBbMinY + l], iy);
        atomicMax(&misc[kBbMaxY + l], iy);
    }
    __syncthreads();
    if (tid < kTL) {
        const int l = tid;
        const int wc = tile_w_cap(l), hc = tile_h_cap(l);
        const int ox = misc[kBbMinX + l] - kHalo, oy = misc[kBbMinY + l] - kHalo;
        misc[kOx + l] = ox;
        misc[kOy + l] = oy;
        misc[kTw + l] = min(wc, misc[kBbMaxX + l] - misc[kBbMinX + l] + 2 * kHalo + 2);
        misc[kTh + l] = min(hc, misc[kBbMaxY + l] - misc[kBbMinY + l] + 2 * kHalo + 2);
    }
    __syncthreads();

    // ---- stage the four windows: 16-byte chunks, zeros outside the image ----
    const char *vbase = p.value + ((int64_t)b * p.M + m) * p.Nv * (int64_t)kPixBytes;
#pragma unroll
    for (int l = 0; l < kTL; ++l) {
        const int tw = misc[kTw + l], th = misc[kTh + l], ox = misc[kOx + l], oy = misc[kOy + l];
        const int W = misc[kLvW + l], H = misc[kLvH + l], start = misc[kLvS + l];
        const int rowc = tw * 4;  // 16-byte chunks per window row
        const int total = th * rowc;
        unsigned char *tl = tile + tile_base_px(l) * kPixBytes;
        for (int i0 = tid; i0 < total; i0 += kBlock * 4) {
            uint4 v[4];
            int dst[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * kBlock;
                dst[u] = -1;
                v[u] = make_uint4(0u, 0u, 0u, 0u);
                if (i < total) {
                    const int y = i / rowc;
                    const int rem = i - y * rowc;
                    const int x = rem >> 2, part = rem & 3;
                    const int gy = oy + y, gx = ox + x;
                    dst[u] = (y * tile_w_cap(l) + x) * kPixBytes + part * 16;
                    if (gy >= 0 && gy < H && gx >= 0 && gx < W)
                        v[u] = *reinterpret_cast<const uint4 *>(vbase + (int64_t)(start + gy * W + gx) * kPixBytes + part * 16);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (dst[u] >= 0) *reinterpret_cast<uint4 *>(tl + dst[u]) = v[u];
        }
    }
    __syncthreads();

    // ---- rows: a quad per (query, head); lane j of the quad owns level j's four samples ----
    const int lane = tid & 63, wave = tid >> 6;
    const int qd = lane >> 2, j = lane & 3;
    float4 *myW = dW + wave * (8 * 16);
    uint32_t *myO = dO + wave * (8 * 16);
    const int Wl = misc[kLvW + j], Hl = misc[kLvH + j], startl = misc[kLvS + j];
    const int oxl = misc[kOx + j], oyl = misc[kOy + j], twl = misc[kTw + j], thl = misc[kTh + j];
    const uint32_t lvl_base16 = (uint32_t)((j == 0 ? tile_base_px(0) : j == 1 ? tile_base_px(1)
                                            : j == 2 ? tile_base_px(2) : tile_base_px(3)) * (kPixBytes / 16));
    const uint32_t pitch16 = (uint32_t)((j == 0 ? tile_w_cap(0) : j == 1 ? tile_w_cap(1)
                                         : j == 2 ? tile_w_cap(2) : tile_w_cap(3)) * (kPixBytes / 16));
    const int LP = kTL * kTP;
    const int nchunks = (nq + 63) >> 6;
    for (int c = 0; c < nchunks; ++c) {
        const int i = c * 64 + wave * 16 + qd;
        const bool active = i < nq;
        const int q = active ? ord[i] : 0;
        const int64_t bq = (int64_t)b * p.Nq + q;

        // this lane's four samples (level j, points 0..3): offsets + logits
        float ox[4], oy[4], lg[4];
        {
            const int64_t o_idx = bq * p.proj_stride + ((int64_t)m * LP + j * kTP) * 2;
            const int64_t l_idx = bq * p.proj_stride + (int64_t)p.M * LP * 2 + (int64_t)m * LP + j * kTP;
            if (p.proj_bf16) {
                const bf16_t *pp = reinterpret_cast<const bf16_t *>(p.proj);
                const uint4 o = *reinterpret_cast<const uint4 *>(pp + o_idx);
                const uint2 g = *reinterpret_cast<const uint2 *>(pp + l_idx);
                ox[0] = bf16_lo(o.x); oy[0] = bf16_hi(o.x); ox[1] = bf16_lo(o.y); oy[1] = bf16_hi(o.y);
                ox[2] = bf16_lo(o.z); oy[2] = bf16_hi(o.z); ox[3] = bf16_lo(o.w); oy[3] = bf16_hi(o.w);
                lg[0] = bf16_lo(g.x); lg[1] = bf16_hi(g.x); lg[2] = bf16_lo(g.y); lg[3] = bf16_hi(g.y);
            } else {
                const float *pp = reinterpret_cast<const float *>(p.proj);
                const float4 o0 = *reinterpret_cast<const float4 *>(pp + o_idx);
                const float4 o1 = *reinterpret_cast<const float4 *>(pp + o_idx + 4);
                const float4 g = *reinterpret_cast<const float4 *>(pp + l_idx);
                ox[0] = o0.x; oy[0] = o0.y; ox[1] = o0.z; oy[1] = o0.w;
                ox[2] = o1.x; oy[2] = o1.y; ox[3] = o1.z; oy[3] = o1.w;
                lg[0] = g.x; lg[1] = g.y; lg[2] = g.z; lg[3] = g.w;
            }
        }
        // softmax over the quad's 16 logits
        float mx = fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3]));
        mx = fmaxf(mx, __shfl_xor(mx, 1, 4));
        mx = fmaxf(mx, __shfl_xor(mx, 2, 4));
        float e[4], sum = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            e[t] = __expf(lg[t] - mx);
            sum += e[t];
        }
        sum += __shfl_xor(sum, 1, 4);
        sum += __shfl_xor(sum, 2, 4);
        const float inv = 1.f / sum;

        const float *r = p.ref + (bq * kTL + j) * p.ref_dim;
        const float rx = r[0], ry = r[1];
        float rw = 0.f, rh = 0.f;
        if (p.ref_dim == 4) {
            rw = r[2];
            rh = r[3];
        }
        // descriptors of my four samples
        float dw[4][4];
        uint32_t doff[4];
        float fx[4], fy[4], fa[4];  // kept for the fallback path
        uint32_t fb_mask = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float a = e[t] * inv;
            float x, y;
            if (p.ref_dim == 2) {
                x = rx + ox[t] / (float)Wl;
                y = ry + oy[t] / (float)Hl;
            } else {
                x = rx + ox[t] / (float)kTP * rw * 0.5f;
                y = ry + oy[t] / (float)kTP * rh * 0.5f;
            }
            const float w_im = x * (float)Wl - 0.5f, h_im = y * (float)Hl - 0.5f;
            const bool inside = h_im > -1.f && w_im > -1.f && h_im < (float)Hl && w_im < (float)Wl;
            const float ffx = floorf(w_im), ffy = floorf(h_im);
            const int x0 = (int)ffx, y0 = (int)ffy;
            const float lx = w_im - ffx, ly = h_im - ffy;
            if (!inside || !active) a = 0.f;
            const int tx = x0 - oxl, ty = y0 - oyl;
            const bool in_tile = tx >= 0 && ty >= 0 && tx + 1 < twl && ty + 1 < thl;
            const float hy = 1.f - ly, hx = 1.f - lx;
            dw[t][0] = hy * hx * a;
            dw[t][1] = hy * lx * a;
            dw[t][2] = ly * hx * a;
            dw[t][3] = ly * lx * a;
            if (a == 0.f || in_tile) {
                const uint32_t r0 = (a == 0.f && !in_tile) ? lvl_base16 : lvl_base16 + (uint32_t)ty * pitch16 + (uint32_t)tx * 4u;
                const uint32_t r1 = (a == 0.f && !in_tile) ? lvl_base16 : r0 + pitch16;
                doff[t] = r0 | (r1 << 16);
                if (a == 0.f && !in_tile) dw[t][0] = dw[t][1] = dw[t][2] = dw[t][3] = 0.f;
            } else {
                doff[t] = 0xffffffffu;  // outside the staged window: global fallback
                fb_mask |= 1u << t;
            }
            fx[t] = w_im;
            fy[t] = h_im;
            fa[t] = a;
        }

        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;

#pragma unroll
        for (int round = 0; round < 2; ++round) {
            // lanes 2*round, 2*round+1 of every quad publish their samples: slot u = (j & 1) * 4 + t
            if ((j >> 1) == round) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int u = (j & 1) * 4 + t;
                    myW[u * 16 + qd] = make_float4(dw[t][0], dw[t][1], dw[t][2], dw[t][3]);
                    myO[u * 16 + qd] = doff[t];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float4 w = myW[u * 16 + qd];
                const uint32_t o = myO[u * 16 + qd];
                if (o != 0xffffffffu) {
                    const unsigned char *a0 = tile + (o & 0xffffu) * 16u + j * 16;
                    const unsigned char *a1 = tile + (o >> 16) * 16u + j * 16;
                    const uint4 v00 = *reinterpret_cast<const uint4 *>(a0);
                    const uint4 v01 = *reinterpret_cast<const uint4 *>(a0 + kPixBytes);
                    const uint4 v10 = *reinterpret_cast<const uint4 *>(a1);
                    const uint4 v11 = *reinterpret_cast<const uint4 *>(a1 + kPixBytes);
                    fma8(acc, v00, w.x);
                    fma8(acc, v01, w.y);
                    fma8(acc, v10, w.z);
                    fma8(acc, v11, w.w);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }

        // ---- rare: samples outside the staged windows, straight from global memory ----
        if (__any(fb_mask != 0)) {
#pragma unroll
            for (int owner = 0; owner < 4; ++owner) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int src = (lane & ~3) | owner;
                    const int flag = __shfl((int)((fb_mask >> t) & 1u), src, 64);
                    if (!__any(flag)) continue;
                    const float w_im = __shfl(fx[t], src, 64), h_im = __shfl(fy[t], src, 64);
                    const float a = __shfl(fa[t], src, 64);
                    const int W = misc[kLvW + owner], H = misc[kLvH + owner], start = misc[kLvS + owner];
                    if (flag) {
                        const float ffx = floorf(w_im), ffy = floorf(h_im);
                        const int x0 = (int)ffx, y0 = (int)ffy, x1 = x0 + 1, y1 = y0 + 1;
                        const float lx = w_im - ffx, ly = h_im - ffy, hx = 1.f - lx, hy = 1.f - ly;
                        const char *lb = vbase + j * 16;
                        if (y0 >= 0 && x0 >= 0)
                            fma8(acc, *reinterpret_cast<const uint4 *>(lb + (int64_t)(start + y0 * W + x0) * kPixBytes), hy * hx * a);
                        if (y0 >= 0 && x1 <= W - 1)
                            fma8(acc, *reinterpret_cast<const uint4 *>(lb + (int64_t)(start + y0 * W + x1) * kPixBytes), hy * lx * a);
                        if (y1 <= H - 1 && x0 >= 0)
                            fma8(acc, *reinterpret_cast<const uint4 *>(lb + (int64_t)(start + y1 * W + x0) * kPixBytes), ly * hx * a);
                        if (y1 <= H - 1 && x1 <= W - 1)
                            fma8(acc, *reinterpret_cast<const uint4 *>(lb + (int64_t)(start + y1 * W + x1) * kPixBytes), ly * lx * a);
                    }
                }
            }
        }

        if (active) {
            const int64_t o = (bq * p.M + m) * kTD + j * 8;
            if (p.out_bf16) {
                *reinterpret_cast<uint4 *>(reinterpret_cast<bf16_t *>(p.out) + o) =
                    make_uint4(pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]), pack_bf16x2(acc[4], acc[5]),
                               pack_bf16x2(acc[6], acc[7]));
            } else {
                float *out = reinterpret_cast<float *>(p.out) + o;
                *reinterpret_cast<float4 *>(out) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                *reinterpret_cast<float4 *>(out + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
            }
        }
    }
}

}  // namespace sdetr

using namespace sdetr;

extern "C" void sdetr_tiled_config(int *region_w, int *region_h, int *halo)
{
    if (region_w) *region_w = kTX;
    if (region_h) *region_h = kTY;
    if (halo) *halo = kHalo;
}

extern "C" int sdetr_region_bucket(sdetr_stream_t stream, const float *ref_points, int ref_dim, int B, int Nq, int L,
                                   int level0_h, int level0_w, int32_t *order, int32_t *region_start)
{
    if (B < 0 || Nq < 0 || L <= 0 || level0_h <= 0 || level0_w <= 0) return fail("region_bucket: bad dims");
    if (ref_dim != 2 && ref_dim != 4)
        return fail("Last dim of reference_points must be 2 or 4, but get %d instead.", ref_dim);
    if (!ref_points || !order || !region_start) return fail("region_bucket: null pointer");
    BucketArgs a{};
    a.ref = ref_points; a.ref_dim = ref_dim; a.B = B; a.Nq = Nq; a.L = L; a.H0 = level0_h; a.W0 = level0_w;
    a.RX = (level0_w + kTX - 1) / kTX;
    a.RY = (level0_h + kTY - 1) / kTY;
    a.order = order; a.region_start = region_start;
    const int R = a.RX * a.RY;
    if (R > 8192) return fail("region_bucket: %d regions exceed the LDS histogram", R);
    if (B == 0) return 0;
    hipLaunchKernelGGL(region_bucket_kernel, dim3((unsigned)B), dim3(1024), (size_t)(R + 16) * 4, stream, a);
    return check_launch("region_bucket");
}

extern "C" int sdetr_msda_tiled_forward(sdetr_stream_t stream, const void *value_hm, const int64_t *shapes,
                                        const int64_t *lsi, const float *ref, int ref_dim, const void *proj,
                                        int proj_dtype, int64_t proj_row_stride, const int32_t *order,
                                        const int32_t *region_start, int num_regions, int B, int Nv, int M, int D,
                                        int L, int Nq, int P, void *out, int out_dtype)
{
    if (B < 0 || Nv < 0 || M <= 0 || Nq < 0 || num_regions <= 0) return fail("msda_tiled_forward: bad dims");
    if (D != kTD || L != kTL || P != kTP)
        return fail("msda_tiled_forward: only head_dim=32, 4 levels, 4 points (got D=%d L=%d P=%d)", D, L, P);
    if (ref_dim != 2 && ref_dim != 4)
        return fail("Last dim of reference_points must be 2 or 4, but get %d instead.", ref_dim);
    if (!value_hm || !shapes || !lsi || !ref || !proj || !order || !region_start || !out)
        return fail("msda_tiled_forward: null pointer");
    if (proj_row_stride < (int64_t)M * L * P * 3 || (proj_row_stride % 8) != 0)
        return fail("msda_tiled_forward: proj row stride must be >= 3*M*L*P and a multiple of 8");
    if ((int64_t)B * Nq == 0) return 0;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(msda_tiled_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kTiledLds);
        attr_set = true;
    }
    TiledArgs a{};
    a.value = reinterpret_cast<const char *>(value_hm); a.shapes = shapes; a.lsi = lsi; a.ref = ref; a.ref_dim = ref_dim;
    a.proj = proj; a.proj_bf16 = (proj_dtype == SDETR_BF16); a.proj_stride = proj_row_stride; a.order = order;
    a.region_start = region_start; a.R = num_regions; a.out = out; a.out_bf16 = (out_dtype == SDETR_BF16);
    a.B = B; a.Nv = Nv; a.M = M; a.Nq = Nq;
    const int64_t blocks = (int64_t)B * num_regions * M;
    if (blocks > 0x7fffffffLL) return fail("msda_tiled_forward: grid too large");
    hipLaunchKernelGGL(msda_tiled_kernel, dim3((unsigned)blocks), dim3(kBlock), kTiledLds, stream, a);
    return check_launch("msda_tiled");
}
