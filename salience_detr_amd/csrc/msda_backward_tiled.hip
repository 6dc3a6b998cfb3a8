// Multi-scale deformable attention backward with grad_value accumulated in LDS (gfx950).
//
// The first backward (msda_backward.hip) issues one global fp32 atomic per (sample, corner, channel):
// 372 M atomics for encoder layer 0 at batch 2, measured 2.5-4.2 ms per launch (0.7 % of the HBM roofline) --
// the same design as the reference's kernels (models/bricks/ops/cuda/ms_deform_im2col_cuda.cuh:76-148).
// Here the scatter is made local first:
//
//  * queries are bucketed by the level-0 region of their (head-averaged) sampling position -- the same
//    region_bucket_kernel as the LDS-staged forward;
//  * one 512-thread workgroup = (image, head, region) keeps an fp32 gradient window of every level in LDS
//    (1020 pixels x 32 channels x 4 B = 127.5 KB), zeroed once;
//  * 8 lanes own one (query, head) row, 4 channels each: per sample they read the four value corners
//    (16 B per lane, needed for grad_attn_weight / grad_sampling_loc), reduce the three per-sample sums across
//    the 8 lanes with shuffles, and add  w_corner * attn * grad_out  into the LDS window with ds_add_f32 --
//    no global atomic in the loop; samples outside the window fall back to global atomics (rare);
//  * at the end the window is flushed with coalesced global atomics (windows of neighbouring regions overlap
//    by the halo): 75 M adds in full cache lines instead of 372 M scattered ones.
//
// Shape support: fp32 reference layout [B,Nv,M,32], L = 4, P = 4; other shapes keep the atomic kernel.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "common.h"
#include "tiled_geometry.h"

namespace sdetr {

int launch_region_bucket_from_loc(hipStream_t stream, const float *loc, const int64_t *shapes, int B, int Nq, int M, int L,
                                  int P, int level0_h, int level0_w, int32_t *order, int32_t *region_start,
                                  int32_t *region_box);

constexpr int kBT = 512;                         // threads per workgroup (8 waves)
constexpr int kBWaves = kBT / 64;
constexpr int kGradTileBytes = kTilePx * kTD * 4;  // 130 560
constexpr int kBDesc = kBWaves * 8 * 8 * 32;       // [wave][sample][row] x (uint4 + float4)
constexpr int kBwdLds = kGradTileBytes + kBDesc;
static_assert(kBwdLds <= 160 * 1024, "one workgroup per CU");

struct BwdTiledArgs {
    const float *grad_out;  // [B,Nq,M*32]
    const float *value;     // [B,Nv,M,32]
    const int64_t *shapes;
    const int64_t *lsi;
    const float *loc;       // [B,Nq,M,4,4,2]
    const float *aw;        // [B,Nq,M,4,4]
    const int32_t *order;
    const int32_t *region_start;
    const int32_t *region_box;
    int R;
    float *grad_value;
    float *grad_loc;
    float *grad_aw;
    int B, Nv, M, Nq;
};

__global__ void __launch_bounds__(kBT) msda_col2im_tiled_kernel(BwdTiledArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float *tile = reinterpret_cast<float *>(lds);
    uint4 *dA = reinterpret_cast<uint4 *>(lds + kGradTileBytes);            // [wave][8][8]
    float4 *dB = reinterpret_cast<float4 *>(lds + kGradTileBytes + kBDesc / 2);

    const int tid = threadIdx.x;
    const int m = blockIdx.x % p.M;
    const int br = blockIdx.x / p.M;
    const int b = br / p.R;
    const int region = br - b * p.R;
    const int rs = p.region_start[(int64_t)b * (p.R + 1) + region];
    const int nq = p.region_start[(int64_t)b * (p.R + 1) + region + 1] - rs;
    if (nq <= 0) return;
    const int32_t *ord = p.order + (int64_t)b * p.Nq + rs;

    int LW[kTL], LH[kTL], LS[kTL], OX[kTL], OY[kTL], TW[kTL], TH[kTL];
    {
        const int4 *boxes = reinterpret_cast<const int4 *>(p.region_box) + ((int64_t)b * p.R + region) * kTL;
#pragma unroll
        for (int l = 0; l < kTL; ++l) {
            LH[l] = (int)p.shapes[2 * l];
            LW[l] = (int)p.shapes[2 * l + 1];
            LS[l] = (int)p.lsi[l];
            const int4 bx = boxes[l];
            OX[l] = bx.x - kHalo;
            OY[l] = bx.y - kHalo;
            TW[l] = min(tile_w_cap(l), bx.z - bx.x + 2 * kHalo + 2);
            TH[l] = min(tile_h_cap(l), bx.w - bx.y + 2 * kHalo + 2);
        }
    }
    // zero the gradient windows
    for (int i = tid; i < kTilePx * kTD / 4; i += kBT) reinterpret_cast<float4 *>(tile)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int r8 = lane >> 3, j = lane & 7;  // row within the wave, 4-channel group
    uint4 *myA = dA + wave * 64;
    float4 *myB = dB + wave * 64;
    const int64_t pix_stride = (int64_t)p.M * kTD;  // floats between pixels of the token-major value
    const float *vbase = p.value + ((int64_t)b * p.Nv) * pix_stride + m * kTD + j * 4;
    float *gbase = p.grad_value + ((int64_t)b * p.Nv) * pix_stride + m * kTD + j * 4;

    const int nchunks = (nq + 63) >> 6;
    for (int c = 0; c < nchunks; ++c) {
        const int i = c * 64 + wave * 8 + r8;
        const bool active = i < nq;
        const int q = active ? ord[i] : 0;
        const int64_t row = ((int64_t)b * p.Nq + q) * p.M + m;
        const float4 go = active ? *reinterpret_cast<const float4 *>(p.grad_out + row * kTD + j * 4)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
        for (int round = 0; round < 2; ++round) {
            // lane j of the row prepares sample s = round*8 + j
            {
                const int s = round * 8 + j;
                const int l = s >> 2;
                const int W = l == 0 ? LW[0] : l == 1 ? LW[1] : l == 2 ? LW[2] : LW[3];
                const int H = l == 0 ? LH[0] : l == 1 ? LH[1] : l == 2 ? LH[2] : LH[3];
                const int S = l == 0 ? LS[0] : l == 1 ? LS[1] : l == 2 ? LS[2] : LS[3];
                const int ox = l == 0 ? OX[0] : l == 1 ? OX[1] : l == 2 ? OX[2] : OX[3];
                const int oy = l == 0 ? OY[0] : l == 1 ? OY[1] : l == 2 ? OY[2] : OY[3];
                const int tw = l == 0 ? TW[0] : l == 1 ? TW[1] : l == 2 ? TW[2] : TW[3];
                const int th = l == 0 ? TH[0] : l == 1 ? TH[1] : l == 2 ? TH[2] : TH[3];
                const int wcap = l == 0 ? tile_w_cap(0) : l == 1 ? tile_w_cap(1) : l == 2 ? tile_w_cap(2) : tile_w_cap(3);
                const int lbase = l == 0 ? tile_base_px(0) : l == 1 ? tile_base_px(1) : l == 2 ? tile_base_px(2) : tile_base_px(3);
                const float2 xy = reinterpret_cast<const float2 *>(p.loc)[row * 16 + s];
                float a = p.aw[row * 16 + s];
                const float h_im = xy.y * (float)H - 0.5f, w_im = xy.x * (float)W - 0.5f;
                const bool inside = active && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
                const float fy = floorf(h_im), fx = floorf(w_im);
                const int y0 = inside ? (int)fy : 0, x0 = inside ? (int)fx : 0;
                uint32_t flags = 0;
                if (inside) {
                    flags = (uint32_t)(y0 >= 0 && x0 >= 0) | ((uint32_t)(y0 >= 0 && x0 + 1 <= W - 1) << 1) |
                            ((uint32_t)(y0 + 1 <= H - 1 && x0 >= 0) << 2) | ((uint32_t)(y0 + 1 <= H - 1 && x0 + 1 <= W - 1) << 3);
                } else {
                    a = 0.f;
                }
                const int tx = x0 - ox, ty = y0 - oy;
                const bool in_tile = tx >= 0 && ty >= 0 && tx + 1 < tw && ty + 1 < th;
                if (in_tile) flags |= 16u;
                const uint32_t toff = in_tile ? (uint32_t)(lbase + ty * wcap + tx) : 0u;  // window pixel index of (y0,x0)
                myA[j * 8 + r8] = make_uint4(toff | ((uint32_t)wcap << 16), (uint32_t)(S + y0 * W + x0),
                                             flags | ((uint32_t)l << 8), (uint32_t)W);
                myB[j * 8 + r8] = make_float4(h_im - fy, w_im - fx, a, 0.f);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
#pragma unroll 2
            for (int u = 0; u < 8; ++u) {
                const uint4 A = myA[u * 8 + r8];
                const float4 Bf = myB[u * 8 + r8];
                const float ly = Bf.x, lx = Bf.y, a = Bf.z;
                const uint32_t flags = A.z;
                const int W = (int)A.w;
                const int gpix = (int)A.y;
                const float hy = 1.f - ly, hx = 1.f - lx;
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 v00 = (flags & 1u) ? *reinterpret_cast<const float4 *>(vbase + (int64_t)gpix * pix_stride) : z;
                const float4 v01 = (flags & 2u) ? *reinterpret_cast<const float4 *>(vbase + (int64_t)(gpix + 1) * pix_stride) : z;
                const float4 v10 = (flags & 4u) ? *reinterpret_cast<const float4 *>(vbase + (int64_t)(gpix + W) * pix_stride) : z;
                const float4 v11 = (flags & 8u) ? *reinterpret_cast<const float4 *>(vbase + (int64_t)(gpix + W + 1) * pix_stride) : z;
                const float w00 = hy * hx, w01 = hy * lx, w10 = ly * hx, w11 = ly * lx;
                float s_aw = 0.f, s_x = 0.f, s_y = 0.f;
#define SDETR_CH(cc)                                                                             \
    {                                                                                            \
        const float gc = go.cc;                                                                  \
        s_aw += gc * (w00 * v00.cc + w01 * v01.cc + w10 * v10.cc + w11 * v11.cc);                \
        s_x += gc * (hy * (v01.cc - v00.cc) + ly * (v11.cc - v10.cc));                           \
        s_y += gc * (hx * (v10.cc - v00.cc) + lx * (v11.cc - v01.cc));                           \
    }
                SDETR_CH(x) SDETR_CH(y) SDETR_CH(z) SDETR_CH(w)
#undef SDETR_CH
                const float4 ga = make_float4(go.x * a, go.y * a, go.z * a, go.w * a);
                if (flags & 16u) {  // whole 2x2 footprint inside the LDS window (out-of-image cells are dropped at the flush)
                    const int wcap = (int)(A.x >> 16);
                    float *t00 = tile + (A.x & 0xffffu) * kTD + j * 4;
                    float *t10 = t00 + wcap * kTD;
#define SDETR_LADD(dst, wgt)                                                                     \
    atomicAdd((dst) + 0, (wgt) * ga.x); atomicAdd((dst) + 1, (wgt) * ga.y);                     \
    atomicAdd((dst) + 2, (wgt) * ga.z); atomicAdd((dst) + 3, (wgt) * ga.w);
                    SDETR_LADD(t00, w00) SDETR_LADD(t00 + kTD, w01) SDETR_LADD(t10, w10) SDETR_LADD(t10 + kTD, w11)
#undef SDETR_LADD
                } else if (flags & 15u) {  // outside the window: global atomics, as the first backward
#define SDETR_GADD(bit, pix, wgt)                                                                \
    if (flags & bit) {                                                                           \
        float *dst = gbase + (int64_t)(pix) * pix_stride;                                        \
        unsafeAtomicAdd(dst + 0, (wgt) * ga.x); unsafeAtomicAdd(dst + 1, (wgt) * ga.y);          \
        unsafeAtomicAdd(dst + 2, (wgt) * ga.z); unsafeAtomicAdd(dst + 3, (wgt) * ga.w);          \
    }
                    SDETR_GADD(1u, gpix, w00) SDETR_GADD(2u, gpix + 1, w01) SDETR_GADD(4u, gpix + W, w10)
                    SDETR_GADD(8u, gpix + W + 1, w11)
#undef SDETR_GADD
                }
#pragma unroll
                for (int sh = 1; sh < 8; sh <<= 1) {
                    s_aw += __shfl_xor(s_aw, sh, 8);
                    s_x += __shfl_xor(s_x, sh, 8);
                    s_y += __shfl_xor(s_y, sh, 8);
                }
                if (active && j == u) {
                    const int l = (int)((flags >> 8) & 0xffu);
                    const int H = l == 0 ? LH[0] : l == 1 ? LH[1] : l == 2 ? LH[2] : LH[3];
                    const int64_t si = row * 16 + round * 8 + u;
                    p.grad_aw[si] = s_aw;
                    reinterpret_cast<float2 *>(p.grad_loc)[si] = make_float2((float)W * s_x * a, (float)H * s_y * a);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();

    // ---- flush the windows: coalesced global atomics, in-image pixels only, untouched cells skipped ----
#pragma unroll
    for (int l = 0; l < kTL; ++l) {
        const int tw = TW[l], th = TH[l];
        const int cells = th * tw * kTD;
        for (int i = tid; i < cells; i += kBT) {
            const int ch = i & (kTD - 1);
            const int px = i >> 5;
            const int y = px / tw, x = px - y * tw;
            const int gy = OY[l] + y, gx = OX[l] + x;
            if (gy < 0 || gy >= LH[l] || gx < 0 || gx >= LW[l]) continue;
            const float v = tile[(tile_base_px(l) + y * tile_w_cap(l) + x) * kTD + ch];
            if (v != 0.f)
                unsafeAtomicAdd(p.grad_value + ((int64_t)b * p.Nv + LS[l] + gy * LW[l] + gx) * pix_stride + m * kTD + ch, v);
        }
    }
}

}  // namespace sdetr

using namespace sdetr;

// Workspace layout (int32): order [B*Nq] | region_start [B*(R+1)] | region_box [B*R*16]
extern "C" size_t sdetr_msda_col2im_tiled_workspace_bytes(int B, int Nq, int level0_h, int level0_w)
{
    const int R = ((level0_w + kTX - 1) / kTX) * ((level0_h + kTY - 1) / kTY);
    return ((size_t)B * Nq + (size_t)B * (R + 1) + (size_t)B * R * 16 + 16) * sizeof(int32_t);
}

extern "C" int sdetr_msda_col2im_tiled_f32(sdetr_stream_t stream, const float *grad_col, const float *value,
                                           const int64_t *shapes, const int64_t *lsi, const float *loc, const float *aw,
                                           int B, int Nv, int M, int D, int L, int Nq, int P, int level0_h, int level0_w,
                                           float *grad_value, float *grad_loc, float *grad_aw, void *workspace,
                                           size_t workspace_bytes)
{
    if (B < 0 || Nv < 0 || M <= 0 || Nq < 0 || level0_h <= 0 || level0_w <= 0) return fail("msda_col2im_tiled: bad dims");
    if (D != kTD || L != kTL || P != kTP)
        return fail("msda_col2im_tiled: only head_dim=32, 4 levels, 4 points (got D=%d L=%d P=%d)", D, L, P);
    if (!grad_col || !value || !shapes || !lsi || !loc || !aw || !grad_value || !grad_loc || !grad_aw)
        return fail("msda_col2im_tiled: null pointer");
    if ((int64_t)B * Nq == 0) return 0;
    const size_t need = sdetr_msda_col2im_tiled_workspace_bytes(B, Nq, level0_h, level0_w);
    if (!workspace || workspace_bytes < need)
        return fail("msda_col2im_tiled: needs %zu bytes of workspace, got %zu", need, workspace_bytes);
    const int R = ((level0_w + kTX - 1) / kTX) * ((level0_h + kTY - 1) / kTY);
    int32_t *order = reinterpret_cast<int32_t *>(workspace);
    int32_t *region_start = order + (size_t)B * Nq;
    int32_t *region_box = region_start + (((size_t)B * (R + 1) + 3) & ~(size_t)3);  // int4-aligned
    if (int e = launch_region_bucket_from_loc(stream, loc, shapes, B, Nq, M, L, P, level0_h, level0_w, order, region_start,
                                              region_box))
        return e;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(msda_col2im_tiled_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kBwdLds);
        attr_set = true;
    }
    BwdTiledArgs a{};
    a.grad_out = grad_col; a.value = value; a.shapes = shapes; a.lsi = lsi; a.loc = loc; a.aw = aw;
    a.order = order; a.region_start = region_start; a.region_box = region_box; a.R = R;
    a.grad_value = grad_value; a.grad_loc = grad_loc; a.grad_aw = grad_aw;
    a.B = B; a.Nv = Nv; a.M = M; a.Nq = Nq;
    const int64_t blocks = (int64_t)B * R * M;
    hipLaunchKernelGGL(msda_col2im_tiled_kernel, dim3((unsigned)blocks), dim3(kBT), kBwdLds, stream, a);
    return check_launch("msda_col2im_tiled");
}
