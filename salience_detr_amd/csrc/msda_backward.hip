// Multi-scale deformable attention, backward (grad_value scatter, grad_sampling_loc,
// grad_attn_weight), for gfx950.  Arithmetic follows the reference
// (models/bricks/ops/cuda/ms_deform_im2col_cuda.cuh:76-148: grad_value += w_i*g*aw,
// grad_aw = g*val, grad_loc = (W*gw, H*gh)*g*aw, summed over the D channels of the head).
//
// Structure (not the reference's 32-thread blocks with serial thread-0 reductions, :290-392): sample set-up
// is done once per sample by one lane and broadcast through LDS; the three per-sample channel sums are
// reduced with wavefront shuffles (no barrier); grad_value goes out as hardware fp32 atomics
// (global_atomic_add_f32).  Summation order of those atomics is not deterministic, exactly as in the reference.
//
// Two lane mappings:
//  * msda_col2im_chan_kernel (D = 32 / 64, the shapes that matter): ONE CHANNEL PER LANE, D adjacent lanes own a
//    (b,q,m) row, so every atomic wave-instruction covers whole 128-byte lines (2 rows x 32 channels).  Measured
//    on MI355X the L2 retires atomics per line-sized request, not per element: 64 consecutive floats per
//    instruction sustain ~680 G adds/s, the 8-lanes-x-4-channels mapping below (8 partial lines per instruction)
//    only ~88 G adds/s -- 4.4 ms vs ~0.6 ms for encoder layer 0 at batch 2.  (Accumulating in LDS instead was
//    tried and dropped: ds_add_f32 is far slower still, 2.6 ms for the same work.)
//  * msda_col2im_kernel: G = D/4 lanes per row, 4 channels per lane -- kept for the remaining head dims.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "common.h"

namespace sdetr {

void note_backward_kernel(int which);  // abi.hip

constexpr int kBChunk = 16;

struct BackwardArgs {
    const float *grad_out;
    const char *value;
    const int64_t *shapes;
    const int64_t *lsi;
    const float *loc;
    const float *aw;
    char *grad_value;
    float *grad_loc;
    float *grad_aw;
    int B, Nv, M, L, Nq, P;
    int nchunk;
};

template <int D, bool HEAD_MAJOR>
__global__ void __launch_bounds__(kBlock) msda_col2im_kernel(BackwardArgs p)
{
    constexpr int G = D / 4;
    constexpr int GPB = kBlock / G;
    constexpr int DSTRIDE = kBChunk * 8 + 4;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *desc = smem;
    int *lvl_tab = reinterpret_cast<int *>(smem + GPB * DSTRIDE);

    const int tid = threadIdx.x;
    const int m = blockIdx.x % p.M;
    const int chunk_global = blockIdx.x / p.M;
    const int b = chunk_global / p.nchunk;
    const int chunk = chunk_global - b * p.nchunk;
    const int g = tid / G, j = tid - g * G;
    const int q = chunk * GPB + g;
    const bool active = q < p.Nq;

    if (tid < p.L) {
        lvl_tab[tid * 3 + 0] = (int)p.shapes[2 * tid];
        lvl_tab[tid * 3 + 1] = (int)p.shapes[2 * tid + 1];
        lvl_tab[tid * 3 + 2] = (int)p.lsi[tid];
    }
    __syncthreads();

    const int LP = p.L * p.P;
    const int64_t row = ((int64_t)b * p.Nq + (active ? q : 0)) * p.M + m;
    const uint32_t pixel_bytes = HEAD_MAJOR ? (uint32_t)(D * 4) : (uint32_t)(p.M * D * 4);
    const int64_t img_off = HEAD_MAJOR ? ((int64_t)b * p.M + m) * p.Nv * (int64_t)(D * 4)
                                       : (int64_t)b * p.Nv * (int64_t)pixel_bytes;
    const uint32_t lane_off = (HEAD_MAJOR ? 0u : (uint32_t)(m * D * 4)) + (uint32_t)(j * 16);
    const char *vbase = p.value + img_off + lane_off;
    char *gbase = p.grad_value + img_off + lane_off;

    float4 go = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) go = *reinterpret_cast<const float4 *>(p.grad_out + row * D + j * 4);

    uint32_t *my_desc = desc + g * DSTRIDE;
    for (int c0 = 0; c0 < LP; c0 += kBChunk) {
        const int ns = min(kBChunk, LP - c0);
        if (c0 > 0) __syncthreads();
        if (active) {
            for (int t = j; t < ns; t += G) {
                const int s = c0 + t;
                const int l = s / p.P;
                const int H = lvl_tab[l * 3], W = lvl_tab[l * 3 + 1], start = lvl_tab[l * 3 + 2];
                const float2 xy = reinterpret_cast<const float2 *>(p.loc)[row * LP + s];
                const float a = p.aw[row * LP + s];
                const float h_im = xy.y * (float)H - 0.5f, w_im = xy.x * (float)W - 0.5f;
                const bool inside = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
                const float fy = floorf(h_im), fx = floorf(w_im);
                int y0 = inside ? (int)fy : 0, x0 = inside ? (int)fx : 0;
                const int y1 = y0 + 1, x1 = x0 + 1;
                const bool y0ok = inside && y0 >= 0, x0ok = inside && x0 >= 0;
                const bool y1ok = inside && y1 <= H - 1, x1ok = inside && x1 <= W - 1;
                const int y0c = y0 >= 0 ? y0 : 0, x0c = x0 >= 0 ? x0 : 0;
                const int y1c = y1 <= H - 1 ? y1 : H - 1, x1c = x1 <= W - 1 ? x1 : W - 1;
                const uint32_t r0 = (uint32_t)(start + y0c * W), r1 = (uint32_t)(start + y1c * W);
                const uint32_t flags = (uint32_t)(y0ok && x0ok) | ((uint32_t)(y0ok && x1ok) << 1) |
                                       ((uint32_t)(y1ok && x0ok) << 2) | ((uint32_t)(y1ok && x1ok) << 3) |
                                       ((uint32_t)l << 8);
                *reinterpret_cast<uint4 *>(my_desc + t * 8) =
                    make_uint4((r0 + x0c) * pixel_bytes, (r0 + x1c) * pixel_bytes, (r1 + x0c) * pixel_bytes,
                               (r1 + x1c) * pixel_bytes);
                *reinterpret_cast<uint4 *>(my_desc + t * 8 + 4) =
                    make_uint4(__float_as_uint(h_im - fy), __float_as_uint(w_im - fx), __float_as_uint(a), flags);
            }
        }
        __syncthreads();
        if (active) {
            for (int t = 0; t < ns; ++t) {
                const uint4 o = *reinterpret_cast<const uint4 *>(my_desc + t * 8);
                const uint4 e = *reinterpret_cast<const uint4 *>(my_desc + t * 8 + 4);
                const float ly = __uint_as_float(e.x), lx = __uint_as_float(e.y), a = __uint_as_float(e.z);
                const uint32_t flags = e.w;
                const float hy = 1.f - ly, hx = 1.f - lx;
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 v00 = (flags & 1u) ? *reinterpret_cast<const float4 *>(vbase + o.x) : z;
                const float4 v01 = (flags & 2u) ? *reinterpret_cast<const float4 *>(vbase + o.y) : z;
                const float4 v10 = (flags & 4u) ? *reinterpret_cast<const float4 *>(vbase + o.z) : z;
                const float4 v11 = (flags & 8u) ? *reinterpret_cast<const float4 *>(vbase + o.w) : z;
                const float w00 = hy * hx, w01 = hy * lx, w10 = ly * hx, w11 = ly * lx;
                float s_aw = 0.f, s_x = 0.f, s_y = 0.f;
#define SDETR_CH(c)                                                                              \
    {                                                                                            \
        const float gc = go.c;                                                                   \
        s_aw += gc * (w00 * v00.c + w01 * v01.c + w10 * v10.c + w11 * v11.c);                    \
        s_x += gc * (hy * (v01.c - v00.c) + ly * (v11.c - v10.c));                               \
        s_y += gc * (hx * (v10.c - v00.c) + lx * (v11.c - v01.c));                               \
    }
                SDETR_CH(x) SDETR_CH(y) SDETR_CH(z) SDETR_CH(w)
#undef SDETR_CH
                const float4 ga = make_float4(go.x * a, go.y * a, go.z * a, go.w * a);
#define SDETR_ATOM(bit, off, wgt)                                                                \
    if (flags & bit) {                                                                           \
        float *dst = reinterpret_cast<float *>(gbase + off);                                     \
        unsafeAtomicAdd(dst + 0, wgt * ga.x);                                                    \
        unsafeAtomicAdd(dst + 1, wgt * ga.y);                                                    \
        unsafeAtomicAdd(dst + 2, wgt * ga.z);                                                    \
        unsafeAtomicAdd(dst + 3, wgt * ga.w);                                                    \
    }
                SDETR_ATOM(1u, o.x, w00) SDETR_ATOM(2u, o.y, w01) SDETR_ATOM(4u, o.z, w10) SDETR_ATOM(8u, o.w, w11)
#undef SDETR_ATOM
#pragma unroll
                for (int sh = G / 2; sh > 0; sh >>= 1) {
                    s_aw += __shfl_xor(s_aw, sh, G);
                    s_x += __shfl_xor(s_x, sh, G);
                    s_y += __shfl_xor(s_y, sh, G);
                }
                if (j == (t & (G - 1))) {
                    const int l = (int)(flags >> 8);
                    const int64_t si = row * LP + c0 + t;
                    p.grad_aw[si] = s_aw;
                    reinterpret_cast<float2 *>(p.grad_loc)[si] =
                        make_float2((float)lvl_tab[l * 3 + 1] * s_x * a, (float)lvl_tab[l * 3] * s_y * a);
                }
            }
        }
    }
}


// ---- one channel per lane ------------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(kBlock) msda_col2im_chan_kernel(BackwardArgs p)
{
    static_assert(D == 32 || D == 64, "lane-per-channel mapping");
    constexpr int RPB = kBlock / D;          // rows per block
    constexpr int DSTRIDE = kBChunk * 8 + 4;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *desc = smem;
    int *lvl_tab = reinterpret_cast<int *>(smem + RPB * DSTRIDE);

    const int tid = threadIdx.x;
    const int m = blockIdx.x % p.M;
    const int chunk_global = blockIdx.x / p.M;
    const int b = chunk_global / p.nchunk;
    const int chunk = chunk_global - b * p.nchunk;
    const int g = tid / D, c = tid - g * D;   // row in block, channel
    const int q = chunk * RPB + g;
    const bool active = q < p.Nq;

    if (tid < p.L) {
        lvl_tab[tid * 3 + 0] = (int)p.shapes[2 * tid];
        lvl_tab[tid * 3 + 1] = (int)p.shapes[2 * tid + 1];
        lvl_tab[tid * 3 + 2] = (int)p.lsi[tid];
    }
    __syncthreads();

    const int LP = p.L * p.P;
    const int64_t row = ((int64_t)b * p.Nq + (active ? q : 0)) * p.M + m;
    const int64_t pix_stride = (int64_t)p.M * D;  // floats between pixels (reference layout)
    const float *vbase = reinterpret_cast<const float *>(p.value) + (int64_t)b * p.Nv * pix_stride + m * D + c;
    float *gbase = reinterpret_cast<float *>(p.grad_value) + (int64_t)b * p.Nv * pix_stride + m * D + c;
    const float go = active ? p.grad_out[row * D + c] : 0.f;

    uint32_t *my_desc = desc + g * DSTRIDE;
    for (int c0 = 0; c0 < LP; c0 += kBChunk) {
        const int ns = min(kBChunk, LP - c0);
        if (c0 > 0) __syncthreads();
        if (active && c < ns) {  // lane c of the row prepares sample c0 + c
            const int s = c0 + c;
            const int l = s / p.P;
            const int H = lvl_tab[l * 3], W = lvl_tab[l * 3 + 1], start = lvl_tab[l * 3 + 2];
            const float2 xy = reinterpret_cast<const float2 *>(p.loc)[row * LP + s];
            const float a = p.aw[row * LP + s];
            const float h_im = xy.y * (float)H - 0.5f, w_im = xy.x * (float)W - 0.5f;
            const bool inside = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
            const float fy = floorf(h_im), fx = floorf(w_im);
            const int y0 = inside ? (int)fy : 0, x0 = inside ? (int)fx : 0;
            const uint32_t flags = !inside ? 0u
                : ((uint32_t)(y0 >= 0 && x0 >= 0) | ((uint32_t)(y0 >= 0 && x0 + 1 <= W - 1) << 1) |
                   ((uint32_t)(y0 + 1 <= H - 1 && x0 >= 0) << 2) | ((uint32_t)(y0 + 1 <= H - 1 && x0 + 1 <= W - 1) << 3));
            *reinterpret_cast<uint4 *>(my_desc + c * 8) =
                make_uint4((uint32_t)(start + y0 * W + x0), (uint32_t)W, flags | ((uint32_t)l << 8), 0u);
            *reinterpret_cast<uint4 *>(my_desc + c * 8 + 4) =
                make_uint4(__float_as_uint(h_im - fy), __float_as_uint(w_im - fx), __float_as_uint(a), 0u);
        }
        __syncthreads();
        if (active) {
            for (int t = 0; t < ns; ++t) {
                const uint4 o = *reinterpret_cast<const uint4 *>(my_desc + t * 8);
                const uint4 e = *reinterpret_cast<const uint4 *>(my_desc + t * 8 + 4);
                const int pix = (int)o.x, W = (int)o.y;
                const uint32_t flags = o.z;
                const float ly = __uint_as_float(e.x), lx = __uint_as_float(e.y), a = __uint_as_float(e.z);
                const float hy = 1.f - ly, hx = 1.f - lx;
                const float v00 = (flags & 1u) ? vbase[(int64_t)pix * pix_stride] : 0.f;
                const float v01 = (flags & 2u) ? vbase[(int64_t)(pix + 1) * pix_stride] : 0.f;
                const float v10 = (flags & 4u) ? vbase[(int64_t)(pix + W) * pix_stride] : 0.f;
                const float v11 = (flags & 8u) ? vbase[(int64_t)(pix + W + 1) * pix_stride] : 0.f;
                const float w00 = hy * hx, w01 = hy * lx, w10 = ly * hx, w11 = ly * lx;
                const float ga = go * a;
                // whole 128-byte lines per instruction: lanes c = 0..D-1 hit consecutive floats
                if (flags & 1u) unsafeAtomicAdd(gbase + (int64_t)pix * pix_stride, w00 * ga);
                if (flags & 2u) unsafeAtomicAdd(gbase + (int64_t)(pix + 1) * pix_stride, w01 * ga);
                if (flags & 4u) unsafeAtomicAdd(gbase + (int64_t)(pix + W) * pix_stride, w10 * ga);
                if (flags & 8u) unsafeAtomicAdd(gbase + (int64_t)(pix + W + 1) * pix_stride, w11 * ga);
                float s_aw = go * (w00 * v00 + w01 * v01 + w10 * v10 + w11 * v11);
                float s_x = go * (hy * (v01 - v00) + ly * (v11 - v10));
                float s_y = go * (hx * (v10 - v00) + lx * (v11 - v01));
#pragma unroll
                for (int sh = D / 2; sh > 0; sh >>= 1) {
                    s_aw += __shfl_xor(s_aw, sh, D);
                    s_x += __shfl_xor(s_x, sh, D);
                    s_y += __shfl_xor(s_y, sh, D);
                }
                if (c == (t & (D - 1))) {
                    const int l = (int)(flags >> 8);
                    const int64_t si = row * LP + c0 + t;
                    p.grad_aw[si] = s_aw;
                    reinterpret_cast<float2 *>(p.grad_loc)[si] =
                        make_float2((float)W * s_x * a, (float)lvl_tab[l * 3] * s_y * a);
                }
            }
        }
    }
}

// Generic fallback (any head dim, fp32/fp64, reference layout): one thread per (b,q,m) row.
template <typename S>
__global__ void __launch_bounds__(kBlock) msda_col2im_generic_kernel(int64_t rows, const S *grad_out, const S *value,
                                                                     const int64_t *shapes, const int64_t *lsi,
                                                                     const S *loc, const S *aw, int Nv, int M, int D,
                                                                     int L, int Nq, int P, S *grad_value, S *grad_loc,
                                                                     S *grad_aw)
{
    for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < rows;
         row += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(row % M);
        const int b = (int)(row / M / Nq);
        const int LP = L * P;
        const S *g = grad_out + row * D;
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            const int64_t base = ((int64_t)b * Nv + lsi[l]) * M * D + (int64_t)m * D;
            for (int pp = 0; pp < P; ++pp) {
                const int64_t si = row * LP + l * P + pp;
                const S x = loc[si * 2] * (S)W - (S)0.5, y = loc[si * 2 + 1] * (S)H - (S)0.5;
                const S a = aw[si];
                S s_aw = 0, s_x = 0, s_y = 0;
                if (y > (S)-1 && x > (S)-1 && y < (S)H && x < (S)W) {
                    const S fy = floor(y), fx = floor(x);
                    const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
                    const S ly = y - fy, lx = x - fx, hy = (S)1 - ly, hx = (S)1 - lx;
                    const bool k00 = y0 >= 0 && x0 >= 0, k01 = y0 >= 0 && x1 <= W - 1;
                    const bool k10 = y1 <= H - 1 && x0 >= 0, k11 = y1 <= H - 1 && x1 <= W - 1;
                    const int64_t sp = (int64_t)M * D;
                    const int64_t o00 = base + ((int64_t)y0 * W + x0) * sp, o01 = base + ((int64_t)y0 * W + x1) * sp;
                    const int64_t o10 = base + ((int64_t)y1 * W + x0) * sp, o11 = base + ((int64_t)y1 * W + x1) * sp;
                    for (int c = 0; c < D; ++c) {
                        const S gc = g[c], ga = gc * a;
                        const S v00 = k00 ? value[o00 + c] : (S)0, v01 = k01 ? value[o01 + c] : (S)0;
                        const S v10 = k10 ? value[o10 + c] : (S)0, v11 = k11 ? value[o11 + c] : (S)0;
                        if (k00) unsafeAtomicAdd(grad_value + o00 + c, hy * hx * ga);
                        if (k01) unsafeAtomicAdd(grad_value + o01 + c, hy * lx * ga);
                        if (k10) unsafeAtomicAdd(grad_value + o10 + c, ly * hx * ga);
                        if (k11) unsafeAtomicAdd(grad_value + o11 + c, ly * lx * ga);
                        s_aw += gc * (hy * hx * v00 + hy * lx * v01 + ly * hx * v10 + ly * lx * v11);
                        s_x += ga * (hy * (v01 - v00) + ly * (v11 - v10));
                        s_y += ga * (hx * (v10 - v00) + lx * (v11 - v01));
                    }
                }
                grad_aw[si] = s_aw;
                grad_loc[si * 2] = (S)W * s_x;
                grad_loc[si * 2 + 1] = (S)H * s_y;
            }
        }
    }
}

template <int D, bool HM>
static int launch_bwd(hipStream_t stream, BackwardArgs &a)
{
    constexpr int GPB = kBlock / (D / 4);
    a.nchunk = (a.Nq + GPB - 1) / GPB;
    const int64_t blocks = (int64_t)a.B * a.nchunk * a.M;
    if (blocks == 0) return 0;
    if (blocks > 0x7fffffffLL) return fail("msda backward: grid too large");
    const size_t lds = (size_t)(GPB * (kBChunk * 8 + 4) + kMaxLevels * 3) * 4;
    hipLaunchKernelGGL((msda_col2im_kernel<D, HM>), dim3((unsigned)blocks), dim3(kBlock), lds, stream, a);
    note_backward_kernel(SDETR_KERNEL_MSDA_BWD_DIRECT);
    return check_launch("msda_col2im");
}

template <int D>
static int launch_bwd_chan(hipStream_t stream, BackwardArgs &a)
{
    constexpr int RPB = kBlock / D;
    a.nchunk = (a.Nq + RPB - 1) / RPB;
    const int64_t blocks = (int64_t)a.B * a.nchunk * a.M;
    if (blocks == 0) return 0;
    if (blocks > 0x7fffffffLL) return fail("msda backward: grid too large");
    const size_t lds = (size_t)(RPB * (kBChunk * 8 + 4) + kMaxLevels * 3) * 4;
    hipLaunchKernelGGL((msda_col2im_chan_kernel<D>), dim3((unsigned)blocks), dim3(kBlock), lds, stream, a);
    note_backward_kernel(SDETR_KERNEL_MSDA_BWD_DIRECT);
    return check_launch("msda_col2im_chan");
}

template <bool HM>
static int dispatch_bwd(hipStream_t stream, BackwardArgs &a, int D)
{
    if (!HM && D == 32) return launch_bwd_chan<32>(stream, a);
    if (!HM && D == 64) return launch_bwd_chan<64>(stream, a);
    switch (D) {
        case 4: return launch_bwd<4, HM>(stream, a);
        case 8: return launch_bwd<8, HM>(stream, a);
        case 16: return launch_bwd<16, HM>(stream, a);
        case 32: return launch_bwd<32, HM>(stream, a);
        case 64: return launch_bwd<64, HM>(stream, a);
        case 128: return launch_bwd<128, HM>(stream, a);
        default: break;
    }
    return fail("msda backward: head dim %d not supported by the tiled kernel", D);
}

template <typename S>
static int generic_bwd(hipStream_t stream, const S *grad_col, const S *value, const int64_t *shapes,
                       const int64_t *lsi, const S *loc, const S *aw, int B, int Nv, int M, int D, int L, int Nq, int P,
                       S *gv, S *gl, S *ga)
{
    const int64_t rows = (int64_t)B * Nq * M;
    if (rows == 0) return 0;
    const int64_t blocks = (rows + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(msda_col2im_generic_kernel<S>, dim3((unsigned)(blocks > 1048576 ? 1048576 : blocks)),
                       dim3(kBlock), 0, stream, rows, grad_col, value, shapes, lsi, loc, aw, Nv, M, D, L, Nq, P, gv, gl,
                       ga);
    note_backward_kernel(SDETR_KERNEL_MSDA_BWD_DIRECT);
    return check_launch("msda_col2im_generic");
}

}  // namespace sdetr

using namespace sdetr;

extern "C" int sdetr_msda_col2im_f32(sdetr_stream_t stream, const float *grad_col, const float *value,
                                     const int64_t *shapes, const int64_t *lsi, const float *loc, const float *aw,
                                     int B, int Nv, int M, int D, int L, int Nq, int P, float *grad_value,
                                     float *grad_loc, float *grad_aw)
{
    if (B < 0 || Nv < 0 || M <= 0 || D <= 0 || L <= 0 || Nq < 0 || P <= 0) return fail("msda_col2im_f32: bad dims");
    if (!grad_col || !value || !shapes || !lsi || !loc || !aw || !grad_value || !grad_loc || !grad_aw)
        return fail("msda_col2im_f32: null pointer");
    const bool tiled = (D == 4 || D == 8 || D == 16 || D == 32 || D == 64 || D == 128) && L <= kMaxLevels &&
                       (int64_t)Nv * M * D * 4 < 0xffffffffLL;
    if (!tiled)
        return generic_bwd<float>(stream, grad_col, value, shapes, lsi, loc, aw, B, Nv, M, D, L, Nq, P, grad_value,
                                  grad_loc, grad_aw);
    BackwardArgs a{};
    a.grad_out = grad_col; a.value = reinterpret_cast<const char *>(value); a.shapes = shapes; a.lsi = lsi;
    a.loc = loc; a.aw = aw; a.grad_value = reinterpret_cast<char *>(grad_value); a.grad_loc = grad_loc;
    a.grad_aw = grad_aw; a.B = B; a.Nv = Nv; a.M = M; a.L = L; a.Nq = Nq; a.P = P;
    return dispatch_bwd<false>(stream, a, D);
}

extern "C" int sdetr_msda_col2im_f64(sdetr_stream_t stream, const double *grad_col, const double *value,
                                     const int64_t *shapes, const int64_t *lsi, const double *loc, const double *aw,
                                     int B, int Nv, int M, int D, int L, int Nq, int P, double *grad_value,
                                     double *grad_loc, double *grad_aw)
{
    if (B < 0 || Nv < 0 || M <= 0 || D <= 0 || L <= 0 || Nq < 0 || P <= 0) return fail("msda_col2im_f64: bad dims");
    if (!grad_col || !value || !shapes || !lsi || !loc || !aw || !grad_value || !grad_loc || !grad_aw)
        return fail("msda_col2im_f64: null pointer");
    return generic_bwd<double>(stream, grad_col, value, shapes, lsi, loc, aw, B, Nv, M, D, L, Nq, P, grad_value,
                               grad_loc, grad_aw);
}
