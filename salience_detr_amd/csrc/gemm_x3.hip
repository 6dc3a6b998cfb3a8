// fp32 GEMM on the bf16 matrix cores at fp32 accuracy ("bf16 x 3"), for the training step's Linear layers (gfx950).
//
// The fp32-input MFMA peaks at 157 TFLOP/s and the library's fp32 GEMMs reach ~56 TFLOP/s on the step's skinny shapes
// (K = 256: profiles/r02_train_kernel_stats.csv, 12.8 ms of the 27 ms step).  v_mfma_f32_32x32x16_bf16 is 16 times
// faster per multiply-add.  Every fp32 operand element is split exactly into three bf16 terms, x = x0 + x1 + x2
// (24 mantissa bits = 3 x 8), and a.b is taken as the six bf16 products a0b0 + a0b1 + a1b0 + a1b1 + a0b2 + a2b0:
// each is exact in fp32, accumulation is fp32, and the dropped terms are below 2^-24 of the product -- the rounding
// an fp32 multiply makes anyway.  Same technique as stage 1 of the salience head (salience_head.hip), here as a
// general tiled GEMM: 128 x 128 x 32 tiles, global fp32 -> registers (two steps ahead of the MFMAs) -> fp32 tiles in
// LDS (row stride 144 bytes: conflict-free 16-byte fragment reads) -> every wave splits the fragments it reads ->
// 48 MFMAs per wave and step.  Either operand may have the reduction index as
// its contiguous one or as its row index (the LDS store transposes 4 x 4 blocks in registers), which covers the three
// products of a Linear layer without a transposed copy:
//     y  = x  w^T      A = x  [T,K]  k-major,   B = w [N,K]  k-major
//     dx = dy w        A = dy [T,N]  k-major,   B = w [N,K]  reduction index = row
//     dw = dy^T x      A = dy [T,N]  reduction index = row,  B = x [T,K]  reduction index = row   (split over T)
// A reduction split over blockIdx.z accumulates with fp32 atomics into a zeroed C (the few-tile weight gradients).
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "common.h"

namespace sdetr {

constexpr int kGTile = 128, kGK = 32, kGThreads = 256;
constexpr int kGRow = kGK * 4 + 16;              // bytes per fp32 row of an operand tile (16 bytes of padding:
                                                 // conflict-free 16-byte fragment reads at stride 36 dwords)
constexpr int kGOperand = kGTile * kGRow;        // 18 432
constexpr int kGLds = 2 * kGOperand;             // 36 864
constexpr int kGPreRow = kGK * 2 + 16;           // bytes per row of a pre-split plane tile (bf16)
constexpr int kGPrePlane = kGTile * kGPreRow;    // 10 240
constexpr int kGLdsPre = kGOperand + 3 * kGPrePlane;   // 49 152: fp32 A tile + three B planes

typedef __bf16 g_bf16x8_t __attribute__((ext_vector_type(8)));
typedef float g_f32x16_t __attribute__((ext_vector_type(16)));

struct GemmArgs {
    const float *a, *b;
    float *c;
    const float *bias;     // [N] or NULL (added by the blockIdx.z == 0 slice)
    int64_t lda, ldb, ldc;
    int M, N, K;
    int k_per_split;       // reduction elements per blockIdx.z slice (multiple of 32)
    int atomic;            // accumulate into C with atomics (split reduction)
    int64_t b_plane;       // pre-split B: elements between its three planes
    float *a_row_sum;      // optional [M]: sum_k A(m, k) accumulated with atomics (zero on entry) -- the bias gradient
                           // sum_t dy[t][n] next to dw = dy^T x; reduction-major A only
};

__device__ __forceinline__ g_f32x16_t g_mfma(u32x4_t a, u32x4_t b, g_f32x16_t c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(g_bf16x8_t, a), __builtin_bit_cast(g_bf16x8_t, b), c, 0, 0, 0);
}

// Exact three-way split of an MFMA operand fragment (8 fp32 values consecutive along the reduction index) into
// three bf16x8 fragments, by TRUNCATION: h = top 16 bits of x (a bf16), r1 = x - h exactly, m = top 16 bits of r1,
// r2 = r1 - m exactly and with at most 8 significant bits left, i.e. already a bf16.  5.5 VALU operations per
// element (and, sub, and, sub and three half-instruction byte permutes that pack the high halves of two floats) --
// the rounding converter (compare / select / add chains of f32_to_bf16_bits) costs ~25.
// The operands stay fp32 in LDS and every wave splits the fragments it reads: splitting before the LDS store (three
// bf16 planes per operand) moves 1.6x the bytes through LDS and made LDS bandwidth the bound of the first version
// (82-103 TFLOP/s fp32-equivalent).
__device__ __forceinline__ uint32_t g_pack_hi(float lo, float hi)   // bf16(lo) | bf16(hi) << 16, both by truncation
{
    return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
struct Frag3 {
    u32x4_t p[3];
};
__device__ __forceinline__ Frag3 g_split(const float4 lo, const float4 hi)
{
    const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    float r1[8], r2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        r1[i] = x[i] - __uint_as_float(__float_as_uint(x[i]) & 0xffff0000u);
        r2[i] = r1[i] - __uint_as_float(__float_as_uint(r1[i]) & 0xffff0000u);
    }
    Frag3 f;
    f.p[0] = u32x4_t{g_pack_hi(x[0], x[1]), g_pack_hi(x[2], x[3]), g_pack_hi(x[4], x[5]), g_pack_hi(x[6], x[7])};
    f.p[1] = u32x4_t{g_pack_hi(r1[0], r1[1]), g_pack_hi(r1[2], r1[3]), g_pack_hi(r1[4], r1[5]), g_pack_hi(r1[6], r1[7])};
    f.p[2] = u32x4_t{g_pack_hi(r2[0], r2[1]), g_pack_hi(r2[2], r2[3]), g_pack_hi(r2[4], r2[5]), g_pack_hi(r2[6], r2[7])};
    return f;
}

// One operand's 128 x 32 tile: 16 floats per thread.  KMAJOR: element (row, k) at src[row * ld + k]; otherwise at
// src[k * ld + row].  Out-of-range rows / reduction indices read as zero.  The LDS tile is [row][k] fp32 either way.
template <bool KMAJOR>
struct TileLoad {
    float4 v0, v1, v2, v3;   // (named members: an array here ends up in scratch memory)
    __device__ __forceinline__ static float4 ld4(const float *ptr, bool ok)
    {
        return ok ? *reinterpret_cast<const float4 *>(ptr) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __device__ __forceinline__ void load(const float *src, int64_t ld, int row0, int rows, int k0, int kend, int tid)
    {
        if (KMAJOR) {
            const int c4 = tid & 7, r = row0 + (tid >> 3);
            const int k = k0 + 4 * c4;
            const float *q = src + (int64_t)r * ld + k;
            const bool kok = k < kend;
            v0 = ld4(q, kok && r < rows);
            v1 = ld4(q + 32 * ld, kok && r + 32 < rows);
            v2 = ld4(q + 64 * ld, kok && r + 64 < rows);
            v3 = ld4(q + 96 * ld, kok && r + 96 < rows);
        } else {
            const int kb = tid >> 5, mb = tid & 31;
            const int row = row0 + 4 * mb, k = k0 + 4 * kb;
            const float *q = src + (int64_t)k * ld + row;
            const bool rok = row < rows;
            v0 = ld4(q, rok && k < kend);
            v1 = ld4(q + ld, rok && k + 1 < kend);
            v2 = ld4(q + 2 * ld, rok && k + 2 < kend);
            v3 = ld4(q + 3 * ld, rok && k + 3 < kend);
        }
    }
    __device__ __forceinline__ void store(char *tile, int tid) const
    {
        if (KMAJOR) {
            const int c4 = tid & 7, r = tid >> 3;
            char *d = tile + r * kGRow + 16 * c4;
            *reinterpret_cast<float4 *>(d) = v0;
            *reinterpret_cast<float4 *>(d + 32 * kGRow) = v1;
            *reinterpret_cast<float4 *>(d + 64 * kGRow) = v2;
            *reinterpret_cast<float4 *>(d + 96 * kGRow) = v3;
        } else {
            const int kb = tid >> 5, mb = tid & 31;   // 4 x 4 block transposed in registers
            char *d = tile + (4 * mb) * kGRow + 16 * kb;
            *reinterpret_cast<float4 *>(d) = make_float4(v0.x, v1.x, v2.x, v3.x);
            *reinterpret_cast<float4 *>(d + kGRow) = make_float4(v0.y, v1.y, v2.y, v3.y);
            *reinterpret_cast<float4 *>(d + 2 * kGRow) = make_float4(v0.z, v1.z, v2.z, v3.z);
            *reinterpret_cast<float4 *>(d + 3 * kGRow) = make_float4(v0.w, v1.w, v2.w, v3.w);
        }
    }
};

// A PRE-SPLIT operand: three bf16 planes [3][rows][K] (k-major), made once per call by gemm_x3_presplit_kernel -- a
// weight is shared by all row tiles of the other operand, so splitting it in every workgroup that reads it repeats
// the work M / 128 times (and the split is what the kernel's vector ALUs spend their time on).
struct PreTileLoad {
    uint4 q[3][2];
    __device__ __forceinline__ void load(const uint16_t *src, int64_t ld, int64_t plane, int row0, int rows, int k0, int kend,
                                         int tid)
    {
        const int r = row0 + (tid >> 1), k = k0 + 16 * (tid & 1);
        const uint16_t *ptr = src + (int64_t)r * ld + k;
        const bool ok0 = r < rows && k < kend, ok1 = r < rows && k + 8 < kend;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            q[pl][0] = ok0 ? *reinterpret_cast<const uint4 *>(ptr + pl * plane) : make_uint4(0u, 0u, 0u, 0u);
            q[pl][1] = ok1 ? *reinterpret_cast<const uint4 *>(ptr + pl * plane + 8) : make_uint4(0u, 0u, 0u, 0u);
        }
    }
    __device__ __forceinline__ void store(char *planes, int tid) const
    {
        char *d = planes + (tid >> 1) * kGPreRow + 32 * (tid & 1);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            *reinterpret_cast<uint4 *>(d + pl * kGPrePlane) = q[pl][0];
            *reinterpret_cast<uint4 *>(d + pl * kGPrePlane + 16) = q[pl][1];
        }
    }
};

__device__ __forceinline__ int g_acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

template <bool A_KMAJOR, bool B_KMAJOR>
__device__ __forceinline__ void gemm_x3_step(const GemmArgs &p, TileLoad<A_KMAJOR> &ta, TileLoad<B_KMAJOR> &tb, char *pa,
                                             char *pb, const char *fa, const char *fb, int m0, int n0, int k0, int kend,
                                             int tid, g_f32x16_t (&acc)[2][2], float4 &row_sum, bool want_row_sum)
{
    if (!A_KMAJOR && want_row_sum) {   // my 4 rows x 4 reduction indices of this step's A tile
        row_sum.x += (ta.v0.x + ta.v1.x) + (ta.v2.x + ta.v3.x);
        row_sum.y += (ta.v0.y + ta.v1.y) + (ta.v2.y + ta.v3.y);
        row_sum.z += (ta.v0.z + ta.v1.z) + (ta.v2.z + ta.v3.z);
        row_sum.w += (ta.v0.w + ta.v1.w) + (ta.v2.w + ta.v3.w);
    }

    ta.store(pa, tid);
    tb.store(pb, tid);
    __syncthreads();
    ta.load(p.a, p.lda, m0, p.M, k0 + 2 * kGK, kend, tid);   // (reads as zeros past the end)
    tb.load(p.b, p.ldb, n0, p.N, k0 + 2 * kGK, kend, tid);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        Frag3 a[2], b[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float4 *qa = reinterpret_cast<const float4 *>(fa + t * 32 * kGRow + kk * 64);
            const float4 *qb = reinterpret_cast<const float4 *>(fb + t * 32 * kGRow + kk * 64);
            a[t] = g_split(qa[0], qa[1]);
            b[t] = g_split(qb[0], qb[1]);
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                g_f32x16_t c = acc[rt][ct];
                c = g_mfma(a[rt].p[2], b[ct].p[0], c);   // smallest terms first
                c = g_mfma(a[rt].p[0], b[ct].p[2], c);
                c = g_mfma(a[rt].p[1], b[ct].p[1], c);
                c = g_mfma(a[rt].p[1], b[ct].p[0], c);
                c = g_mfma(a[rt].p[0], b[ct].p[1], c);
                c = g_mfma(a[rt].p[0], b[ct].p[0], c);
                acc[rt][ct] = c;
            }
    }
    __syncthreads();   // every wave is done with this step's tiles
}

template <bool A_KMAJOR, bool B_KMAJOR>
__global__ void __launch_bounds__(kGThreads, 2) gemm_x3_kernel(GemmArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *pa = lds, *pb = lds + kGOperand;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * kGTile, n0 = blockIdx.x * kGTile;
    const int kbeg = blockIdx.z * p.k_per_split, kend = min(p.K, kbeg + p.k_per_split);

    g_f32x16_t acc[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rt][ct][i] = 0.f;

    // operand tiles travel TWO steps ahead of the MFMAs that consume them (one step = ~0.7 us of matrix work per
    // wave, an L2 / HBM trip under load 1-2 us): two register sets, the loop unrolled by two
    TileLoad<A_KMAJOR> ta0, ta1;
    TileLoad<B_KMAJOR> tb0, tb1;
    ta0.load(p.a, p.lda, m0, p.M, kbeg, kend, tid);
    tb0.load(p.b, p.ldb, n0, p.N, kbeg, kend, tid);
    ta1.load(p.a, p.lda, m0, p.M, kbeg + kGK, kend, tid);
    tb1.load(p.b, p.ldb, n0, p.N, kbeg + kGK, kend, tid);
    const char *fa = pa + (64 * wm + (lane & 31)) * kGRow + (lane >> 5) * 32;
    const char *fb = pb + (64 * wn + (lane & 31)) * kGRow + (lane >> 5) * 32;

    float4 row_sum = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool want_row_sum = !A_KMAJOR && p.a_row_sum != nullptr && blockIdx.x == 0;
    for (int k0 = kbeg; k0 < kend; k0 += 2 * kGK) {
        gemm_x3_step<A_KMAJOR, B_KMAJOR>(p, ta0, tb0, pa, pb, fa, fb, m0, n0, k0, kend, tid, acc, row_sum, want_row_sum);
        if (k0 + kGK < kend)
            gemm_x3_step<A_KMAJOR, B_KMAJOR>(p, ta1, tb1, pa, pb, fa, fb, m0, n0, k0 + kGK, kend, tid, acc, row_sum, want_row_sum);
    }

    if (want_row_sum) {   // (uniform per workgroup) 8 threads hold pieces of each row's sum: meet in LDS, one atomic per row
        float *red = reinterpret_cast<float *>(pa);   // [8][128]; the operand tiles are no longer needed
        const int kb = tid >> 5, mb = tid & 31;
        *reinterpret_cast<float4 *>(red + kb * 128 + 4 * mb) = row_sum;
        __syncthreads();
        if (tid < 128 && m0 + tid < p.M) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) t += red[j * 128 + tid];
            unsafeAtomicAdd(p.a_row_sum + m0 + tid, t);
        }
    }
    const bool add_bias = p.bias && blockIdx.z == 0;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int n = n0 + 64 * wn + 32 * ct + (lane & 31);
        if (n >= p.N) continue;
        const float bias = add_bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = m0 + 64 * wm + 32 * rt + g_acc_row(i, lane);
                if (m < p.M) {
                    float *dst = p.c + (int64_t)m * p.ldc + n;
                    if (p.atomic) unsafeAtomicAdd(dst, acc[rt][ct][i] + bias);
                    else *dst = acc[rt][ct][i] + bias;
                }
            }
    }
}

template <bool A_KMAJOR>
__device__ __forceinline__ void gemm_x3_pre_step(const GemmArgs &p, TileLoad<A_KMAJOR> &ta, PreTileLoad &tb, char *pa, char *pb,
                                                 const char *fa, const char *fb, int m0, int n0, int k0, int kend, int tid,
                                                 g_f32x16_t (&acc)[2][2])
{
    ta.store(pa, tid);
    tb.store(pb, tid);
    __syncthreads();
    ta.load(p.a, p.lda, m0, p.M, k0 + 2 * kGK, kend, tid);
    tb.load(reinterpret_cast<const uint16_t *>(p.b), p.ldb, p.b_plane, n0, p.N, k0 + 2 * kGK, kend, tid);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        Frag3 a[2];
        u32x4_t b[2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float4 *qa = reinterpret_cast<const float4 *>(fa + t * 32 * kGRow + kk * 64);
            a[t] = g_split(qa[0], qa[1]);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                b[t][pl] = *reinterpret_cast<const u32x4_t *>(fb + t * 32 * kGPreRow + pl * kGPrePlane + kk * 32);
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                g_f32x16_t c = acc[rt][ct];
                c = g_mfma(a[rt].p[2], b[ct][0], c);   // smallest terms first
                c = g_mfma(a[rt].p[0], b[ct][2], c);
                c = g_mfma(a[rt].p[1], b[ct][1], c);
                c = g_mfma(a[rt].p[1], b[ct][0], c);
                c = g_mfma(a[rt].p[0], b[ct][1], c);
                c = g_mfma(a[rt].p[0], b[ct][0], c);
                acc[rt][ct] = c;
            }
    }
    __syncthreads();
}

template <bool A_KMAJOR>
__global__ void __launch_bounds__(kGThreads, 2) gemm_x3_pre_kernel(GemmArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *pa = lds, *pb = lds + kGOperand;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * kGTile, n0 = blockIdx.x * kGTile;
    const int kbeg = blockIdx.z * p.k_per_split, kend = min(p.K, kbeg + p.k_per_split);
    g_f32x16_t acc[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rt][ct][i] = 0.f;
    TileLoad<A_KMAJOR> ta0, ta1;
    PreTileLoad tb0, tb1;
    const uint16_t *bp = reinterpret_cast<const uint16_t *>(p.b);
    ta0.load(p.a, p.lda, m0, p.M, kbeg, kend, tid);
    tb0.load(bp, p.ldb, p.b_plane, n0, p.N, kbeg, kend, tid);
    ta1.load(p.a, p.lda, m0, p.M, kbeg + kGK, kend, tid);
    tb1.load(bp, p.ldb, p.b_plane, n0, p.N, kbeg + kGK, kend, tid);
    const char *fa = pa + (64 * wm + (lane & 31)) * kGRow + (lane >> 5) * 32;
    const char *fb = pb + (64 * wn + (lane & 31)) * kGPreRow + (lane >> 5) * 16;
    for (int k0 = kbeg; k0 < kend; k0 += 2 * kGK) {
        gemm_x3_pre_step<A_KMAJOR>(p, ta0, tb0, pa, pb, fa, fb, m0, n0, k0, kend, tid, acc);
        if (k0 + kGK < kend) gemm_x3_pre_step<A_KMAJOR>(p, ta1, tb1, pa, pb, fa, fb, m0, n0, k0 + kGK, kend, tid, acc);
    }
    const bool add_bias = p.bias && blockIdx.z == 0;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int n = n0 + 64 * wn + 32 * ct + (lane & 31);
        if (n >= p.N) continue;
        const float bias = add_bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = m0 + 64 * wm + 32 * rt + g_acc_row(i, lane);
                if (m < p.M) {
                    float *dst = p.c + (int64_t)m * p.ldc + n;
                    if (p.atomic) unsafeAtomicAdd(dst, acc[rt][ct][i] + bias);
                    else *dst = acc[rt][ct][i] + bias;
                }
            }
    }
}

// planes[pl][i][j] = plane pl of (transpose ? w[j][i] : w[i][j]); rows_out x cols_out = transpose ? cols x rows : rows x cols
__global__ void __launch_bounds__(256) gemm_x3_presplit_kernel(const float *w, int64_t ld, int rows, int cols, int transpose,
                                                               uint16_t *out)
{
    const int ro = transpose ? cols : rows, co = transpose ? rows : cols;
    const int64_t total = (int64_t)ro * co;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(t / co), j = (int)(t - (int64_t)i * co);
        const float x = transpose ? w[(int64_t)j * ld + i] : w[(int64_t)i * ld + j];
        const float r1 = x - __uint_as_float(__float_as_uint(x) & 0xffff0000u);
        const float r2 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
        out[t] = (uint16_t)(__float_as_uint(x) >> 16);
        out[total + t] = (uint16_t)(__float_as_uint(r1) >> 16);
        out[2 * total + t] = (uint16_t)(__float_as_uint(r2) >> 16);
    }
}

}  // namespace sdetr

using namespace sdetr;

template <bool AK, bool BK>
static int launch_gemm_x3(hipStream_t s, const GemmArgs &a, int splits)
{
    static DeviceOnce once;
    allow_dynamic_lds(gemm_x3_kernel<AK, BK>, once, kGLds);
    const dim3 grid((unsigned)((a.N + kGTile - 1) / kGTile), (unsigned)((a.M + kGTile - 1) / kGTile), (unsigned)splits);
    hipLaunchKernelGGL((gemm_x3_kernel<AK, BK>), grid, dim3(kGThreads), kGLds, s, a);
    return check_launch("gemm_x3");
}

// C[M,N] = sum_k A(m,k) B(n,k) (+ bias[n]).  a_kmajor: A(m,k) = a[m * lda + k], else a[k * lda + m]; b likewise with n.
// reduction_splits > 1: the reduction is cut into that many slices whose partial products are accumulated into C with
// fp32 atomics -- C must be zero on entry.  Alignment: every operand 16-byte aligned, its leading dimension a multiple
// of 4; a k-major operand needs K % 4 == 0, the other kind its row count % 4 == 0.  a_row_sum (optional, [M], zero on
// entry, reduction-major A only): receives sum_k A(m, k) -- the bias gradient that comes with dw = dy^T x.
extern "C" int sdetr_gemm_x3_f32(sdetr_stream_t stream, const float *a, int64_t lda, int a_kmajor, const float *b,
                                 int64_t ldb, int b_kmajor, float *c, int64_t ldc, int M, int N, int K,
                                 const float *bias, int reduction_splits, float *a_row_sum)
{
    if (M < 0 || N < 0 || K < 0) return fail("gemm_x3: negative size");
    if (M == 0 || N == 0) return 0;
    if (!a || !b || !c) return fail("gemm_x3: null pointer");
    if (b_kmajor == 2) {   // pre-split B planes [3][N][K] bf16 (sdetr_gemm_x3_presplit), rows ldb elements apart
        if ((lda & 3) || ((uintptr_t)a & 15) || ((uintptr_t)b & 15) || (ldb & 7) || (K & 7) || ldb < K)
            return fail("gemm_x3: a pre-split B needs K % 8 == 0, rows of >= K elements, 16-byte aligned operands");
        if ((a_kmajor && (K & 3)) || (!a_kmajor && (M & 3))) return fail("gemm_x3: M=%d K=%d do not meet A's alignment rule", M, K);
        if (a_row_sum) return fail("gemm_x3: a_row_sum is not available with a pre-split B");
        if (reduction_splits < 1) reduction_splits = 1;
        GemmArgs g{};
        g.a = a; g.b = b; g.c = c; g.bias = bias; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
        g.b_plane = (int64_t)N * ldb;
        const int steps = (K + kGK - 1) / kGK;
        int splits = reduction_splits > steps ? (steps > 0 ? steps : 1) : reduction_splits;
        g.k_per_split = ((steps + splits - 1) / splits) * kGK;
        splits = g.k_per_split > 0 ? (K + g.k_per_split - 1) / g.k_per_split : 1;
        if (splits < 1) splits = 1;
        g.atomic = splits > 1;
        const dim3 grid((unsigned)((N + kGTile - 1) / kGTile), (unsigned)((M + kGTile - 1) / kGTile), (unsigned)splits);
        hipStream_t hs = static_cast<hipStream_t>(stream);
        if (a_kmajor) hipLaunchKernelGGL((gemm_x3_pre_kernel<true>), grid, dim3(kGThreads), kGLdsPre, hs, g);
        else hipLaunchKernelGGL((gemm_x3_pre_kernel<false>), grid, dim3(kGThreads), kGLdsPre, hs, g);
        return check_launch("gemm_x3");
    }
    if ((lda & 3) || (ldb & 3) || ((uintptr_t)a & 15) || ((uintptr_t)b & 15))
        return fail("gemm_x3: operands must be 16-byte aligned with leading dimensions that are multiples of 4");
    if ((a_kmajor && (K & 3)) || (!a_kmajor && (M & 3)) || (b_kmajor && (K & 3)) || (!b_kmajor && (N & 3)))
        return fail("gemm_x3: M=%d N=%d K=%d do not meet the alignment rule of the chosen layouts", M, N, K);
    if (reduction_splits < 1) reduction_splits = 1;
    if (a_row_sum && a_kmajor) return fail("gemm_x3: a_row_sum needs a reduction-major A");
    GemmArgs g{};
    g.a_row_sum = a_row_sum;
    g.a = a; g.b = b; g.c = c; g.bias = bias; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    const int steps = (K + kGK - 1) / kGK;
    int splits = reduction_splits > steps ? (steps > 0 ? steps : 1) : reduction_splits;
    g.k_per_split = ((steps + splits - 1) / splits) * kGK;
    splits = g.k_per_split > 0 ? (K + g.k_per_split - 1) / g.k_per_split : 1;
    if (splits < 1) splits = 1;
    g.atomic = splits > 1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (a_kmajor && b_kmajor) return launch_gemm_x3<true, true>(s, g, splits);
    if (a_kmajor && !b_kmajor) return launch_gemm_x3<true, false>(s, g, splits);
    if (!a_kmajor && b_kmajor) return launch_gemm_x3<false, true>(s, g, splits);
    return launch_gemm_x3<false, false>(s, g, splits);
}

// Three bf16 planes of an fp32 matrix (exact split by truncation), optionally transposed: the form in which
// sdetr_gemm_x3_f32 takes a B operand with b_kmajor = 2.  out: 3 * rows * cols bf16.
extern "C" int sdetr_gemm_x3_presplit(sdetr_stream_t stream, const float *w, int64_t ld, int rows, int cols, int transpose,
                                      void *out)
{
    if (rows < 0 || cols < 0) return fail("gemm_x3_presplit: negative size");
    if ((int64_t)rows * cols == 0) return 0;
    if (!w || !out || ld < cols) return fail("gemm_x3_presplit: bad arguments");
    const int64_t total = (int64_t)rows * cols;
    const int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(gemm_x3_presplit_kernel, dim3((unsigned)(blocks > 65536 ? 65536 : blocks)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), w, ld, rows, cols, transpose, static_cast<uint16_t *>(out));
    return check_launch("gemm_x3_presplit");
}
