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
    uint32_t a_bytes, b_bytes;   // extents of the operands (second-generation kernel: buffer resources)
    int epilogue;          // SDETR_GEMM_EPI_*: 0 none; 1 ReLU on (acc + bias); 2 gate: C = gate(m, n) <= 0 ? 0 : acc + bias
    const float *gate;     // [M, N] rows ldg apart (epilogue 2): the ReLU OUTPUT whose backward this product is
    int64_t ldg;
};

// The feed-forward's ReLU lives in the products around it (round 4): forward h = relu(x w1^T + b1) as epilogue 1, backward
// dh = (dy w2) * (h > 0) as epilogue 2 -- torch's clamp_min_ / threshold_backward passes over the [T, 2048] hidden state
// (33 + 50 us per layer at 22 726 tokens) disappear.  NaN behaviour as torch's: relu(NaN) = NaN, a NaN gate lets the
// gradient through.  Unsplit reductions only (an atomic partial sum cannot be clamped).
__device__ __forceinline__ float gemm_epilogue(int epilogue, float v, float gate)
{
    if (epilogue == 1) return v < 0.f ? 0.f : v;
    if (epilogue == 2) return gate <= 0.f ? 0.f : v;
    return v;
}

// (benchmarks/micro/gemm_x3_ablate.hip compiles this file with SDETR_GX3_ABLATE = 1: no MFMAs, 2: no operand split,
// 3: no global loads after the first tiles, 4: no LDS tile stores, 5: no barrier, 6: no fragment reads (second-generation
// kernel) -- where the kernel's time goes.  0 / undefined in the library.)
#ifndef SDETR_GX3_ABLATE
#define SDETR_GX3_ABLATE 0
#endif

__device__ __forceinline__ g_f32x16_t g_mfma(u32x4_t a, u32x4_t b, g_f32x16_t c)
{
#if SDETR_GX3_ABLATE == 1
    c[0] += __uint_as_float(a[0] ^ b[0]);   // keeps the operands alive
    return c;
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(g_bf16x8_t, a), __builtin_bit_cast(g_bf16x8_t, b), c, 0, 0, 0);
#endif
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
#if SDETR_GX3_ABLATE == 2
    Frag3 raw;
    raw.p[0] = u32x4_t{__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3])};
    raw.p[1] = u32x4_t{__float_as_uint(x[4]), __float_as_uint(x[5]), __float_as_uint(x[6]), __float_as_uint(x[7])};
    raw.p[2] = raw.p[0];
    return raw;
#endif
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
            // (the reduction index varies fastest over the lanes: the 4 x 4 blocks of 8 neighbouring lanes land on 32
            // consecutive dwords of an LDS row -- with the row block varying fastest, 32 lanes met on 8 banks: 77 % of the
            // LDS cycles of the weight-gradient product were bank conflicts)
            const int kb = tid & 7, mb = tid >> 3;
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
            const int kb = tid & 7, mb = tid >> 3;   // 4 x 4 block transposed in registers
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
#if SDETR_GX3_ABLATE != 3
    ta.load(p.a, p.lda, m0, p.M, k0 + 2 * kGK, kend, tid);   // (reads as zeros past the end)
    tb.load(p.b, p.ldb, n0, p.N, k0 + 2 * kGK, kend, tid);
#endif
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

// C tile of one wave (2 x 2 MFMA accumulators of 32 x 32) to memory, lanes along n.  The gate values of an accumulator
// (epilogue 2) are requested together before its stores: interleaved, every load waits behind the store in front of it
// (the compiler must assume C and the gate alias): +77 us on the 22 726 x 2048 product that way.
__device__ __forceinline__ void gemm_store_c(const GemmArgs &p, const g_f32x16_t (&acc)[2][2], int m_base, int n_base,
                                             int lane, bool add_bias)
{
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int n = n_base + 32 * ct + (lane & 31);
        if (n >= p.N) continue;
        const float bias = add_bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            float gv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) gv[i] = 1.f;
            if (p.epilogue == 2) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int m = m_base + 32 * rt + g_acc_row(i, lane);
                    if (m < p.M) gv[i] = __builtin_nontemporal_load(p.gate + (int64_t)m * p.ldg + n);
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = m_base + 32 * rt + g_acc_row(i, lane);
                if (m < p.M) {
                    float *dst = p.c + (int64_t)m * p.ldc + n;
                    if (p.atomic) unsafeAtomicAdd(dst, acc[rt][ct][i] + bias);
                    else *dst = gemm_epilogue(p.epilogue, acc[rt][ct][i] + bias, gv[i]);
                }
            }
        }
    }
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
        const int kb = tid & 7, mb = tid >> 3;
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
    gemm_store_c(p, acc, m0 + 64 * wm, n0 + 64 * wn, lane, add_bias);
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
    gemm_store_c(p, acc, m0 + 64 * wm, n0 + 64 * wn, lane, add_bias);
}

// ---- second generation (round 3): 256 x 128 x 32 tiles, one 8-wave workgroup per CU ---------------------------------
// Counters and ablations of the kernels above on the feed-forward's first product (22 726 x 256 -> 2048, 201 us): the
// matrix pipes are busy 35 % of the time and the vector ALUs 46 %, one after the other -- and WITHOUT the operand split
// it still takes 177 us where its MFMAs alone are 71 (benchmarks/micro/gemm_x3_ablate.hip).  The hardware does overlap
// the two pipes, across the wavefronts of a SIMD and inside one (benchmarks/micro/mfma_valu_overlap.hip: 176 VALU + 24
// MFMA per wavefront, two per SIMD: 940 cycles each where the sum is 1630).  What the first generation loses is the
// barrier pair around every 32-deep step with ONE tile buffer: the four wavefronts of a workgroup share their SIMDs with
// another workgroup's in an unrelated phase, so at every barrier three of them wait for the one that found its matrix
// pipe taken (a 1536-cycle burst).  Here both wavefronts of a SIMD belong to the SAME workgroup (no foreign phase), the
// tiles are double-buffered in LDS (one barrier per step, between its two k-halves), and a wavefront reads and splits the
// fragments of the NEXT k-half while the MFMAs of the current one run (same-wavefront overlap; loads through buffer
// resources, out-of-range pieces by offset select: no branch inside a step, so the scheduler sees one block).
constexpr int kV2M = 256, kV2N = 128, kV2Threads = 512;
constexpr int kV2A = kV2M * kGRow;                   // 36 864 bytes: the A tile, fp32 rows of 144 bytes
constexpr int kV2Bp = 3 * kV2N * kGPreRow;           // 30 720: B as three bf16 planes (rows of 80 bytes)
constexpr int kV2PrePlane = kV2N * kGPreRow;         // 10 240

// exact three-way split of four consecutive values: plane pl gets two words (bf16 pairs, lower index in the low half)
__device__ __forceinline__ void g_split4(const float4 x, uint2 (&p)[3])
{
    const float v[4] = {x.x, x.y, x.z, x.w};
    float r1[4], r2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        r1[i] = v[i] - __uint_as_float(__float_as_uint(v[i]) & 0xffff0000u);
        r2[i] = r1[i] - __uint_as_float(__float_as_uint(r1[i]) & 0xffff0000u);
    }
    p[0] = make_uint2(g_pack_hi(v[0], v[1]), g_pack_hi(v[2], v[3]));
    p[1] = make_uint2(g_pack_hi(r1[0], r1[1]), g_pack_hi(r1[2], r1[3]));
    p[2] = make_uint2(g_pack_hi(r2[0], r2[1]), g_pack_hi(r2[2], r2[3]));
}

// byte offset or, when the piece is out of range, an offset no buffer contains (the load then returns zeros)
__device__ __forceinline__ uint32_t v2_off(bool ok, uint32_t off) { return ok ? off : 0xfffffff0u; }

// A tile (256 x 32 fp32): 16 floats per thread; threads 0-255 rows 0-127, threads 256-511 rows 128-255
template <bool KMAJOR>
struct V2LoadA {
    uint4 v0, v1, v2, v3;
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, uint32_t ld, int row0, int rows, int k0, int kend, int tid)
    {
        const int u = tid & 255, half = tid >> 8;
        if (KMAJOR) {
            const int c4 = u & 7, r = row0 + 128 * half + (u >> 3), k = k0 + 4 * c4;
            const bool kok = k < kend;
            const uint32_t o = ((uint32_t)r * ld + (uint32_t)k) * 4u, step = 32u * ld * 4u;
            v0 = buffer_load16(rs, v2_off(kok && r < rows, o));
            v1 = buffer_load16(rs, v2_off(kok && r + 32 < rows, o + step));
            v2 = buffer_load16(rs, v2_off(kok && r + 64 < rows, o + 2u * step));
            v3 = buffer_load16(rs, v2_off(kok && r + 96 < rows, o + 3u * step));
        } else {
            const int kb = u & 7, mb = u >> 3, row = row0 + 128 * half + 4 * mb, k = k0 + 4 * kb;   // (see TileLoad)
            const bool rok = row < rows;
            const uint32_t o = ((uint32_t)k * ld + (uint32_t)row) * 4u, step = ld * 4u;
            v0 = buffer_load16(rs, v2_off(rok && k < kend, o));
            v1 = buffer_load16(rs, v2_off(rok && k + 1 < kend, o + step));
            v2 = buffer_load16(rs, v2_off(rok && k + 2 < kend, o + 2u * step));
            v3 = buffer_load16(rs, v2_off(rok && k + 3 < kend, o + 3u * step));
        }
    }
    __device__ __forceinline__ void store(char *tile, int tid) const
    {
        const int u = tid & 255, half = tid >> 8;
        if (KMAJOR) {
            char *d = tile + (128 * half + (u >> 3)) * kGRow + 16 * (u & 7);
            *reinterpret_cast<uint4 *>(d) = v0;
            *reinterpret_cast<uint4 *>(d + 32 * kGRow) = v1;
            *reinterpret_cast<uint4 *>(d + 64 * kGRow) = v2;
            *reinterpret_cast<uint4 *>(d + 96 * kGRow) = v3;
        } else {
            char *d = tile + (128 * half + 4 * (u >> 3)) * kGRow + 16 * (u & 7);   // 4 x 4 block transposed in registers
            *reinterpret_cast<uint4 *>(d) = make_uint4(v0.x, v1.x, v2.x, v3.x);
            *reinterpret_cast<uint4 *>(d + kGRow) = make_uint4(v0.y, v1.y, v2.y, v3.y);
            *reinterpret_cast<uint4 *>(d + 2 * kGRow) = make_uint4(v0.z, v1.z, v2.z, v3.z);
            *reinterpret_cast<uint4 *>(d + 3 * kGRow) = make_uint4(v0.w, v1.w, v2.w, v3.w);
        }
    }
    __device__ __forceinline__ float4 row_sums() const   // reduction-major form: my 4 rows over my 4 reduction indices
    {
        const float4 a = __builtin_bit_cast(float4, v0), b = __builtin_bit_cast(float4, v1), c = __builtin_bit_cast(float4, v2),
                     d = __builtin_bit_cast(float4, v3);
        return make_float4((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w));
    }
};

// B tile (128 x 32): BMODE 0 fp32 with the reduction index as the source's row, 1 fp32 k-major, 2 three bf16 planes
template <int BMODE>
struct V2LoadB {
    uint4 q0, q1, q2;
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, uint32_t ld, uint32_t plane_bytes, int row0, int rows, int k0,
                                         int kend, int tid)
    {
        if (BMODE == 1) {
            const int c4 = tid & 7, r = row0 + (tid >> 3), k = k0 + 4 * c4;
            const bool kok = k < kend;
            const uint32_t o = ((uint32_t)r * ld + (uint32_t)k) * 4u;
            q0 = buffer_load16(rs, v2_off(kok && r < rows, o));
            q1 = buffer_load16(rs, v2_off(kok && r + 64 < rows, o + 64u * ld * 4u));
        } else if (BMODE == 0) {
            const int kb = tid & 15, mb = tid >> 4, row = row0 + 4 * mb, k = k0 + 2 * kb;   // (reduction index fastest)
            const bool rok = row < rows;
            const uint32_t o = ((uint32_t)k * ld + (uint32_t)row) * 4u;
            q0 = buffer_load16(rs, v2_off(rok && k < kend, o));
            q1 = buffer_load16(rs, v2_off(rok && k + 1 < kend, o + ld * 4u));
        } else {
            const int r = row0 + (tid >> 2), k = k0 + 8 * (tid & 3);
            const bool ok = r < rows && k < kend;
            const uint32_t o = ((uint32_t)r * ld + (uint32_t)k) * 2u;
            q0 = buffer_load16(rs, v2_off(ok, o));
            q1 = buffer_load16(rs, v2_off(ok, o + plane_bytes));
            q2 = buffer_load16(rs, v2_off(ok, o + 2u * plane_bytes));
        }
    }
    // The LDS tile is three bf16 planes in every mode: an fp32 B is split HERE, once per element -- a B fragment is read
    // by the four row-waves of the workgroup, and splitting it in each of them made the vector ALUs as busy as the matrix
    // pipes (counters at 22 726 x 2048 x 2048: MFMA 56 % of the cycles, VALU 56 %, barely overlapped).
    __device__ __forceinline__ void store(char *tile, int tid) const
    {
        if (BMODE == 1) {   // rows r, r + 64; reduction indices 4 c4 .. + 3: 8 bytes per plane and row
            char *d = tile + (tid >> 3) * kGPreRow + 8 * (tid & 7);
            const float4 x0 = __builtin_bit_cast(float4, q0), x1 = __builtin_bit_cast(float4, q1);
            uint2 p0[3], p1[3];
            g_split4(x0, p0);
            g_split4(x1, p1);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                *reinterpret_cast<uint2 *>(d + pl * kV2PrePlane) = p0[pl];
                *reinterpret_cast<uint2 *>(d + pl * kV2PrePlane + 64 * kGPreRow) = p1[pl];
            }
        } else if (BMODE == 0) {   // rows n .. n + 3, reduction indices 2 kb, 2 kb + 1: 4 bytes per plane and row
            char *d = tile + (4 * (tid >> 4)) * kGPreRow + 4 * (tid & 15);
            const float4 k0 = __builtin_bit_cast(float4, q0), k1 = __builtin_bit_cast(float4, q1);
            uint2 pa[3], pb[3];   // (row n, n + 1) and (row n + 2, n + 3), each word = the row's two reduction indices
            g_split4(make_float4(k0.x, k1.x, k0.y, k1.y), pa);
            g_split4(make_float4(k0.z, k1.z, k0.w, k1.w), pb);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                *reinterpret_cast<uint32_t *>(d + pl * kV2PrePlane) = pa[pl].x;
                *reinterpret_cast<uint32_t *>(d + pl * kV2PrePlane + kGPreRow) = pa[pl].y;
                *reinterpret_cast<uint32_t *>(d + pl * kV2PrePlane + 2 * kGPreRow) = pb[pl].x;
                *reinterpret_cast<uint32_t *>(d + pl * kV2PrePlane + 3 * kGPreRow) = pb[pl].y;
            }
        } else {
            char *d = tile + (tid >> 2) * kGPreRow + 16 * (tid & 3);
            *reinterpret_cast<uint4 *>(d) = q0;
            *reinterpret_cast<uint4 *>(d + kV2PrePlane) = q1;
            *reinterpret_cast<uint4 *>(d + 2 * kV2PrePlane) = q2;
        }
    }
};

// the fragments of one k-half (16 reduction indices): A rows 64 wm + {0, 32} + lane % 32, B rows 64 wn + ...
template <int BMODE>
struct V2Frags {
    Frag3 a[2], b[2];
    __device__ __forceinline__ void fetch(const char *fa, const char *fb, int kk)
    {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float4 *qa = reinterpret_cast<const float4 *>(fa + t * 32 * kGRow + kk * 64);
            a[t] = g_split(qa[0], qa[1]);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                b[t].p[pl] = *reinterpret_cast<const u32x4_t *>(fb + t * 32 * kGPreRow + pl * kV2PrePlane + kk * 32);
        }
    }
    __device__ __forceinline__ void mfmas(g_f32x16_t (&acc)[2][2]) const
    {
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
};

template <bool A_KMAJOR, int BMODE>
__global__ void __launch_bounds__(kV2Threads, 1) gemm_x3_v2_kernel(GemmArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int kStage = kV2A + kV2Bp;
    constexpr int kBRow = kGPreRow;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * kV2M, n0 = blockIdx.x * kV2N;
    const int kbeg = blockIdx.z * p.k_per_split, kend = min(p.K, kbeg + p.k_per_split);
    const __amdgpu_buffer_rsrc_t ra = make_uniform_rsrc(reinterpret_cast<const char *>(p.a), p.a_bytes);
    const __amdgpu_buffer_rsrc_t rb = make_uniform_rsrc(reinterpret_cast<const char *>(p.b), p.b_bytes);
    const uint32_t lda = (uint32_t)p.lda, ldb = (uint32_t)p.ldb, plane_bytes = (uint32_t)(p.b_plane * 2);

    g_f32x16_t acc[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rt][ct][i] = 0.f;

    // Tiles travel TWO steps ahead of the MFMAs that consume them (a step is ~1.5 us of matrix work per SIMD, a trip to
    // memory under load 1-2.5 us: one step ahead left every step waiting ~2.5 us for its tile): two register sets, the
    // loop unrolled by two.
    V2LoadA<A_KMAJOR> ta0, ta1;
    V2LoadB<BMODE> tb0, tb1;
    const bool want_row_sum = !A_KMAJOR && p.a_row_sum != nullptr && blockIdx.x == 0;
    float4 row_sum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto add_row_sum = [&](const V2LoadA<A_KMAJOR> &t) {
        if (!A_KMAJOR && want_row_sum) {
            const float4 r = t.row_sums();
            row_sum.x += r.x; row_sum.y += r.y; row_sum.z += r.z; row_sum.w += r.w;
        }
    };
    // tile 0 into stage 0, tiles 1 and 2 into the registers
    ta0.load(ra, lda, m0, p.M, kbeg, kend, tid);
    tb0.load(rb, ldb, plane_bytes, n0, p.N, kbeg, kend, tid);
    ta1.load(ra, lda, m0, p.M, kbeg + kGK, kend, tid);
    tb1.load(rb, ldb, plane_bytes, n0, p.N, kbeg + kGK, kend, tid);
    add_row_sum(ta0);
    ta0.store(lds, tid);
    tb0.store(lds + kV2A, tid);
    ta0.load(ra, lda, m0, p.M, kbeg + 2 * kGK, kend, tid);
    tb0.load(rb, ldb, plane_bytes, n0, p.N, kbeg + 2 * kGK, kend, tid);
    __syncthreads();
    const int fa_off = (64 * wm + (lane & 31)) * kGRow + (lane >> 5) * 32;
    const int fb_off = kV2A + (64 * wn + (lane & 31)) * kBRow + (lane >> 5) * 16;
    V2Frags<BMODE> f0, f1;
    f0.fetch(lds + fa_off, lds + fb_off, 0);
    f1 = f0;   // (defined also when an ablation build skips the fetches)
    // one step: `cur` holds the step's tile; (ta, tb) hold the NEXT step's tile (stored to `nxt` now) and are reloaded
    // with the tile three steps on
    auto step = [&](V2LoadA<A_KMAJOR> &ta, V2LoadB<BMODE> &tb, char *cur, char *nxt, int k0) {
        add_row_sum(ta);
#if SDETR_GX3_ABLATE != 4
        ta.store(nxt, tid);   // (nobody reads `nxt`: its last readers fetched their fragments before the previous barrier)
        tb.store(nxt + kV2A, tid);
#endif
#if SDETR_GX3_ABLATE != 3
        ta.load(ra, lda, m0, p.M, k0 + 3 * kGK, kend, tid);
        tb.load(rb, ldb, plane_bytes, n0, p.N, k0 + 3 * kGK, kend, tid);
#endif
        // first k-half: its MFMAs run while the second half's fragments are read and split
#if SDETR_GX3_ABLATE != 6
        f1.fetch(cur + fa_off, cur + fb_off, 1);
#endif
        f0.mfmas(acc);
#if SDETR_GX3_ABLATE != 5
        __syncthreads();      // the next tile is complete in `nxt`; everybody holds what it needs from `cur`
#endif
#if SDETR_GX3_ABLATE != 6
        f0.fetch(nxt + fa_off, nxt + fb_off, 0);
#endif
        f1.mfmas(acc);
    };
    for (int k0 = kbeg; k0 < kend; k0 += 2 * kGK) {
        step(ta1, tb1, lds, lds + kStage, k0);
        if (k0 + kGK < kend) step(ta0, tb0, lds + kStage, lds, k0 + kGK);
    }

    if (want_row_sum) {   // (uniform per workgroup) 8 threads hold pieces of each row's sum: meet in LDS, one atomic per row
        __syncthreads();
        float *red = reinterpret_cast<float *>(lds);   // [8][256]
        const int u = tid & 255, half = tid >> 8, kb = u & 7, mb = u >> 3;
        *reinterpret_cast<float4 *>(red + kb * 256 + 128 * half + 4 * mb) = row_sum;
        __syncthreads();
        if (tid < 256 && m0 + tid < p.M) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) t += red[j * 256 + tid];
            unsafeAtomicAdd(p.a_row_sum + m0 + tid, t);
        }
    }
    const bool add_bias = p.bias && blockIdx.z == 0;
    gemm_store_c(p, acc, m0 + 64 * wm, n0 + 64 * wn, lane, add_bias);
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

// the second-generation kernel (256 x 128 tiles, 8 waves): operands below 2 GB (32-bit buffer offsets)
template <bool AK, int BMODE>
static int launch_gemm_x3_v2(hipStream_t s, const GemmArgs &a, int splits)
{
    constexpr int lds_bytes = 2 * (kV2A + kV2Bp);
    static DeviceOnce once;
    allow_dynamic_lds(gemm_x3_v2_kernel<AK, BMODE>, once, lds_bytes);
    const dim3 grid((unsigned)((a.N + kV2N - 1) / kV2N), (unsigned)((a.M + kV2M - 1) / kV2M), (unsigned)splits);
    hipLaunchKernelGGL((gemm_x3_v2_kernel<AK, BMODE>), grid, dim3(kV2Threads), lds_bytes, s, a);
    return check_launch("gemm_x3");
}

// Which generation takes a product.  Measured (benchmarks/gemm_x3_bench.py, MI355X): the 256 x 128 tiles win where an
// output or reduction dimension is long (feed-forward shapes: 154-190 us against 186-217), the 128 x 128 tiles with two
// workgroups per CU where the output is a few tiles wide (256 / 384 features: 56-66 us against 62-101).
// sdetr_gemm_x3_generation(1 | 2) pins the first / second generation for the calling thread (0 = this rule): both
// generations compute the same six-term products, the parity tests run every shape on each.
static thread_local int t_generation = 0;
extern "C" int sdetr_gemm_x3_generation(int generation)
{
    const int before = t_generation;
    if (generation >= 0 && generation <= 2) t_generation = generation;
    return before;
}
static bool gemm_x3_use_v2(int M, int N, int K)
{
    if (t_generation) return t_generation == 2;
    const int longest = N > K ? N : K;
    return M >= 128 && longest >= 1024;   // (a 128-row output still wins on the 256-row tiles when the reduction is long: 40 vs 55 us)
}

// extent in bytes of a 2-d operand (rows x width elements of `elem` bytes, rows `ld` elements apart); 0 when it does not
// fit the 31 bits a buffer resource's offsets are trusted with here
static uint32_t operand_bytes(int64_t rows, int64_t width, int64_t ld, int elem, int64_t extra = 0)
{
    const int64_t b = (rows > 0 ? ((rows - 1) * ld + width) : 0) * elem + extra;
    return b > 0 && b < ((int64_t)1 << 31) ? (uint32_t)b : 0u;
}

// C[M,N] = sum_k A(m,k) B(n,k) (+ bias[n]).  a_kmajor: A(m,k) = a[m * lda + k], else a[k * lda + m]; b likewise with n.
// reduction_splits > 1: the reduction is cut into that many slices whose partial products are accumulated into C with
// fp32 atomics -- C must be zero on entry.  Alignment: every operand 16-byte aligned, its leading dimension a multiple
// of 4; a k-major operand needs K % 4 == 0, the other kind its row count % 4 == 0.  a_row_sum (optional, [M], zero on
// entry, reduction-major A only): receives sum_k A(m, k) -- the bias gradient that comes with dw = dy^T x.
// epilogue (SDETR_GEMM_EPI_NONE / _RELU / _GATE, include/salience_hip.h) with its gate matrix [M, N] (rows ldg apart; _GATE
// only): unsplit reductions only.
extern "C" int sdetr_gemm_x3_epilogue_f32(sdetr_stream_t stream, const float *a, int64_t lda, int a_kmajor, const float *b,
                                          int64_t ldb, int b_kmajor, float *c, int64_t ldc, int M, int N, int K,
                                          const float *bias, int reduction_splits, float *a_row_sum, int epilogue,
                                          const float *gate, int64_t ldg)
{
    if (M < 0 || N < 0 || K < 0) return fail("gemm_x3: negative size");
    if (M == 0 || N == 0) return 0;
    if (!a || !b || !c) return fail("gemm_x3: null pointer");
    if (epilogue < 0 || epilogue > 2) return fail("gemm_x3: unknown epilogue %d", epilogue);
    if (epilogue && reduction_splits > 1) return fail("gemm_x3: an epilogue needs an unsplit reduction");
    if (epilogue == 2 && (!gate || ldg < N)) return fail("gemm_x3: the gate epilogue needs a gate matrix with rows of >= N elements");
    if (b_kmajor == 2) {   // pre-split B planes [3][N][K] bf16 (sdetr_gemm_x3_presplit), rows ldb elements apart
        if ((lda & 3) || ((uintptr_t)a & 15) || ((uintptr_t)b & 15) || (ldb & 7) || (K & 7) || ldb < K)
            return fail("gemm_x3: a pre-split B needs K % 8 == 0, rows of >= K elements, 16-byte aligned operands");
        if ((a_kmajor && (K & 3)) || (!a_kmajor && (M & 3))) return fail("gemm_x3: M=%d K=%d do not meet A's alignment rule", M, K);
        if (a_row_sum) return fail("gemm_x3: a_row_sum is not available with a pre-split B");
        if (reduction_splits < 1) reduction_splits = 1;
        GemmArgs g{};
        g.a = a; g.b = b; g.c = c; g.bias = bias; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
        g.b_plane = (int64_t)N * ldb;
        g.epilogue = epilogue; g.gate = gate; g.ldg = ldg;
        const int steps = (K + kGK - 1) / kGK;
        int splits = reduction_splits > steps ? (steps > 0 ? steps : 1) : reduction_splits;
        g.k_per_split = ((steps + splits - 1) / splits) * kGK;
        splits = g.k_per_split > 0 ? (K + g.k_per_split - 1) / g.k_per_split : 1;
        if (splits < 1) splits = 1;
        g.atomic = splits > 1;
        hipStream_t hs = static_cast<hipStream_t>(stream);
        g.a_bytes = a_kmajor ? operand_bytes(M, K, lda, 4) : operand_bytes(K, M, lda, 4);
        g.b_bytes = operand_bytes(N, K, ldb, 2, 2 * g.b_plane * 2);
        if (gemm_x3_use_v2(M, N, K) && g.a_bytes && g.b_bytes) {
            if (a_kmajor) return launch_gemm_x3_v2<true, 2>(hs, g, splits);
            return launch_gemm_x3_v2<false, 2>(hs, g, splits);
        }
        const dim3 grid((unsigned)((N + kGTile - 1) / kGTile), (unsigned)((M + kGTile - 1) / kGTile), (unsigned)splits);
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
    g.epilogue = epilogue; g.gate = gate; g.ldg = ldg;
    g.a = a; g.b = b; g.c = c; g.bias = bias; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    const int steps = (K + kGK - 1) / kGK;
    int splits = reduction_splits > steps ? (steps > 0 ? steps : 1) : reduction_splits;
    g.k_per_split = ((steps + splits - 1) / splits) * kGK;
    splits = g.k_per_split > 0 ? (K + g.k_per_split - 1) / g.k_per_split : 1;
    if (splits < 1) splits = 1;
    g.atomic = splits > 1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    g.a_bytes = a_kmajor ? operand_bytes(M, K, lda, 4) : operand_bytes(K, M, lda, 4);
    g.b_bytes = b_kmajor ? operand_bytes(N, K, ldb, 4) : operand_bytes(K, N, ldb, 4);
    if (gemm_x3_use_v2(M, N, K) && g.a_bytes && g.b_bytes) {
        if (a_kmajor && b_kmajor) return launch_gemm_x3_v2<true, 1>(s, g, splits);
        if (a_kmajor && !b_kmajor) return launch_gemm_x3_v2<true, 0>(s, g, splits);
        if (!a_kmajor && b_kmajor) return launch_gemm_x3_v2<false, 1>(s, g, splits);
        return launch_gemm_x3_v2<false, 0>(s, g, splits);
    }
    if (a_kmajor && b_kmajor) return launch_gemm_x3<true, true>(s, g, splits);
    if (a_kmajor && !b_kmajor) return launch_gemm_x3<true, false>(s, g, splits);
    if (!a_kmajor && b_kmajor) return launch_gemm_x3<false, true>(s, g, splits);
    return launch_gemm_x3<false, false>(s, g, splits);
}

extern "C" int sdetr_gemm_x3_f32(sdetr_stream_t stream, const float *a, int64_t lda, int a_kmajor, const float *b,
                                 int64_t ldb, int b_kmajor, float *c, int64_t ldc, int M, int N, int K,
                                 const float *bias, int reduction_splits, float *a_row_sum)
{
    return sdetr_gemm_x3_epilogue_f32(stream, a, lda, a_kmajor, b, ldb, b_kmajor, c, ldc, M, N, K, bias, reduction_splits,
                                      a_row_sum, 0, nullptr, 0);
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
