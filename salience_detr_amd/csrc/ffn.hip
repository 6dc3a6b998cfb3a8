// The encoder layer's feed-forward block in one launch (models/bricks/salience_transformer.py:347-351):
//
//     out = LayerNorm(x + W2 relu(W1 x + b1) + b2)            x [T,256] bf16, W1 [F,256], W2 [256,F], F = 2048
//
// The framework path is two library GEMMs with the [T,F] hidden state written to and re-read from HBM between
// them (93 MB each way for the 22 726 queries of encoder layer 0) plus a residual/LayerNorm pass; with K = 256 those
// GEMMs are epilogue-bound and reach 300-500 TFLOP/s.  Here the hidden state never leaves the register file:
//
//   * a wave owns 32 tokens for the whole kernel: their activations X^T live in 64 VGPRs as the B operands of
//     v_mfma_f32_32x32x16_bf16 (lane (t,h) holds X[t][16k+8h .. +7]), the output Y^T [256 x 32] in 128 accumulator
//     registers;
//   * the hidden dimension is walked in chunks of 32: H^T = W1[chunk] X^T (16 MFMAs), bias + ReLU + bf16 pack in
//     registers, Y^T += W2[:,chunk] H^T (16 MFMAs).  Computing the TRANSPOSED products makes the accumulator layout
//     of the first product (lane = token column, registers = hidden rows 8g+4h+{0..3}) exactly a B operand of the
//     second once W2's contraction index is permuted to match -- no LDS round trip, no cross-lane traffic;
//   * the weights are the only stream: both matrices are pre-packed (sdetr_ffn_pack_bf16) into 32 KB chunks of
//     lane-ordered 1 KB MFMA A-fragments.  Four LOADER waves (one per SIMD, next to the compute wave) bring them
//     global -> registers -> LDS: a chunk is requested four iterations before it is written into one of four LDS
//     buffers (inline-asm loads with hand-counted s_waitcnt; see the loader).  Each fragment is read from LDS once per
//     compute wave; L2 sees 2 MB per 128 tokens.
//   * the main loop alternates the first product of chunk jt+1 with the second product of chunk jt (two independent
//     MFMA streams), fragments reach the MFMAs through a 4-deep register ring.
//   * epilogue in registers: the accumulator starts as b2, the residual comes out of the X^T operand registers through
//     v_permlane32_swap, LayerNorm over the 256 channels a lane pair holds, bf16 store.
//
// Bound: bf16 MFMA peak -- 4*F*256 flops per token; one wave issues 64 chunks x 32 MFMAs (1024 cycles per chunk).
// Measured on MI355X (hipGraph replay, benchmarks/ffn_hidden_sweep.py, F = 2048): 63 us at 22 726 tokens (178 blocks),
// 51 us at 1800 (15 blocks) = ~13 us of launch + prologue + epilogue and 0.63-0.78 us (1500-1850 cycles) per chunk.
// History of the loop, each step measured:
//   * LDS-DMA (global_load_lds_dwordx4) issued by the compute waves: 64 us total; by dedicated loader waves: 54 us.
//   * PMC (matrix pipe busy 37 %) and ablations then showed that NOTHING inside the loop set its pace: without the
//     MFMAs, without the fragment reads, without any loader work the kernel took the same 73-77 us.  A sweep over F
//     separated the costs: 26 us were fixed -- an epilogue that reloaded the residual with 32 row-strided loads per lane
//     and spilled (tuple copies of the accumulators) -- and the per-chunk time is what the LDS read latency (~185 cycles)
//     allows a 4-deep ring (8 would spill inside the loop): 32 reads x latency / 4.
//   * what did NOT help and was dropped: rotating the chunk order per block (L2 channel spreading); in the epilogue (its
//     32 row-strided 8-byte stores are 4.2 of the 14 us of fixed cost at 178 blocks, 2.7 at 15) both an LDS-staged
//     coalesced store of whole rows (16.5 us fixed) and v_permlane32_swap + 16-byte stores (17.1 us).
//
// A block keeps a CU for the whole hidden dimension, so small token counts leave most of the chip idle: the hidden
// dimension can be cut into `nsplit` pieces per token block (fp32 partial products + ffn_reduce_ln_kernel).
#include "common.h"
#include "class_head_core.h"

namespace sdetr {

constexpr int kFE = 256;               // embed dim
constexpr int kFChunk = 32;            // hidden units per chunk
constexpr int kFChunkBytes = 32768;    // 16 KB of W1 rows + 16 KB of W2 columns, as 1 KB A-fragments
constexpr int kFTokWave = 32, kFTokBlock = 128;

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

// End-of-layer row bookkeeping (sdetr_advance_rows): row i of image b goes to sorted_result[b,i] when live
// (i < count[b]) and to next_query[b,i] when i < next_rows -- live rows the layer output, the others the never-updated
// original tokens[b, sorted_index[b,i]].
struct FfnAdvance {
    bf16_t *sorted_result;        // [B, sorted_rows, 256]
    bf16_t *next_query;           // [B, next_rows, 256] or NULL
    const bf16_t *tokens;         // [B, spatial_size, 256]
    const int64_t *sorted_index;  // rows index_batch_stride apart
    int64_t index_batch_stride;
    const int64_t *count;         // [B] or NULL
    int rows, sorted_rows, next_rows, spatial_size;
};

struct FfnArgs {
    const bf16_t *x;        // [T, 256]
    const char *pw;         // packed weights, nchunk * 32 KB
    const float *b1;        // [F]
    const float *b2, *gamma, *beta;   // [256]
    float eps;
    bf16_t *out;            // [T, 256]
    int T, nchunk;
    int nsplit;             // hidden-dimension pieces per token block (1: the block finishes its tokens itself)
    float *partial;         // nsplit > 1: [nsplit, T, 256] fp32 partial products, finished by ffn_reduce_ln_kernel
    // TAIL form (ffn_fused_kernel<true>): the deformable attention's tail runs in front of the feed-forward,
    //     x = LayerNorm1(res + Wo s + bo)        (salience_transformer.py:390-391, ms_deform_attn.py:375)
    // `pw` then starts with kTailChunks chunks of Wo fragments, `x` is not read.
    const bf16_t *s;        // [T, 256] sampled heads (the MSDA kernel's output)
    const bf16_t *res;      // [T, 256] the layer's queries (residual)
    const float *bo, *g1, *be1;   // [256] output_proj bias, norm1 weight / bias
    float eps1;
    // NEXT form (ffn_fused_kernel<., true>, nsplit == 1): the epilogue does the row bookkeeping itself (`adv`, no `out`)
    // and computes the NEXT layer's class score max_c(Wc q + bc) * fg of the rows it hands on
    // (salience_transformer.py:462, 366) -- `pw` then ends with kClsChunks chunks of Wc fragments.
    FfnAdvance adv;
    const float *cls_bias;  // [96]: the class head's bias, -inf on the padded classes
    const float *fg;        // [B, >= next_rows] foreground scores of the rows, images fg_bs apart
    int64_t fg_bs;
    float *cmax;            // [B, next_rows]
};

constexpr int kClsChunks = 2;    // Wc [96 x 256] (91 classes, zero padded) as 48 fragments: f = 3 (k-step) + tile
constexpr int kTailChunks = 4;   // Wo [256 x 256] as 128 1-KB A-fragments: chunk a = k-steps 4a .. 4a+3 of all 8 output tiles

__device__ __forceinline__ f32x16_t mfma_bf16(uint4 a, uint4 b, f32x16_t c)
{
    return mfma_act_32x32x16(a, b, c);
}

typedef short i16x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const char *lds_cptr_t;

// The first product accumulates in ARCHITECTURAL VGPRs (the VALU that follows reads it; the builtin would put it in
// AGPRs, which costs a v_accvgpr_read per value plus, here, a 32-register shuffle of the output tile whose AGPRs the
// allocator reuses).  Inline asm is the only way to choose the register class.
__device__ __forceinline__ void mfma_bf16_vgpr(uint4 a, uint4 b, f32x16_t &c)
{
    const u32x4_t av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
    asm volatile("v_mfma_f32_32x32x16_" SDETR_ACT_MFMA_SUFFIX " %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv));
}

__device__ __forceinline__ uint4 lds_read16(lds_cptr_t p)
{
    const u32x4_t v = *reinterpret_cast<const __attribute__((address_space(3))) u32x4_t *>(p);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// ReLU on a packed bf16 pair: as signed 16-bit integers every negative value (sign bit set, -0.0 included) is < 0
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t v)
{
    const i16x2_t z = {0, 0};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, v), z));
}

// Block = 4 compute waves (32 tokens each) + 4 LOADER waves, one of each per SIMD.  (All waves get the same register
// allocation, so the kernel has to fit 256 registers for two waves per SIMD.)
constexpr int kFThreads = 512;

// TAIL: the layer's attention tail (output_proj + residual + norm1) in front, see FfnArgs.  The weight STREAM of a
// block is then [tail chunks 0..3][its piece of the hidden dimension]: stream chunk c lives in LDS buffer c & 3 and the
// loaders / barriers below count stream chunks; the feed-forward part addresses its chunk j as stream chunk NT + j.
template <bool TAIL, bool NEXT = false>
__global__ void __launch_bounds__(kFThreads, 1) ffn_fused_kernel(FfnArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *wbuf = lds;                                                  // 4 chunk buffers
    float *b1s = reinterpret_cast<float *>(lds + 4 * kFChunkBytes);    // [F]
    float *par = b1s + p.nchunk * kFChunk;                             // b2 | gamma | beta (| bo | gamma1 | beta1)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // hidden split: block = (token block, piece); piece sp walks chunks [c0, c0 + nloc) of the hidden dimension
    const int tblock = blockIdx.x / p.nsplit, sp = blockIdx.x - tblock * p.nsplit;
    const int c0 = (int)((int64_t)sp * p.nchunk / p.nsplit);
    const int nloc = (int)((int64_t)(sp + 1) * p.nchunk / p.nsplit) - c0;
    constexpr int NT = TAIL ? kTailChunks : 0;
    constexpr int NC = NEXT ? kClsChunks : 0;
    const int nl = nloc + NT + NC;                                     // chunks in this block's stream
    const char *pw = p.pw;
    // stream chunk -> chunk of the packed buffer [tail][feed-forward: nchunk][class head]
    auto chunk_at = [&](int c) {
        return c < NT ? c : (c < NT + nloc ? NT + c0 + (c - NT) : NT + p.nchunk + (c - NT - nloc));
    };

    if (wave >= 4) {
        // ---- loader wave: a quarter (8 KB) of every 32 KB chunk, global -> registers -> LDS, FOUR chunk buffers ----
        // (LDS-DMA, global_load_lds_dwordx4, would need no registers, but a copy then has to complete within the one
        // iteration between two barriers; ordinary loads into the loader's own, otherwise idle registers can run four
        // chunks ahead.)
        // Chunk c lives in buffer c & 3.  During main-loop iteration jt the compute waves read chunk jt+1 (first
        // product of the NEXT chunk), chunk jt (second product) and the head of chunk jt+2 (for the register ring);
        // chunk jt+3 is written into the buffer chunk jt-1 left, and chunk jt+4 is on its way from L2.
        const int lw = wave - 4;
        const uint4 *src = reinterpret_cast<const uint4 *>(pw) + lw * 512 + lane;
        char *dst = wbuf + lw * 8192 + lane * 16;
        // Chunk c travels in register set c & 3 and is requested FOUR iterations before it is written to LDS.
        // The loads are inline asm with hand-counted waits: hipcc's own wait insertion merges the conditional paths of
        // the tail and then drains every outstanding load (vmcnt(0)) in front of each store.  Loads complete in order,
        // so "at most 24 outstanding" = the oldest set has arrived.  (Plain named registers: an array passed to a
        // helper stays in scratch memory under the asm memory clobbers.)
#define SDETR_GLD(dst, base, off) asm volatile("global_load_dwordx4 %0, %1, off offset:" #off : "=v"(dst) : "v"(base) : "memory")
#define SDETR_FETCH8(P, c)                                                                                        \
    {                                                                                                             \
        const uint4 *q0_ = src + (int64_t)chunk_at(c) * 2048, *q1_ = q0_ + 256;                                   \
        SDETR_GLD(P##0, q0_, 0); SDETR_GLD(P##1, q0_, 1024); SDETR_GLD(P##2, q0_, 2048); SDETR_GLD(P##3, q0_, 3072); \
        SDETR_GLD(P##4, q1_, 0); SDETR_GLD(P##5, q1_, 1024); SDETR_GLD(P##6, q1_, 2048); SDETR_GLD(P##7, q1_, 3072); \
    }
#define SDETR_STASH8(P, c)                                                                                        \
    {                                                                                                             \
        char *d_ = dst + ((c) & 3) * kFChunkBytes;                                                                \
        *reinterpret_cast<uint4 *>(d_) = P##0; *reinterpret_cast<uint4 *>(d_ + 1024) = P##1;                      \
        *reinterpret_cast<uint4 *>(d_ + 2048) = P##2; *reinterpret_cast<uint4 *>(d_ + 3072) = P##3;               \
        *reinterpret_cast<uint4 *>(d_ + 4096) = P##4; *reinterpret_cast<uint4 *>(d_ + 5120) = P##5;               \
        *reinterpret_cast<uint4 *>(d_ + 6144) = P##6; *reinterpret_cast<uint4 *>(d_ + 7168) = P##7;               \
    }
        uint4 ra0, ra1, ra2, ra3, ra4, ra5, ra6, ra7, rb0, rb1, rb2, rb3, rb4, rb5, rb6, rb7;
        uint4 rc0, rc1, rc2, rc3, rc4, rc5, rc6, rc7, rd0, rd1, rd2, rd3, rd4, rd5, rd6, rd7;
        // prologue: chunks 0..2 into LDS, chunks 3..6 on their way
        SDETR_FETCH8(ra, 0);
        SDETR_FETCH8(rb, nl > 1 ? 1 : 0);
        SDETR_FETCH8(rc, nl > 2 ? 2 : 0);
        SDETR_FETCH8(rd, nl > 3 ? 3 : 0);
        asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        SDETR_STASH8(ra, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SDETR_FETCH8(ra, nl > 4 ? 4 : 0);
        asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        if (nl > 1) SDETR_STASH8(rb, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SDETR_FETCH8(rb, nl > 5 ? 5 : 0);
        asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        if (nl > 2) SDETR_STASH8(rc, 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SDETR_FETCH8(rc, nl > 6 ? 6 : 0);
        __builtin_amdgcn_s_barrier();
        // iteration j: past its barrier every compute wave is done with chunk j-1, whose buffer takes chunk j+3; the
        // register set that held it goes back to L2 for chunk j+7 (the tail re-requests chunk 0: every phase issues
        // exactly 8 loads, which is what keeps the wait count a constant)
#define SDETR_PHASE(P, j)                                                                                         \
    {                                                                                                             \
        __builtin_amdgcn_s_barrier();                                                                             \
        asm volatile("s_waitcnt vmcnt(24)" ::: "memory");                                                         \
        if ((j) + 3 < nl) SDETR_STASH8(P, (j) + 3);                                                             \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                        \
        SDETR_FETCH8(P, (j) + 7 < nl ? (j) + 7 : 0);                                                            \
    }
        for (int jt = 0; jt + 1 < nl; jt += 4) {
            SDETR_PHASE(rd, jt);
            if (jt + 2 >= nl) break;
            SDETR_PHASE(ra, jt + 1);
            if (jt + 3 >= nl) break;
            SDETR_PHASE(rb, jt + 2);
            if (jt + 4 >= nl) break;
            SDETR_PHASE(rc, jt + 3);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the dummy requests of the tail)
#undef SDETR_PHASE
#undef SDETR_FETCH8
#undef SDETR_STASH8
#undef SDETR_GLD
        return;
    }

    const int t = lane & 31, h = lane >> 5;
    const int tok = tblock * kFTokBlock + wave * kFTokWave + t;
    const bool valid = tok < p.T;
    const int64_t row = (int64_t)(valid ? tok : p.T - 1) * kFE;

    for (int i = tid; i < p.nchunk * kFChunk; i += kBlock) b1s[i] = p.b1[i];
    par[tid] = p.b2[tid];
    par[kFE + tid] = p.gamma[tid];
    par[2 * kFE + tid] = p.beta[tid];
    if (TAIL) {
        par[3 * kFE + tid] = p.bo[tid];
        par[4 * kFE + tid] = p.g1[tid];
        par[5 * kFE + tid] = p.be1[tid];
    }
    if (NEXT && tid < 96) par[6 * kFE + tid] = p.cls_bias[tid];

    uint4 xb[16];   // X^T as B operands: k-step ks covers channels 16ks + 8h .. +7 of my token (TAIL: first S^T)
    {
        const bf16_t *xin = TAIL ? p.s : p.x;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) xb[ks] = *reinterpret_cast<const uint4 *>(xin + row + 16 * ks + 8 * h);
    }
    // Consume the loads HERE: hipcc places the wait for a pending load at its first use, which would be inside the
    // main loop -- a vmcnt(0) there every iteration also drains the LDS copies it cannot see.
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) asm volatile("" : "+v"(xb[ks].x), "+v"(xb[ks].y), "+v"(xb[ks].z), "+v"(xb[ks].w));

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the staged vectors; the loaders bring chunks 0..2
    __builtin_amdgcn_s_barrier();

#ifndef SDETR_FFN_RING
#define SDETR_FFN_RING 4   // (benchmarks/lib_variant.sh ... -DSDETR_FFN_RING=5 | 6: 72-156 bytes of scratch per lane and the
                           // same launch times, 78.5 / 71.9 / 47.2 -> 78.3 / 71.6 / 47.0 (5) and 80.0 / 72.2 / 47.1 (6) us on
                           // one box, round 6 -- the ring's depth is not what the loop waits for)
#endif
    constexpr int R = SDETR_FFN_RING;
    uint4 ring[R];
    const lds_cptr_t lbase = (lds_cptr_t)wbuf + lane * 16;
    auto chunk_lds = [&](int c) { return lbase + (c & 3) * kFChunkBytes; };   // c = STREAM chunk
    // x <-> the form in which lane (t, h) holds the channels of ITS accumulator quads: upper dwords of row 0 <-> lower
    // dwords of row 1 (v_permlane32_swap); afterwards (x, y) of piece ks are quad g = 2 (ks & 1) of e-tile ks >> 1 and
    // (z, w) quad g + 1.  The exchange is its own inverse.
    auto swap_halves = [&]() {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const auto s0 = __builtin_amdgcn_permlane32_swap(xb[ks].x, xb[ks].z, false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(xb[ks].y, xb[ks].w, false, false);
            xb[ks] = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
    };
    auto residual = [&](int et, int g, float (&r)[4]) {   // (after swap_halves)
        const uint4 q = xb[2 * et + (g >> 1)];
        const uint32_t d0 = (g & 1) ? q.z : q.x, d1 = (g & 1) ? q.w : q.y;
        r[0] = act_lo(d0); r[1] = act_hi(d0); r[2] = act_lo(d1); r[3] = act_hi(d1);
    };

    f32x16_t yacc[8];
    if (TAIL) {
        // ---- Z^T = Wo S^T + bo in the output accumulators: chunk a holds k-steps 4a .. 4a+3 of the 8 output tiles
        // (fragment f = 8 (k-step) + tile: eight independent accumulator chains).  Once a chunk's MFMAs are issued its
        // four k-steps of S^T are dead and the registers take the RESIDUAL rows -- their trip to memory runs under the
        // rest of the product. ----
#pragma unroll
        for (int et = 0; et < 8; ++et)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *reinterpret_cast<const float4 *>(par + 3 * kFE + 32 * et + 8 * g + 4 * h);
                yacc[et][4 * g] = bv.x; yacc[et][4 * g + 1] = bv.y; yacc[et][4 * g + 2] = bv.z; yacc[et][4 * g + 3] = bv.w;
            }
#pragma unroll
        for (int a = 0; a < kTailChunks; ++a) {
            if (a > 0) __builtin_amdgcn_s_barrier();   // stream chunk a is in LDS; the loaders learn that a - 1 is done with
            const lds_cptr_t ca = chunk_lds(a);
#pragma unroll
            for (int f = 0; f < R; ++f) ring[f] = lds_read16(ca + f * 1024);
#pragma unroll
            for (int f = 0; f < 32; ++f) {
                yacc[f & 7] = mfma_bf16(ring[f % R], xb[4 * a + (f >> 3)], yacc[f & 7]);
                if (f + R < 32) ring[f % R] = lds_read16(ca + (f + R) * 1024);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                xb[4 * a + q] = *reinterpret_cast<const uint4 *>(p.res + row + 16 * (4 * a + q) + 8 * h);
        }
        __builtin_amdgcn_s_barrier();   // done with the tail chunks (the feed-forward's first chunks are in LDS)
        // ---- x = LayerNorm1(z + res), two passes over read-only accumulators as in the epilogue below ----
        swap_halves();
        float sum = 0.f, sq = 0.f;
#pragma unroll
        for (int et = 0; et < 8; ++et) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float r[4];
                residual(et, g, r);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = yacc[et][4 * g + i] + r[i];
                    sum += v;
                    sq = fmaf(v, v, sq);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        sum += __shfl_xor(sum, 32);
        sq += __shfl_xor(sq, 32);
        const float mean = sum * (1.f / kFE);
        const float rstd = rsqrtf(fmaxf(sq * (1.f / kFE) - mean * mean, 0.f) + p.eps1);
        const float shift = -mean * rstd;
        // (the residual pieces are made opaque between the passes: otherwise the compiler keeps pass 1's 128 unpacked
        // residual values alive across the reduction instead of unpacking again -- 300 spilled registers)
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) asm volatile("" : "+v"(xb[ks].x), "+v"(xb[ks].y), "+v"(xb[ks].z), "+v"(xb[ks].w));
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int et = ks >> 1, g0 = 2 * (ks & 1);
            uint32_t d[4];
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
                const int g = g0 + gg, e0 = 32 * et + 8 * g + 4 * h;
                const float4 gv = *reinterpret_cast<const float4 *>(par + 4 * kFE + e0);
                const float4 be = *reinterpret_cast<const float4 *>(par + 5 * kFE + e0);
                float r[4];
                residual(et, g, r);
                const float x0 = fmaf(fmaf(yacc[et][4 * g] + r[0], rstd, shift), gv.x, be.x);
                const float x1 = fmaf(fmaf(yacc[et][4 * g + 1] + r[1], rstd, shift), gv.y, be.y);
                const float x2 = fmaf(fmaf(yacc[et][4 * g + 2] + r[2], rstd, shift), gv.z, be.z);
                const float x3 = fmaf(fmaf(yacc[et][4 * g + 3] + r[3], rstd, shift), gv.w, be.w);
                d[2 * gg] = pack_act2(x0, x1);
                d[2 * gg + 1] = pack_act2(x2, x3);
            }
            xb[ks] = make_uint4(d[0], d[1], d[2], d[3]);   // x (bf16) in the accumulator-quad form
            // (pinned here: the code-sinking pass otherwise moves this arithmetic down to the first reader of x, past
            // the scheduling fence, and the gamma / beta reads above -- which cannot follow -- stay alive and spill)
            asm volatile("" : "+v"(xb[ks].x), "+v"(xb[ks].y), "+v"(xb[ks].z), "+v"(xb[ks].w));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // the output accumulator STARTS as b2 (a piece of a split hidden dimension starts at zero: its second pass adds b2
    // and the residual -- except in the TAIL form, where the residual x exists nowhere but here: piece 0 starts at
    // b2 + x and the second pass only sums)
#pragma unroll
    for (int et = 0; et < 8; ++et) {
        f32x16_t start;   // (a fresh tuple: element writes into the old one would tie the new value to z's registers)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 bv = *reinterpret_cast<const float4 *>(par + 32 * et + 8 * g + 4 * h);
            if (p.nsplit > 1) {
                if (TAIL && sp == 0) {
                    float r[4];
                    residual(et, g, r);
                    bv = make_float4(bv.x + r[0], bv.y + r[1], bv.z + r[2], bv.w + r[3]);
                } else {
                    bv = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            start[4 * g] = bv.x; start[4 * g + 1] = bv.y; start[4 * g + 2] = bv.z; start[4 * g + 3] = bv.w;
        }
        yacc[et] = start;
    }
    if (TAIL) {
        __builtin_amdgcn_sched_barrier(0);
        swap_halves();   // x^T as B operands
    }

    // SOFTWARE PIPELINE over the chunks.  A chunk is two dependent products (H = relu(W1 X + b1), Y += W2 H).  The first
    // is a chain of 16 MFMAs into ONE accumulator and the second pairs its MFMAs on each of 8 accumulators: issued
    // back to back, a dependent MFMA waits for the full 16-pass latency of its predecessor, not the 8-pass issue
    // interval (PMC: the matrix pipe was busy 37 % of the kernel, SQ_WAIT_INST_ANY 38 % of the compute waves' cycles).
    // So iteration jt ALTERNATES the first product of chunk jt+1 with the second product of chunk jt -- two independent
    // streams, every dependent pair now two or more MFMAs apart -- and walks the second product k-block-major (the 8
    // e-tiles of k-block 0, then those of k-block 1).  The conversion of the hidden tile (ReLU, bf16) sits at the
    // iteration boundary.
    //
    // The accumulator of the first product STARTS as the bias (four 16-byte LDS reads land in its register quads -- no
    // zeroing, no adds), ReLU runs on packed bf16 pairs.  A fragments come from LDS through a 4-deep ring of registers:
    // slot s of an iteration (even: W1[jt+1] k-step s/2, odd: W2[jt] fragment of (k-block, e-tile) = (s>>4, (s>>1)&7))
    // is requested 4 MFMAs before its use and the slot is refilled right AFTER the MFMA that consumed it; the last four
    // requests fetch the head of the next iteration's stream.
    const lds_cptr_t bias_base = (lds_cptr_t)(const char *)b1s + 16 * h;
    f32x16_t hacc;
    auto load_bias = [&](int chunk) {
        const lds_cptr_t bb = bias_base + chunk * (kFChunk * 4);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint4 bv = lds_read16(bb + 32 * g);
            hacc[4 * g] = __uint_as_float(bv.x);
            hacc[4 * g + 1] = __uint_as_float(bv.y);
            hacc[4 * g + 2] = __uint_as_float(bv.z);
            hacc[4 * g + 3] = __uint_as_float(bv.w);
        }
    };
    uint4 hp[2];   // packed relu(H) of the chunk in its second product: k-blocks 0 and 1
    // ReLU + bf16 of the finished hidden tile, then the NEXT chunk's bias into the freed accumulator registers
    auto convert = [&](int next_bias_chunk) {
        // the VALU below reads registers an inline-asm MFMA wrote: hipcc cannot see that, so the wait states
        // (XDL write -> VALU read) are spelled out; the accumulator is an operand so that its readers stay below
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(hacc));
        hp[0] = make_uint4(relu_bf16x2(pack_act2(hacc[0], hacc[1])), relu_bf16x2(pack_act2(hacc[2], hacc[3])),
                           relu_bf16x2(pack_act2(hacc[4], hacc[5])), relu_bf16x2(pack_act2(hacc[6], hacc[7])));
        hp[1] = make_uint4(relu_bf16x2(pack_act2(hacc[8], hacc[9])), relu_bf16x2(pack_act2(hacc[10], hacc[11])),
                           relu_bf16x2(pack_act2(hacc[12], hacc[13])), relu_bf16x2(pack_act2(hacc[14], hacc[15])));
        // pin the conversion HERE: once it has run the accumulator registers are free and the next bias can land in
        // them -- were its live range to reach past load_bias, the allocator would have to move the accumulator
        // between two asm MFMAs, a copy that reads registers the MFMA before it has not written yet
        asm volatile("" : "+v"(hp[0].x), "+v"(hp[0].y), "+v"(hp[0].z), "+v"(hp[0].w), "+v"(hp[1].x), "+v"(hp[1].y), "+v"(hp[1].z), "+v"(hp[1].w));
        load_bias(next_bias_chunk);
    };
    // byte offset inside a chunk of the W2 fragment that the q-th MFMA of the k-block-major walk uses
    auto w2_frag = [](int q) { return (16 + 2 * (q & 7) + (q >> 3)) * 1024; };

    // ---- prologue: first product + conversion of chunk 0 (a bare dependent chain, once per piece) ----
#pragma unroll
    for (int f = 0; f < R; ++f) ring[f] = lds_read16(chunk_lds(NT) + f * 1024);
    load_bias(c0);
    {
        const lds_cptr_t a0 = chunk_lds(NT), a1 = chunk_lds(NT + 1);
        const bool more = nloc > 1;   // then the interleaved stream follows: W1[1] k-step, W2[0] fragment, ...
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            mfma_bf16_vgpr(ring[f % R], xb[f], hacc);
            if (f + R < 16) {
                ring[f % R] = lds_read16(a0 + (f + R) * 1024);
            } else {
                const int s2 = f + R - 16;   // slot of the next stream
                const lds_cptr_t inter = (s2 & 1) ? a0 + w2_frag(s2 >> 1) : a1 + (s2 >> 1) * 1024;
                ring[f % R] = lds_read16(more ? inter : a0 + w2_frag(s2));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    convert(c0 + (nloc > 1 ? 1 : 0));

    for (int jt = 0; jt + 1 < nloc; ++jt) {
        // chunks jt+1 and jt+2 are in LDS (the head of jt+2 feeds the ring at the end of this iteration); the loaders
        // learn that chunk jt-1 is done with
        __builtin_amdgcn_s_barrier();
        const lds_cptr_t ca = chunk_lds(NT + jt + 1), cb = chunk_lds(NT + jt), cn = chunk_lds(NT + jt + 2);
        const bool more = jt + 2 < nloc;   // is the next iteration interleaved as well (or the bare tail)?
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            if (s & 1) {
                const int q = s >> 1;
                yacc[q & 7] = mfma_bf16(ring[s % R], hp[q >> 3], yacc[q & 7]);
            } else {
                mfma_bf16_vgpr(ring[s % R], xb[s >> 1], hacc);
            }
            if (s + R < 32) {
                ring[s % R] = (s & 1) ? lds_read16(cb + w2_frag((s + R) >> 1)) : lds_read16(ca + ((s + R) >> 1) * 1024);
            } else {
                const int s2 = s + R - 32;
                const lds_cptr_t inter = (s2 & 1) ? ca + w2_frag(s2 >> 1) : cn + (s2 >> 1) * 1024;
                ring[s % R] = lds_read16(more ? inter : ca + w2_frag(s2));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        convert(c0 + (more ? jt + 2 : jt + 1));
    }
    // ---- tail: second product of the last chunk ----
    {
        const lds_cptr_t cb = chunk_lds(NT + nloc - 1);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            yacc[q & 7] = mfma_bf16(ring[q % R], hp[q >> 3], yacc[q & 7]);
            if (q + R < 16) ring[q % R] = lds_read16(cb + w2_frag(q + R));
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    if (p.nsplit > 1) {
        // a piece of the hidden dimension: raw partial product, 16 bytes per (e-tile, quad) and lane
        if (valid) {
            float *prow = p.partial + ((int64_t)sp * p.T + tok) * kFE;
#pragma unroll
            for (int et = 0; et < 8; ++et)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4 *>(prow + 32 * et + 8 * g + 4 * h) =
                        make_float4(yacc[et][4 * g], yacc[et][4 * g + 1], yacc[et][4 * g + 2], yacc[et][4 * g + 3]);
        }
        return;
    }

    // ---- + residual, LayerNorm over the 256 channels held by lanes (t, 0) and (t, 1), store ----
    // The residual is X itself and X^T is still in the B-operand registers: lane (t, h) holds channels 16ks + 8h .. +7
    // of its token and needs 32et + 8g + 4h .. +3, i.e. half of each of its own pieces and half of its partner's
    // (t, 1-h) -- v_permlane32_swap exchanges exactly those halves between the two 32-lane rows.  No reload from
    // memory (32 row-strided 8-byte loads per lane and the registers to hold them).
    // The accumulators are only READ from here on (writing elements of a 16-register tuple makes the allocator copy
    // tuples, and with 128 + 64 registers live that spilled: the epilogue ran 13 us): pass 1 sums v = y + x and v^2,
    // pass 2 recomputes v and writes (v - mean) * rstd * gamma + beta.
    swap_halves();
    float sum = 0.f, sq = 0.f;
#pragma unroll
    for (int et = 0; et < 8; ++et) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float r[4];
            residual(et, g, r);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = yacc[et][4 * g + i] + r[i];
                sum += v;
                sq = fmaf(v, v, sq);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    sum += __shfl_xor(sum, 32);
    sq += __shfl_xor(sq, 32);
    const float mean = sum * (1.f / kFE);
    const float rstd = rsqrtf(fmaxf(sq * (1.f / kFE) - mean * mean, 0.f) + p.eps);
    const float shift = -mean * rstd;
    if (!NEXT) {
        if (valid) {
            bf16_t *orow = p.out + row;
#pragma unroll
            for (int et = 0; et < 8; ++et) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int e0 = 32 * et + 8 * g + 4 * h;
                    const float4 gv = *reinterpret_cast<const float4 *>(par + kFE + e0);
                    const float4 be = *reinterpret_cast<const float4 *>(par + 2 * kFE + e0);
                    float r[4];
                    residual(et, g, r);
                    // (v - mean) * rstd * gamma + beta = (v * rstd + shift) * gamma + beta
                    const float y0 = fmaf(fmaf(yacc[et][4 * g] + r[0], rstd, shift), gv.x, be.x);
                    const float y1 = fmaf(fmaf(yacc[et][4 * g + 1] + r[1], rstd, shift), gv.y, be.y);
                    const float y2 = fmaf(fmaf(yacc[et][4 * g + 2] + r[2], rstd, shift), gv.z, be.z);
                    const float y3 = fmaf(fmaf(yacc[et][4 * g + 3] + r[3], rstd, shift), gv.w, be.w);
                    *reinterpret_cast<uint2 *>(orow + e0) = make_uint2(pack_act2(y0, y1), pack_act2(y2, y3));
                }
                // one e-tile at a time: hoisting all 64 gamma / beta reads above the arithmetic costs 256 registers
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        return;
    }
    // ---- NEXT form: y (bf16) replaces the residual in the operand registers, piece by piece (cf. the TAIL's second
    // pass: opaque residual pieces between the passes, every piece pinned where it is computed) ----
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) asm volatile("" : "+v"(xb[ks].x), "+v"(xb[ks].y), "+v"(xb[ks].z), "+v"(xb[ks].w));
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const int et = ks >> 1, g0 = 2 * (ks & 1);
        uint32_t d[4];
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
            const int g = g0 + gg, e0 = 32 * et + 8 * g + 4 * h;
            const float4 gv = *reinterpret_cast<const float4 *>(par + kFE + e0);
            const float4 be = *reinterpret_cast<const float4 *>(par + 2 * kFE + e0);
            float r[4];
            residual(et, g, r);
            const float y0 = fmaf(fmaf(yacc[et][4 * g] + r[0], rstd, shift), gv.x, be.x);
            const float y1 = fmaf(fmaf(yacc[et][4 * g + 1] + r[1], rstd, shift), gv.y, be.y);
            const float y2 = fmaf(fmaf(yacc[et][4 * g + 2] + r[2], rstd, shift), gv.z, be.z);
            const float y3 = fmaf(fmaf(yacc[et][4 * g + 3] + r[3], rstd, shift), gv.w, be.w);
            d[2 * gg] = pack_act2(y0, y1);
            d[2 * gg + 1] = pack_act2(y2, y3);
        }
        xb[ks] = make_uint4(d[0], d[1], d[2], d[3]);
        asm volatile("" : "+v"(xb[ks].x), "+v"(xb[ks].y), "+v"(xb[ks].z), "+v"(xb[ks].w));
        __builtin_amdgcn_sched_barrier(0);
    }
    swap_halves();   // y^T: lane (t, h) holds channels 16ks + 8h .. +7 of its row -- 16-byte stores, and the class head's B operands
    {
        const FfnAdvance &a = p.adv;
        const int tk = valid ? tok : p.T - 1;
        const int b = tk / a.rows, i = tk - b * a.rows;
        const bool live = !a.count || i < a.count[b];
        const bool feeds = i < a.next_rows;
        if (live) {
            if (valid) {
                bf16_t *o = a.sorted_result + ((int64_t)b * a.sorted_rows + i) * kFE + 8 * h;
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) *reinterpret_cast<uint4 *>(o + 16 * ks) = xb[ks];
            }
        } else if (feeds) {
            // not part of this image's focus set: the next layer sees (and scores) the original token
            const bf16_t *src = a.tokens + ((int64_t)b * a.spatial_size + a.sorted_index[(int64_t)b * a.index_batch_stride + i]) * kFE + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) xb[ks] = *reinterpret_cast<const uint4 *>(src + 16 * ks);
        }
        if (valid && feeds) {
            bf16_t *o = a.next_query + ((int64_t)b * a.next_rows + i) * kFE + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) *reinterpret_cast<uint4 *>(o + 16 * ks) = xb[ks];
        }
        // ---- class head of the NEXT layer on the rows just handed on: logits^T [96 x 32] = Wc q^T + bc ----
        __builtin_amdgcn_s_barrier();   // both class-head chunks are in LDS (stashed during the last two iterations)
        const lds_cptr_t cc = chunk_lds(NT + nloc), cd = chunk_lds(NT + nloc + 1);
        f32x16_t cacc[3];
#pragma unroll
        for (int et = 0; et < 3; ++et)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *reinterpret_cast<const float4 *>(par + 6 * kFE + 32 * et + 8 * g + 4 * h);
                cacc[et][4 * g] = bv.x; cacc[et][4 * g + 1] = bv.y; cacc[et][4 * g + 2] = bv.z; cacc[et][4 * g + 3] = bv.w;
            }
        auto cls_frag = [&](int f) { return f < 32 ? cc + f * 1024 : cd + (f - 32) * 1024; };
#pragma unroll
        for (int f = 0; f < R; ++f) ring[f] = lds_read16(cls_frag(f));
#pragma unroll
        for (int f = 0; f < 48; ++f) {
            cacc[f % 3] = mfma_bf16(ring[f % R], xb[f / 3], cacc[f % 3]);
            if (f + R < 48) ring[f % R] = lds_read16(cls_frag(f + R));
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_barrier();   // (the loaders' last phase)
        float mx = -INFINITY;
#pragma unroll
        for (int et = 0; et < 3; ++et)
#pragma unroll
            for (int e = 0; e < 16; ++e) mx = fmaxf(mx, cacc[et][e]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (valid && feeds && h == 0) p.cmax[(int64_t)b * a.next_rows + i] = mx * p.fg[(int64_t)b * p.fg_bs + i];
    }
}

// Second pass of the hidden-split form: out = LayerNorm(x + b2 + sum_s partial[s]).  One wave per token, lane l owns
// channels 4l .. 4l+3 (1 KB coalesced row reads).
__global__ void __launch_bounds__(256) ffn_reduce_ln_kernel(const float *partial, int nsplit, int T, const bf16_t *x,
                                                            const float *b2, const float *gamma, const float *beta,
                                                            float eps, bf16_t *out)
{
    const int lane = threadIdx.x & 63;
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= T) return;
    const int64_t o = (int64_t)tok * kFE + 4 * lane;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (x) {   // (NULL: piece 0 of the partial products already carries bias + residual -- the TAIL form)
        const uint2 r = *reinterpret_cast<const uint2 *>(x + o);
        const float4 bv = *reinterpret_cast<const float4 *>(b2 + 4 * lane);
        v0 = bv.x + act_lo(r.x); v1 = bv.y + act_hi(r.x); v2 = bv.z + act_lo(r.y); v3 = bv.w + act_hi(r.y);
    }
    for (int s = 0; s < nsplit; ++s) {
        const float4 pv = *reinterpret_cast<const float4 *>(partial + (int64_t)s * T * kFE + o);
        v0 += pv.x; v1 += pv.y; v2 += pv.z; v3 += pv.w;
    }
    float sum = (v0 + v1) + (v2 + v3);
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) sum += __shfl_xor(sum, m);
    const float mean = sum * (1.f / kFE);
    const float d0 = v0 - mean, d1 = v1 - mean, d2 = v2 - mean, d3 = v3 - mean;
    float sq = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) sq += __shfl_xor(sq, m);
    const float rstd = rsqrtf(sq * (1.f / kFE) + eps);
    const float4 gv = *reinterpret_cast<const float4 *>(gamma + 4 * lane);
    const float4 be = *reinterpret_cast<const float4 *>(beta + 4 * lane);
    *reinterpret_cast<uint2 *>(out + o) = make_uint2(pack_act2(d0 * rstd * gv.x + be.x, d1 * rstd * gv.y + be.y),
                                                     pack_act2(d2 * rstd * gv.z + be.z, d3 * rstd * gv.w + be.w));
}

// The same second pass with the end-of-layer bookkeeping of sdetr_advance_rows in its store (the four hidden-split
// layers of the encoder: one launch less each): row i of image b goes to sorted_result[b,i] when live (i < count[b])
// and to next_query[b,i] when i < next_rows -- live rows the LayerNorm output, the others the never-updated original
// tokens[b, sorted_index[b,i]].

__global__ void __launch_bounds__(256) ffn_reduce_ln_advance_kernel(const float *partial, int nsplit, int T, const bf16_t *x,
                                                                    const float *b2, const float *gamma,
                                                                    const float *beta, float eps, FfnAdvance a)
{
    const int lane = threadIdx.x & 63;
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= T) return;
    const int b = tok / a.rows, i = tok - b * a.rows;
    const bool live = !a.count || i < a.count[b];
    const bool feeds_next = a.next_query && i < a.next_rows;
    if (!live) {   // the row is not part of this image's focus set: the next layer sees the original token
        if (feeds_next)
            *reinterpret_cast<uint2 *>(a.next_query + ((int64_t)b * a.next_rows + i) * kFE + 4 * lane) =
                *reinterpret_cast<const uint2 *>(a.tokens + ((int64_t)b * a.spatial_size +
                                                             a.sorted_index[(int64_t)b * a.index_batch_stride + i]) * kFE + 4 * lane);
        return;
    }
    const int64_t o = (int64_t)tok * kFE + 4 * lane;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (x) {   // (NULL: piece 0 of the partial products already carries bias + residual -- the TAIL form)
        const uint2 r = *reinterpret_cast<const uint2 *>(x + o);
        const float4 bv = *reinterpret_cast<const float4 *>(b2 + 4 * lane);
        v0 = bv.x + act_lo(r.x); v1 = bv.y + act_hi(r.x); v2 = bv.z + act_lo(r.y); v3 = bv.w + act_hi(r.y);
    }
    for (int s = 0; s < nsplit; ++s) {
        const float4 pv = *reinterpret_cast<const float4 *>(partial + (int64_t)s * T * kFE + o);
        v0 += pv.x; v1 += pv.y; v2 += pv.z; v3 += pv.w;
    }
    float sum = (v0 + v1) + (v2 + v3);
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) sum += __shfl_xor(sum, m);
    const float mean = sum * (1.f / kFE);
    const float d0 = v0 - mean, d1 = v1 - mean, d2 = v2 - mean, d3 = v3 - mean;
    float sq = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) sq += __shfl_xor(sq, m);
    const float rstd = rsqrtf(sq * (1.f / kFE) + eps);
    const float4 gv = *reinterpret_cast<const float4 *>(gamma + 4 * lane);
    const float4 be = *reinterpret_cast<const float4 *>(beta + 4 * lane);
    const uint2 y = make_uint2(pack_act2(d0 * rstd * gv.x + be.x, d1 * rstd * gv.y + be.y),
                               pack_act2(d2 * rstd * gv.z + be.z, d3 * rstd * gv.w + be.w));
    *reinterpret_cast<uint2 *>(a.sorted_result + ((int64_t)b * a.sorted_rows + i) * kFE + 4 * lane) = y;
    if (feeds_next) *reinterpret_cast<uint2 *>(a.next_query + ((int64_t)b * a.next_rows + i) * kFE + 4 * lane) = y;
}

// ... and with the NEXT layer's class score of the rows handed on (the NEXT form's epilogue, for a split hidden dimension:
// the four hidden-split layers of the encoder used to launch the class head on its own, 6.2-6.6 us each on a chip that
// waits for it).  32 rows per 512-thread block: every wave finishes four rows exactly as the kernel above (same sums in
// the same order: identical rows), the handed-on rows also go to an LDS tile [32][256] (rows 528 bytes apart: the B-operand
// reads of a 16-lane group hit 64 distinct banks); then waves 0..2 take one 32-class tile each -- logits^T [32 x 32] =
// Wc[tile] q^T + bc, sixteen MFMAs on one accumulator in k order (the chain of the NEXT form) from the class head's
// packed fragments (class_head_pack_kernel; requested first, they arrive under the row sums) -- and the maximum over the
// three tiles times the foreground score is the row's score.
constexpr int kRcRows = kClsTileRows, kRcThreads = 512, kRcRowBytes = kClsRowBytes;

__global__ void __launch_bounds__(kRcThreads) ffn_reduce_ln_advance_cls_kernel(
    const float *partial, int nsplit, int T, const bf16_t *x, const float *b2, const float *gamma, const float *beta,
    float eps, FfnAdvance a, const char *cls_pw, const float *cls_bias, const float *fg, int64_t fg_bs, float *cmax)
{
    __shared__ __attribute__((aligned(16))) unsigned char ytile[kRcRows * kRcRowBytes];
    __shared__ float red[3][kRcRows];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tok0 = blockIdx.x * kRcRows;
    uint4 af[16];
    if (wave < 3) class_frag_load<0, 8>(af, cls_pw, wave, lane);
    constexpr int RW = kRcRows / (kRcThreads / 64);   // rows per wave: 4
    float4 v[RW];
    int bb[RW], ii[RW];
    bool live[RW], feeds[RW];
#pragma unroll
    for (int k = 0; k < RW; ++k) {
        const int tok = tok0 + wave * RW + k;
        const bool valid = tok < T;
        const int tk = valid ? tok : T - 1;
        bb[k] = tk / a.rows;
        ii[k] = tk - bb[k] * a.rows;
        live[k] = valid && (!a.count || ii[k] < a.count[bb[k]]);
        feeds[k] = valid && a.next_query && ii[k] < a.next_rows;
        v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // the never-updated original tokens of the rows that are handed on without being part of the focus set
    uint2 orig[RW];
#pragma unroll
    for (int k = 0; k < RW; ++k) {
        orig[k] = make_uint2(0u, 0u);
        if (!live[k] && feeds[k])
            orig[k] = *reinterpret_cast<const uint2 *>(
                a.tokens + ((int64_t)bb[k] * a.spatial_size + a.sorted_index[(int64_t)bb[k] * a.index_batch_stride + ii[k]]) * kFE +
                4 * lane);
    }
    if (x) {   // (NULL: piece 0 of the partial products already carries bias + residual -- the TAIL form)
        const float4 bv = *reinterpret_cast<const float4 *>(b2 + 4 * lane);
#pragma unroll
        for (int k = 0; k < RW; ++k)
            if (live[k]) {
                const uint2 r = *reinterpret_cast<const uint2 *>(x + (int64_t)(tok0 + wave * RW + k) * kFE + 4 * lane);
                v[k] = make_float4(bv.x + act_lo(r.x), bv.y + act_hi(r.x), bv.z + act_lo(r.y), bv.w + act_hi(r.y));
            }
    }
    // two pieces of every row in flight (eight 16-byte loads per lane and trip)
    for (int s = 0; s < nsplit; s += 2) {
        float4 p0[RW], p1[RW];
        const bool two = s + 1 < nsplit;
#pragma unroll
        for (int k = 0; k < RW; ++k) {
            const int64_t o = (int64_t)(tok0 + wave * RW + k) * kFE + 4 * lane;
            p0[k] = p1[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live[k]) {
                p0[k] = *reinterpret_cast<const float4 *>(partial + (int64_t)s * T * kFE + o);
                if (two) p1[k] = *reinterpret_cast<const float4 *>(partial + (int64_t)(s + 1) * T * kFE + o);
            }
        }
#pragma unroll
        for (int k = 0; k < RW; ++k) {
            v[k].x += p0[k].x; v[k].y += p0[k].y; v[k].z += p0[k].z; v[k].w += p0[k].w;
            if (two) { v[k].x += p1[k].x; v[k].y += p1[k].y; v[k].z += p1[k].z; v[k].w += p1[k].w; }
        }
    }
    // (the second half of the fragments takes the registers the pieces have left: 128 registers, two blocks per CU)
    if (wave < 3) class_frag_load<8, 16>(af, cls_pw, wave, lane);
    const float4 gv = *reinterpret_cast<const float4 *>(gamma + 4 * lane);
    const float4 be = *reinterpret_cast<const float4 *>(beta + 4 * lane);
#pragma unroll
    for (int k = 0; k < RW; ++k) {
        uint2 y = orig[k];
        if (live[k]) {
            float sum = (v[k].x + v[k].y) + (v[k].z + v[k].w);
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) sum += __shfl_xor(sum, m);
            const float mean = sum * (1.f / kFE);
            const float d0 = v[k].x - mean, d1 = v[k].y - mean, d2 = v[k].z - mean, d3 = v[k].w - mean;
            float sq = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) sq += __shfl_xor(sq, m);
            const float rstd = rsqrtf(sq * (1.f / kFE) + eps);
            y = make_uint2(pack_act2(d0 * rstd * gv.x + be.x, d1 * rstd * gv.y + be.y),
                           pack_act2(d2 * rstd * gv.z + be.z, d3 * rstd * gv.w + be.w));
            *reinterpret_cast<uint2 *>(a.sorted_result + ((int64_t)bb[k] * a.sorted_rows + ii[k]) * kFE + 4 * lane) = y;
        }
        if (feeds[k]) *reinterpret_cast<uint2 *>(a.next_query + ((int64_t)bb[k] * a.next_rows + ii[k]) * kFE + 4 * lane) = y;
        // (a column of the class product depends on its own row only: what rows that are not handed on leave here is unused)
        *reinterpret_cast<uint2 *>(ytile + (wave * RW + k) * kRcRowBytes + 8 * lane) = y;
    }
    __syncthreads();
    if (wave < 3) {
        const float mx = class_tile_max(af, ytile, cls_bias, wave, lane);
        if (lane < 32) red[wave][lane] = mx;
    }
    __syncthreads();
    if (tid < kRcRows) {
        const int tok = tok0 + tid;
        if (tok < T) {
            const int b = tok / a.rows, i = tok - b * a.rows;
            if (i < a.next_rows) cmax[(int64_t)b * a.next_rows + i] = fmaxf(fmaxf(red[0][tid], red[1][tid]), red[2][tid]) * fg[(int64_t)b * fg_bs + i];
        }
    }
}

// hidden index inside a 16-wide k-block that MFMA operand slot (h, s) stands for: the accumulator rows a lane of
// half h holds in registers 0..7 (kb = 0) / 8..15 (kb = 1)
__device__ __forceinline__ int acc_hidden(int h, int s) { return s < 4 ? 4 * h + s : 8 + 4 * h + (s - 4); }

__global__ void ffn_pack_kernel(const bf16_t *w1, const bf16_t *w2, int F, bf16_t *out)
{
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over nchunk * 16384 elements
    if (o >= (int64_t)(F / kFChunk) * 16384) return;
    const int jt = (int)(o / 16384), idx = (int)(o % 16384);
    const int s = idx & 7, l = (idx >> 3) & 63, frag = (idx & 8191) >> 9;   // 512 elements per fragment
    const int h = l >> 5, r = l & 31;
    if (idx < 8192) {
        // W1 fragment `frag` = k-step: A[j = r][k = 16 frag + 8h + s]
        out[o] = w1[(int64_t)(kFChunk * jt + r) * kFE + 16 * frag + 8 * h + s];
    } else {
        const int et = frag >> 1, kb = frag & 1;
        out[o] = w2[(int64_t)(32 * et + r) * F + kFChunk * jt + 16 * kb + acc_hidden(h, s)];
    }
}

// Wo [256 out, 256 in] -> kTailChunks chunks of 32 A-fragments: chunk a, fragment f = 8 ksl + et holds
// A[m = 32 et + (lane & 31)][k = 16 (4 a + ksl) + 8 (lane >> 5) .. +7] as 16 bytes per lane.
__global__ void attn_tail_pack_kernel(const bf16_t *wo, bf16_t *out)
{
    const int o = blockIdx.x * blockDim.x + threadIdx.x;   // over 4 * 16384 elements
    if (o >= kTailChunks * 16384) return;
    const int a = o / 16384, idx = o % 16384;
    const int s = idx & 7, l = (idx >> 3) & 63, f = idx >> 9;
    const int ksl = f >> 3, et = f & 7;
    out[o] = wo[(int64_t)(32 * et + (l & 31)) * kFE + 16 * (4 * a + ksl) + 8 * (l >> 5) + s];
}

// Wc [num_classes <= 96, 256] -> kClsChunks chunks: fragment f = 3 ks + et (48 of the 64 slots used) holds
// A[m = 32 et + (lane & 31)][k = 16 ks + 8 (lane >> 5) .. +7], zero rows for the padded classes.
__global__ void class_head_pack_kernel(const bf16_t *wc, int num_classes, bf16_t *out)
{
    const int o = blockIdx.x * blockDim.x + threadIdx.x;   // over 2 * 16384 elements
    if (o >= kClsChunks * 16384) return;
    const int s = o & 7, l = (o >> 3) & 63, f = o >> 9;
    const int ks = f / 3, et = f - 3 * ks, m = 32 * et + (l & 31);
    out[o] = (f < 48 && m < num_classes) ? wc[(int64_t)m * kFE + 16 * ks + 8 * (l >> 5) + s] : (bf16_t)0;
}

}  // namespace sdetr

using namespace sdetr;

extern "C" int64_t sdetr_class_head_packed_bytes(void) { return (int64_t)kClsChunks * kFChunkBytes; }

extern "C" int sdetr_class_head_pack_bf16(sdetr_stream_t stream, const void *weight, int num_classes, int embed_dim, void *packed)
{
    if (embed_dim != kFE) return fail("class_head_pack: built for embed_dim %d (got %d)", kFE, embed_dim);
    if (num_classes <= 0 || num_classes > 96) return fail("class_head_pack: 1..96 classes (got %d)", num_classes);
    if (!weight || !packed) return fail("class_head_pack: null pointer");
    hipLaunchKernelGGL(class_head_pack_kernel, dim3((kClsChunks * 16384 + 255) / 256), dim3(256), 0,
                       static_cast<hipStream_t>(stream), (const bf16_t *)weight, num_classes, (bf16_t *)packed);
    return check_launch("class_head_pack");
}

extern "C" int64_t sdetr_attn_tail_packed_bytes(void) { return (int64_t)kTailChunks * kFChunkBytes; }

extern "C" int sdetr_attn_tail_pack_bf16(sdetr_stream_t stream, const void *weight_o, int embed_dim, void *packed)
{
    if (embed_dim != kFE) return fail("attn_tail_pack: built for embed_dim %d (got %d)", kFE, embed_dim);
    if (!weight_o || !packed) return fail("attn_tail_pack: null pointer");
    hipLaunchKernelGGL(attn_tail_pack_kernel, dim3((kTailChunks * 16384 + 255) / 256), dim3(256), 0,
                       static_cast<hipStream_t>(stream), (const bf16_t *)weight_o, (bf16_t *)packed);
    return check_launch("attn_tail_pack");
}

extern "C" int64_t sdetr_ffn_packed_bytes(int hidden) { return hidden > 0 ? (int64_t)(hidden / kFChunk) * kFChunkBytes : 0; }

extern "C" int sdetr_ffn_pack_bf16(sdetr_stream_t stream, const void *weight1, const void *weight2, int embed_dim,
                                   int hidden, void *packed)
{
    if (embed_dim != kFE) return fail("ffn_pack: built for embed_dim %d (got %d)", kFE, embed_dim);
    if (hidden <= 0 || hidden % kFChunk) return fail("ffn_pack: hidden (%d) must be a positive multiple of %d", hidden, kFChunk);
    if (!weight1 || !weight2 || !packed) return fail("ffn_pack: null pointer");
    const int64_t total = (int64_t)(hidden / kFChunk) * 16384;
    hipLaunchKernelGGL(ffn_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), (const bf16_t *)weight1, (const bf16_t *)weight2, hidden,
                       (bf16_t *)packed);
    return check_launch("ffn_pack");
}

static int device_cus()
{
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return cus;
}

// A block keeps a whole CU (128 KB of weight buffers, one compute wave per SIMD): measured on MI355X (hipGraph replay,
// benchmarks/ffn_hidden_sweep.py) a piece costs ~0.7 us per 32-unit chunk plus ~13 us of launch + prologue + epilogue;
// the second pass of a split launch reads nsplit + 1 KB per token.  With one piece per token block the kernel runs ~60 us
// whether 15 or 178 CUs have work: splitting the hidden dimension fills the chip for the encoder's smaller layers.
extern "C" int sdetr_ffn_auto_splits(int tokens, int hidden)
{
    if (tokens <= 0 || hidden <= 0) return 1;
    const int nchunk = hidden / kFChunk, cus = device_cus();
    const int tblocks = (tokens + kFTokBlock - 1) / kFTokBlock;
    int best = 1;
    double best_us = 1e30;
    for (int s = 1; s <= nchunk && s <= 16; ++s) {
        const int rounds = (tblocks * s + cus - 1) / cus;
        const int chunks = (nchunk + s - 1) / s;
        double us = rounds * (chunks * 0.7 + 13.0);
        if (s > 1) us += 4.0 + (double)tokens * (s + 1) * 1024.0 / 3.0e6;   // second launch + its traffic at ~3 TB/s
        if (us < best_us - 1e-9) { best_us = us; best = s; }
    }
    return best;
}

extern "C" int64_t sdetr_ffn_workspace_bytes(int tokens, int hidden_splits)
{
    return hidden_splits > 1 && tokens > 0 ? (int64_t)hidden_splits * tokens * kFE * 4 : 0;
}

extern "C" int sdetr_ffn_fused_bf16(sdetr_stream_t stream, const void *x, const void *packed_weights, const float *bias1,
                                    const float *bias2, const float *norm_weight, const float *norm_bias, float norm_eps,
                                    int tokens, int embed_dim, int hidden, void *out, int hidden_splits, void *workspace,
                                    int64_t workspace_bytes)
{
    if (embed_dim != kFE) return fail("ffn_fused: built for embed_dim %d (got %d)", kFE, embed_dim);
    if (hidden <= 0 || hidden % kFChunk) return fail("ffn_fused: hidden (%d) must be a positive multiple of %d", hidden, kFChunk);
    if (tokens < 0) return fail("ffn_fused: negative token count");
    if (hidden_splits < 1 || hidden_splits > hidden / kFChunk) return fail("ffn_fused: hidden_splits must be in 1 .. hidden/32");
    if (tokens == 0) return 0;
    if (!x || !packed_weights || !bias1 || !bias2 || !norm_weight || !norm_bias || !out) return fail("ffn_fused: null pointer");
    if (hidden_splits > 1 && (!workspace || workspace_bytes < sdetr_ffn_workspace_bytes(tokens, hidden_splits)))
        return fail("ffn_fused: %d hidden splits need a workspace of %lld bytes", hidden_splits,
                    (long long)sdetr_ffn_workspace_bytes(tokens, hidden_splits));
    const size_t lds = 4 * (size_t)kFChunkBytes + (size_t)hidden * 4 + 3 * kFE * 4;
    if (lds > 160 * 1024) return fail("ffn_fused: hidden %d needs %zu bytes of LDS", hidden, lds);
    static DeviceOnce lds_once1;
    allow_dynamic_lds(ffn_fused_kernel<false>, lds_once1, 160 * 1024);
    FfnArgs a;
    a.x = (const bf16_t *)x; a.pw = (const char *)packed_weights; a.b1 = bias1; a.b2 = bias2; a.gamma = norm_weight;
    a.beta = norm_bias; a.eps = norm_eps; a.out = (bf16_t *)out; a.T = tokens; a.nchunk = hidden / kFChunk;
    a.nsplit = hidden_splits; a.partial = hidden_splits > 1 ? (float *)workspace : nullptr;
    const int64_t tblocks = (tokens + kFTokBlock - 1) / kFTokBlock;
    hipLaunchKernelGGL(ffn_fused_kernel<false>, dim3((unsigned)(tblocks * hidden_splits)), dim3(kFThreads), lds,
                       static_cast<hipStream_t>(stream), a);
    if (hidden_splits > 1)
        hipLaunchKernelGGL(ffn_reduce_ln_kernel, dim3((unsigned)((tokens + 3) / 4)), dim3(256), 0,
                           static_cast<hipStream_t>(stream), (const float *)workspace, hidden_splits, tokens,
                           (const bf16_t *)x, bias2, norm_weight, norm_bias, norm_eps, (bf16_t *)out);
    return check_launch("ffn_fused");
}

extern "C" int sdetr_advance_rows(sdetr_stream_t stream, const void *layer_out, void *sorted_result, void *next_query,
                                  const void *tokens, const int64_t *sorted_index, int64_t index_batch_stride,
                                  const int64_t *count, int batch_size, int rows, int sorted_rows, int next_rows,
                                  int spatial_size, int row_bytes);

struct TailArgs {   // the attention tail in front of the feed-forward (NULL sampled = plain feed-forward)
    const void *sampled, *residual;
    const float *bias_o, *norm1_weight, *norm1_bias;
    float norm1_eps;
    // the next layer's class score out of the epilogue (NULL next_score = none; only with one hidden piece)
    const float *cls_bias, *fg;
    int64_t fg_bs;
    float *next_score;
};

static int ffn_advance_impl(sdetr_stream_t stream, const void *x, const TailArgs &tail, const void *packed_weights,
                            const float *bias1, const float *bias2, const float *norm_weight, const float *norm_bias,
                            float norm_eps, int batch_size, int rows, int embed_dim, int hidden, int hidden_splits,
                            void *workspace, int64_t workspace_bytes, void *sorted_result, void *next_query,
                            const void *tokens, const int64_t *sorted_index, int64_t index_batch_stride,
                            const int64_t *count, int sorted_rows, int next_rows, int spatial_size)
{
    const bool with_tail = tail.sampled != nullptr;
    if (batch_size < 0 || rows < 0) return fail("ffn_fused_advance: negative size");
    if (next_rows < 0 || next_rows > rows || rows > sorted_rows)
        return fail("ffn_fused_advance: need next_rows <= rows <= sorted_rows");
    if ((int64_t)batch_size * rows > 0x7fffffffLL) return fail("ffn_fused_advance: too many rows");
    const int tokens_total = batch_size * rows;
    if (tokens_total == 0) return 0;
    if (!sorted_result || !tokens || !sorted_index || (next_rows > 0 && !next_query))
        return fail("ffn_fused_advance: null pointer");
    if (index_batch_stride < rows) return fail("ffn_fused_advance: index batch stride too small");
    if (with_tail && (!tail.residual || !tail.bias_o || !tail.norm1_weight || !tail.norm1_bias))
        return fail("attn_tail_ffn_advance: null pointer");
    // the layer output itself: the head of the workspace (after the split partials), never seen by the caller
    const int64_t partial_bytes = sdetr_ffn_workspace_bytes(tokens_total, hidden_splits);
    const int64_t out_bytes = hidden_splits > 1 ? 0 : (int64_t)tokens_total * kFE * 2;
    if (workspace_bytes < partial_bytes + out_bytes || (partial_bytes + out_bytes > 0 && !workspace))
        return fail("ffn_fused_advance: workspace of %lld bytes needed", (long long)(partial_bytes + out_bytes));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hidden_splits == 1 && !with_tail) {
        void *out = static_cast<char *>(workspace) + partial_bytes;
        if (int rc = sdetr_ffn_fused_bf16(stream, x, packed_weights, bias1, bias2, norm_weight, norm_bias, norm_eps,
                                          tokens_total, embed_dim, hidden, out, 1, nullptr, 0))
            return rc;
        return sdetr_advance_rows(stream, out, sorted_result, next_query, tokens, sorted_index, index_batch_stride, count,
                                  batch_size, rows, sorted_rows, next_rows, spatial_size, kFE * 2);
    }
    if (embed_dim != kFE) return fail("ffn_fused: built for embed_dim %d (got %d)", kFE, embed_dim);
    if (hidden <= 0 || hidden % kFChunk) return fail("ffn_fused: hidden (%d) must be a positive multiple of %d", hidden, kFChunk);
    if (hidden_splits < 1 || hidden_splits > hidden / kFChunk) return fail("ffn_fused: hidden_splits must be in 1 .. hidden/32");
    if ((!with_tail && !x) || !packed_weights || !bias1 || !bias2 || !norm_weight || !norm_bias) return fail("ffn_fused: null pointer");
    const size_t lds = 4 * (size_t)kFChunkBytes + (size_t)hidden * 4 + (with_tail ? 6 : 3) * kFE * 4;
    if (lds > 160 * 1024) return fail("ffn_fused: hidden %d needs %zu bytes of LDS", hidden, lds);
    FfnArgs a{};
    a.x = (const bf16_t *)x; a.pw = (const char *)packed_weights; a.b1 = bias1; a.b2 = bias2; a.gamma = norm_weight;
    a.beta = norm_bias; a.eps = norm_eps; a.out = nullptr; a.T = tokens_total; a.nchunk = hidden / kFChunk;
    a.nsplit = hidden_splits; a.partial = hidden_splits > 1 ? (float *)workspace : nullptr;
    a.s = (const bf16_t *)tail.sampled; a.res = (const bf16_t *)tail.residual; a.bo = tail.bias_o;
    a.g1 = tail.norm1_weight; a.be1 = tail.norm1_bias; a.eps1 = tail.norm1_eps;
    const int64_t tblocks = (tokens_total + kFTokBlock - 1) / kFTokBlock;
    if (hidden_splits == 1) a.out = reinterpret_cast<bf16_t *>(static_cast<char *>(workspace) + partial_bytes);
    FfnAdvance adv;
    adv.sorted_result = (bf16_t *)sorted_result; adv.next_query = next_rows > 0 ? (bf16_t *)next_query : nullptr;
    adv.tokens = (const bf16_t *)tokens; adv.sorted_index = sorted_index; adv.index_batch_stride = index_batch_stride;
    adv.count = count; adv.rows = rows; adv.sorted_rows = sorted_rows; adv.next_rows = next_rows;
    adv.spatial_size = spatial_size;
    if (with_tail && tail.next_score && hidden_splits == 1 && next_rows > 0) {
        // layer end + row bookkeeping + the next layer's class score: ONE launch
        if (!tail.cls_bias || !tail.fg || tail.fg_bs < next_rows) return fail("attn_tail_ffn_advance: bad class-score operands");
        a.adv = adv; a.out = nullptr; a.cls_bias = tail.cls_bias; a.fg = tail.fg; a.fg_bs = tail.fg_bs; a.cmax = tail.next_score;
        static DeviceOnce lds_once4;
        allow_dynamic_lds(ffn_fused_kernel<true, true>, lds_once4, 160 * 1024);
        hipLaunchKernelGGL((ffn_fused_kernel<true, true>), dim3((unsigned)tblocks), dim3(kFThreads), lds + 96 * 4, s, a);
        return check_launch("attn_tail_ffn_next");
    }
    if (with_tail) {
        static DeviceOnce lds_once3;
        allow_dynamic_lds(ffn_fused_kernel<true>, lds_once3, 160 * 1024);
        hipLaunchKernelGGL(ffn_fused_kernel<true>, dim3((unsigned)(tblocks * hidden_splits)), dim3(kFThreads), lds, s, a);
    } else {
        static DeviceOnce lds_once2;
        allow_dynamic_lds(ffn_fused_kernel<false>, lds_once2, 160 * 1024);
        hipLaunchKernelGGL(ffn_fused_kernel<false>, dim3((unsigned)(tblocks * hidden_splits)), dim3(kFThreads), lds, s, a);
    }
    if (hidden_splits == 1) {
        if (int rc = check_launch("attn_tail_ffn")) return rc;
        return sdetr_advance_rows(stream, a.out, sorted_result, next_query, tokens, sorted_index, index_batch_stride, count,
                                  batch_size, rows, sorted_rows, next_rows, spatial_size, kFE * 2);
    }
    // (TAIL: piece 0 of the partial products carries b2 + x, the pass only sums)
    if (with_tail && tail.next_score && next_rows > 0) {
        // ... and the next layer's class score of the rows handed on: the class head's fragments close the packed buffer
        if (!tail.cls_bias || !tail.fg || tail.fg_bs < next_rows) return fail("attn_tail_ffn_advance: bad class-score operands");
        const char *cls_pw = static_cast<const char *>(packed_weights) + (int64_t)(kTailChunks + hidden / kFChunk) * kFChunkBytes;
        hipLaunchKernelGGL(ffn_reduce_ln_advance_cls_kernel, dim3((unsigned)((tokens_total + kRcRows - 1) / kRcRows)),
                           dim3(kRcThreads), 0, s, (const float *)workspace, hidden_splits, tokens_total,
                           (const bf16_t *)nullptr, bias2, norm_weight, norm_bias, norm_eps, adv, cls_pw, tail.cls_bias, tail.fg,
                           tail.fg_bs, tail.next_score);
        return check_launch("attn_tail_ffn_advance_score");
    }
    hipLaunchKernelGGL(ffn_reduce_ln_advance_kernel, dim3((unsigned)((tokens_total + 3) / 4)), dim3(256), 0, s,
                       (const float *)workspace, hidden_splits, tokens_total, with_tail ? nullptr : (const bf16_t *)x, bias2,
                       norm_weight, norm_bias, norm_eps, adv);
    return check_launch("ffn_fused_advance");
}

extern "C" int sdetr_ffn_fused_advance_bf16(sdetr_stream_t stream, const void *x, const void *packed_weights,
                                            const float *bias1, const float *bias2, const float *norm_weight,
                                            const float *norm_bias, float norm_eps, int batch_size, int rows,
                                            int embed_dim, int hidden, int hidden_splits, void *workspace,
                                            int64_t workspace_bytes, void *sorted_result, void *next_query,
                                            const void *tokens, const int64_t *sorted_index,
                                            int64_t index_batch_stride, const int64_t *count, int sorted_rows,
                                            int next_rows, int spatial_size)
{
    return ffn_advance_impl(stream, x, TailArgs{}, packed_weights, bias1, bias2, norm_weight, norm_bias, norm_eps, batch_size,
                            rows, embed_dim, hidden, hidden_splits, workspace, workspace_bytes, sorted_result, next_query,
                            tokens, sorted_index, index_batch_stride, count, sorted_rows, next_rows, spatial_size);
}

extern "C" int sdetr_attn_tail_ffn_advance_bf16(
    sdetr_stream_t stream, const void *sampled, const void *residual, const void *packed_tail_ffn, const float *bias_o,
    const float *norm1_weight, const float *norm1_bias, float norm1_eps, const float *bias1, const float *bias2,
    const float *norm_weight, const float *norm_bias, float norm_eps, int batch_size, int rows, int embed_dim, int hidden,
    int hidden_splits, void *workspace, int64_t workspace_bytes, void *sorted_result, void *next_query, const void *tokens,
    const int64_t *sorted_index, int64_t index_batch_stride, const int64_t *count, int sorted_rows, int next_rows,
    int spatial_size, const float *class_bias_padded, const float *foreground, int64_t foreground_batch_stride,
    float *next_class_score)
{
    if (!sampled) return fail("attn_tail_ffn_advance: null pointer");
    TailArgs t{sampled, residual, bias_o, norm1_weight, norm1_bias, norm1_eps, class_bias_padded, foreground,
               foreground_batch_stride, next_class_score};
    return ffn_advance_impl(stream, nullptr, t, packed_tail_ffn, bias1, bias2, norm_weight, norm_bias, norm_eps, batch_size, rows,
                            embed_dim, hidden, hidden_splits, workspace, workspace_bytes, sorted_result, next_query, tokens,
                            sorted_index, index_batch_stride, count, sorted_rows, next_rows, spatial_size);
}
