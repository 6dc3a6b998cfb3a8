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
//     lane-ordered 1 KB MFMA A-fragments, which the four waves of a block copy global -> LDS with
//     global_load_lds_dwordx4 (no registers, lane-linear destination = fragment order), triple-buffered with a
//     counted s_waitcnt so one chunk is always in flight across the block barrier.  Each fragment is read from LDS
//     once per wave (128 B/clk/CU at full MFMA rate, half the LDS peak); L2 sees 2 MB per 128 tokens.
//   * epilogue in registers: + b2 + residual, LayerNorm over the 256 channels a lane pair holds, bf16 store.
//
// Bound: bf16 MFMA peak -- 4*F*256 flops per token; one wave issues 64 chunks x 32 MFMAs.  Measured on MI355X (in-kernel
// s_memtime, F = 2048) with the copies still issued by the compute waves: ~1800 cycles per chunk against 1056 for the
// bare MFMAs -- 550 + 590 for the two products, ~180 for MFMA drain + ReLU/bf16, ~490 for the barrier and the wave's
// 8 LDS-DMA issues (~45 cycles each, they stall the issuing wave): 64 us for any token count up to 32 768 (one compute
// wave per SIMD, one round).  With the loader waves below: 54 us.  Next: interleaving chunk jt's second product with
// chunk jt+1's first (hides the drain + ReLU phase and most of the barrier).
#include "common.h"

namespace sdetr {

constexpr int kFE = 256;               // embed dim
constexpr int kFChunk = 32;            // hidden units per chunk
constexpr int kFChunkBytes = 32768;    // 16 KB of W1 rows + 16 KB of W2 columns, as 1 KB A-fragments
constexpr int kFTokWave = 32, kFTokBlock = 128;

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

struct FfnArgs {
    const bf16_t *x;        // [T, 256]
    const char *pw;         // packed weights, nchunk * 32 KB
    const float *b1;        // [F]
    const float *b2, *gamma, *beta;   // [256]
    float eps;
    bf16_t *out;            // [T, 256]
    int T, nchunk;
};

__device__ __forceinline__ f32x16_t mfma_bf16(uint4 a, uint4 b, f32x16_t c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c,
                                                   0, 0, 0);
}

typedef short i16x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const char *lds_cptr_t;

// The first product accumulates in ARCHITECTURAL VGPRs (the VALU that follows reads it; the builtin would put it in
// AGPRs, which costs a v_accvgpr_read per value plus, here, a 32-register shuffle of the output tile whose AGPRs the
// allocator reuses).  Inline asm is the only way to choose the register class.
__device__ __forceinline__ void mfma_bf16_vgpr(uint4 a, uint4 b, f32x16_t &c)
{
    const u32x4_t av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv));
}

// one 32 KB chunk global -> LDS: a wave copies its 8 KB as 8 LDS-DMA instructions of 1 KB (destination = M0 base +
// instruction offset + lane * 16; the instruction offset advances the global source as well, so two M0 values cover
// the 8 pieces).  Issued as inline asm on purpose: hipcc does not know which LDS bytes an LDS-DMA instruction it can
// see will write and drains ALL of them (s_waitcnt vmcnt(0)) in front of the next ds_read -- which would serialise
// the copy of chunk jt+2 with the MFMAs of chunk jt.  The counted waits in the main loop are the synchronisation.
__device__ __forceinline__ void issue_chunk(const char *chunk, uint32_t voff, uint32_t dst_lds)
{
    const uint32_t d0 = __builtin_amdgcn_readfirstlane(dst_lds), d1 = d0 + 4096;
    const uint32_t voff1 = voff + 4096;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %4\n\t"
                 "global_load_lds_dwordx4 %0, %4 offset:1024\n\t"
                 "global_load_lds_dwordx4 %0, %4 offset:2048\n\t"
                 "global_load_lds_dwordx4 %0, %4 offset:3072\n\t"
                 "s_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %4\n\t"
                 "global_load_lds_dwordx4 %1, %4 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %4 offset:2048\n\t"
                 "global_load_lds_dwordx4 %1, %4 offset:3072"
                 :
                 : "v"(voff), "v"(voff1), "s"(d0), "s"(d1), "s"(chunk)
                 : "memory", "m0");
}

__device__ __forceinline__ uint32_t lds_address(const void *p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
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

// Block = 4 compute waves (32 tokens each) + 4 LOADER waves, one of each per SIMD.  An LDS-DMA instruction stalls the
// wave that issues it for ~45 cycles; issued by the compute waves the 8 copies per chunk cost a fifth of the loop
// (measured: 490 cycles of 1800 for barrier + copies).  The loaders do nothing else: wait for their copies, meet the
// compute waves at the chunk barrier, issue the next chunk.  (All waves get the same register allocation, so the
// kernel has to fit 256 registers for two waves per SIMD.)
constexpr int kFThreads = 512;

__global__ void __launch_bounds__(kFThreads, 1) ffn_fused_kernel(FfnArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *wbuf = lds;                                                  // 3 chunk buffers
    float *b1s = reinterpret_cast<float *>(lds + 3 * kFChunkBytes);    // [F]
    float *par = b1s + p.nchunk * kFChunk;                             // b2 | gamma | beta
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t wbuf_lds = lds_address(wbuf);

    if (wave >= 4) {
        // ---- loader wave: a quarter (8 KB = 8 LDS-DMA instructions) of every 32 KB chunk ----
        const int lw = wave - 4;
        const uint32_t voff = (uint32_t)(lw * 8192 + lane * 16);
        const uint32_t wave_lds = wbuf_lds + lw * 8192;
        issue_chunk(p.pw, voff, wave_lds);
        if (p.nchunk > 1) issue_chunk(p.pw + kFChunkBytes, voff, wave_lds + kFChunkBytes);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // chunks 0 and 1 are in LDS
        int buf = 0;
        for (int jt = 0; jt + 1 < p.nchunk; ++jt) {
            // chunk jt+1 (requested one iteration ago) has landed; at the barrier every compute wave is past chunk
            // jt-1, whose buffer the copy of chunk jt+2 may now overwrite
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (jt + 2 < p.nchunk) {
                const int pbuf = buf == 0 ? 2 : buf - 1;
                issue_chunk(p.pw + (int64_t)(jt + 2) * kFChunkBytes, voff, wave_lds + pbuf * kFChunkBytes);
            }
            buf = buf == 2 ? 0 : buf + 1;
        }
        return;
    }

    const int t = lane & 31, h = lane >> 5;
    const int tok = blockIdx.x * kFTokBlock + wave * kFTokWave + t;
    const bool valid = tok < p.T;
    const int64_t row = (int64_t)(valid ? tok : p.T - 1) * kFE;

    for (int i = tid; i < p.nchunk * kFChunk; i += kBlock) b1s[i] = p.b1[i];
    par[tid] = p.b2[tid];
    par[kFE + tid] = p.gamma[tid];
    par[2 * kFE + tid] = p.beta[tid];

    uint4 xb[16];   // X^T as B operands: k-step ks covers channels 16ks + 8h .. +7 of my token
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) xb[ks] = *reinterpret_cast<const uint4 *>(p.x + row + 16 * ks + 8 * h);
    // Consume the loads HERE: hipcc places the wait for a pending load at its first use, which would be inside the
    // main loop -- a vmcnt(0) there every iteration also drains the LDS copies it cannot see.
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) asm volatile("" : "+v"(xb[ks].x), "+v"(xb[ks].y), "+v"(xb[ks].z), "+v"(xb[ks].w));

    f32x16_t yacc[8];
#pragma unroll
    for (int et = 0; et < 8; ++et)
#pragma unroll
        for (int i = 0; i < 16; ++i) yacc[et][i] = 0.f;

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // chunks 0 and 1, biases
    __builtin_amdgcn_s_barrier();

    // The loop is ISSUE-bound if it is not kept lean: one wave per SIMD hides ~5 other instructions behind each
    // 32-cycle MFMA, so a chunk (32 MFMAs) may cost ~160: 32 fragment reads + 4 bias reads, their waits, 16 VALU for
    // ReLU + bf16, ~16 for the LDS copy of chunk jt+2, the loop itself.  Hence: the accumulator of the first product
    // STARTS as the bias (four 16-byte LDS reads land in the four register quads -- no zeroing, no adds), ReLU runs
    // on packed bf16 pairs, and the chunk copy uses one scalar base + immediate offsets.
    //
    // A fragments come from LDS through a 4-deep ring of registers (8 would not fit the 256-register budget the loader
    // waves impose): fragment f of a chunk (0..15 = W1 k-steps, 16..31 = W2 (e-tile, k-block)) is requested 4 MFMAs
    // (~130 cycles, the LDS latency) before it is used, and a slot is refilled
    // right AFTER the MFMA that consumed it, so no value ever needs a second register (nothing to copy on the loop's
    // back edge).  The last requests of a chunk fetch the first fragments of the next one.
    constexpr int R = 4;
    uint4 ring[R];
    lds_cptr_t cb = (lds_cptr_t)wbuf + lane * 16;
    const lds_cptr_t bias_base = (lds_cptr_t)(const char *)b1s + 16 * h;
#pragma unroll
    for (int f = 0; f < R; ++f) ring[f] = lds_read16(cb + f * 1024);
    int buf = 0;   // buffer of chunk jt
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
    load_bias(0);

    for (int jt = 0; jt < p.nchunk; ++jt) {
        const int nbuf = buf == 2 ? 0 : buf + 1;                       // buffer of chunk jt+1
        const lds_cptr_t cn = (lds_cptr_t)wbuf + nbuf * kFChunkBytes + lane * 16;
        // H^T[j][t] = b1[32jt + j] + sum_k W1[32jt + j][k] X[t][k]; registers 4g..4g+3 are rows 8g + 4h + {0..3}
        // (hacc already holds the bias: loaded during the previous chunk's second product)
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            mfma_bf16_vgpr(ring[f % R], xb[f], hacc);
            ring[f % R] = lds_read16(cb + (f + R) * 1024);
            __builtin_amdgcn_sched_barrier(0);
        }
        // the VALU below reads registers the last MFMA is still writing: hipcc cannot see that through the inline
        // asm, so the required wait states (8-pass XDL write -> VALU read) are spelled out
        // (the accumulator is an operand of the statement so that its readers cannot be scheduled above it)
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(hacc));
        // ReLU, bf16
        uint4 hp[2];
        hp[0] = make_uint4(relu_bf16x2(pack_bf16x2(hacc[0], hacc[1])), relu_bf16x2(pack_bf16x2(hacc[2], hacc[3])),
                           relu_bf16x2(pack_bf16x2(hacc[4], hacc[5])), relu_bf16x2(pack_bf16x2(hacc[6], hacc[7])));
        hp[1] = make_uint4(relu_bf16x2(pack_bf16x2(hacc[8], hacc[9])), relu_bf16x2(pack_bf16x2(hacc[10], hacc[11])),
                           relu_bf16x2(pack_bf16x2(hacc[12], hacc[13])), relu_bf16x2(pack_bf16x2(hacc[14], hacc[15])));
        // pin the conversion HERE: once it has run, the accumulator registers are free and the next chunk's bias can
        // land in them -- were its live range to reach past load_bias, the allocator would have to move the
        // accumulator between two of the MFMAs above, a copy that reads registers the MFMA before it has not written yet
        asm volatile("" : "+v"(hp[0].x), "+v"(hp[0].y), "+v"(hp[0].z), "+v"(hp[0].w), "+v"(hp[1].x), "+v"(hp[1].y), "+v"(hp[1].z), "+v"(hp[1].w));
        load_bias(jt + 1 < p.nchunk ? jt + 1 : jt);   // next chunk's bias into the now free accumulator registers
        // chunk jt+1 is in LDS once the loaders reach this barrier; passing it also tells them chunk jt-1 is done with
        if (jt + 1 < p.nchunk) __builtin_amdgcn_s_barrier();   // (the loader waves' side: see the top of the kernel)
        // Y^T[e][t] += sum_j W2[e][32jt + j] H^T[j][t]   (W2's j order permuted to the accumulator layout above)
#pragma unroll
        for (int f = 16; f < 32; ++f) {
            const int et = (f - 16) >> 1;
            yacc[et] = mfma_bf16(ring[f % R], hp[f & 1], yacc[et]);
            if (f + R < 32) ring[f % R] = lds_read16(cb + (f + R) * 1024);
            else ring[f % R] = lds_read16(cn + (f + R - 32) * 1024);   // (unused after the last chunk)
            __builtin_amdgcn_sched_barrier(0);
        }
        cb = cn;
        buf = nbuf;
    }

    // ---- + b2 + residual, LayerNorm over the 256 channels held by lanes (t, 0) and (t, 1), store ----
    // all 32 residual pieces are requested before the first is used (one by one in front of their adds each would
    // cost a full memory latency)
    uint2 rbuf[32];
#pragma unroll
    for (int et = 0; et < 8; ++et)
#pragma unroll
        for (int g = 0; g < 4; ++g) rbuf[et * 4 + g] = *reinterpret_cast<const uint2 *>(p.x + row + 32 * et + 8 * g + 4 * h);
    __builtin_amdgcn_sched_barrier(0);
    float sum = 0.f;
#pragma unroll
    for (int et = 0; et < 8; ++et)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int e0 = 32 * et + 8 * g + 4 * h;
            const uint2 r = rbuf[et * 4 + g];
            const float4 bv = *reinterpret_cast<const float4 *>(par + e0);
            yacc[et][4 * g] += bv.x + bf16_lo(r.x);
            yacc[et][4 * g + 1] += bv.y + bf16_hi(r.x);
            yacc[et][4 * g + 2] += bv.z + bf16_lo(r.y);
            yacc[et][4 * g + 3] += bv.w + bf16_hi(r.y);
            sum += (yacc[et][4 * g] + yacc[et][4 * g + 1]) + (yacc[et][4 * g + 2] + yacc[et][4 * g + 3]);
        }
    sum += __shfl_xor(sum, 32);
    const float mean = sum * (1.f / kFE);
    float sq = 0.f;
#pragma unroll
    for (int et = 0; et < 8; ++et)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float d = yacc[et][i] - mean;
            sq += d * d;
        }
    sq += __shfl_xor(sq, 32);
    const float rstd = rsqrtf(sq * (1.f / kFE) + p.eps);
    if (valid) {
        bf16_t *orow = p.out + row;
#pragma unroll
        for (int et = 0; et < 8; ++et)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int e0 = 32 * et + 8 * g + 4 * h;
                const float4 gv = *reinterpret_cast<const float4 *>(par + kFE + e0);
                const float4 be = *reinterpret_cast<const float4 *>(par + 2 * kFE + e0);
                const float y0 = (yacc[et][4 * g] - mean) * rstd * gv.x + be.x;
                const float y1 = (yacc[et][4 * g + 1] - mean) * rstd * gv.y + be.y;
                const float y2 = (yacc[et][4 * g + 2] - mean) * rstd * gv.z + be.z;
                const float y3 = (yacc[et][4 * g + 3] - mean) * rstd * gv.w + be.w;
                *reinterpret_cast<uint2 *>(orow + e0) = make_uint2(pack_bf16x2(y0, y1), pack_bf16x2(y2, y3));
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

}  // namespace sdetr

using namespace sdetr;

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

extern "C" int sdetr_ffn_fused_bf16(sdetr_stream_t stream, const void *x, const void *packed_weights, const float *bias1,
                                    const float *bias2, const float *norm_weight, const float *norm_bias, float norm_eps,
                                    int tokens, int embed_dim, int hidden, void *out)
{
    if (embed_dim != kFE) return fail("ffn_fused: built for embed_dim %d (got %d)", kFE, embed_dim);
    if (hidden <= 0 || hidden % kFChunk) return fail("ffn_fused: hidden (%d) must be a positive multiple of %d", hidden, kFChunk);
    if (tokens < 0) return fail("ffn_fused: negative token count");
    if (tokens == 0) return 0;
    if (!x || !packed_weights || !bias1 || !bias2 || !norm_weight || !norm_bias || !out) return fail("ffn_fused: null pointer");
    const size_t lds = 3 * (size_t)kFChunkBytes + (size_t)hidden * 4 + 3 * kFE * 4;
    if (lds > 160 * 1024) return fail("ffn_fused: hidden %d needs %zu bytes of LDS", hidden, lds);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ffn_fused_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    FfnArgs a;
    a.x = (const bf16_t *)x; a.pw = (const char *)packed_weights; a.b1 = bias1; a.b2 = bias2; a.gamma = norm_weight;
    a.beta = norm_bias; a.eps = norm_eps; a.out = (bf16_t *)out; a.T = tokens; a.nchunk = hidden / kFChunk;
    hipLaunchKernelGGL(ffn_fused_kernel, dim3((unsigned)((tokens + kFTokBlock - 1) / kFTokBlock)), dim3(kFThreads), lds,
                       static_cast<hipStream_t>(stream), a);
    return check_launch("ffn_fused");
}
