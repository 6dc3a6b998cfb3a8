// The token-resident linear kernel's body (see token_linear.hip), as a device function so that another launch can carry
// it next to other work (fused_head_value.hip).
#pragma once
#include "common.h"

namespace sdetr {

constexpr int kTLK = 256;
constexpr int kTLTileBytes = 16384;    // 32 output features x 256 k, as 16 1-KB A fragments
constexpr int kTLTokWave = 32, kTLTokBlock = 128;

typedef __bf16 tl_bf16x8_t __attribute__((ext_vector_type(8)));
typedef float tl_f32x16_t __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) const char *tl_lds_cptr_t;

enum { kStore = 0, kHeadMajor = 1, kClassMax = 2 };

struct TLArgs {
    const bf16_t *x;          // [T, 256]
    const bf16_t *x2;         // optional addend, rows_per_batch rows per image, images x2_batch_stride elements apart
    int64_t x2_batch_stride;
    int rows_per_batch;       // tokens per image (x2 / scale / head-major addressing)
    const char *pw;           // packed weights, ntiles * 16 KB
    const float *bias;        // [ntiles * 32] (zero padded)
    int T, N, ntiles;
    // kStore
    bf16_t *out;
    int64_t out_row_stride;
    int group;                // > 0: features per group, out is [B][N/group][rows_per_batch][group] (head-major)
    // rows another part of the same launch writes instead (fused_head_value.hip): token (img, ri) is skipped when
    // skip_hint[img][ri] = m in 1..skip_n and skip_sel[img][m - 1] == ri (hints are validated, stale ones are harmless)
    const int32_t *skip_hint;
    int64_t skip_hint_bs;
    const int64_t *skip_sel;
    int skip_n;
    // kHeadMajor
    const uint8_t *pad;       // [T] or NULL
    void *hm;                 // [groups][B][M][hm_records][32]
    int heads, batch, hm_f16;
    // bordered destination (include/salience_hip.h sdetr_bordered_layout): token ri of an image -> record hm_pixel_map[ri]
    // (NULL: record ri, hm_records = rows_per_batch); the launch's `hm_blocks` token blocks share the zero fill of the
    // hm_num_border border records of every map
    const int32_t *hm_pixel_map;
    const int32_t *hm_border;
    int hm_num_border, hm_records, hm_blocks;
    // kClassMax
    const float *scale;       // [B, rows_per_batch] with batch stride
    int64_t scale_batch_stride;
    float *cmax;              // [T]
};

// destination layout of a head-major job: plain (`bordered` == NULL) or bordered; `blocks` = token blocks of the launch
inline int tl_set_bordered(TLArgs &a, const sdetr_bordered_layout *bordered, int spatial_size, int blocks)
{
    a.hm_pixel_map = nullptr; a.hm_border = nullptr; a.hm_num_border = 0; a.hm_records = spatial_size; a.hm_blocks = blocks;
    if (!bordered) return 0;
    if (!bordered->pixel_map || !bordered->border || bordered->num_border <= 0 || bordered->records < spatial_size + bordered->num_border)
        return fail("value projection: bad bordered layout (records %d, border %d, tokens %d)", bordered->records,
                    bordered->num_border, spatial_size);
    a.hm_pixel_map = bordered->pixel_map; a.hm_border = bordered->border; a.hm_num_border = bordered->num_border;
    a.hm_records = bordered->records;
    return 0;
}

__device__ __forceinline__ tl_f32x16_t tl_mfma(uint4 a, uint4 b, tl_f32x16_t c)
{
    return mfma_act_32x32x16(a, b, c);
}

// Weights move in STEPS of four tiles (64 KB): one block barrier and one round of LDS-DMA latency per 64 MFMAs
// instead of per 16 (with a barrier per tile the 12 tiles of the 256 -> 384 projection took 17 us, ~1.3 us each for
// 0.26 us of MFMA work).  A wave copies its quarter of the step, 16 KB = 16 LDS-DMA instructions of 1 KB
// (destination = M0 base + instruction offset + lane * 16; the offset also advances the global source).
constexpr int kTLStepTiles = 4;
constexpr int kTLStepBytes = kTLStepTiles * kTLTileBytes;

template <int GROUPS = 4>   // 4 KB groups per wave: 4 with four waves per block, 2 with eight
__device__ __forceinline__ void tl_issue_step(const char *step, uint32_t voff, uint32_t dst_lds)
{
#pragma unroll
    for (int q = 0; q < GROUPS; ++q) {
        const uint32_t d = __builtin_amdgcn_readfirstlane(dst_lds + q * 4096);
        const uint32_t v = voff + q * 4096;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, %2\n\t"
                     "global_load_lds_dwordx4 %0, %2 offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, %2 offset:2048\n\t"
                     "global_load_lds_dwordx4 %0, %2 offset:3072"
                     :
                     : "v"(v), "s"(d), "s"(step)
                     : "memory", "m0");
    }
}

__device__ __forceinline__ uint4 tl_lds_read16(tl_lds_cptr_t p)
{
    const u32x4_t v = *reinterpret_cast<const __attribute__((address_space(3))) u32x4_t *>(p);
    return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ uint32_t add_bf16x2(uint32_t a, uint32_t b)
{
    return pack_act2(act_lo(a) + act_lo(b), act_hi(a) + act_hi(b));
}

// WAVES = 4: 128 tokens per block (the per-layer projections: <= 256 blocks, one per CU); WAVES = 8: 256 tokens per
// block, two waves per SIMD sharing one weight stream (the value projection over all 44 646 tokens: 175 blocks in one
// round instead of 349 in two, half the copy issues per wave, and the second wave's MFMAs cover the first's stores).
// 16 named uint4 registers (a loader wave's 16 KB of a 64 KB step); named, because an array handed to a helper stays in
// scratch memory under the asm memory clobbers
#define SDETR_TL_DECL16(P) uint4 P##0, P##1, P##2, P##3, P##4, P##5, P##6, P##7, P##8, P##9, P##10, P##11, P##12, P##13, P##14, P##15
#define SDETR_TL_GLD(dst, base, off) asm volatile("global_load_dwordx4 %0, %1, off offset:" #off : "=v"(dst) : "v"(base) : "memory")
#define SDETR_TL_FETCH16(P, step)                                                                                  \
    {                                                                                                              \
        const uint4 *q0_ = src + (int64_t)(step) * (kTLStepBytes / 16), *q1_ = q0_ + 256, *q2_ = q0_ + 512, *q3_ = q0_ + 768; \
        SDETR_TL_GLD(P##0, q0_, 0); SDETR_TL_GLD(P##1, q0_, 1024); SDETR_TL_GLD(P##2, q0_, 2048); SDETR_TL_GLD(P##3, q0_, 3072);   \
        SDETR_TL_GLD(P##4, q1_, 0); SDETR_TL_GLD(P##5, q1_, 1024); SDETR_TL_GLD(P##6, q1_, 2048); SDETR_TL_GLD(P##7, q1_, 3072);   \
        SDETR_TL_GLD(P##8, q2_, 0); SDETR_TL_GLD(P##9, q2_, 1024); SDETR_TL_GLD(P##10, q2_, 2048); SDETR_TL_GLD(P##11, q2_, 3072); \
        SDETR_TL_GLD(P##12, q3_, 0); SDETR_TL_GLD(P##13, q3_, 1024); SDETR_TL_GLD(P##14, q3_, 2048); SDETR_TL_GLD(P##15, q3_, 3072); \
    }
#define SDETR_TL_ST(P, i) *reinterpret_cast<uint4 *>(d_ + (i) * 1024) = P##i
#define SDETR_TL_STASH16(P, step)                                                                                  \
    {                                                                                                              \
        char *d_ = dst + ((step) & 1) * kTLStepBytes;                                                              \
        SDETR_TL_ST(P, 0); SDETR_TL_ST(P, 1); SDETR_TL_ST(P, 2); SDETR_TL_ST(P, 3); SDETR_TL_ST(P, 4); SDETR_TL_ST(P, 5);       \
        SDETR_TL_ST(P, 6); SDETR_TL_ST(P, 7); SDETR_TL_ST(P, 8); SDETR_TL_ST(P, 9); SDETR_TL_ST(P, 10); SDETR_TL_ST(P, 11);     \
        SDETR_TL_ST(P, 12); SDETR_TL_ST(P, 13); SDETR_TL_ST(P, 14); SDETR_TL_ST(P, 15);                                         \
    }

// Block = WAVES compute waves (32 tokens each) + 4 LOADER waves.  WAVES = 4: 128 tokens per block; WAVES = 8: 256
// tokens per block, two compute waves per SIMD sharing one weight stream (the value projection over all 44 646 tokens:
// 175 blocks in one round instead of 349 in two).
// The loaders bring the weights global -> registers -> LDS in steps of four tiles (64 KB, two LDS buffers), running two
// steps ahead.  (When the compute waves issued the copies themselves -- LDS-DMA plus s_waitcnt vmcnt(0) at every step --
// the wait also covered their own row-strided output STORES of the step before: ~1 us per tile against 0.22 us of MFMA
// work, the same at 36 and at 178 blocks.)  The compute waves now never wait on memory inside the loop.
// (`block` = index of the token block: the kernel's blockIdx.x, or its position inside a launch that also carries other
// work -- fused_head_value.hip)
template <int EPI, bool ADD2, int WAVES>
__device__ __forceinline__ void token_linear_body(const TLArgs &p, int block)
{
    constexpr int kThreads = 64 * WAVES;   // compute threads
    // two compute waves per SIMD hide each other's LDS latency and leave 168 registers per wave: a ring of 4 there, 8 with
    // one wave per SIMD
    constexpr int R = WAVES == 8 ? 4 : 8;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *wbuf = lds;                                                   // 2 step buffers
    float *bs = reinterpret_cast<float *>(lds + 2 * kTLStepBytes);      // [nsteps * 128]
    const int nsteps = (p.ntiles + kTLStepTiles - 1) / kTLStepTiles;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef TL_STAMPS   // (benchmark builds: cycle stamps of compute wave 0, printed by two workgroups)
    long long tl_t[12];
    int tl_n = 0;
#define TL_STAMP() tl_t[tl_n++] = clock64()
#else
#define TL_STAMP()
#endif
    TL_STAMP();

    if (wave >= WAVES) {
        // ---- loader wave: a quarter (16 KB) of every 64 KB step.  Step s travels in register set s & 1 and is written
        // into LDS buffer s & 1 while the compute waves work on step s - 1.  The packed buffer is padded to whole steps.
        const int lw = wave - WAVES;
        const uint4 *src = reinterpret_cast<const uint4 *>(p.pw) + lw * 1024 + lane;
        char *dst = wbuf + lw * 16384 + lane * 16;
        SDETR_TL_DECL16(ra);
        SDETR_TL_DECL16(rb);
        SDETR_TL_FETCH16(ra, 0);
        SDETR_TL_FETCH16(rb, nsteps > 1 ? 1 : 0);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        SDETR_TL_STASH16(ra, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SDETR_TL_FETCH16(ra, nsteps > 2 ? 2 : 0);
        __builtin_amdgcn_s_barrier();                       // step 0 is in LDS
        for (int st = 0; st + 1 < nsteps; st += 2) {
            // compute is on step st (even): step st+1 (set rb) goes into the other buffer, step st+3 is requested
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            SDETR_TL_STASH16(rb, st + 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SDETR_TL_FETCH16(rb, st + 3 < nsteps ? st + 3 : 0);
            __builtin_amdgcn_s_barrier();                   // step st+1 is in LDS, everyone is done with step st
            if (st + 2 >= nsteps) break;
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            SDETR_TL_STASH16(ra, st + 2);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SDETR_TL_FETCH16(ra, st + 4 < nsteps ? st + 4 : 0);
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (the dummy requests of the tail)
        return;
    }

    if (EPI == kHeadMajor && p.hm_border) {
        // this block's share of the border records: 64 bytes of zeros in every (group, head) map of the entry's image
        // (16-byte stores, fire and forget -- nothing in this launch reads them)
        const int total = p.batch * p.hm_num_border;
        const int per = (total + p.hm_blocks - 1) / p.hm_blocks;
        const int e0 = block * per, e1 = min(total, e0 + per);
        const int pieces = (e1 - e0) * p.ntiles * 4;
        for (int i = tid; i < pieces; i += kThreads) {
            const int q4 = i & 3, rest = i >> 2;
            const int nt = rest % p.ntiles, e = e0 + rest / p.ntiles;
            const int bi = e / p.hm_num_border, rec = p.hm_border[e - bi * p.hm_num_border];
            const int grp = nt / p.heads, hm = nt - grp * p.heads;
            const int64_t pix = (((int64_t)grp * p.batch + bi) * p.heads + hm) * p.hm_records + rec;
            *reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(p.hm) + pix * 32 + q4 * 8) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    const int t = lane & 31, h = lane >> 5;
    const int tok = block * (kTLTokWave * WAVES) + wave * kTLTokWave + t;
    const bool valid = tok < p.T;
    const int tk = valid ? tok : p.T - 1;
    const int img = tk / p.rows_per_batch, ri = tk - img * p.rows_per_batch;
    const int hm_rec = (EPI == kHeadMajor && p.hm_pixel_map) ? p.hm_pixel_map[ri] : ri;   // record of my token in its map

    for (int i = tid; i < nsteps * 128; i += kThreads) bs[i] = p.bias[i];
    uint4 xb[16];   // X^T as B operands: k-step ks covers channels 16ks + 8h .. +7 of my token
#ifndef TL_ROW_STRIDED_X
    constexpr bool kLinearX = WAVES == 4;   // (8 compute waves: two rounds through half the staging space were slower, +8 us per launch)
#else
    constexpr bool kLinearX = false;
#endif
    if (!kLinearX) {
        const bf16_t *xr = p.x + (int64_t)tk * kTLK + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) xb[ks] = *reinterpret_cast<const uint4 *>(xr + 16 * ks);
        if (ADD2) {
            const bf16_t *x2r = p.x2 + (int64_t)img * p.x2_batch_stride + (int64_t)ri * kTLK + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const uint4 o = *reinterpret_cast<const uint4 *>(x2r + 16 * ks);
                xb[ks] = make_uint4(add_bf16x2(xb[ks].x, o.x), add_bf16x2(xb[ks].y, o.y), add_bf16x2(xb[ks].z, o.z),
                                    add_bf16x2(xb[ks].w, o.w));
            }
        }
    } else {
        // Round 6.  Read as operands (lane = token, 16 pieces of 16 bytes 32 bytes apart along its row) the wave's 16 KB
        // of activations were 32 cache lines per load instruction, 1024 line requests through the CU's L1 for 256 lines
        // of data: cycle stamps put 16 000 of a workgroup's 36 000 cycles here.  The wave's 32 rows are CONTIGUOUS in
        // memory, so they are loaded as they lie (a KB per instruction: two rows, lane l = row l / 32, piece l % 32),
        // x2 is added in that layout, and the pieces change lanes through LDS -- the second weight buffer, which the
        // loaders do not write before the first barrier; piece p of row r sits in slot p ^ r of the row: every LDS
        // instruction conflict-free.  With 8 compute waves a wave's share of that buffer is 8 KB: two rounds of 16 rows.
        constexpr int kRows = WAVES == 8 ? 16 : 32;                 // rows per round
        char *stage = wbuf + kTLStepBytes + wave * (kTLStepBytes / WAVES);
        const int piece = lane & 31, half = lane >> 5;
        const int tok0 = block * (kTLTokWave * WAVES) + wave * kTLTokWave;
        uint4 lin[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int tr = min(tok0 + 2 * i + half, p.T - 1);
            lin[i] = *reinterpret_cast<const uint4 *>(p.x + (int64_t)tr * kTLK + 8 * piece);
        }
        if (ADD2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int tr = min(tok0 + 2 * i + half, p.T - 1);
                const int im = tr / p.rows_per_batch, rr = tr - im * p.rows_per_batch;
                const uint4 o = *reinterpret_cast<const uint4 *>(p.x2 + (int64_t)im * p.x2_batch_stride + (int64_t)rr * kTLK + 8 * piece);
                lin[i] = make_uint4(add_bf16x2(lin[i].x, o.x), add_bf16x2(lin[i].y, o.y), add_bf16x2(lin[i].z, o.z),
                                    add_bf16x2(lin[i].w, o.w));
            }
        }
#pragma unroll
        for (int round = 0; round < 32 / kRows; ++round) {
            if (round > 0) {   // (the first round's reads are done before its slots are written again)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (int i = 0; i < kRows / 2; ++i) {
                const int r = 2 * i + half;            // row of the round
                *reinterpret_cast<uint4 *>(stage + r * 512 + ((piece ^ r) & 31) * 16) = lin[round * (kRows / 2) + i];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int r = t - round * kRows;           // my token's row in this round, if it is in it
            if (r >= 0 && r < kRows) {
#pragma unroll
                for (int ks = 0; ks < 16; ++ks)
                    xb[ks] = *reinterpret_cast<const uint4 *>(stage + r * 512 + (((2 * ks + h) ^ r) & 31) * 16);
            }
        }
    }
    // consume the loads here so that hipcc's wait for them is not placed inside the loop (it would drain the LDS
    // copies it cannot see on every iteration)
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) asm volatile("" : "+v"(xb[ks].x), "+v"(xb[ks].y), "+v"(xb[ks].z), "+v"(xb[ks].w));

    float run_max = -INFINITY;
    const bool masked = EPI == kHeadMajor && p.pad && p.pad[tk];
    // (a row the attention kernel of this launch projects itself is skipped: a mark in `skip_hint` that the selection
    // confirms.  The confirmation is a trip to memory BEHIND the mark's: it is requested after the first barrier and
    // travels under step 0's products -- in front of the barrier it kept every wave of the workgroup waiting, 6 000 of the
    // workgroup's 36 000 cycles by the stamps)
    bool skip = false;
    int skip_mark = 0;
    if (EPI == kStore && p.skip_hint) skip_mark = p.skip_hint[(int64_t)img * p.skip_hint_bs + ri];

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // my activations and the staged bias
    TL_STAMP();   // 1: my activations are back
    __builtin_amdgcn_s_barrier();
    TL_STAMP();   // 2: step 0 is in LDS
    if (EPI == kStore && p.skip_hint && skip_mark > 0 && skip_mark <= p.skip_n)
        skip = p.skip_sel[(int64_t)img * p.skip_n + skip_mark - 1] == ri;

    // A fragments come from LDS through a ring of R registers (requested R MFMAs before use, refilled right after the
    // MFMA that consumed the slot): with one wave per SIMD nothing else hides the LDS latency.
    for (int st = 0; st < nsteps; ++st) {
        if (st > 0) __builtin_amdgcn_s_barrier();   // step st is in LDS; the loaders learn that step st-1 is done with
        const tl_lds_cptr_t cb = (tl_lds_cptr_t)wbuf + (st & 1) * kTLStepBytes + lane * 16;
        // The four tiles of a step are walked k-step-major: MFMA 4 ks + j feeds tile j, so two MFMAs into the same
        // accumulator are four issues apart.  (Tile-major, each tile was a chain of 16 dependent MFMAs and a dependent
        // MFMA waits for the full latency of its predecessor, twice the issue interval: ~1 us per tile measured against
        // 0.22 us of MFMA work.)  Fragment of consumption slot f: tile f & 3, k-step f >> 2.
        auto frag = [](int f) { return ((f & 3) * 16 + (f >> 2)) * 1024; };
        uint4 ring[R];
#pragma unroll
        for (int f = 0; f < R; ++f) ring[f] = tl_lds_read16(cb + frag(f));
        tl_f32x16_t accs[kTLStepTiles];   // start as the bias: registers 4g..4g+3 are features 8g + 4h + {0..3} of the tile
#pragma unroll
        for (int j = 0; j < kTLStepTiles; ++j) {
            const tl_lds_cptr_t bb = (tl_lds_cptr_t)(const char *)bs + (st * kTLStepTiles + j) * 128 + 16 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint4 bv = tl_lds_read16(bb + 32 * g);
                accs[j][4 * g] = __uint_as_float(bv.x);
                accs[j][4 * g + 1] = __uint_as_float(bv.y);
                accs[j][4 * g + 2] = __uint_as_float(bv.z);
                accs[j][4 * g + 3] = __uint_as_float(bv.w);
            }
        }
#pragma unroll
        for (int f = 0; f < kTLStepTiles * 16; ++f) {
            accs[f & 3] = tl_mfma(ring[f % R], xb[f >> 2], accs[f & 3]);
            if (f + R < kTLStepTiles * 16) ring[f % R] = tl_lds_read16(cb + frag(f + R));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (st < 4) TL_STAMP();   // 3 + 2 st: products of step st
        if (EPI == kClassMax) {
#pragma unroll
            for (int j = 0; j < kTLStepTiles; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int n = (st * kTLStepTiles + j) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (n < p.N) run_max = fmaxf(run_max, accs[j][i]);
                }
        } else {
            // Output.  Straight from the accumulator layout a lane owns 4 features of one token (8-byte stores, 32 rows per
            // instruction, ~110 cycles of issue each: the 16 of a step cost as much as its 64 MFMAs).  The two lanes of a
            // token therefore first exchange 8-byte pieces with v_permlane32_swap so that each owns 8 consecutive
            // features, and stores 16 bytes: half the store instructions, no LDS.  (A per-wave LDS staging tile for fully
            // coalesced rows was slower: every extra LDS instruction between the MFMAs costs ~130 cycles of issue --
            // benchmarks/micro/mfma_rate.hip.)
#pragma unroll
            for (int j = 0; j < kTLStepTiles; ++j) {
                const int nt = st * kTLStepTiles + j;
                if (nt >= p.ntiles) break;
                const tl_f32x16_t acc = accs[j];
                uint32_t d[8];   // d[2g], d[2g+1] = features 8g + 4h + {0,1}, {2,3} of the tile
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (EPI == kHeadMajor)
                        d[i] = masked ? 0u : (p.hm_f16 ? pack_f16x2(acc[2 * i], acc[2 * i + 1]) : pack_bf16x2(acc[2 * i], acc[2 * i + 1]));   // (value maps: their own explicit type)
                    else
                        d[i] = pack_act2(acc[2 * i], acc[2 * i + 1]);
                }
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        // row 1's first piece <-> row 0's second piece of the 16-feature half m
                        const auto r = __builtin_amdgcn_permlane32_swap(d[4 * m + w], d[4 * m + 2 + w], false, false);
                        d[4 * m + w] = r[0];
                        d[4 * m + 2 + w] = r[1];
                    }
                if (!valid || skip) continue;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    // d[4m .. 4m+3] = features 16 m + 8 h + {0..7} of tile nt
                    const uint4 v = make_uint4(d[4 * m], d[4 * m + 1], d[4 * m + 2], d[4 * m + 3]);
                    const int n0 = nt * 32 + 16 * m + 8 * h;
                    if (EPI == kHeadMajor) {
                        const int grp = nt / p.heads, hm = nt - grp * p.heads;
                        const int64_t pix = (((int64_t)grp * p.batch + img) * p.heads + hm) * p.hm_records + hm_rec;
                        *reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(p.hm) + pix * 32 + 16 * m + 8 * h) = v;
                    } else if (p.group > 0) {
                        // feature-group-major [B][N/group][rows][group]: the group size is a multiple of 4
                        const int ngroups = p.N / p.group;
                        if ((p.group & 7) == 0 && n0 + 8 <= p.N) {   // the piece lies inside one group
                            const int gi = n0 / p.group, within = n0 - gi * p.group;
                            *reinterpret_cast<uint4 *>(p.out + (((int64_t)img * ngroups + gi) * p.rows_per_batch + ri) * p.group + within) = v;
                            continue;
                        }
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int n = n0 + 4 * q;
                            if (n >= p.N) continue;
                            const int gi = n / p.group, within = n - gi * p.group;
                            *reinterpret_cast<uint2 *>(p.out + (((int64_t)img * ngroups + gi) * p.rows_per_batch + ri) * p.group + within) =
                                q ? make_uint2(v.z, v.w) : make_uint2(v.x, v.y);
                        }
                    } else {
                        bf16_t *o = p.out + (int64_t)tok * p.out_row_stride + n0;
                        if (n0 + 8 <= p.N && (p.out_row_stride & 7) == 0) *reinterpret_cast<uint4 *>(o) = v;
                        else {
                            if (n0 < p.N) *reinterpret_cast<uint2 *>(o) = make_uint2(v.x, v.y);
                            if (n0 + 4 < p.N) *reinterpret_cast<uint2 *>(o + 4) = make_uint2(v.z, v.w);
                        }
                    }
                }
            }
        }
        if (st < 4) TL_STAMP();   // 4 + 2 st: stores of step st
    }
#ifdef TL_STAMPS
    if (EPI == kHeadMajor && tid == 0 && (block == 0 || block == 50) && nsteps == 2)
        printf("value projection blk=%d cycles: x %lld | wait step0 %lld | s0 mfma %lld st %lld | s1 wait+mfma %lld st %lld | total %lld\n",
               block, tl_t[1] - tl_t[0], tl_t[2] - tl_t[1], tl_t[3] - tl_t[2], tl_t[4] - tl_t[3], tl_t[5] - tl_t[4], tl_t[6] - tl_t[5],
               tl_t[6] - tl_t[0]);
    if (EPI == kStore && tid == 0 && (block == 0 || block == 50) && nsteps == 3)
        printf("token_linear blk=%d cycles: x %lld | wait step0 %lld | s0 mfma %lld st %lld | s1 wait+mfma %lld st %lld | s2 wait+mfma %lld st %lld | total %lld\n",
               block, tl_t[1] - tl_t[0], tl_t[2] - tl_t[1], tl_t[3] - tl_t[2], tl_t[4] - tl_t[3], tl_t[5] - tl_t[4], tl_t[6] - tl_t[5],
               tl_t[7] - tl_t[6], tl_t[8] - tl_t[7], tl_t[8] - tl_t[0]);
#endif
    if (EPI == kClassMax) {
        run_max = fmaxf(run_max, __shfl_xor(run_max, 32));
        if (valid && h == 0) {
            // The score that picks the layer's top-300 rows comes straight from the fp32 accumulators.  (Until round 3
            // the maximum was first rounded to bf16, "as the reference's autocast Linear would": a quantum of 2^-6 on
            // logits of 2..4 against ~750 candidates per unit of score at the cut -- 39 % of the selections then
            // differed from the fp32 reference's, ~5 % without the rounding.)
            p.cmax[tok] = run_max * p.scale[(int64_t)img * p.scale_batch_stride + ri];
        }
    }
}

template <int EPI, bool ADD2, int WAVES>
__global__ void __launch_bounds__(64 * (WAVES + 4), 1) token_linear_kernel(TLArgs p)
{
    token_linear_body<EPI, ADD2, WAVES>(p, (int)blockIdx.x);
}

}  // namespace sdetr
