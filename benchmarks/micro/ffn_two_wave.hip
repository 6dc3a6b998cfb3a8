// How fast can the feed-forward kernel's inner loop (csrc/ffn.hip) go with TWO compute waves per SIMD?
//
// The product kernel is 4 compute waves (32 tokens each, one per SIMD) + 4 loader waves per block; all waves of a
// kernel get the same register budget, so the loaders take half the register file and a compute wave has nobody to
// hide its LDS latency behind: 32 fragment reads per 32-unit chunk through a ring of 4 = ~0.65 us per chunk against
// 0.43 us of MFMA issue.  This benchmark runs the same loop shape (per chunk and wave: 16 MFMAs of the first product
// alternating with 16 of the second, every A fragment one 16-byte LDS read through a ring of 4, packed-bf16
// conversion of the hidden tile between chunks, four 32 KB chunk buffers) in two block shapes:
//   MODE 0: 4 compute waves + 4 loader waves, the loaders issue the LDS-DMA copies (8 KB per wave and chunk);
//   MODE 1: 8 compute waves (256 tokens per block), every wave issues 4 KB of each chunk's copy itself.
// A chunk's copy is issued three iterations before its first use and waited for (vmcnt) two iterations later.
// Numbers are meaningless (weights are whatever the buffer holds); only time is measured.
//   hipcc --offload-arch=gfx950 -O3 -o ffn_two_wave benchmarks/micro/ffn_two_wave.hip && ./ffn_two_wave
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const char *lds_cptr_t;

constexpr int kChunkBytes = 32768;

__device__ __forceinline__ f32x16_t mfma(uint4 a, uint4 b, f32x16_t c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ uint4 lds_read16(lds_cptr_t p)
{
    const u32x4_t v = *reinterpret_cast<const __attribute__((address_space(3))) u32x4_t *>(p);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t pack2(float a, float b)
{
    return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u);
}
// 4 KB: four LDS-DMA instructions of 1 KB (64 lanes x 16 bytes); the instruction offset advances source and destination
__device__ __forceinline__ void dma4k(const char *src, uint32_t voff, uint32_t dst_lds)
{
    const uint32_t d = __builtin_amdgcn_readfirstlane(dst_lds);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %2\n\t"
                 "global_load_lds_dwordx4 %0, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %0, %2 offset:2048\n\t"
                 "global_load_lds_dwordx4 %0, %2 offset:3072"
                 :
                 : "v"(voff), "s"(d), "s"(src)
                 : "memory", "m0");
}

template <int MODE>
__global__ void __launch_bounds__(512, 1) ffn_like(const char *pw, int nchunk, float *out)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)lds;
    constexpr int kCopyWaves = MODE == 0 ? 4 : 8;              // waves that issue the copies
    constexpr int kPerWave = kChunkBytes / kCopyWaves;         // 8 KB or 4 KB
    const bool copier = MODE == 1 || wave >= 4;
    const int cw = MODE == 1 ? wave : wave - 4;
    auto issue = [&](int c) {
        const char *src = pw + (int64_t)c * kChunkBytes;
        const uint32_t dst = lds_base + (uint32_t)((c & 3) * kChunkBytes + cw * kPerWave);
#pragma unroll
        for (int q = 0; q < kPerWave / 4096; ++q) dma4k(src, (uint32_t)(cw * kPerWave + q * 4096 + lane * 16), dst + q * 4096);
    };
    constexpr int kIssues = kPerWave / 1024;                   // DMA instructions per wave and chunk

    if (copier) { issue(0); issue(1); issue(2); }
    if (MODE == 0 && wave >= 4) {
        // loader wave: wait for chunk j + 1, barrier, request chunk j + 3
        for (int j = 0; j + 1 < nchunk; ++j) {
            if (kIssues == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            issue(j + 3);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // compute wave: X^T operands, output accumulators, hidden accumulator
    uint4 xb[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) xb[k] = make_uint4(0x3f803f80u + k, 0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u);
    f32x16_t yacc[8], hacc;
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int i = 0; i < 16; ++i) yacc[e][i] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) hacc[i] = 0.f;
    uint4 hp[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    constexpr int R = 4;
    const lds_cptr_t lbase = (lds_cptr_t)lds + lane * 16;
    auto w2_frag = [](int q) { return (16 + 2 * (q & 7) + (q >> 3)) * 1024; };

    for (int j = 0; j + 1 < nchunk; ++j) {
        if (MODE == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // my share of chunk j + 1 has landed
        __builtin_amdgcn_s_barrier();
        if (MODE == 1) issue(j + 3);
        const lds_cptr_t ca = lbase + ((j + 1) & 3) * kChunkBytes, cb = lbase + (j & 3) * kChunkBytes;
        uint4 ring[R];
#pragma unroll
        for (int f = 0; f < R; ++f) ring[f] = (f & 1) ? lds_read16(cb + w2_frag(f >> 1)) : lds_read16(ca + (f >> 1) * 1024);
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            if (s & 1) {
                const int q = s >> 1;
                yacc[q & 7] = mfma(ring[s % R], hp[q >> 3], yacc[q & 7]);
            } else {
                hacc = mfma(ring[s % R], xb[s >> 1], hacc);
            }
            if (s + R < 32)
                ring[s % R] = ((s + R) & 1) ? lds_read16(cb + w2_frag((s + R) >> 1)) : lds_read16(ca + ((s + R) >> 1) * 1024);
            __builtin_amdgcn_sched_barrier(0);
        }
        // conversion of the hidden tile (packed bf16, ReLU as an integer max), accumulator restarts
        hp[0] = make_uint4(pack2(hacc[0], hacc[1]), pack2(hacc[2], hacc[3]), pack2(hacc[4], hacc[5]), pack2(hacc[6], hacc[7]));
        hp[1] = make_uint4(pack2(hacc[8], hacc[9]), pack2(hacc[10], hacc[11]), pack2(hacc[12], hacc[13]), pack2(hacc[14], hacc[15]));
#pragma unroll
        for (int i = 0; i < 16; ++i) hacc[i] = 1.f;
    }
    if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float t = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int i = 0; i < 16; ++i) t += yacc[e][i];
    if (t == 123.456f) out[tid] = t;   // (keeps the work alive)
}

template <int MODE>
static float run(const char *pw, int nchunk, int blocks, float *out, int reps)
{
    const size_t ldsb = 4 * kChunkBytes;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ffn_like<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(ffn_like<MODE>, dim3(blocks), dim3(512), ldsb, 0, pw, nchunk, out);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(ffn_like<MODE>, dim3(blocks), dim3(512), ldsb, 0, pw, nchunk, out);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main()
{
    const int maxchunk = 64 + 4;
    char *pw;
    float *out;
    (void)hipMalloc(&pw, (size_t)maxchunk * kChunkBytes);
    (void)hipMemset(pw, 0x3c, (size_t)maxchunk * kChunkBytes);
    (void)hipMalloc(&out, 4096);
    printf("{\"ffn_two_wave\": [\n");
    const int blocks_list[] = {64, 178, 256};
    bool first = true;
    for (int blocks : blocks_list)
        for (int nchunk : {16, 32, 64}) {
            const float a = run<0>(pw, nchunk, blocks, out, 20), b = run<1>(pw, nchunk, blocks, out, 20);
            printf("%s {\"blocks\": %d, \"chunks\": %d, \"us_4compute_4loader_128tok\": %.2f, \"us_8compute_256tok\": %.2f}",
                   first ? "" : ",\n", blocks, nchunk, a, b);
            first = false;
        }
    printf("\n]}\n");
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "device error\n"); return 1; }
    return 0;
}
