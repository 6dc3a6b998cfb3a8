// Can a SIMD's vector ALU work while its matrix pipe runs (gfx950)?  The fp32-accurate GEMM (csrc/gemm_x3.hip) spends
// 35 % of its cycles in MFMAs and 46 % in the vector instructions that split fp32 operands into bf16 terms -- the
// counters say the two do not overlap.  Two wavefronts per SIMD (a 512-thread workgroup per CU):
//   0: waves 0-3 issue 24 MFMAs (6-long dependent chains on 4 accumulators) per iteration, waves 4-7 exit
//   1: waves 4-7 issue 176 independent-chain v_fma_f32 per iteration, waves 0-3 exit
//   2: both at once (different waves of the same SIMD)
//   3: every wave does 176 VALU then 24 MFMAs per iteration (the GEMM's shape), two waves per SIMD
//   4: one wave per SIMD, 7 VALU after every MFMA (same wave, independent registers)
//   5: as 3 but the second wave of each SIMD starts half an iteration late (VALU phase against MFMA phase)
// Prints cycles per iteration (s_memtime deltas of wave 0 / wave 4, max over the workgroup's stamps).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap benchmarks/micro/mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ f32x16_t mfma(u32x4_t a, u32x4_t b, f32x16_t c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

struct Regs {
    f32x16_t acc[4];
    float v[8];
    u32x4_t a, b;
};

__device__ __forceinline__ void mfma_phase(Regs &r)
{
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 6; ++t) r.acc[j] = mfma(r.a, r.b, r.acc[j]);
}

__device__ __forceinline__ void valu_phase(Regs &r, float x)
{
#pragma unroll
    for (int u = 0; u < 22; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(r.v[i]) : "v"(x));
}

template <int MODE>
__global__ void __launch_bounds__(512, 1) k(int iters, unsigned long long *cycles, float *sink)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool second = wave >= 4;
    Regs r;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) r.acc[j][i] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = (float)i;
    r.a = u32x4_t{1u + lane, 2u, 3u, 4u};
    r.b = u32x4_t{5u, 6u + lane, 7u, 8u};
    const float x = 1.0f + lane * 1e-7f;
    if (MODE == 0 && second) return;
    if (MODE == 1 && !second) return;
    if (MODE == 4 && second) return;
    if (MODE == 5 && second) valu_phase(r, x);   // half an iteration of head start for the first wave's MFMAs
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || (MODE == 2 && !second)) mfma_phase(r);
        else if (MODE == 1 || (MODE == 2 && second)) valu_phase(r, x);
        else if (MODE == 3 || MODE == 5) { valu_phase(r, x); mfma_phase(r); }
        else if (MODE == 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    r.acc[j] = mfma(r.a, r.b, r.acc[j]);
#pragma unroll
                    for (int i = 0; i < 7; ++i) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(r.v[i]) : "v"(x));
                }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += r.acc[j][0] + r.acc[j][7];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += r.v[i];
    if (lane == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
    if (s == 123.456f) sink[0] = s;
}

template <int MODE>
static int run(const char *what, unsigned long long *cyc, float *sink, bool last)
{
    const int iters = 2000, blocks = 256;
    CK(hipMemset(cyc, 0, blocks * 8 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, 10, cyc, sink);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, iters, cyc, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[256 * 8];
    CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    unsigned long long mx = 0;
    for (int i = 0; i < 256 * 8; ++i) mx = h[i] > mx ? h[i] : mx;
    printf(" {\"mode\": \"%s\", \"ticks_per_iteration\": %.1f, \"ns_per_iteration\": %.1f}%s\n", what, (double)mx / iters,
           ms * 1e6 / iters, last ? "" : ",");
    return 0;
}

int main()
{
    unsigned long long *cyc;
    float *sink;
    CK(hipMalloc(&cyc, 256 * 8 * 8));
    CK(hipMalloc(&sink, 16));
    printf("{\"mfma_valu_overlap\": [\n");
    if (run<0>("24 MFMA per iteration, one wave per SIMD", cyc, sink, false)) return 1;
    if (run<1>("176 VALU per iteration, one wave per SIMD", cyc, sink, false)) return 1;
    if (run<2>("MFMA wave + VALU wave on every SIMD", cyc, sink, false)) return 1;
    if (run<3>("two waves per SIMD, each 176 VALU then 24 MFMA", cyc, sink, false)) return 1;
    if (run<4>("one wave per SIMD, 7 VALU after every MFMA", cyc, sink, false)) return 1;
    if (run<5>("as [3], second wave half an iteration late", cyc, sink, true)) return 1;
    printf("]}\n");
    return 0;
}
