// Issue rate of the vector instructions a bilinear gather could be built from (gfx950): wave-instructions per cycle
// and SIMD for v_fma_f32, v_fma_mix_f32 (fp16 source, fp32 accumulate), v_pk_fma_f32, v_pk_fma_f16, v_dot2_f32_f16,
// each as 8 independent accumulator chains, 4 waves per SIMD, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate benchmarks/micro/valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(1024) rate_kernel(int iters, float *out, long long *cycles)
{
    float a[8];
    float2_t p[8];
    half2_t h[8];
    const float x = 1.0f + threadIdx.x * 1e-7f;
    const half2_t hx = {(_Float16)1.0f, (_Float16)(threadIdx.x * 1e-3f)};
    uint32_t hbits = __builtin_bit_cast(uint32_t, hx);
    for (int i = 0; i < 8; ++i) { a[i] = i; p[i] = float2_t{(float)i, 1.f}; h[i] = hx; }
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(x));
                if (KIND == 1) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(hbits), "v"(x));
                if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "v"(p[(i + 2) & 7]));
                if (KIND == 3) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(h[i]) : "v"(hbits), "v"(hbits));
                if (KIND == 4) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(hbits), "v"(hbits));
            }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + (float)h[i].x;
    if (s == 12345.678f) out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int KIND>
static void run(const char *name, float *out, long long *cyc, bool last)
{
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(256), dim3(1024), 0, 0, iters, out, cyc);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(256), dim3(1024), 0, 0, iters, out, cyc);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long c = 0;
    (void)hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    // 1024 threads = 16 waves per CU = 4 per SIMD; each wave issues iters * 32 instructions
    const double per_simd = 4.0 * iters * 32;
    printf(" {\"instr\": \"%s\", \"us\": %.1f, \"wave_instr_per_simd_per_us\": %.1f, \"shader_clock_counter_ticks\": %lld}%s\n", name, ms * 1e3,
           per_simd / (ms * 1e3), c, last ? "" : ",");
}

int main()
{
    float *out;
    long long *cyc;
    (void)hipMalloc(&out, 4096);
    (void)hipMalloc(&cyc, 8);
    printf("{\"valu_rate\": [\n");
    run<0>("v_fma_f32", out, cyc, false);
    run<1>("v_fma_mix_f32 (f16 src)", out, cyc, false);
    run<2>("v_pk_fma_f32", out, cyc, false);
    run<3>("v_pk_fma_f16", out, cyc, false);
    run<4>("v_dot2_f32_f16", out, cyc, true);
    printf("]}\n");
    return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
