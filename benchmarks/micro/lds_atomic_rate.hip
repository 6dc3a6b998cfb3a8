// Ground-truth rates on gfx950 for accumulating fp32 in LDS (the MSDA backward's grad_value scatter):
// ds_add_f32 with different lane -> address patterns, against global_atomic_add_f32 on the same pattern.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o lds_atomic_rate benchmarks/micro/lds_atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// (rows must be a power of two: the index math is kept to a mask so that it does not bound the loop)
// PATTERN 0: a wave-instruction covers 2 random rows x 32 consecutive floats  (one channel per lane, D = 32)
// PATTERN 1: 4 random rows x 16 consecutive floats
// PATTERN 2: 16 random rows x 4 consecutive floats (lane = 4 channels apart ...), worst case of the quad mapping
// PATTERN 3: all 64 lanes the same 64 consecutive floats every time (no randomness: best case)
template <int PATTERN, bool GLOBAL>
__global__ void __launch_bounds__(1024, 1) k(int iters, int rows, float *gbuf, float *out)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < rows * 32; i += 1024) lds[i] = 0.f;
    __syncthreads();
    uint32_t s = (blockIdx.x * 16 + wave) * 2654435761u + 12345u;
    float *base = GLOBAL ? gbuf + (size_t)(blockIdx.x % 8) * rows * 32 : lds;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s = s * 1664525u + 1013904223u;
            uint32_t r = s >> 8;
            int idx;
            if (PATTERN == 0) idx = ((r + (lane >> 5) * 7919u) & (rows - 1)) * 32 + (lane & 31);
            else if (PATTERN == 1) idx = ((r + (lane >> 4) * 7919u) & (rows - 1)) * 32 + (lane & 15) + 16 * (u & 1);
            else if (PATTERN == 2) idx = ((r + (lane >> 2) * 7919u) & (rows - 1)) * 32 + (lane & 3) + 4 * (u & 7);
            else idx = lane;
            if (GLOBAL) unsafeAtomicAdd(base + idx, 1.0f);
            else __hip_atomic_fetch_add(base + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    if (!GLOBAL) {
        float t = 0.f;
        for (int i = tid; i < rows * 32; i += 1024) t += lds[i];
        atomicAdd(out, t);
    }
}

// integer / 64-bit / packed-half LDS atomics on the 2 rows x 32 pattern, and the non-atomic read-add-write a wave
// may use when it knows its lanes' addresses are distinct
template <int KIND>
__global__ void __launch_bounds__(1024, 1) k2(int iters, int rows, float *out)   // (launched with 1024 or 512 threads)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < rows * 32 * (KIND == 1 || KIND == 4 ? 2 : 1); i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    uint32_t s = (blockIdx.x * 16 + wave) * 2654435761u + 12345u;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s = s * 1664525u + 1013904223u;
            uint32_t r = s >> 8;
            const int idx = ((r + (lane >> 5) * 7919u) & (rows - 1)) * 32 + (lane & 31);
            if (KIND == 0) __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(lds) + idx, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (KIND == 1) __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(lds) + idx, 3ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (KIND == 2) { float v = lds[idx]; lds[idx] = v + 1.0f; }                       // b32 read-add-write
            else if (KIND == 3) {                                                                    // b128 read-add-write: 8 rows x 4 floats x ... per instruction
                const int i4 = ((r + (lane >> 3) * 7919u) & (rows - 1)) * 8 + (lane & 7);
                float4 v = reinterpret_cast<float4 *>(lds)[i4];
                v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
                reinterpret_cast<float4 *>(lds)[i4] = v;
            } else if (KIND == 5) {   // the backward kernel's scatter: 8 rows x 8 lanes, octet (row & 3) ^ u of a random pixel per row
                const int row = lane >> 3, kk = lane & 7;
                const int pix = (r + row * 7919u) & (rows - 1);
                __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(lds) + pix * 32 + 8 * ((row & 3) ^ (u & 3)) + kk, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (KIND == 6) {   // the same with 64-bit adds: a lane owns two neighbouring channels, half (row & 1) ^ u of the pixel
                const int row = lane >> 3, kk = lane & 7;
                const int pix = (r + row * 7919u) & (rows - 1);
                __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(lds) + pix * 16 + 8 * ((row & 1) ^ (u & 1)) + kk, 3ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (KIND == 7) {   // 64-bit adds, 4 rows x 16 lanes: a lane owns two neighbouring channels of a whole pixel
                const int row = lane >> 4, kk = lane & 15;
                const int pix = (r + row * 7919u) & (rows - 1);
                __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(lds) + pix * 16 + kk, 3ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (KIND == 4) {                                                                  // b64 read-add-write
                const int i2 = ((r + (lane >> 4) * 7919u) & (rows - 1)) * 16 + (lane & 15);
                float2 v = reinterpret_cast<float2 *>(lds)[i2];
                v.x += 1.f; v.y += 1.f;
                reinterpret_cast<float2 *>(lds)[i2] = v;
            }
        }
    }
    __syncthreads();
    float t = acc.x;
    for (int i = tid; i < rows * 32; i += blockDim.x) t += lds[i];
    atomicAdd(out, t);
}

template <int KIND>
void run2(const char *name, int rows, int per_lane, int threads = 1024)
{
    float *out;
    hipMalloc(&out, 4); hipMemset(out, 0, 4);
    const int iters = 2000, blocks = 256;
    const int bytes = rows * 32 * 4 * (KIND == 1 ? 2 : 1);
    hipFuncSetAttribute((const void *)k2<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k2<KIND><<<blocks, threads, bytes>>>(10, rows, out);
    hipEventRecord(e0);
    k2<KIND><<<blocks, threads, bytes>>>(iters, rows, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double adds = (double)blocks * threads * iters * 8 * per_lane;
    printf("%-58s rows %5d  %8.3f ms  %8.1f G adds/s  (%5.2f adds/clk/CU at 2.1 GHz)\n", name, rows, ms, adds / ms * 1e-6,
           adds / ms * 1e-6 / 256 / 2.1);
    hipFree(out);
}

template <int PATTERN, bool GLOBAL>
void run(const char *name, int rows)
{
    float *out, *gbuf;
    hipMalloc(&out, 4); hipMemset(out, 0, 4);
    hipMalloc(&gbuf, (size_t)8 * rows * 32 * 4); hipMemset(gbuf, 0, (size_t)8 * rows * 32 * 4);
    const int iters = 2000, blocks = 256;
    hipFuncSetAttribute((const void *)k<PATTERN, GLOBAL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<PATTERN, GLOBAL><<<blocks, 1024, rows * 32 * 4>>>(10, rows, gbuf, out);
    hipEventRecord(e0);
    k<PATTERN, GLOBAL><<<blocks, 1024, rows * 32 * 4>>>(iters, rows, gbuf, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double adds = (double)blocks * 1024 * iters * 8;
    printf("%-58s rows %5d  %8.3f ms  %8.1f G adds/s  (%5.2f lanes/clk/CU at 2.1 GHz)\n", name, rows, ms, adds / ms * 1e-6,
           adds / ms * 1e-6 / 256 / 2.1);
    hipFree(out); hipFree(gbuf);
}

int main()
{
    run<3, false>("lds  same 64 floats", 1024);
    run<0, false>("lds  2 rows x 32 floats", 1024);
    run<1, false>("lds  4 rows x 16 floats", 1024);
    run<2, false>("lds  16 rows x 4 floats", 1024);
    run<0, false>("lds  2 rows x 32 floats, 64 rows (conflicts across waves)", 64);
    run2<0>("lds  ds_add_u32, 2 rows x 32", 1024, 1);
    run2<1>("lds  ds_add_u64, 2 rows x 32", 512, 1);
    run2<5>("lds  ds_add_u32, 8 rows x 8 lanes (rotated octets), 16 waves", 1024, 1);
    run2<5>("lds  ds_add_u32, 8 rows x 8 lanes (rotated octets), 8 waves", 1024, 1, 512);
    run2<0>("lds  ds_add_u32, 2 rows x 32, 8 waves", 1024, 1, 512);
    run2<6>("lds  ds_add_u64, 8 rows x 8 lanes x 2 (rotated halves), 16 w", 1024, 2);
    run2<6>("lds  ds_add_u64, 8 rows x 8 lanes x 2 (rotated halves), 8 w", 1024, 2, 512);
    run2<7>("lds  ds_add_u64, 4 rows x 16 lanes x 2, 8 waves", 1024, 2, 512);
    run2<2>("lds  read b32 / add / write b32 (racy between waves)", 1024, 1);
    run2<4>("lds  read b64 / add / write b64", 1024, 2);
    run2<3>("lds  read b128 / add / write b128", 1024, 4);
    run<0, true>("glob 2 rows x 32 floats", 1024);
    run<1, true>("glob 4 rows x 16 floats", 1024);
    return 0;
}
