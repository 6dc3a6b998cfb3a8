// The two memory legs of the MSDA gather in isolation (gfx950): 64-byte records (one head's 32 fp16 channels of a
// pixel) read at random, four lanes per record and 16 bytes per lane, 16 waves per CU --
//   from an 85 KB LDS slab (ds_read_b128), and from a 22 MB array in global memory (L2 / Infinity Cache resident,
//   global_load_dwordx4 through the L1).  Reported: bytes per CU and microsecond.
//   hipcc --offload-arch=gfx950 -O3 -o gather_rate benchmarks/micro/gather_rate.hip && ./gather_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <bool GLOBAL>
__global__ void __launch_bounds__(1024, 1) gather_kernel(const uint4 *g, int records, int iters, uint32_t *out)
{
    extern __shared__ __attribute__((aligned(16))) uint4 slab[];
    const int tid = threadIdx.x, lane = tid & 63;
    if (!GLOBAL) {
        for (int i = tid; i < records * 4; i += 1024) slab[i] = make_uint4(i, i, i, i);
        __syncthreads();
    }
    uint32_t s = (blockIdx.x * 1024 + (tid >> 2)) * 2654435761u + 12345u;   // one stream per 4-lane group
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s = s * 1664525u + 1013904223u;
            const uint32_t r = (uint32_t)(((uint64_t)(s >> 4) * (uint32_t)records) >> 28);
            v[u] = GLOBAL ? g[(size_t)r * 4 + (lane & 3)] : slab[r * 4 + (lane & 3)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[tid] = acc;
}

template <bool GLOBAL>
static void run(const char *name, const uint4 *g, int records, uint32_t *out, bool last)
{
    const int iters = 400, blocks = 256;
    const size_t ldsb = GLOBAL ? 0 : (size_t)records * 64;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gather_kernel<GLOBAL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(gather_kernel<GLOBAL>, dim3(blocks), dim3(1024), ldsb, 0, g, records, iters, out);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(gather_kernel<GLOBAL>, dim3(blocks), dim3(1024), ldsb, 0, g, records, iters, out);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes_per_cu = 1024.0 * 16 * 8 * iters;
    printf(" {\"source\": \"%s\", \"records\": %d, \"us\": %.1f, \"bytes_per_cu_per_us\": %.0f}%s\n", name, records, ms * 1e3,
           bytes_per_cu / (ms * 1e3), last ? "" : ",");
}

int main()
{
    const int grec = 22 * 1024 * 1024 / 64;
    uint4 *g;
    uint32_t *out;
    (void)hipMalloc(&g, (size_t)grec * 64);
    (void)hipMemset(g, 1, (size_t)grec * 64);
    (void)hipMalloc(&out, 4096);
    printf("{\"gather_rate\": [\n");
    run<false>("LDS slab", g, 1323, out, false);
    run<true>("global, 22 MB", g, grec, out, false);
    run<true>("global, 1.4 MB (one head, levels 0+1)", g, 20900, out, true);
    printf("]}\n");
    return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
