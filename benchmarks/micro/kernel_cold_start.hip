// What the first trips to memory cost a small kernel inside a chain of dependent launches (replayed hipGraph): chains
// of 200 launches that alternate between two buffers, each kernel doing 0, 1, 2 or 3 DEPENDENT global loads before its
// store (pointer chase through small index arrays), 1 workgroup and 64 workgroups.
//   hipcc --offload-arch=gfx950 -O3 -o kernel_cold_start benchmarks/micro/kernel_cold_start.hip && ./kernel_cold_start
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int TRIPS>
__global__ void chase_kernel(const int *idx, int *out)
{
    int v = threadIdx.x + blockIdx.x * blockDim.x;
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) v = idx[v];
    out[threadIdx.x + blockIdx.x * blockDim.x] = v;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int TRIPS>
static int run(int blocks, int *a, int *b, hipStream_t s, bool last)
{
    hipGraph_t g;
    hipGraphExec_t exec;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 200; ++i)   // each launch reads what the previous one wrote: a genuine dependency chain
        hipLaunchKernelGGL(chase_kernel<TRIPS>, dim3(blocks), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(exec, s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(exec, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf(" {\"dependent_loads\": %d, \"workgroups\": %d, \"us_per_launch\": %.2f}%s\n", TRIPS, blocks, ms * 1e3f / 10 / 200,
           last ? "" : ",");
    return 0;
}

int main()
{
    const int n = 64 * 256;
    std::vector<int> h(n);
    for (int i = 0; i < n; ++i) h[i] = (i * 7919 + 13) % n;   // a permutation-like scatter of indices inside [0, n)
    int *a, *b;
    CK(hipMalloc(&a, n * 4));
    CK(hipMalloc(&b, n * 4));
    CK(hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, h.data(), n * 4, hipMemcpyHostToDevice));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    printf("{\"kernel_cold_start\": [\n");
    for (int blocks : {1, 64}) {
        if (run<0>(blocks, a, b, s, false)) return 1;
        if (run<1>(blocks, a, b, s, false)) return 1;
        if (run<2>(blocks, a, b, s, false)) return 1;
        if (run<3>(blocks, a, b, s, blocks == 64)) return 1;
    }
    printf("]}\n");
    return 0;
}
