// What a dependent kernel launch costs inside a replayed hipGraph on this stack (the hot path's step is a chain of ~70):
// chains of N kernels captured from one stream, replayed; time per node for
//   empty kernel, 1 workgroup | empty kernel, 1024 workgroups | a kernel that reads and writes 1 MB (256 workgroups)
//   | the same chain issued as plain stream launches (no graph).
//   hipcc --offload-arch=gfx950 -O3 -o graph_launch_floor benchmarks/micro/graph_launch_floor.hip && ./graph_launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void empty_kernel() {}
__global__ void touch_kernel(float4 *buf)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // 256 x 256 threads x 16 B = 1 MB
    float4 v = buf[i];
    v.x += 1.f;
    buf[i] = v;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <typename F>
static int time_chain(const char *name, int n, bool graph, F launch, hipStream_t s, bool last)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipGraphExec_t exec = nullptr;
    if (graph) {
        hipGraph_t g;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < n; ++i) launch(s);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
    }
    auto once = [&]() { if (graph) (void)hipGraphLaunch(exec, s); else for (int i = 0; i < n; ++i) launch(s); };
    for (int r = 0; r < 3; ++r) once();
    CK(hipStreamSynchronize(s));
    const int reps = 20;
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) once();
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf(" {\"chain\": \"%s\", \"graph\": %s, \"nodes\": %d, \"us_per_node\": %.2f}%s\n", name, graph ? "true" : "false", n,
           ms * 1e3f / reps / n, last ? "" : ",");
    return 0;
}

int main()
{
    hipStream_t s;
    CK(hipStreamCreate(&s));
    float4 *buf;
    CK(hipMalloc(&buf, 1 << 20));
    CK(hipMemset(buf, 0, 1 << 20));
    printf("{\"graph_launch_floor\": [\n");
    for (int graph = 1; graph >= 0; --graph) {
        if (time_chain("empty, 1 workgroup", 200, graph, [](hipStream_t st) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st); }, s, false)) return 1;
        if (time_chain("empty, 1024 workgroups x 256", 200, graph, [](hipStream_t st) { hipLaunchKernelGGL(empty_kernel, dim3(1024), dim3(256), 0, st); }, s, false)) return 1;
        if (time_chain("1 MB read + write, 256 workgroups", 200, graph, [buf](hipStream_t st) { hipLaunchKernelGGL(touch_kernel, dim3(256), dim3(256), 0, st, buf); }, s, graph == 0)) return 1;
    }
    printf("]}\n");
    return 0;
}
