// Does an XCD's L2 keep, across a kernel boundary, lines that the PREVIOUS kernel touched on that XCD?  (The question
// behind "let the launch in front of the MSDA gather warm the L2 with the head's value maps": in the step each layer's
// maps were written ~0.5 ms earlier, so the gather's first touch of every line goes to the Infinity Cache.)
//
// Chain replayed as a hipGraph:  flush (every XCD streams 48 MB: its 4 MB L2 holds none of X afterwards, X itself stays
// in the 256 MB Infinity Cache)  ->  [prefetch: 8 blocks per XCD read slab (block % 8) of X with discarded loads]  ->
// consumer (256 blocks x 1024 threads, block b gathers random 64-byte records from slab b % 8 -- the MSDA mapping).
// Reported: consumer time = chain with it minus chain without it, for no prefetch / prefetch on the consumer's XCD /
// prefetch on the WRONG XCD (slab (b + 1) % 8: must not help if the block -> XCD mapping is what we think it is), and
// the prefetch kernel's own time.
//   hipcc --offload-arch=gfx950 -O3 -o l2_prefetch benchmarks/micro/l2_prefetch.hip && ./l2_prefetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(1024) flush_kernel(const uint4 *buf, size_t n16, uint4 *sink, int never)
{
    // every XCD (block % 8) reads the whole buffer
    const int xcd = blockIdx.x & 7, part = blockIdx.x >> 3, parts = gridDim.x >> 3;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)part * 1024 + threadIdx.x; i < n16; i += (size_t)parts * 1024) {
        const uint4 v = buf[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if (never && acc.x == 0x12345678u) sink[xcd] = acc;   // (`never` is 0 at run time: keeps the loads alive)
}

__global__ void __launch_bounds__(1024) prefetch_kernel(const uint4 *x, size_t slab16, int shift, uint4 *sink, int never)
{
    const int slab = ((blockIdx.x & 7) + shift) & 7, part = blockIdx.x >> 3, parts = gridDim.x >> 3;
    const uint4 *src = x + (size_t)slab * slab16;
    uint4 acc = make_uint4(0, 0, 0, 0);
    // one 16-byte load per 128-byte line is enough to bring the line in
    for (size_t line = (size_t)part * 1024 + threadIdx.x; line * 8 < slab16; line += (size_t)parts * 1024) {
        const uint4 v = src[line * 8];
        acc.x ^= v.x;
    }
    if (never && acc.x == 0x12345678u) sink[slab] = acc;
}

__global__ void __launch_bounds__(1024) consumer_kernel(const uint4 *x, size_t slab16, int iters, uint4 *out)
{
    const int slab = blockIdx.x & 7;
    const uint4 *src = x + (size_t)slab * slab16;
    const uint32_t records = (uint32_t)(slab16 / 4);
    const int quad = (blockIdx.x * 1024 + threadIdx.x) >> 2, j = threadIdx.x & 3;
    uint32_t state = quad * 2654435761u + 12345u;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int i = 0; i < iters; i += 4) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            state = state * 1664525u + 1013904223u;
            const uint32_t r = (uint32_t)(((uint64_t)(state >> 8) * records) >> 24);
            v[u] = src[(size_t)r * 4 + j];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc.x ^= v[u].x; acc.y += v[u].y; acc.z ^= v[u].z; acc.w += v[u].w; }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

struct Chain {
    bool prefetch;
    int shift;
    int consume;   // how many consumer launches follow (the second one finds the first one's lines in L2)
};

static int time_chain(hipStream_t s, const Chain &c, const uint4 *flushbuf, size_t flush16, const uint4 *x, size_t slab16,
                      int pf_blocks, int iters, uint4 *sink, uint4 *out, float *us)
{
    hipGraph_t g;
    hipGraphExec_t exec;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(flush_kernel, dim3(256), dim3(1024), 0, s, flushbuf, flush16, sink, 0);
        if (c.prefetch) hipLaunchKernelGGL(prefetch_kernel, dim3(pf_blocks), dim3(1024), 0, s, x, slab16, c.shift, sink, 0);
        for (int n = 0; n < c.consume; ++n)
            hipLaunchKernelGGL(consumer_kernel, dim3(256), dim3(1024), 0, s, x, slab16, iters, out);
    }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int r = 0; r < 2; ++r) CK(hipGraphLaunch(exec, s));
    float best = 1e30f;
    for (int t = 0; t < 5; ++t) {
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(exec, s));
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    *us = best * 1e3f / reps;
    CK(hipGraphExecDestroy(exec));
    CK(hipGraphDestroy(g));
    return 0;
}

int main()
{
    const size_t flush_bytes = 48u << 20;
    uint4 *flushbuf, *x, *sink, *out;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    CK(hipMalloc(&flushbuf, flush_bytes));
    CK(hipMemset(flushbuf, 1, flush_bytes));
    CK(hipMalloc(&sink, 1024));
    CK(hipMalloc(&out, 256 * 1024 * 16));
    printf("{\"l2_prefetch\": [\n");
    bool first = true;
    // slab per XCD: 2.7 MB = one head's level-0/1 maps of two images; 1.4 MB = of one image
    for (size_t slab_bytes : {(size_t)2700 << 10, (size_t)1400 << 10}) {
        CK(hipMalloc(&x, slab_bytes * 8));
        CK(hipMemset(x, 2, slab_bytes * 8));
        const size_t slab16 = slab_bytes / 16;
        for (int iters : {64, 16}) {
            for (int pf_blocks : {64, 128}) {
                float base, cold, twice, pf_only, pf_cons, wrong_only, wrong_cons;
                if (time_chain(s, {false, 0, 0}, flushbuf, flush_bytes / 16, x, slab16, pf_blocks, iters, sink, out, &base)) return 1;
                if (time_chain(s, {false, 0, 1}, flushbuf, flush_bytes / 16, x, slab16, pf_blocks, iters, sink, out, &cold)) return 1;
                if (time_chain(s, {false, 0, 2}, flushbuf, flush_bytes / 16, x, slab16, pf_blocks, iters, sink, out, &twice)) return 1;
                if (time_chain(s, {true, 0, 0}, flushbuf, flush_bytes / 16, x, slab16, pf_blocks, iters, sink, out, &pf_only)) return 1;
                if (time_chain(s, {true, 0, 1}, flushbuf, flush_bytes / 16, x, slab16, pf_blocks, iters, sink, out, &pf_cons)) return 1;
                if (time_chain(s, {true, 1, 0}, flushbuf, flush_bytes / 16, x, slab16, pf_blocks, iters, sink, out, &wrong_only)) return 1;
                if (time_chain(s, {true, 1, 1}, flushbuf, flush_bytes / 16, x, slab16, pf_blocks, iters, sink, out, &wrong_cons)) return 1;
                printf("%s {\"slab_KB\": %zu, \"gathers_per_thread\": %d, \"prefetch_blocks\": %d, \"flush_us\": %.2f, "
                       "\"consumer_cold_us\": %.2f, \"consumer_again_us\": %.2f, \"prefetch_us\": %.2f, \"consumer_after_prefetch_us\": %.2f, "
                       "\"consumer_after_wrong_xcd_prefetch_us\": %.2f}",
                       first ? "" : ",\n", slab_bytes >> 10, iters, pf_blocks, base, cold - base, twice - cold, pf_only - base,
                       pf_cons - pf_only, wrong_cons - wrong_only);
                first = false;
            }
        }
        CK(hipFree(x));
    }
    printf("\n]}\n");
    return 0;
}
