// How fast does the chip retire whole-line fp32 atomics (no return) on device memory -- the flush of the LDS backward
// (msda_backward_tiled.hip): 1.8 M lines of 128 bytes per launch at 11 363 queries, every wave instruction two lines.
// Pattern: 256 workgroups x 8 waves; a workgroup walks "windows" of 26 rows x 42 pixels at pseudo-random places of a
// [2 x 22223 pixels x 8 heads x 32] fp32 buffer (pixel pitch 1 KB: the head's 128 bytes of a pixel are one line), a wave
// takes rows wave, wave + 8, ..., a lane is (pixel parity, channel).  Compared with plain stores of the same lines.
#include <hip/hip_runtime.h>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(512) k(float *buf, int windows_per_block, int W, int H, int64_t image_floats)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
    uint32_t s = blockIdx.x * 2654435761u + 12345u;
    for (int it = 0; it < windows_per_block; ++it) {
        s = s * 1664525u + 1013904223u;
        const int b = (s >> 8) & 1, m = (s >> 9) & 7;
        const int ox = (int)((s >> 12) % (uint32_t)(W - 42)), oy = (int)((s >> 20) % (uint32_t)(H - 26));
        float *base = buf + b * image_floats + m * 32;
        for (int wy = wave; wy < 26; wy += 8) {
            float *rowp = base + (int64_t)((oy + wy) * W + ox) * 256 + c;
            for (int wx = h; wx < 42; wx += 2) {
                if (MODE == 0) unsafeAtomicAdd(rowp + (int64_t)wx * 256, 1.0f);
                else if (MODE == 1) rowp[(int64_t)wx * 256] = 1.0f;
                else if (MODE == 2) atomicAdd(reinterpret_cast<int *>(rowp + (int64_t)wx * 256), 1);
            }
        }
    }
}

template <int MODE>
void run(const char *name, float *buf, int wpb)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int W = 168, H = 100;
    const int64_t image_floats = (int64_t)22223 * 256;
    k<MODE><<<256, 512>>>(buf, 1, W, H, image_floats);
    hipEventRecord(e0);
    k<MODE><<<256, 512>>>(buf, wpb, W, H, image_floats);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double lines = 256.0 * wpb * 26 * 42;
    printf("%-40s %8.1f us  %6.2f M lines  %7.1f M lines/s per CU  %6.1f G lane-ops/s\n", name, ms * 1e3, lines * 1e-6,
           lines / ms * 1e-3 / 256, lines * 32 / ms * 1e-6);
}

int main()
{
    float *buf;
    hipMalloc(&buf, (size_t)2 * 22223 * 256 * 4);
    hipMemset(buf, 0, (size_t)2 * 22223 * 256 * 4);
    for (int wpb : {1, 6, 24}) {
        printf("windows per workgroup: %d\n", wpb);
        run<0>("fp32 atomic add, no return", buf, wpb);
        run<2>("int32 atomic add, no return", buf, wpb);
        run<1>("plain store", buf, wpb);
    }
    return 0;
}
