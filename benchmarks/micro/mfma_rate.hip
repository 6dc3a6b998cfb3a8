// Ground-truth issue rates on gfx950 for the building blocks of the token-resident kernels: one wave per SIMD issuing
// v_mfma_f32_32x32x16_bf16 (independent / dependent accumulators), with and without an LDS fragment ring.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate benchmarks/micro/mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16_t mfma(u32x4_t a, u32x4_t b, f32x16_t c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// MODE 0: NACC independent accumulators, operands in registers
// MODE 1: as 0 + every MFMA's A operand comes from LDS through a ring of depth R (refilled after use)
template <int NACC, int MODE, int R>
__global__ void __launch_bounds__(256, 1) k(int iters, uint64_t *cycles, float *sink)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<uint32_t *>(lds)[i] = i * 2654435761u;
    __syncthreads();
    f32x16_t acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    u32x4_t a = {1u + lane, 2u, 3u, 4u}, b = {5u, 6u + lane, 7u, 8u};
    u32x4_t ring[R > 0 ? R : 1];
    const __attribute__((address_space(3))) char *base = (const __attribute__((address_space(3))) char *)lds + lane * 16;
    if (MODE == 1) {
#pragma unroll
        for (int f = 0; f < R; ++f) ring[f] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4_t *>(base + f * 1024);
    }
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int f = 0; f < 64; ++f) {
            if (MODE == 1) {
                acc[f % NACC] = mfma(ring[f % R], b, acc[f % NACC]);
                ring[f % R] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4_t *>(base + ((f + R) & 63) * 1024);
            } else {
                acc[f % NACC] = mfma(a, b, acc[f % NACC]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][7];
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// The token-resident kernels' step loop, feature by feature: 4 compute waves (64 MFMAs per step, A through an 8-deep
// ring from LDS) + optionally 4 loader waves that rewrite 64 KB of LDS per step and meet the compute waves at a
// barrier, + optionally the row-strided 8-byte output stores of four 32-feature tiles per step.
//   FEAT bit 0: loader waves + barrier per step;  bit 1: loaders really write LDS;  bit 2: global stores;
//   bit 3: per-tile bias reads (4 x ds_read_b128) and bf16 packing of the accumulators
template <int FEAT>
__global__ void __launch_bounds__(512, 1) step_kernel(int steps, uint64_t *cycles, float *sink, uint16_t *out, int out_stride)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768; i += 512) reinterpret_cast<uint32_t *>(lds)[i] = i * 2654435761u;
    __syncthreads();
    // FEAT bit 5 (32): hand-off through LDS flags instead of s_barrier -- flags[b] = loader arrivals for buffer b (4 per
    // step written into it), flags[2 + b] = compute waves finished with a step held in buffer b
    volatile __attribute__((address_space(3))) int *flags = (volatile __attribute__((address_space(3))) int *)(lds + 131072 - 64);
    if (FEAT & 32) {
        if (threadIdx.x < 4) flags[threadIdx.x] = 0;
        __syncthreads();
    }
    if (wave >= 4 && (FEAT & 32)) {
        u32x4_t v = {1u, 2u, 3u, 4u};
        __attribute__((address_space(3))) char *dst = (__attribute__((address_space(3))) char *)lds + (wave - 4) * 16000 + lane * 16;
        for (int st = 0; st + 1 < steps; ++st) {
            const int b = (st + 1) & 1;            // buffer of step st+1; last used by step st-1
            if (st >= 1) {
                const int need = 4 * ((st - 1) / 2 + 1);
                while (flags[2 + b] < need) __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int i = 0; i < 15; ++i)
                *reinterpret_cast<__attribute__((address_space(3))) u32x4_t *>(dst + b * 65536 + i * 1024) = v;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add((__attribute__((address_space(3))) int *)&flags[b], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return;
    }
    if (wave >= 4) {
        if (!(FEAT & 1)) return;
        u32x4_t v = {1u, 2u, 3u, 4u};
        __attribute__((address_space(3))) char *dst = (__attribute__((address_space(3))) char *)lds + (wave - 4) * 16384 + lane * 16;
        for (int st = 0; st < steps; ++st) {
            if (FEAT & 2) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    *reinterpret_cast<__attribute__((address_space(3))) u32x4_t *>(dst + ((st & 1) ^ 1) * 65536 + i * 1024) = v;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
        }
        return;
    }
    f32x16_t acc[4];
    u32x4_t b = {5u, 6u + lane, 7u, 8u};
    u32x4_t ring[8];
    const uint64_t t0 = __builtin_readcyclecounter();
    const int t = lane & 31, h = lane >> 5;
    uint16_t *orow = out + ((size_t)(blockIdx.x * 4 + wave) * 32 + t) * out_stride + 4 * h;
    for (int st = 0; st < steps; ++st) {
        const __attribute__((address_space(3))) char *base =
            (const __attribute__((address_space(3))) char *)lds + (st & 1) * 65536 + lane * 16;
        if ((FEAT & 32) && st >= 1) {
            const int need = 4 * ((st - 1) / 2 + 1);   // the four loaders have written step st into this buffer
            while (flags[st & 1] < need) __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int f = 0; f < 8; ++f) ring[f] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4_t *>(base + f * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (FEAT & 8) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const u32x4_t bv = *reinterpret_cast<const __attribute__((address_space(3))) u32x4_t *>(base + 60000 + j * 128 + 32 * g);
                    acc[j][4 * g] = __uint_as_float(bv.x); acc[j][4 * g + 1] = __uint_as_float(bv.y);
                    acc[j][4 * g + 2] = __uint_as_float(bv.z); acc[j][4 * g + 3] = __uint_as_float(bv.w);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
            }
        }
#pragma unroll
        for (int f = 0; f < 64; ++f) {
            acc[f & 3] = mfma(ring[f % 8], b, acc[f & 3]);
            if (f + 8 < 64) ring[f % 8] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4_t *>(base + (f + 8) * 1024);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (FEAT & 16) {
            // halves exchange 8-byte pieces so that a lane owns features 16m .. 16m+7 (+8 for the upper half) of its token:
            // two 16-byte stores per tile instead of four 8-byte ones, no LDS
            typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
            typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t d[8];
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    d[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector((f2){acc[j][2 * i], acc[j][2 * i + 1]}, bf2));
                // d[2g], d[2g+1] = features 8g + 4h + {0..3}.  swap (h=0: g odd) <-> (h=1: g even)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        const auto r = __builtin_amdgcn_permlane32_swap(d[4 * m + w], d[4 * m + 2 + w], false, false);
                        d[4 * m + w] = r[0];
                        d[4 * m + 2 + w] = r[1];
                    }
                // now: h = 0 holds features 16m + 0..7 in d[4m .. 4m+3]; h = 1 holds 16m + 8..15
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    *reinterpret_cast<u32x4_t *>(orow + ((st & 3) * 4 + j) * 32 + 16 * m + 4 * h) = (u32x4_t){d[4 * m], d[4 * m + 1], d[4 * m + 2], d[4 * m + 3]};
            }
        } else if (FEAT & 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 v;
                    if (FEAT & 8) {
                        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
                        typedef float f2 __attribute__((ext_vector_type(2)));
                        v.x = __builtin_bit_cast(uint32_t, __builtin_convertvector((f2){acc[j][4 * g], acc[j][4 * g + 1]}, bf2));
                        v.y = __builtin_bit_cast(uint32_t, __builtin_convertvector((f2){acc[j][4 * g + 2], acc[j][4 * g + 3]}, bf2));
                    } else {
                        v = make_uint2(__float_as_uint(acc[j][4 * g]), __float_as_uint(acc[j][4 * g + 2]));
                    }
                    *reinterpret_cast<uint2 *>(orow + ((st & 3) * 4 + j) * 32 + 8 * g) = v;
                }
        } else {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][9];
            if (s == 123.456f) sink[0] = s;
        }
        if (FEAT & 32) {
            if (lane == 0) __hip_atomic_fetch_add((__attribute__((address_space(3))) int *)&flags[2 + (st & 1)], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (FEAT & 1) __builtin_amdgcn_s_barrier();
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// The same step with its output SOFTWARE-PIPELINED: while the 64 MFMAs of step s run, the accumulators of step s-1 go
// through a per-wave LDS staging tile (8-byte writes in accumulator order, 16-byte reads row-major) to coalesced global
// stores, a few instructions per MFMA slot.  Loader waves rewrite LDS and meet the compute waves at a barrier as before.
struct Stage {
    __attribute__((address_space(3))) char *tile;   // [32][136]
    uint16_t *orow;                                 // this lane's first output row (row-major piece addressing)
    int out_stride, t, h, lane;
};

template <int OPT>
__device__ __forceinline__ void epilogue_slot(const f32x16_t (&acc)[4], const Stage &sg, int st, int slot, u32x4_t (&pend)[4])
{
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    // per part (32 slots): slots 0..7 writes, 14..17 reads, 22..25 stores (the read data is consumed 8 slots later)
    const int part = slot >> 5, s = slot & 31;
    if (s < 8 && !(OPT & 32)) {
        const int jj = s >> 2, g = s & 3;
        const f32x16_t &a = acc[2 * part + jj];
        u32x2_t v;
        v.x = __builtin_bit_cast(uint32_t, __builtin_convertvector((f2){a[4 * g], a[4 * g + 1]}, bf2));
        v.y = __builtin_bit_cast(uint32_t, __builtin_convertvector((f2){a[4 * g + 2], a[4 * g + 3]}, bf2));
        *reinterpret_cast<__attribute__((address_space(3))) u32x2_t *>(sg.tile + sg.t * 136 + (jj * 32 + 8 * g + 4 * sg.h) * 2) = v;
    } else if (s >= 14 && s < 18 && !(OPT & 16)) {
        const int i = s - 14;
        const int row = (sg.lane >> 3) + 8 * i, piece = sg.lane & 7;
        pend[i] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4_t *>(sg.tile + row * 136 + piece * 16);
    } else if (s >= 22 && s < 26 && !(OPT & 16)) {
        const int i = s - 22;
        const int row = (sg.lane >> 3) + 8 * i, piece = sg.lane & 7;
        const u32x4_t v = pend[i];
        if (sg.orow) *reinterpret_cast<u32x4_t *>(sg.orow + (size_t)row * sg.out_stride + ((st & 3) * 4 + 2 * part) * 32 + piece * 8) = v;
        else if (v.x == 0x12345678u) *reinterpret_cast<__attribute__((address_space(3))) uint32_t *>(sg.tile) = v.y;
    }
}

template <bool PIPE, int OPT>   // OPT bit 0: loaders do not write LDS; bit 1: no barrier; bit 2: no global stores; bit 3: no staging at all
__global__ void __launch_bounds__(512, 1) piped_kernel(int steps, uint64_t *cycles, float *sink, uint16_t *out, int out_stride)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768; i += 512) reinterpret_cast<uint32_t *>(lds)[i] = i * 2654435761u;
    __syncthreads();
    if (wave >= 4) {
        u32x4_t v = {1u, 2u, 3u, 4u};
        __attribute__((address_space(3))) char *dst = (__attribute__((address_space(3))) char *)lds + (wave - 4) * 16384 + lane * 16;
        if (OPT & 2) return;
        for (int st = 0; st < steps; ++st) {
            if (!(OPT & 1)) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    *reinterpret_cast<__attribute__((address_space(3))) u32x4_t *>(dst + ((st & 1) ^ 1) * 65536 + i * 1024) = v;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
        }
        return;
    }
    Stage sg;
    sg.lane = lane; sg.t = lane & 31; sg.h = lane >> 5; sg.out_stride = out_stride;
    sg.tile = (__attribute__((address_space(3))) char *)lds + 131072 + wave * (32 * 136);
    sg.orow = (OPT & 4) ? nullptr : out + (size_t)(blockIdx.x * 4 + wave) * 32 * out_stride;
    f32x16_t accA[4], accB[4];
    u32x4_t b = {5u, 6u + lane, 7u, 8u};
    u32x4_t ring[8];
    u32x4_t pend[4];
    auto step = [&](f32x16_t (&cur)[4], const f32x16_t (&prev)[4], int st, bool drain) {
        const __attribute__((address_space(3))) char *base =
            (const __attribute__((address_space(3))) char *)lds + (st & 1) * 65536 + lane * 16;
#pragma unroll
        for (int f = 0; f < 8; ++f) ring[f] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4_t *>(base + f * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const u32x4_t bv = *reinterpret_cast<const __attribute__((address_space(3))) u32x4_t *>(base + 60000 + j * 128 + 32 * g);
                cur[j][4 * g] = __uint_as_float(bv.x); cur[j][4 * g + 1] = __uint_as_float(bv.y);
                cur[j][4 * g + 2] = __uint_as_float(bv.z); cur[j][4 * g + 3] = __uint_as_float(bv.w);
            }
#pragma unroll
        for (int f = 0; f < 64; ++f) {
            cur[f & 3] = mfma(ring[f % 8], b, cur[f & 3]);
            if (f + 8 < 64) ring[f % 8] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4_t *>(base + (f + 8) * 1024);
            if (PIPE && drain && !(OPT & 8)) epilogue_slot<OPT>(prev, sg, st - 1, f, pend);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!PIPE && !(OPT & 8)) {
#pragma unroll
            for (int f = 0; f < 64; ++f) epilogue_slot<OPT>(cur, sg, st, f, pend);
        }
        if (!(OPT & 2)) __builtin_amdgcn_s_barrier();
    };
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int st = 0; st < steps; st += 2) {
        step(accA, accB, st, st > 0);
        step(accB, accA, st + 1, true);
    }
    if (PIPE) {
#pragma unroll
        for (int f = 0; f < 64; ++f) epilogue_slot<OPT>(accB, sg, steps - 1, f, pend);
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (accA[0][0] == 123.456f) sink[0] = 1.f;
}

template <bool PIPE, int OPT = 0>
void run_piped(const char *name, int blocks)
{
    uint64_t *d; float *sink; uint16_t *out;
    const int steps = 48, stride = 512;
    hipMalloc(&d, blocks * 8); hipMalloc(&sink, 4); hipMalloc(&out, (size_t)blocks * 128 * stride * 2);
    const int lds = 131072 + 4 * 32 * 136;
    hipFuncSetAttribute((const void *)piped_kernel<PIPE, OPT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((piped_kernel<PIPE, OPT>), dim3(blocks), dim3(512), lds, 0, steps, d, sink, out, stride);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((piped_kernel<PIPE, OPT>), dim3(blocks), dim3(512), lds, 0, steps, d, sink, out, stride);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(blocks);
    hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    printf("%-62s blocks %3d: %.0f ticks / step (2048 = MFMA bound), %.2f us / tile (wall)\n", name, blocks, (double)h[0] / steps,
           ms * 1e3 / (steps * 4));
    hipFree(d); hipFree(sink); hipFree(out);
}

template <int FEAT>
void run_step(const char *name, int blocks)
{
    uint64_t *d; float *sink; uint16_t *out;
    const int steps = 48, stride = 512;
    hipMalloc(&d, blocks * 8); hipMalloc(&sink, 4); hipMalloc(&out, (size_t)blocks * 128 * stride * 2);
    hipFuncSetAttribute((const void *)step_kernel<FEAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((step_kernel<FEAT>), dim3(blocks), dim3(512), 131072, 0, steps, d, sink, out, stride);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((step_kernel<FEAT>), dim3(blocks), dim3(512), 131072, 0, steps, d, sink, out, stride);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(blocks);
    hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    printf("%-62s blocks %3d: %.0f ticks / step (2048 = MFMA bound), %.2f us / tile (wall)\n", name, blocks, (double)h[0] / steps,
           ms * 1e3 / (steps * 4));
    hipFree(d); hipFree(sink); hipFree(out);
}

template <int NACC, int MODE, int R>
void run(const char *name, int blocks)
{
    uint64_t *d; float *sink;
    hipMalloc(&d, blocks * 8); hipMalloc(&sink, 4);
    const int iters = 200;
    hipFuncSetAttribute((const void *)k<NACC, MODE, R>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NACC, MODE, R>), dim3(blocks), dim3(256), 65536, 0, iters, d, sink);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, MODE, R>), dim3(blocks), dim3(256), 65536, 0, iters, d, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(blocks);
    hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    const double n = 64.0 * iters;
    printf("%-46s blocks %3d: %.1f counter ticks / MFMA, %.1f ns / MFMA (wall), %.0f TFLOP/s\n", name, blocks, h[0] / n,
           ms * 1e6 / n, blocks * 4 * n * 32768.0 / (ms * 1e-3) / 1e12);
    hipFree(d); hipFree(sink);
}

// plain streaming kernels: what the memory system gives a kernel that does nothing else
__global__ void __launch_bounds__(256) stream_write(u32x4_t *dst, size_t n)
{
    const u32x4_t v = {1u, 2u, 3u, (uint32_t)threadIdx.x};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = v;
}
__global__ void __launch_bounds__(256) stream_read(const u32x4_t *src, size_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const u32x4_t v = src[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void __launch_bounds__(256) stream_copy(const u32x4_t *src, u32x4_t *dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

void run_stream(size_t mb)
{
    const size_t n = mb * 1024 * 1024 / 16;
    u32x4_t *a, *b; uint32_t *sink;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&sink, 4);
    hipMemset(a, 1, n * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int which = 0; which < 3; ++which) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(stream_write, dim3(2048), dim3(256), 0, 0, b, n);
            if (which == 1) hipLaunchKernelGGL(stream_read, dim3(2048), dim3(256), 0, 0, a, n, sink);
            if (which == 2) hipLaunchKernelGGL(stream_copy, dim3(2048), dim3(256), 0, 0, a, b, n);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const char *names[3] = {"write", "read", "copy (read + write bytes)"};
        printf("stream %-26s %5zu MB: %.1f us, %.2f TB/s\n", names[which], mb, ms * 1e3, (which == 2 ? 2.0 : 1.0) * n * 16 / (ms * 1e-3) / 1e12);
    }
    hipFree(a); hipFree(b); hipFree(sink);
}

int main()
{
    for (size_t mb : {16, 64, 137, 1024}) run_stream(mb);
    for (int blocks : {16, 256}) {
        run<4, 0, 0>("4 independent accumulators, register operands", blocks);
        run<2, 0, 0>("2 accumulators (dependent distance 2)", blocks);
        run<1, 0, 0>("1 accumulator (fully dependent chain)", blocks);
        run<4, 1, 4>("4 accumulators, A from LDS, ring depth 4", blocks);
        run<4, 1, 8>("4 accumulators, A from LDS, ring depth 8", blocks);
        run<1, 1, 8>("1 accumulator, A from LDS, ring depth 8", blocks);
    }
    for (int blocks : {178}) {
        run_step<0>("step loop: MFMAs + ring only", blocks);
        run_step<8>("+ bias reads / bf16 packing", blocks);
        run_step<1>("+ loader waves at a barrier (idle)", blocks);
        run_step<3>("+ loader waves rewriting 64 KB of LDS per step", blocks);
        run_step<4>("+ row-strided 8-byte stores (no loaders)", blocks);
        run_step<12>("+ stores + packing (no loaders)", blocks);
        run_step<15>("everything", blocks);
        run_step<16 + 8>("permlane32_swap + 16-byte row-strided stores (no loaders)", blocks);
        run_step<16 + 8 + 3>("permlane32_swap + 16-byte row-strided stores + loaders", blocks);
        run_step<32 + 16 + 8>("  ... with LDS-flag hand-off instead of s_barrier", blocks);
        run_step<32 + 4 + 8>("8-byte stores + loaders with LDS-flag hand-off", blocks);
        run_step<32 + 8>("no stores, loaders with LDS-flag hand-off", blocks);
        run_piped<false>("loaders + LDS-staged coalesced stores after the MFMAs", blocks);
        run_piped<true>("loaders + staged stores of step s-1 under the MFMAs of step s", blocks);
        run_piped<true, 1>("  ... loaders idle", blocks);
        run_piped<true, 2>("  ... no barrier, no loaders", blocks);
        run_piped<true, 4>("  ... no global stores (staging only)", blocks);
        run_piped<true, 8>("  ... no epilogue at all", blocks);
        run_piped<true, 4 + 16>("  ... staging writes only", blocks);
        run_piped<true, 4 + 32>("  ... staging reads only", blocks);
    }
    return 0;
}
