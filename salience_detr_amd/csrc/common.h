// Shared helpers for the gfx950 kernels of libsalience_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/salience_hip.h"

namespace sdetr {

constexpr int kWave = 64;        // CDNA wavefront
constexpr int kBlock = 256;      // 4 waves, one per SIMD
constexpr int kMaxLevels = 16;   // levels handled by the LDS-table kernels

// thread-local error text returned by sdetr_last_error()
char *error_buffer();

inline int fail(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return SDETR_EINVAL;
}

inline int check_launch(const char *what)
{
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        snprintf(error_buffer(), 512, "%s: launch failed: %s", what, hipGetErrorString(err));
        return (int)err;
    }
    return 0;
}

// ---- bf16 <-> f32 (storage type is a raw 16-bit pattern) -------------------------------------
using bf16_t = uint16_t;

__device__ __forceinline__ float bf16_lo(uint32_t packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t packed) { return __uint_as_float(packed & 0xffff0000u); }

// round-to-nearest-even, NaN kept quiet (same rule as torch's float -> bfloat16)
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f)
{
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi)
{
    return f32_to_bf16_bits(lo) | (f32_to_bf16_bits(hi) << 16);
}

}  // namespace sdetr
