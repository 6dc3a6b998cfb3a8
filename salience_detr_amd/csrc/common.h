// Shared helpers for the gfx950 kernels of libsalience_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

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

// A/B switches of the benchmark scripts (kernel forms, tile sizes, warm-ups): read from the environment ONLY in the
// benchmark build (-DSDETR_AB_SWITCHES, csrc/build.py build_ablations -> benchmarks/libsalience_hip_ablate.so).  In the two
// product libraries this returns NULL for every name: what they compute never depends on the environment; the choices
// a test needs (the gather's accumulation form, the GEMM generation) are explicit entry-point arguments.
inline const char *ab_env(const char *name)
{
#ifdef SDETR_AB_SWITCHES
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
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

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: raise it once per (kernel, device), not once
// per process (a second GPU driven from the same process would otherwise launch with the 64 KiB default and fail).
// `slots` is a per-kernel static array of 64 flags indexed by the current device.
struct DeviceOnce {
    std::atomic<bool> done[64];
};
template <typename K>
inline void allow_dynamic_lds(K kernel, DeviceOnce &slots, int bytes)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!slots.done[dev].load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        slots.done[dev].store(true, std::memory_order_release);
    }
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
// two floats -> packed bf16 pair with the hardware converter (v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays
// NaN -- the rule of f32_to_bf16_bits above, without its compare / select chain)
typedef float f32x2_vec_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_vec_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi)
{
    const f32x2_vec_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_vec_t));
}

// ---- IEEE fp16 storage (distinct C++ type so templates can tell it from bf16) -----------------
struct half_t {
    uint16_t bits;
};
typedef _Float16 f16x2_pair_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi)
{
    // round-to-nearest-even, saturating at +-65504 so a large activation never becomes inf in storage (a NaN lands on
    // -65504: v_med3_f32 returns the minimum of the other two).  Three instructions: two v_med3_f32, one v_cvt_pk_f16_f32.
    const f32x2_vec_t v = {__builtin_amdgcn_fmed3f(lo, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(hi, -65504.f, 65504.f)};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_pair_t));
}
__device__ __forceinline__ float f16_lo(uint32_t packed) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(packed & 0xffffu)); }
__device__ __forceinline__ float f16_hi(uint32_t packed) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(packed >> 16)); }
// acc += f32(half of `packed`) * w as ONE instruction (v_fma_mix_f32: f16 source converted exactly, single
// rounding) -- no separate unpack, which is a third of the bf16 gather's instruction stream
__device__ __forceinline__ float fma_f16lo(uint32_t packed, float w, float acc)
{
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(packed), "v"(w));
    return acc;
}
__device__ __forceinline__ float fma_f16hi(uint32_t packed, float w, float acc)
{
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(packed), "v"(w));
    return acc;
}

// ---- the library's 16-bit ACTIVATION type (round 5) --------------------------------------------------------------------
// Token rows, projection slabs, attention / feed-forward operands and 16-bit outputs are "activations".  The product
// library libsalience_hip.so keeps them in bfloat16; the SAME sources built with -DSDETR_ACT_F16 give
// libsalience_hip_f16.so, in which every activation is IEEE half (the reference's `--mixed-precision fp16`, main.py:24-56,
// BASELINE.json configs[4]): v_mfma_f32_*_f16 runs at the bf16 rate, accumulators / LayerNorm / softmax / class scores /
// sampling locations stay fp32 in both, and stores SATURATE at +-65504 where fp16's range would bite (pack_f16x2).
// Weight packing permutes 16-bit words and is type-agnostic (the host hands the weights over in the activation type).
// What is NOT an activation and keeps its explicit type in both flavours: the head-major value maps (`half_t` | `bf16_t`
// by their own dtype argument) and the bf16 x 3 split of the fp32-accurate products (salience head, gemm_x3).
typedef float act_f32x16_t __attribute__((ext_vector_type(16)));
typedef float act_f32x4_t __attribute__((ext_vector_type(4)));
#ifdef SDETR_ACT_F16
#define SDETR_ACT_IS_F16 1
#define SDETR_ACT_MFMA_SUFFIX "f16"
typedef _Float16 act_x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float act_lo(uint32_t packed) { return f16_lo(packed); }
__device__ __forceinline__ float act_hi(uint32_t packed) { return f16_hi(packed); }
__device__ __forceinline__ uint32_t pack_act2(float lo, float hi) { return pack_f16x2(lo, hi); }
__device__ __forceinline__ uint32_t f32_to_act_bits(float f) { return pack_f16x2(f, 0.f) & 0xffffu; }
__device__ __forceinline__ act_f32x16_t mfma_act_32x32x16(uint4 a, uint4 b, act_f32x16_t c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(act_x8_t, a), __builtin_bit_cast(act_x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ act_f32x4_t mfma_act_16x16x32(uint4 a, uint4 b, act_f32x4_t c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(act_x8_t, a), __builtin_bit_cast(act_x8_t, b), c, 0, 0, 0);
}
#else
#define SDETR_ACT_IS_F16 0
#define SDETR_ACT_MFMA_SUFFIX "bf16"
typedef __bf16 act_x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float act_lo(uint32_t packed) { return bf16_lo(packed); }
__device__ __forceinline__ float act_hi(uint32_t packed) { return bf16_hi(packed); }
__device__ __forceinline__ uint32_t pack_act2(float lo, float hi) { return pack_bf16x2(lo, hi); }
__device__ __forceinline__ uint32_t f32_to_act_bits(float f) { return f32_to_bf16_bits(f); }
__device__ __forceinline__ act_f32x16_t mfma_act_32x32x16(uint4 a, uint4 b, act_f32x16_t c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(act_x8_t, a), __builtin_bit_cast(act_x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ act_f32x4_t mfma_act_16x16x32(uint4 a, uint4 b, act_f32x4_t c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(act_x8_t, a), __builtin_bit_cast(act_x8_t, b), c, 0, 0, 0);
}
#endif
// storage type of an activation element (a raw 16-bit pattern in both flavours) and the dtype code activation arguments
// carry: SDETR_BF16 in libsalience_hip.so, SDETR_F16 in libsalience_hip_f16.so
using act_t = uint16_t;
constexpr int kActCode = SDETR_ACT_IS_F16 ? SDETR_F16 : SDETR_BF16;

// 16-byte load through a buffer resource: the block-uniform base lives in the (SGPR) descriptor, the lane
// supplies a 32-bit byte offset -- one v_add_u32 per corner instead of a 64-bit VALU address
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_uniform_rsrc(const char *base, uint32_t num_bytes)
{
    // make uniformity provable: the pointer halves go through readfirstlane (cdna_hip_programming.md T20)
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    char *p = reinterpret_cast<char *>(((uint64_t)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(p, /*stride*/ 0, (int)__builtin_amdgcn_readfirstlane(num_bytes), 0x00020000);
}
__device__ __forceinline__ uint4 buffer_load16(__amdgpu_buffer_rsrc_t r, uint32_t byte_off)
{
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}

typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 buffer_load8(__amdgpu_buffer_rsrc_t r, uint32_t byte_off)
{
    const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, 0);
    return make_uint2(v.x, v.y);
}

}  // namespace sdetr
