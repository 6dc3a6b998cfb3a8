// The encoder's class score of a 32-row tile that lies in LDS (models/bricks/salience_transformer.py:462, 366:
// max_c(class_head(q)) -- the caller multiplies by the foreground score): logits^T [96 x 32] = Wc q^T + bc as three
// 32-class tiles, one per wave, sixteen v_mfma_f32_32x32x16 on one accumulator in k order (the chain of ffn.hip's NEXT
// form) from the class head's packed fragments (ffn.hip class_head_pack_kernel: fragment f = 3 ks + tile, 1 KB each).
// Shared by the second pass of the split feed-forward (ffn.hip) and the encoder's entry gather (plumbing.hip).
#pragma once

#include "common.h"

namespace sdetr {

constexpr int kClsTileRows = 32;
constexpr int kClsRowBytes = 528;   // a 512-byte row + 16: the B-operand reads of a 16-lane group hit 64 distinct banks

// fragments ks0 .. ks1-1 of class tile `e` (the wave's 32 classes)
template <int KS0, int KS1>
__device__ __forceinline__ void class_frag_load(uint4 (&af)[16], const char *cls_pw, int e, int lane)
{
#pragma unroll
    for (int ks = KS0; ks < KS1; ++ks) af[ks] = reinterpret_cast<const uint4 *>(cls_pw)[(3 * ks + e) * 64 + lane];
}

// max over the 32 classes of tile `e` for token lane & 31 (both lane halves return it); padded classes carry a bias of
// -inf and zero weights.  `ytile`: [32][kClsRowBytes] 16-bit activation rows.
__device__ __forceinline__ float class_tile_max(const uint4 (&af)[16], const unsigned char *ytile, const float *cls_bias, int e,
                                                int lane)
{
    const int t = lane & 31, h = lane >> 5;
    act_f32x16_t acc;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 bv = *reinterpret_cast<const float4 *>(cls_bias + 32 * e + 8 * g + 4 * h);
        acc[4 * g] = bv.x; acc[4 * g + 1] = bv.y; acc[4 * g + 2] = bv.z; acc[4 * g + 3] = bv.w;
    }
    const unsigned char *qrow = ytile + t * kClsRowBytes + 16 * h;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) acc = mfma_act_32x32x16(af[ks], *reinterpret_cast<const uint4 *>(qrow + 32 * ks), acc);
    float mx = acc[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, acc[i]);
    return fmaxf(mx, __shfl_xor(mx, 32));
}

}  // namespace sdetr
